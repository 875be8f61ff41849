// Launch interface of the half-split (f16 MFMA, fp32-class accuracy) 3x3 convolution (conv_hs.hip).
#pragma once
#include <cmath>

#include "common.h"

namespace pnpx {

constexpr float HS_ASCALE = 16.0f;   // activations are stored as split(16 * v)

struct ConvLayerHs {
  int cin = 0, cout = 0, cin_pad = 0, mt = 0;
  const char* w = nullptr;    // device: [cout/mt][cin_pad/16][9][hi,lo][2][mt][8] f16
  const float* b = nullptr;   // device: [cout] fp32
  float inv_scale = 1.f;      // 1 / (weight_scale * HS_ASCALE)
};

// n / d for small n by one multiply-high (exact for n * d < 2^32; checked at launch): the persistent tile walk
// decodes its work items with these instead of hardware-less integer division sequences.
struct HsFastDiv {
  unsigned m, d;
};
inline HsFastDiv hs_fastdiv(unsigned d) {
  HsFastDiv f;
  f.d = d;
  f.m = d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u;
  return f;
}

struct ConvHsArgs {
  int plain_walk = 0;   // set by the launcher: one tile per workgroup, tile j = blockIdx.x (launches that fit one round)
  int share = 1;        // launch chains running side by side (ConvHsFuse::share): the launch table plans for 256 / share workgroups
  const char* in0;   // HS8 tensor, G0 groups of 8 channels
  const char* in1;   // second source (channel concat), G1 groups
  const char* wpk;
  const float* bias;
  char* out;         // HS8 tensor, nct*MT/8 groups
  char* pool_out;    // optional: fused MaxPool2d(2) of the output, HS8 [B][G][H/2+2][W/2+2] (W >= 32 tiles only)
  // optional fused tail of the network (last conv only, cout == 32): out = clamp(x + outc(v) + b)
  const float* outc_w;   // [32] or null
  const float* outc_b;   // [1]
  const float* x_in;     // [B,1,H,W] network input (residual)
  float* out_img;        // [B,1,H,W]
  float* out_pre;        // [B,1,H,W] or null
  int G0, G1;
  int G0t;           // channel groups per image of the tensor behind in0 (>= G0: a layer may read only its first G0 groups)
  int H, W, Hp, Wp;
  int tilesX, tilesY, nct, B;
  HsFastDiv div_nct, div_tx, div_ty;
  float inv_scale, slope;
  // backward pass (input-gradient convolution): LeakyReLU' taken from the sign of a saved HS8 activation with the
  // geometry of `out` instead of from the result itself
  const char* dmask;
  // residual add before the activation (policy ResNet blocks): HS8 tensor with the geometry of `out`
  const char* res;
  float neg_one;           // -1.0f (kept in a scalar register: selects the fused f16 fma-mix forms in the hi/lo split)
  unsigned* range_flag;    // sticky half-split range guard (host-mapped word) or null
  unsigned long long* trace;   // HS_TRACE builds: s_memtime stamps of workgroup 0 / wave 0 (null otherwise)
  unsigned long long* wgt;     // PNPX_TUNING builds: per-workgroup {start, end} wall-clock stamps (100 MHz) or null
  int abl;                     // PNPX_TUNING builds: ablation bits (1 = drop the record stores, 2 = skip the epilogue, 4 / 8 = planar-layout addressing of loads / stores)
  int w_mt;                    // cout tile the weights were PACKED for (64 while a 32-cout instance runs: "half tiles")
  // fused bilinear x2 (UPS instance): in1 is the low-resolution tensor [B][G1][ups_h+2][ups_w+2], H = 2*ups_h, W = 2*ups_w
  int ups_h, ups_w;
  float ups_sy, ups_sx;        // (h-1)/(2h-1), (w-1)/(2w-1): align_corners=True source step
  // WREG == 2 ("first layer folded in"): the 32-channel input of this layer is not read from `in0` but computed per tile
  // as LeakyReLU(first_b + first_w * [x, sigma]) straight from the fp32 network input (conv_first.h's arithmetic)
  const float* first_x;        // [B,1,H,W]
  const float* first_sigma;    // [B * first_sigma_stride]
  const float* first_w;        // [32][2][9]
  const float* first_b;        // [32]
  const float* first_zero;     // a zero word in device memory (window elements outside the image)
  int first_sigma_stride;
  float first_slope;
};

int conv_hs_mt(int cout);
float pack_conv_weights_hs(const float* w, int cout, int cin, int mt, uint16_t* dst);
// same from w[cout][cin][3][3] but storing only the taps of `tapmask` (in ascending tap order): [..][ntaps][hi,lo][kg][mt][8]
float pack_conv_weights_hs_taps(const float* w, int cout, int cin, int mt, int tapmask, uint16_t* dst);
struct ConvHsFuse {       // optional fused work
  char* pool_out = nullptr;
  const float* outc_w = nullptr;
  const float* outc_b = nullptr;
  const float* x_in = nullptr;
  float* out_img = nullptr;
  float* out_pre = nullptr;
  const char* dmask = nullptr;  // input-gradient mode: out = acc * (dmask > 0 ? 1 : slope), no bias expected (pass zeros)
  float slope = 0.2f;           // in [0, 1]: 1.0 = linear epilogue, 0.0 = ReLU
  const char* res = nullptr;    // out = act(conv + bias + res)
  unsigned* range_flag = nullptr;   // set to 1 by the kernel when a stored value leaves the f16 range (|v| >= 4095) or is NaN
  // fused bilinear x2: `in1` of launch_conv_hs is the LOW-resolution tensor (ups_h x ups_w) and is up-sampled on the fly
  // (cout == 32 layers with G0 >= 4 only: conv_hs_can_fuse_upsample)
  int ups_h = 0, ups_w = 0;
  // cin = cout = 32 single-source layers: weights in registers, one pipeline step per tile (conv_hs_kernel.h WREG):
  // 0 = generic kernel, 1 = four waves x four pixel blocks, 2 = eight waves x two pixel blocks.  Same K order: same bits.
  int wreg = 2;
  // sparse-tap layers (EPI_ACT only): bit mask of the 3x3 taps the layer was PACKED with (pack_conv_weights_hs_taps);
  // 0x1FF = ordinary 3x3.  in0_groups: channel groups per image of the in0 tensor when the layer reads only its first G0.
  int taps = 0x1FF;
  int in0_groups = 0;
  // fold the network's first convolution (2 -> 32 channels, K = 18, vector ALU) into THIS 32 -> 32 layer's tile loader
  // (weights-in-registers instance only): its 32-channel output tensor is then neither written nor read
  const float* first_x = nullptr;
  const float* first_sigma = nullptr;
  const float* first_w = nullptr;
  const float* first_b = nullptr;
  const float* first_zero = nullptr;
  int first_sigma_stride = 0;
  float first_slope = 0.2f;
  // r5: number of independent launch chains the caller runs side by side (unet.hip launch_chains): this launch shares the 256 CUs with
  // share - 1 launches of the same shape, and the launch table picks the tile height for 256 / share workgroups (same bits either way)
  int share = 1;
};
// true when launch_conv_hs will honour ConvHsFuse::first_x for this layer / geometry (else the caller runs conv_first)
bool conv_hs_can_fold_first(const ConvLayerHs& L, int G0, int B, int H, int W, const ConvHsFuse& fuse);
bool conv_hs_can_fuse_upsample(const ConvLayerHs& L, int G0, int G1, int H, int W);
// true when launch_conv_hs will honour ConvHsFuse::pool_out for this geometry
bool conv_hs_can_pool(int H, int W);
int launch_conv_hs(const ConvLayerHs& L, const char* in0, int G0, const char* in1, int G1, char* out, int B, int H,
                   int W, const ConvHsFuse& fuse, hipStream_t s);

}  // namespace pnpx
