// Launch interface of the MFMA 3x3 convolution (conv3x3.hip).
#pragma once
#include "common.h"

namespace pnpx {

struct ConvArgs {
  const float* in0;  // first input tensor (padded planar), C0 channels
  const float* in1;  // second input tensor (channel-concatenated after in0), C1 channels (0 if none)
  const float* wpk;  // packed weights [cout/MT][cin/CC][9][CC][MT]
  const float* bias; // [cout]
  float* out;        // padded planar, nct*MT channels
  int C0, C1;
  int H, W, Hp, Wp;
  int tilesX, tilesY, nct;
  float slope;       // LeakyReLU negative slope
  // epilogue mode: 0 = bias + LeakyReLU (forward); 1 = raw accumulator (no bias, no activation);
  // 2 = raw accumulator x LeakyReLU'(dmask) -- the input-gradient convolution of the backward pass, where dmask is
  // the saved forward activation that this gradient flows into (same [B][Cout][Hp][Wp] geometry as `out`)
  // 4 = like 2, but the saved activation is a half-split HS8 tensor (conv_hs.hip layout: [B][C/8][H+2][W+2] records of
  // hi[8] | lo[8] f16); only the sign of `hi` is used
  int mode;
  const float* dmask;
  const char* dmask_hs = nullptr;
  const float* res = nullptr;   // mode 0: added after the activation (same geometry as `out`; the ResBlock skip of the DRUNet)
};

int conv_pack_mt(int cout);
int conv_pack_cc(int cin);
void pack_conv_weights(const float* w, int cout, int cin, int mt, int cc, float* dst);
int launch_conv3x3(const ConvLayer& L, const float* in0, int C0, const float* in1, int C1, float* out, int B,
                   int H, int W, hipStream_t s);
// ... with a chosen negative slope (0.2 = the UNet's LeakyReLU, 0 = ReLU, 1 = linear) and an optional residual operand
int launch_conv3x3_act(const ConvLayer& L, const float* in0, int C0, const float* in1, int C1, float* out, int B,
                       int H, int W, float slope, const float* res, hipStream_t s);
// 1x1 convolution on the same kernel (centre tap alone): L packed by pack_conv_weights_1x1 (mt 64, cc 8)
int launch_conv1x1_act(const ConvLayer& L, const float* in0, float* out, int B, int H, int W, float slope, const float* res, hipStream_t s);
void pack_conv_weights_1x1(const float* w, int cout, int cin, float* dst);
// Input-gradient convolution: L holds the transposed, tap-flipped weights (pack_conv_weights_transposed).
int launch_conv3x3_grad(const ConvLayer& L, const float* gin, float* gout, const float* dmask, int B, int H, int W,
                        hipStream_t s, const char* dmask_hs = nullptr, float mask_slope = 0.2f);   // mask_slope: LeakyReLU' below zero (0 = ReLU)
// w[cout][cin][3][3] -> packed weights of the adjoint convolution: wt[ci][co][tap] = w[co][ci][8 - tap], with the
// adjoint's output channels (= cin) zero-padded to cout_pad.
// Winograd F(2x2, 3x3) variant of the same layer on the fp32 MFMA (conv3x3_wino.hip; option fp32_winograd).  cout a multiple of 64:
// both sources multiples of 16 channels, H and W multiples of 16; otherwise cout a multiple of 32: sources multiples of 8 channels,
// H a multiple of 16, W of 32.  u: pack_conv_weights_wino's output.
bool conv3x3_wino_ok(int C0, int C1, int cout, int H, int W);
bool conv3x3_wino_packs(int cout, int cin);   // the layer gets Winograd weights at load time
size_t conv3x3_wino_floats(int cout, int cin);
void pack_conv_weights_wino(const float* w, int cout, int cin, float* dst);
int launch_conv3x3_wino(const float* u, const float* bias, int cout, const float* in0, int C0, const float* in1, int C1,
                        float* out, int B, int H, int W, hipStream_t s, float slope = 0.2f, const float* res = nullptr,
                        float* pool_out = nullptr);   // pool_out: also write MaxPool2d(2) of the output (padded planar, H/2 x W/2)
// The UNet's last 3x3 layer (-> 32 channels) with the 1x1 out-conv + input residual + clamp in its epilogue: writes img (clamped)
// and, if given, img_pre; the 32-channel tensor never exists.  Summation order of the 1x1: two 16-channel halves, then their sum.
bool conv3x3_wino_outc_ok(int cin, int cout, int H, int W);
int launch_conv3x3_wino_outc(const float* u, const float* bias, const float* in0, int cin, const float* outc_w, const float* outc_b,
                             const float* x_img, float* img, float* img_pre, int B, int H, int W, hipStream_t s);
// The same layers on the 8-wave Winograd kernel (conv3x3_wino8.hip: the 16 positions of a block split over the two waves of a SIMD;
// same packed weights `u`); option fp32_wino8.
bool conv3x3_wino8_ok(int C0, int C1, int cout, int H, int W);
int launch_conv3x3_wino8(const float* u, const float* bias, int cout, const float* in0, int C0, const float* in1, int C1,
                         float* out, int B, int H, int W, hipStream_t s, float slope = 0.2f, const float* res = nullptr,
                         float* pool_out = nullptr, float* ks_part = nullptr, int ks_rule = 1);
// r6: K-split of the deep levels' layers (a rule of the layer's geometry alone; 1 = no split) and the scratch a caller provides for it
int conv3x3_wino8_ksplit(int C0, int C1, int cout, int H, int W, int rule = 1);
size_t conv3x3_wino8_ksplit_bytes_per_image();
int launch_conv3x3_wino8_grad(const float* u, const float* zero_bias, int cout, const float* gin, int cin, float* gout, const float* dmask,
                              float mask_slope, int B, int H, int W, hipStream_t s);
bool conv3x3_wino8_ups_ok(int C0, int C1, int cout, int H, int W);
int launch_conv3x3_wino8_ups(const float* u, const float* bias, int cout, const float* in0, int C0, const float* in1_lowres, int C1,
                             float* out, int B, int H, int W, hipStream_t s, float slope = 0.2f);
int launch_conv3x3_wino8_outc(const float* u, const float* bias, const float* in0, int cin, const float* outc_w, const float* outc_b,
                              const float* x_img, float* img, float* img_pre, int B, int H, int W, hipStream_t s);
void pack_conv_weights_transposed(const float* w, int cout, int cin, int cout_pad, int mt, int cc, float* dst);

}  // namespace pnpx
