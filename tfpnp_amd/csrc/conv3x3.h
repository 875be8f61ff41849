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
  int mode;
  const float* dmask;
  // ---- policy-network variant only (template POL, launch_conv3x3_policy; mode 3):
  //   v = acc + bias[c] (+ res[same position]);  ReLU unless the cout tile is >= split (linear shortcut branch);
  //   cout tiles >= split are written to out2 (channel index restarts at 0);  s2d: the output is stored
  //   space-to-depth ([4*C][H/2][W/2], channel = (y&1)*2+(x&1) major) for a following stride-2 convolution.
  const unsigned short* tapmask = nullptr;  // [nct][nch] bit t set = tap t of that K-chunk is non-zero
  const float* res = nullptr;
  float* out2 = nullptr;
  int split = 0, C_out1 = 0, C_out2 = 0, s2d = 0;
};

int conv_pack_mt(int cout);
int conv_pack_cc(int cin);
void pack_conv_weights(const float* w, int cout, int cin, int mt, int cc, float* dst);
int launch_conv3x3(const ConvLayer& L, const float* in0, int C0, const float* in1, int C1, float* out, int B,
                   int H, int W, hipStream_t s);
// Input-gradient convolution: L holds the transposed, tap-flipped weights (pack_conv_weights_transposed).
int launch_conv3x3_grad(const ConvLayer& L, const float* gin, float* gout, const float* dmask, int B, int H, int W,
                        hipStream_t s);
// Policy-network convolution (PolicyConv: common.h; epilogue: see ConvArgs).
int launch_conv3x3_policy(const PolicyConv& L, const float* in, float* out, float* out2, const float* res, bool s2d,
                          int B, int H, int W, hipStream_t s);
// w[cout][cin][3][3] -> packed weights of the adjoint convolution: wt[ci][co][tap] = w[co][ci][8 - tap], with the
// adjoint's output channels (= cin) zero-padded to cout_pad.
void pack_conv_weights_transposed(const float* w, int cout, int cin, int cout_pad, int mt, int cc, float* dst);

}  // namespace pnpx
