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

struct ConvHsArgs {
  const char* in0;   // HS8 tensor, G0 groups of 8 channels
  const char* in1;   // second source (channel concat), G1 groups
  const char* wpk;
  const float* bias;
  char* out;         // HS8 tensor, nct*MT/8 groups
  int G0, G1;
  int H, W, Hp, Wp;
  int tilesX, tilesY, nct, B;
  float inv_scale, slope;
};

int conv_hs_mt(int cout);
float pack_conv_weights_hs(const float* w, int cout, int cin, int mt, uint16_t* dst);
int launch_conv_hs(const ConvLayerHs& L, const char* in0, int G0, const char* in1, int G1, char* out, int B, int H,
                   int W, hipStream_t s);

}  // namespace pnpx
