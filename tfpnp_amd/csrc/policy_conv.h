// Convolution kernel of the policy actor (policy_conv.hip).
#pragma once
#include "common.h"

namespace pnpx {

// One (cout tile, K-chunk) work item of a policy convolution: which of the 9 taps are non-zero and where their
// weights live.  Only chunks with a non-zero mask are listed, only present taps are stored.
struct PolStep {
  unsigned short mask;    // bit t = tap t present
  unsigned short chunk;   // K-chunk index (8 input channels)
  unsigned int wofs;      // offset of the first present tap slice, in 512-float ([8][64]) slices
};

// out_hs: both outputs are written as half-split HS8 tensors ([B][C/8][H+2][W+2] records, conv_hs.hip) for the
// stride-1 convolutions of the residual blocks, which run on the f16x3 MFMA kernel.
// fp32 planar tensors of the policy path ([B][C][h+2][pol_wp(w)], interior pixel (y,x) at row y+1, column x+POL_PADL):
// the left border column sits at a 16-byte boundary and rows are whole 16-byte units, so a tile's halo rows are copied
// by 16-byte LDS-DMA (4x fewer DMA instructions than the dword gather of conv3x3.hip).
constexpr int POL_PADL = 5;
__host__ __device__ inline int pol_wp(int w) { return (w + 8 + 3) / 4 * 4; }

int launch_policy_conv(const PolicyConv& L, const float* in, float* out, float* out2, const float* res, bool s2d, int B,
                       int H, int W, hipStream_t s, bool out_hs = false);

}  // namespace pnpx
