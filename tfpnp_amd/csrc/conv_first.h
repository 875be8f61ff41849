// First convolution of a denoiser on the vector ALU (shared by unet.hip and drunet.hip; `static`: one copy per TU).
#pragma once
#include "conv_hs.h"
#include "hs_rec.h"

namespace pnpx {

// First convolution of the network, fused with the input preparation (denoiser/base.py:27-30 + inc.conv-0,
// models/unet.py:8-18): out[c] = LeakyReLU(b[c] + sum_taps w[c][0][t] * x[t] + w[c][1][t] * sigma * [t inside the image]).
// K is only 18, so padding it to the 16-channel chunks of the MFMA kernel wastes 8x the multiplies and a 100 MB
// padded-input round trip; here each thread evaluates 8 output channels of one pixel as exact fp32 FMA chains straight
// from the fp32 image (weights are wave-uniform: scalar loads) and stores one HS8 record.  HBM-write bound.
static __global__ __launch_bounds__(256) void conv_first_hs_kernel(const float* __restrict__ x, const float* __restrict__ sigma,
                                                            int sigma_stride, const float* __restrict__ w,
                                                            const float* __restrict__ bias, HsRec* __restrict__ dst, int H,
                                                            int W, float slope, float oscale) {   // oscale: HS_ASCALE, or HS_ASCALE * 2^-k for a down-scaled pass (drunet.hip)
  const int g = blockIdx.y, b = blockIdx.z;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, xx = p - y * W;
  const float sg = sigma[(size_t)b * sigma_stride];
  const float* xb = x + (size_t)b * H * W;
  float xi[9], si[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xc = xx + t % 3 - 1;
    const bool in = (yy >= 0) && (yy < H) && (xc >= 0) && (xc < W);
    xi[t] = in ? xb[(size_t)yy * W + xc] : 0.f;
    si[t] = in ? sg : 0.f;
  }
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float* wc = w + (size_t)(g * 8 + c) * 18;
    float acc = bias[g * 8 + c];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = fmaf(wc[t], xi[t], acc);
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = fmaf(wc[9 + t], si[t], acc);
    v[c] = fmaxf(acc, acc * slope) * oscale;
  }
  dst[((size_t)(b * gridDim.y + g) * (H + 2) + (y + 1)) * (W + 2) + xx + 1] = hs_pack(v);   // gridDim.y = cout / 8
}

// First convolution of the fp32 family on the vector ALU, fused with the input preparation (denoiser/base.py:27-30 + inc.conv-0,
// models/unet.py:8-18), like conv_first_hs_kernel for the half-split family: K is only 18, the MFMA kernel pads it to an 8-channel chunk
// and runs HBM-write bound at half the rate (0.166 ms at 48 x 256^2 for 403 MB).  One thread = 4 pixels along x of 8 output
// channels: the 3 x 6 input window is loaded once, weights are wave-uniform (scalar loads), one 16-byte store per channel into
// the padded planar tensor.  Sum order: bias, the image's nine taps, the noise map's nine taps (the hs kernel's).
static __global__ __launch_bounds__(256) void conv_first_f32_kernel(const float* __restrict__ x, const float* __restrict__ sigma, int sigma_stride,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ dst, int H, int W, float slope) {
  const int g = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * blockDim.x + threadIdx.x, wq = W / 4;
  if (q >= H * wq) return;
  const int y = q / wq, x0 = (q - y * wq) * 4;
  const float sg = sigma[(size_t)b * sigma_stride];
  const float* xb = x + (size_t)b * H * W;
  float xi[3][6], si[3][6];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int yy = y + r - 1, xc = x0 + c - 1;
      const bool in = (yy >= 0) && (yy < H) && (xc >= 0) && (xc < W);
      xi[r][c] = in ? xb[(size_t)yy * W + xc] : 0.f;
      si[r][c] = in ? sg : 0.f;
    }
  const int Hp = padded_h(H), Wp = padded_w(W);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float* wc = w + (size_t)(g * 8 + c) * 18;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = bias[g * 8 + c];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(wc[t], xi[t / 3][j + t % 3], acc);
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(wc[9 + t], si[t / 3][j + t % 3], acc);
      o[j] = fmaxf(acc, acc * slope);
    }
    *reinterpret_cast<float4*>(dst + ((size_t)(b * gridDim.y * 8 + g * 8 + c) * Hp + (y + 1)) * Wp + x0 + PADL) = make_float4(o[0], o[1], o[2], o[3]);
  }
}


}  // namespace pnpx
