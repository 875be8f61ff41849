// Kernels shared by the denoiser VJPs (unet_bwd.hip, drunet.hip); `static`: one copy per translation unit.
#pragma once
#include "conv_hs.h"
#include "hs_rec.h"

namespace pnpx {

constexpr int SIG_CHUNKS = 64;

static __global__ void sigma_grad_final_kernel(const float* __restrict__ part, float* __restrict__ gsigma, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < SIG_CHUNKS; ++c) s += part[b * SIG_CHUNKS + c];
  gsigma[b] = s;
}

static __global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ g, size_t n, unsigned* __restrict__ bits) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(g[i]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(bits, __float_as_uint(m));   // non-negative floats order like their bits
}
static __global__ void grad_scale_kernel(const unsigned* __restrict__ bits, float2* __restrict__ sc) {
  const float m = __uint_as_float(*bits);
  if (!(m > 0.f) || !isfinite(m)) {
    *sc = make_float2(m > 0.f ? 1.f : 0.f, m > 0.f ? 1.f : 0.f);   // all-zero gradient -> zeros; inf/nan -> pass through
    return;
  }
  int e;
  (void)frexpf(m, &e);                    // m = f * 2^e, f in [0.5, 1)
  *sc = make_float2(ldexpf(1.f, -e), ldexpf(1.f, e));
}

static __global__ __launch_bounds__(256) void input_grad_hs_kernel(const HsRec* __restrict__ g_in0, const float* __restrict__ g_res,
                                                            float* __restrict__ gx, float* __restrict__ part,
                                                            const float2* __restrict__ sc, int H, int W) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int n = H * W, per = (n + SIG_CHUNKS - 1) / SIG_CHUNKS;
  const int lo = chunk * per, hi = min(n, lo + per);
  const HsRec* g0 = g_in0 + (size_t)b * 4 * (H + 2) * (W + 2);     // group 0 of 4: channels 0 (image) and 1 (noise map)
  const float m = sc->y, inv = 1.f / HS_ASCALE;
  float acc = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const int y = i / W, x = i - y * W;
    float v[8];
    hs_unpack(g0[(size_t)(y + 1) * (W + 2) + x + 1], v);
    gx[(size_t)b * n + i] = (v[0] * inv + g_res[(size_t)b * n + i]) * m;
    acc += v[1] * inv;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[b * SIG_CHUNKS + chunk] = ((w[0] + w[1]) + (w[2] + w[3])) * m;
}

}  // namespace pnpx
