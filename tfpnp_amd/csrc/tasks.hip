// Phase retrieval (CDP), single-photon imaging (Poisson prox), sparse-view CT (Radon pair) and PSNR on gfx950.
//   tasks/pr/solver.py:37-76, tasks/spi/solver.py:17-52, tasks/ct/solver.py:17-87,
//   tfpnp/utils/transforms.py:282-320 (cdp), :404-439 (spi_inverse), :447-508 (Radon wrapper),
//   tfpnp/env/base.py:237-242 (torch_psnr).
#include <cmath>

#include "common.h"
#include "fft_lds.h"

namespace pnpx {
namespace {

__device__ __forceinline__ float mulr(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float addr(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float subr(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float divr(float a, float b) { return __fdiv_rn(a, b); }

inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// =================================================================================== phase retrieval
// FFT batch index img = b*S + s.
struct LoadCdp {  // complex_mul(x[b], mask[b,s])                       transforms.py:297-299
  const float2* x;     // item b at x + b*xstride
  size_t xstride;
  const float2* mask;  // [B,S,H,W]
  int S, W, HW;
  __device__ float2 operator()(int img, int y, int xx) const {
    const int b = img / S;
    const size_t r = (size_t)y * W + xx;
    const float2 a = x[(size_t)b * xstride + r], m = mask[(size_t)img * HW + r];
    return make_float2(subr(mulr(a.x, m.x), mulr(a.y, m.y)), addr(mulr(a.x, m.y), mulr(a.y, m.x)));
  }
};
struct MidPrResidual {  // (|Az| - y0) / |Az| * Az, no epsilon            tasks/pr/solver.py:64-67
  const float* y0;      // [B,S,H,W]
  int W, HW;
  __device__ float2 operator()(int img, int ky, int kx, float2 v) const {
    const float yh = sqrtf(addr(mulr(v.x, v.x), mulr(v.y, v.y)));
    const float q = divr(subr(yh, y0[(size_t)img * HW + (size_t)ky * W + kx]), yh);
    return make_float2(mulr(q, v.x), mulr(q, v.y));
  }
};

// g = mean_s(complex_mul(I_s, conj(mask_s)))                              transforms.py:318-320
__device__ __forceinline__ float2 cdp_adjoint_px(const float2* I, const float2* mask, int b, int S, int HW, size_t r) {
  float2 acc = make_float2(0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const size_t o = ((size_t)b * S + s) * HW + r;
    const float2 a = I[o], m = mask[o];
    const float my = -m.y;
    acc.x = addr(acc.x, subr(mulr(a.x, m.x), mulr(a.y, my)));
    acc.y = addr(acc.y, addr(mulr(a.x, my), mulr(a.y, m.x)));
  }
  return make_float2(divr(acc.x, (float)S), divr(acc.y, (float)S));
}
__global__ void cdp_adjoint_kernel(const float2* __restrict__ I, const float2* __restrict__ mask,
                                   float2* __restrict__ out, int S, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const int b = (int)(i / HW);
  out[i] = cdp_adjoint_px(I, mask, b, S, HW, i - (size_t)b * HW);
}
// z = z - tau*(g + mu*(z - (x+u))); u = u + x - z; d = Re(z - u)           tasks/pr/solver.py:69-72
__global__ void pr_update_kernel(const float2* __restrict__ I, const float2* __restrict__ mask,
                                 const float* __restrict__ xr, const float2* zin, const float2* uin, float2* xout,
                                 float2* zout, float2* uout, size_t istride, float* __restrict__ d,
                                 const float* __restrict__ mu, const float* __restrict__ tau, int stride, int S, int HW,
                                 int B, int write_x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const int b = (int)(i / HW);
  const size_t r = i - (size_t)b * HW;
  const float2 g = cdp_adjoint_px(I, mask, b, S, HW, r);
  const float m = mu[(size_t)b * stride], t = tau[(size_t)b * stride];
  const float2 z = zin[(size_t)b * istride + r], u = uin[(size_t)b * istride + r];
  const float xv = xr[i];
  const float2 zn = make_float2(subr(z.x, mulr(t, addr(g.x, mulr(m, subr(z.x, addr(xv, u.x)))))),
                                subr(z.y, mulr(t, addr(g.y, mulr(m, subr(z.y, addr(0.f, u.y)))))));
  const float2 un = make_float2(subr(addr(u.x, xv), zn.x), subr(addr(u.y, 0.f), zn.y));
  zout[(size_t)b * istride + r] = zn;
  uout[(size_t)b * istride + r] = un;
  d[i] = subr(zn.x, un.x);
  if (write_x) xout[(size_t)b * istride + r] = make_float2(xv, 0.f);
}
// The same update as the accumulator of the grouped inverse row pass (fft_lds.h fft256_rows_group_kernel): g is summed over the S masks
// in registers, in the order and with the operations of cdp_adjoint_px, and the pixel is updated right there -- the S image-space
// fields never reach memory.  Bit-identical to cdp rows pass + pr_update_kernel.
struct PrUpdateAcc {
  const float2* mask;     // [B*S][HW]
  const float* xr;        // [B][HW]
  const float2 *zin, *uin;
  float2 *xout, *zout, *uout;
  size_t istride;
  float* d;
  const float *mu, *tau;
  int stride, S, W, HW, write_x;
  typedef float2 State;
  typedef float2 Pre;
  __device__ float2 init() const { return make_float2(0.f, 0.f); }
  __device__ float2 fetch(int img, int y, int x) const { return mask[(size_t)img * HW + (size_t)y * W + x]; }
  __device__ void add(float2& acc, float2 m, float2 a) const {
    const float my = -m.y;
    acc.x = addr(acc.x, subr(mulr(a.x, m.x), mulr(a.y, my)));
    acc.y = addr(acc.y, addr(mulr(a.x, my), mulr(a.y, m.x)));
  }
  __device__ void finish(float2 acc, int b, int y, int x) const {
    const size_t r = (size_t)y * W + x, i = (size_t)b * HW + r;
    const float2 g = make_float2(divr(acc.x, (float)S), divr(acc.y, (float)S));
    const float m = mu[(size_t)b * stride], t = tau[(size_t)b * stride];
    const float2 z = zin[(size_t)b * istride + r], u = uin[(size_t)b * istride + r];
    const float xv = xr[i];
    const float2 zn = make_float2(subr(z.x, mulr(t, addr(g.x, mulr(m, subr(z.x, addr(xv, u.x)))))),
                                  subr(z.y, mulr(t, addr(g.y, mulr(m, subr(z.y, addr(0.f, u.y)))))));
    const float2 un = make_float2(subr(addr(u.x, xv), zn.x), subr(addr(u.y, 0.f), zn.y));
    zout[(size_t)b * istride + r] = zn;
    uout[(size_t)b * istride + r] = un;
    d[i] = subr(zn.x, un.x);
    if (write_x) xout[(size_t)b * istride + r] = make_float2(xv, 0.f);
  }
};
// ---- training path of IADMMSolver_PR (pnpx_pr_iadmm_train / _backward)
struct MidPrResidualSave {  // MidPrResidual that also keeps w = F(mask_s z) of every (item, s)
  const float* y0;
  float2* wsave;   // [B*S][HW]
  int W, HW;
  __device__ float2 operator()(int img, int ky, int kx, float2 v) const {
    const size_t o = (size_t)img * HW + (size_t)ky * W + kx;
    wsave[o] = v;
    const float yh = sqrtf(addr(mulr(v.x, v.x), mulr(v.y, v.y)));
    const float q = divr(subr(yh, y0[o]), yh);
    return make_float2(mulr(q, v.x), mulr(q, v.y));
  }
};
// Adjoint of r = w - y0 w / |w| at the saved w (the Jacobian is symmetric): h = c - y0 (c - wh (wh . c)) / |w|, wh = w / |w|
struct MidPrResidualAdjoint {
  const float* y0;
  const float2* wsave;
  int W, HW;
  __device__ float2 operator()(int img, int ky, int kx, float2 c) const {
    const size_t o = (size_t)img * HW + (size_t)ky * W + kx;
    const float2 w = wsave[o];
    const float yh = sqrtf(w.x * w.x + w.y * w.y);
    const float2 wh = make_float2(w.x / yh, w.y / yh);
    const float dot = wh.x * c.x + wh.y * c.y;
    const float f = y0[o] / yh;
    return make_float2(c.x - f * (c.x - wh.x * dot), c.y - f * (c.y - wh.y * dot));
  }
};
struct LoadCdpDiff {  // complex_mul(a[b] - c[b], mask[b,s]): the cotangent e = gz' - gu' entering the data step's adjoint
  const float2 *a, *c;
  size_t xstride;
  const float2* mask;
  int S, W, HW;
  __device__ float2 operator()(int img, int y, int xx) const {
    const int b = img / S;
    const size_t r = (size_t)y * W + xx;
    const float2 p = a[(size_t)b * xstride + r], q = c[(size_t)b * xstride + r], m = mask[(size_t)img * HW + r];
    const float2 e = make_float2(p.x - q.x, p.y - q.y);
    return make_float2(e.x * m.x - e.y * m.y, e.x * m.y + e.y * m.x);
  }
};
// pr_update_kernel of the training forward: also keeps G = g + mu (z - (x + u)) (= (z - z') / tau) and q2 = z - (x + u)
__global__ void pr_update_save_kernel(const float2* __restrict__ I, const float2* __restrict__ mask,
                                      const float* __restrict__ xr, const float2* zin, const float2* uin, float2* xout,
                                      float2* zout, float2* uout, size_t istride, float* __restrict__ d,
                                      const float* __restrict__ mu, const float* __restrict__ tau, int stride, int S, int HW,
                                      int B, int write_x, float2* __restrict__ Gs, float2* __restrict__ q2s) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const int b = (int)(i / HW);
  const size_t r = i - (size_t)b * HW;
  const float2 g = cdp_adjoint_px(I, mask, b, S, HW, r);
  const float m = mu[(size_t)b * stride], t = tau[(size_t)b * stride];
  const float2 z = zin[(size_t)b * istride + r], u = uin[(size_t)b * istride + r];
  const float xv = xr[i];
  const float2 q2 = make_float2(subr(z.x, addr(xv, u.x)), subr(z.y, addr(0.f, u.y)));
  const float2 G = make_float2(addr(g.x, mulr(m, q2.x)), addr(g.y, mulr(m, q2.y)));
  const float2 zn = make_float2(subr(z.x, mulr(t, G.x)), subr(z.y, mulr(t, G.y)));
  const float2 un = make_float2(subr(addr(u.x, xv), zn.x), subr(addr(u.y, 0.f), zn.y));
  zout[(size_t)b * istride + r] = zn;
  uout[(size_t)b * istride + r] = un;
  d[i] = subr(zn.x, un.x);
  if (write_x) xout[(size_t)b * istride + r] = make_float2(xv, 0.f);
  Gs[i] = G;
  q2s[i] = q2;
}
// Backward of one PR iteration after the data step's adjoint I' (per (item, s), image domain), J = mean_s conj(mask_s) I'_s:
//   e = gz' - gu';  gz = (1 - tau mu) e - tau J;  t = tau mu e;  cotangent of the denoiser output = Re(gx' + gu' + t);
//   gu = gu' + t;  gx = 0 (an iteration never reads its x);  d/d tau = -<e, G>;  d/d mu = -tau <e, q2>
__global__ void pr_adjoint_kernel(const float2* __restrict__ I, const float2* __restrict__ mask, float2* g, size_t istride,
                                  const float2* __restrict__ Gs, const float2* __restrict__ q2s,
                                  const float* __restrict__ mu, const float* __restrict__ tau, int stride, int S, int HW, int B,
                                  float* __restrict__ gxr, float* __restrict__ c_tau, float* __restrict__ c_mu) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const int b = (int)(i / HW);
  const size_t r = i - (size_t)b * HW;
  const float2 J = cdp_adjoint_px(I, mask, b, S, HW, r);
  const float m = mu[(size_t)b * stride], t = tau[(size_t)b * stride];
  float2* gi = g + (size_t)b * istride + r;
  const float2 gx = gi[0], gz = gi[HW], gu = gi[2 * HW];
  const float2 e = make_float2(gz.x - gu.x, gz.y - gu.y);
  const float tm = t * m;
  const float2 G = Gs[i], q2 = q2s[i];
  c_tau[i] = -(e.x * G.x + e.y * G.y);
  c_mu[i] = -t * (e.x * q2.x + e.y * q2.y);
  gxr[i] = gx.x + gu.x + tm * e.x;
  gi[0] = make_float2(0.f, 0.f);
  gi[HW] = make_float2((1.f - tm) * e.x - t * J.x, (1.f - tm) * e.y - t * J.y);
  gi[2 * HW] = make_float2(gu.x + tm * e.x, gu.y + tm * e.y);
}
// ... and after the denoiser's VJP gd (d = Re(z - u)): gz.x += gd, gu.x -= gd
__global__ void pr_adjoint_finish_kernel(const float* __restrict__ gd, float2* g, size_t istride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  float2* gi = g + b * istride + r;
  gi[HW].x += gd[i];
  gi[2 * HW].x -= gd[i];
}
// out[b] = sum of the HW values of item b, fixed summation order (deterministic)
__global__ void __launch_bounds__(256) pr_item_sum_kernel(const float* __restrict__ c, float* __restrict__ out, int HW) {
  __shared__ float sh[256];
  const float* p = c + (size_t)blockIdx.x * HW;
  float a = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) a += p[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}
__global__ void real_of_diff_kernel(const float2* __restrict__ a, const float2* __restrict__ c, size_t istride,
                                    float* __restrict__ d, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  d[i] = subr(a[b * istride + r].x, c[b * istride + r].x);
}

// =================================================================================== SPI
// spi_inverse for one pixel                                                transforms.py:404-439
__device__ __forceinline__ float spi_inverse_px(float zt, float K1, float K, float mu) {
  const float K0 = subr(mulr(K, K), K1);
  float z;
  if (K1 == 0.f) {
    z = subr(zt, divr(K0, mu));
  } else {
    float bmin = 1e-5f, bmax = 1.1f;
    float bave = divr(addr(bmin, bmax), 2.0f);
    bool live = true;
#pragma unroll 1
    for (int it = 0; it < 10; ++it) {
      const float tmp = addr(subr(subr(divr(K1, subr(expf(bave), 1.f)), mulr(mu, bave)), K0), mulr(mu, zt));
      if (live) {
        if (tmp > 0.f) bmin = bave;
        else if (tmp < 0.f) bmax = bave;
        else if (tmp == 0.f) live = false;  // freeze-on-exact-zero branch (transforms.py:430-432)
        if (live) bave = divr(addr(bmin, bmax), 2.0f);
      }
    }
    z = bave;
  }
  return fminf(fmaxf(z, 0.f), 1.f);
}
__global__ void spi_inverse_kernel(const float* __restrict__ zt, const float* __restrict__ K1,
                                   const float* __restrict__ K, const float* __restrict__ mu, float* __restrict__ out,
                                   int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW;
  out[i] = spi_inverse_px(zt[i], K1[i], K[b], mu[b]);
}
// z = spi_inverse(x+u, K1, K, mu); u = u + x - z; d = z - u                tasks/spi/solver.py:41-47
// (x comes with its own item stride: the state's x slot on the first iteration, the denoiser's contiguous output afterwards -- r5: the
//  per-iteration copy of that output into the slot, 29 % of the prox time at config #5, is gone; one copy per call remains)
__global__ void spi_step_kernel(const float* xin, size_t xstride, const float* uin, size_t istride, const float* __restrict__ x0,
                                const float* __restrict__ Kmap, float* zout, float* uout, float* __restrict__ d,
                                const float* __restrict__ mu, int stride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float K = mulr(Kmap[b * HW], 10.f);
  const float K1 = mulr(x0[i], mulr(K, K));
  const float x = xin[b * xstride + r], u = uin[b * istride + r];
  const float z = spi_inverse_px(addr(x, u), K1, K, mu[b * stride]);
  const float un = subr(addr(u, x), z);
  zout[b * istride + r] = z;
  uout[b * istride + r] = un;
  d[i] = subr(z, un);
}
__global__ void copy_real_slot_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t istride, int HW,
                                      int B) {  // dst slot <- contiguous [B,HW]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  dst[b * istride + r] = src[i];
}

// ---- training paths of ADMMSolver_SPI / IADMMSolver_CT / PGSolver_CT (pnpx_{spi_admm,ct_iadmm,ct_pg}_train / _backward)
// spi_step_kernel that also keeps zt = x + u (the argument of spi_inverse)
__global__ void spi_step_save_kernel(const float* xin, size_t xstride, const float* uin, size_t istride, const float* __restrict__ x0,
                                     const float* __restrict__ Kmap, float* zout, float* uout, float* __restrict__ d,
                                     const float* __restrict__ mu, int stride, int HW, int B, float* __restrict__ zts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float K = mulr(Kmap[b * HW], 10.f);
  const float K1 = mulr(x0[i], mulr(K, K));
  const float x = xin[b * xstride + r], u = uin[b * istride + r];
  const float zt = addr(x, u);
  const float z = spi_inverse_px(zt, K1, K, mu[b * stride]);
  const float un = subr(addr(u, x), z);
  zout[b * istride + r] = z;
  uout[b * istride + r] = un;
  d[i] = subr(z, un);
  zts[i] = zt;
}
// one slot of a [B][3][HW] state -> contiguous [B][HW]
__global__ void slot_to_rows_kernel(const float* __restrict__ g, size_t istride, float* __restrict__ out, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  out[i] = g[b * istride + r];
}
// Backward of one SPI iteration after the denoiser's VJP gd (x' = D(z' - u')):
//   gz't = gz' + gd;  gu't = gu' - gd;  (u' = u + x - z')  gx = gu = gu't,  h = gz't - gu't;
//   z' = clamp(K1 == 0 ? zt - K0 / mu : bisection(zt), 0, 1): only the K1 == 0 branch carries a gradient (the bisection result is
//   built from constants in the reference, transforms.py:404-439), inside the clamp's closed interval:
//   gx += h, gu += h, d/d mu += h K0 / mu^2;  gz = 0 (an iteration never reads its z)
__global__ void spi_adjoint_kernel(float* g, size_t istride, const float* __restrict__ gd, const float* __restrict__ zts,
                                   const float* __restrict__ x0, const float* __restrict__ Kmap,
                                   const float* __restrict__ mu, int stride, int HW, int B, float* __restrict__ c_mu) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  float* gi = g + b * istride + r;
  const float gzt = gi[HW] + gd[i], gut = gi[2 * HW] - gd[i];
  const float h = gzt - gut;
  const float K = Kmap[b * HW] * 10.f, K1 = x0[i] * (K * K), m = mu[b * stride];
  const float K0 = K * K - K1;
  const float pre = zts[i] - K0 / m;
  const bool pass = (K1 == 0.f) && pre >= 0.f && pre <= 1.f;
  const float hz = pass ? h : 0.f;
  c_mu[i] = pass ? h * K0 / (m * m) : 0.f;
  gi[0] = gut + hz;
  gi[HW] = 0.f;
  gi[2 * HW] = gut + hz;
}
// ct_update_kernel of the training forward: also keeps G = g + mu (z - (x + u)) and q2 = z - (x + u)
__global__ void ct_update_save_kernel(const float* __restrict__ g, const float* __restrict__ xr, const float* zin,
                                      const float* uin, float* xout, float* zout, float* uout, size_t istride,
                                      float* __restrict__ d, const float* __restrict__ mu, const float* __restrict__ tau,
                                      int stride, int HW, int B, int write_x, float* __restrict__ Gs, float* __restrict__ q2s) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float m = mu[b * stride], t = tau[b * stride];
  const float z = zin[b * istride + r], u = uin[b * istride + r], xv = xr[i];
  const float q2 = subr(z, addr(xv, u));
  const float G = addr(g[i], mulr(m, q2));
  const float zn = subr(z, mulr(t, G));
  const float un = subr(addr(u, xv), zn);
  zout[b * istride + r] = zn;
  uout[b * istride + r] = un;
  d[i] = subr(zn, un);
  if (write_x) xout[b * istride + r] = xv;
  Gs[i] = G;
  q2s[i] = q2;
}
// e = gz' - gu' as a contiguous image batch (the input of the data step's adjoint A^T A / opnorm^2)
__global__ void ct_cotangent_kernel(const float* __restrict__ g, size_t istride, float* __restrict__ e, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  e[i] = g[b * istride + r + HW] - g[b * istride + r + 2 * HW];
}
// Backward of one CT iADMM iteration after J = A^T A e / opnorm^2 (the real-valued twin of pr_adjoint_kernel)
__global__ void ct_adjoint_kernel(const float* __restrict__ e, const float* __restrict__ J, float* g, size_t istride,
                                  const float* __restrict__ Gs, const float* __restrict__ q2s, const float* __restrict__ mu,
                                  const float* __restrict__ tau, int stride, int HW, int B, float* __restrict__ gxr,
                                  float* __restrict__ c_tau, float* __restrict__ c_mu) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float m = mu[b * stride], t = tau[b * stride], tm = t * m;
  float* gi = g + b * istride + r;
  const float ev = e[i], gu = gi[2 * HW];
  c_tau[i] = -(ev * Gs[i]);
  c_mu[i] = -t * (ev * q2s[i]);
  gxr[i] = gi[0] + gu + tm * ev;
  gi[0] = 0.f;
  gi[HW] = (1.f - tm) * ev - t * J[i];
  gi[2 * HW] = gu + tm * ev;
}
__global__ void ct_adjoint_finish_kernel(const float* __restrict__ gd, float* g, size_t istride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  g[b * istride + r + HW] += gd[i];
  g[b * istride + r + 2 * HW] -= gd[i];
}
// CT PG backward: gx = gd - tau J,  d/d tau = -<gd, g_i>
__global__ void ct_pg_adjoint_kernel(const float* __restrict__ gd, const float* __restrict__ J, const float* __restrict__ gsave,
                                     const float* __restrict__ tau, int stride, int HW, int B, float* __restrict__ gx,
                                     float* __restrict__ c_tau) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const float t = tau[(i / HW) * stride];
  c_tau[i] = -(gd[i] * gsave[i]);
  gx[i] = gd[i] - t * J[i];
}

// =================================================================================== PSNR
// partial sums of (clamp(o,0,1) - g)^2 per item                            tfpnp/env/base.py:237-242
constexpr int PSNR_CHUNKS = 32;
__global__ void psnr_partial_kernel(const float* __restrict__ o, const float* __restrict__ g, float* __restrict__ part,
                                    int n) {
  const int b = blockIdx.y, c = blockIdx.x;
  const int per = (n + PSNR_CHUNKS - 1) / PSNR_CHUNKS;
  const int lo = c * per, hi = min(n, lo + per);
  float acc = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float v = fminf(fmaxf(o[(size_t)b * n + i], 0.f), 1.f) - g[(size_t)b * n + i];
    acc = fmaf(v, v, acc);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);  // wavefront (64-lane) reduction
  __shared__ float w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[b * PSNR_CHUNKS + c] = (w[0] + w[1]) + (w[2] + w[3]);
}
__global__ void psnr_final_kernel(const float* __restrict__ part, float* __restrict__ psnr, int n, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < PSNR_CHUNKS; ++c) s += part[b * PSNR_CHUNKS + c];
  const float mse = s / (float)n;
  psnr[b] = 10.f * log10f(1.f / mse);
}

// =================================================================================== CT (own discretisation)
// Ray-driven forward projector.  See oracle/pnp_oracle.py:radon_forward and DESIGN.md for the geometry.  cs = [n_view]
// (cos, sin) pairs computed on the host in double precision.
// EIGHT lanes per ray (b, view, detector): lane dk of a ray takes the samples k0 + dk, k0 + dk + 8, ...; a wave is 8
// adjacent detector bins x 8 consecutive steps, i.e. one gather instruction touches an 8 x 8-pixel rotated patch (<= ~12
// image rows) instead of a 64-pixel line segment (up to 45 rows).  The image is read from two zero-bordered copies made per call
// (radon_pad_kernel: plain and transposed, RADON_PAD pixels of border): with the border standing in for "outside the image"
// a sample needs no validity masks or clamped addresses (an outside pixel contributes v * w = +0, exactly the oracle's skipped
// term); r5: the copies hold row pairs, so all four taps of a sample come from ONE 16-byte gather (r3 / r4: two 8-byte ones; r2: four
// dwords).  History at 32 x 256^2 x 30 views: one lane per ray 302 us (r2), eight lanes per ray 192 + 6 us (r3a),
// an LDS-staged 48 x 48 bounding box per 32 x 32-sample patch 269 us (the box of a rotated square is twice its area and its
// fill costs more instructions than it saves), two 8-byte gathers per sample from padded copies 99 us (r3b / r4), one 16-byte gather from
// row-pair copies 71 + 12 us, images dealt to XCDs by b % 8: 67 + 12 us (r5; the sample loop is bound by the texture-address path: half the
// vector instructions alone bought 7 %).  The eight partial sums are combined by a fixed
// xor-butterfly, so results are deterministic (summation order differs from the oracle's sequential one: ~1e-7).
constexpr int RADON_LPR = 8;   // lanes per ray
constexpr int RADON_PAD = 4;   // zero border of the padded copies: the k interval is computed loosely (widened by 2 steps)
__global__ void radon_forward_kernel(const float* __restrict__ imgP, const float* __restrict__ imgPT,
                                     const float* __restrict__ sub, float* __restrict__ sino,
                                     const float2* __restrict__ cs, int R, int V, int det, int B) {
  // XCD-affine image walk (r5): workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2.  Image b is projected by
  // workgroups of XCD b % 8 only, so its two padded copies (1.1 MiB at 256^2) are fetched into ONE L2 and stay there for all its views
  // (in ray order over the whole batch every XCD touched every image: 132 MB fetched per launch for 36 MB of copies at config #4).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bpi = (int)(((size_t)V * det * RADON_LPR + blockDim.x - 1) / blockDim.x);      // workgroups per image
  const int b = (slot / bpi) * 8 + xcd;
  if (b >= B) return;
  const size_t tid = (size_t)(slot % bpi) * blockDim.x + threadIdx.x;      // within the image
  const size_t n_rays = (size_t)V * det;
  const bool live = (tid / RADON_LPR) < n_rays;
  const size_t ir = live ? tid / RADON_LPR : n_rays - 1;      // dead tail lanes shadow the image's last ray (shuffles stay uniform)
  const size_t i = (size_t)b * n_rays + ir;
  const int dk = (int)(tid % RADON_LPR);
  const int s = (int)(ir % det);
  const int v = (int)(ir / det);
  const float c = cs[v].x, sn = cs[v].y;
  const float half = (float)det / 2.f - 0.5f, off = (float)R / 2.f - 0.5f;
  const float sp = subr((float)s, half);
  // adjacent lanes are adjacent bins: their samples lie 1 px apart along (c, sn).  Read the transposed copy when
  // that direction is closer to the y axis, so a wave's loads stay within few cache lines.
  const bool tr = fabsf(sn) > fabsf(c);
  const int RP = R + 2 * RADON_PAD;
  const float* im = (tr ? imgPT : imgP) + 2 * (size_t)b * RP * RP;      // this image's padded row-pair copy
  const float sc = mulr(sp, c), ss = mulr(sp, sn);
  // Only samples with px, py in [-1, R) touch the image.  Both coordinates are affine in k, so the contributing k
  // form one interval; it is computed loosely (widened by 2: still inside the zero border) and a coordinate that does not
  // move with k (an axis-parallel ray) is checked once.
  int k0 = 0, k1 = det;
  {
    auto clip = [&](float base, float slope) {   // base + (k - half) * slope in [-1, R)
      if (fabsf(slope) < 1e-6f) {
        if (!(base >= -1.f && base < (float)R)) k1 = 0;
        return;
      }
      const float rs = __builtin_amdgcn_rcpf(slope);      // (1 ulp: the interval is widened by two steps below)
      const float ka = (-1.f - base) * rs + half, kb = ((float)R - base) * rs + half;
      const float lo = fminf(ka, kb), hi = fmaxf(ka, kb);
      k0 = max(k0, (int)floorf(lo) - 2);
      k1 = min(k1, (int)ceilf(hi) + 3);
    };
    clip(sc + off, -sn);
    clip(ss + off, c);
  }
  // r5: ONE 16-byte gather per sample instead of two 8-byte ones.  The loop is bound by the texture-address path, not by its
  // instruction count (halving the vector instructions -- fma coordinates, v_cvt_flr / v_fract, lerps -- moved it 99 -> 92 us;
  // halving the gathers 92 -> 71 us at config #4): the padded copies hold ROW PAIRS, element (row, col) = (P[row][col], P[row + 1][col]),
  // so the four taps of a bilinear sample are one aligned-to-8 load.  (a, b) = (column, row) in the copy this ray reads -- plain:
  // (x, y), transposed: (y, x); both coordinates are computed with the oracle's own operation sequence (s c - t sn + off,
  // s sn + t c + off: same floor decisions as the oracle at the loud borders of the shape-sweep test), per-ray operands selected once.
  const float a1 = tr ? ss : sc, a2 = tr ? -c : sn;      // pa = (a1 - t a2) + off
  const float b1 = tr ? sc : ss, b2 = tr ? sn : -c;      // pb = (b1 - t b2) + off
  const float* imo = im + 2 * ((size_t)RADON_PAD * RP + RADON_PAD);
  float acc = 0.f;
  float t = subr((float)(k0 + dk), half);
  for (int k = k0 + dk; k < k1; k += RADON_LPR) {
    const float pa = addr(subr(a1, mulr(t, a2)), off), pb = addr(subr(b1, mulr(t, b2)), off);
    t += (float)RADON_LPR;                       // exact: integers + 0.5
    int a0, b0;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(a0) : "v"(pa));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(b0) : "v"(pb));
    const float fa = subr(pa, floorf(pa)), fb = subr(pb, floorf(pb));
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
    const f32x4u r = *reinterpret_cast<const f32x4u*>(imo + 2 * (ptrdiff_t)(__mul24(b0, RP) + a0));
    const float top = __builtin_fmaf(fa, r[2] - r[0], r[0]);
    const float bot = __builtin_fmaf(fa, r[3] - r[1], r[1]);
    acc += __builtin_fmaf(fb, bot - top, top);
  }
  acc = addr(acc, __shfl_xor(acc, 1, 64));
  acc = addr(acc, __shfl_xor(acc, 2, 64));
  acc = addr(acc, __shfl_xor(acc, 4, 64));
  if (live && dk == 0) sino[i] = sub ? subr(acc, sub[i]) : acc;
}
// [B][R][R] -> zero-bordered plain and transposed ROW-PAIR copies [B][R + 2 PAD][R + 2 PAD][2]: element (row, col) = (P[row][col],
// P[row + 1][col]) of the padded image P resp. its transpose (32 x 32 LDS tiles over the padded grid, one row / column of overlap).
__global__ __launch_bounds__(256) void radon_pad_kernel(const float* __restrict__ img, size_t istride,
                                                        float* __restrict__ outP, float* __restrict__ outPT, int R) {
  __shared__ float tile[33][34];
  const int RP = R + 2 * RADON_PAD;
  const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;   // padded coordinates
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* im = img + (size_t)b * istride;
  float2* oP = reinterpret_cast<float2*>(outP) + (size_t)b * RP * RP;
  float2* oT = reinterpret_cast<float2*>(outPT) + (size_t)b * RP * RP;
  for (int e = threadIdx.x; e < 33 * 33; e += 256) {      // 33 x 33: every element's lower / right neighbour too
    const int r = e / 33, c = e - r * 33;
    const int y = y0 + r - RADON_PAD, x = x0 + c - RADON_PAD;
    tile[r][c] = (y >= 0 && y < R && x >= 0 && x < R) ? im[(size_t)y * R + x] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    if (y0 + r < RP && x0 + tx < RP) oP[(size_t)(y0 + r) * RP + x0 + tx] = make_float2(tile[r][tx], tile[r + 1][tx]);
    if (x0 + r < RP && y0 + tx < RP) oT[(size_t)(x0 + r) * RP + y0 + tx] = make_float2(tile[tx][r], tile[tx][r + 1]);
  }
}
// padded row-pair copies (plain + transposed): 4 * B * (R + 2 PAD)^2 floats at `pad`
static size_t radon_pad_floats(int B, int R) { return 4 * (size_t)B * (R + 2 * RADON_PAD) * (R + 2 * RADON_PAD); }
static void launch_radon_forward(const float* img, size_t istride, const float* sub, float* sino, const float2* cs,
                                 float* pad, int R, int V, int det, int B, hipStream_t s) {
  const int RP = R + 2 * RADON_PAD;
  float* padT = pad + 2 * (size_t)B * RP * RP;
  hipLaunchKernelGGL(radon_pad_kernel, dim3((RP + 31) / 32, (RP + 31) / 32, B), dim3(256), 0, s, img, istride, pad, padT, R);
  const size_t bpi = ((size_t)V * det * RADON_LPR + 255) / 256;      // workgroups per image; images dealt to XCDs by b % 8
  hipLaunchKernelGGL(radon_forward_kernel, dim3((unsigned)(8 * (((size_t)B + 7) / 8) * bpi)), dim3(256), 0, s, pad, padT, sub, sino, cs,
                     R, V, det, B);
}
// Pixel-driven backprojection: one thread per pixel, linear interpolation along the detector.
__global__ void radon_backproject_kernel(const float* __restrict__ sino, float* __restrict__ img,
                                         const float2* __restrict__ cs, int R, int V, int det, int B, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * R * R) return;
  const int x = (int)(i % R), y = (int)((i / R) % R);
  const int b = (int)(i / ((size_t)R * R));
  const float off = (float)R / 2.f - 0.5f, half = (float)det / 2.f - 0.5f;
  const float xs = subr((float)x, off), ys = subr((float)y, off);
  float acc = 0.f;
  for (int v = 0; v < V; ++v) {
    const float sp = addr(addr(mulr(xs, cs[v].x), mulr(ys, cs[v].y)), half);
    const float f0 = floorf(sp);
    const int s0 = (int)f0;
    const float f = subr(sp, f0);
    const float* row = sino + ((size_t)b * V + v) * det;
    if (s0 >= 0 && s0 < det) acc = addr(acc, mulr(row[s0], subr(1.f, f)));
    if (s0 + 1 >= 0 && s0 + 1 < det) acc = addr(acc, mulr(row[s0 + 1], f));
  }
  img[i] = scale == 1.f ? acc : divr(acc, scale);
}
// The same backprojection with the sinogram staged in LDS (r4).  A workgroup = one 32 x 32-pixel tile of one image, four
// pixels per thread.  For a view the tile's pixels project into a window of at most 31 (|cos| + |sin|) + 2 < 48 detector bins:
// the window (64 bins, zeros outside the detector) of EVERY view is copied to LDS once -- V x 256 bytes -- and the per-view
// loop reads its two interpolation taps from there instead of issuing two dependent global gathers per sample (the r3 kernel:
// 55 us at B = 32, 256 x 256, 30 views, bound by the gather rate).  Same arithmetic in the same order per pixel; an
// out-of-range tap multiplies a stored zero instead of being skipped (acc + 0 = acc).
constexpr int RBP_T = 32, RBP_WIN = 64;
__global__ __launch_bounds__(256) void radon_backproject_lds_kernel(const float* __restrict__ sino, float* __restrict__ img,
                                                                    const float2* __restrict__ cs, int R, int V, int det, float scale) {
  extern __shared__ float rb_lds[];          // [V][RBP_WIN] taps, then [V] window origins (as int)
  int* base = reinterpret_cast<int*>(rb_lds + (size_t)V * RBP_WIN);
  const int b = blockIdx.z, x0 = blockIdx.x * RBP_T, y0 = blockIdx.y * RBP_T;
  const float off = (float)R / 2.f - 0.5f, half = (float)det / 2.f - 0.5f;
  for (int e = threadIdx.x; e < V * RBP_WIN; e += 256) {
    const int v = e / RBP_WIN, k = e - v * RBP_WIN;
    const float c = cs[v].x, sn = cs[v].y;
    const float xa = (float)x0 - off, xb = (float)(x0 + RBP_T - 1) - off, ya = (float)y0 - off, yb = (float)(y0 + RBP_T - 1) - off;
    const float lo = fminf(xa * c, xb * c) + fminf(ya * sn, yb * sn) + half;
    const int s_lo = (int)floorf(lo) - 2;      // two bins of slack against the rounding of this bound vs the per-pixel values
    const int sidx = s_lo + k;
    rb_lds[e] = (sidx >= 0 && sidx < det) ? sino[((size_t)b * V + v) * det + sidx] : 0.f;
    if (k == 0) base[v] = s_lo;
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x = x0 + tx;
  const float xs = subr((float)min(x, R - 1), off);     // (pixels past the image edge compute inside the window, are not stored)
  float ys[4], acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ys[j] = subr((float)min(y0 + ty + 8 * j, R - 1), off);
    acc[j] = 0.f;
  }
  for (int v = 0; v < V; ++v) {
    const float2 t = cs[v];
    const float* row = rb_lds + v * RBP_WIN - base[v];
    const float xc = mulr(xs, t.x);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sp = addr(addr(xc, mulr(ys[j], t.y)), half);
      const float f0 = floorf(sp);
      const int s0 = (int)f0;
      const float f = subr(sp, f0);
      acc[j] = addr(acc[j], mulr(row[s0], subr(1.f, f)));
      acc[j] = addr(acc[j], mulr(row[s0 + 1], f));
    }
  }
  if (x < R) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + ty + 8 * j;
      if (y < R) img[((size_t)b * R + y) * R + x] = scale == 1.f ? acc[j] : divr(acc[j], scale);
    }
  }
}
static void launch_radon_backproject(const float* sino, float* img, const float2* cs, int R, int V, int det, int B, float scale,
                                     hipStream_t s) {
  const size_t lds = (size_t)V * RBP_WIN * sizeof(float) + (size_t)V * sizeof(int);
  if (lds <= 60 * 1024 && B <= 65535) {
    hipLaunchKernelGGL(radon_backproject_lds_kernel, dim3((R + RBP_T - 1) / RBP_T, (R + RBP_T - 1) / RBP_T, B), dim3(256), lds, s, sino, img,
                       cs, R, V, det, scale);
  } else {   // very many views: the per-pixel gather kernel
    hipLaunchKernelGGL(radon_backproject_kernel, dim3((unsigned)(((size_t)B * R * R + 255) / 256)), dim3(256), 0, s, sino, img, cs, R, V,
                       det, B, scale);
  }
}
// z = z - tau*(g + mu*(z - (x+u))); u = u + x - z; d = z - u                tasks/ct/solver.py:46-49
__global__ void ct_update_kernel(const float* __restrict__ g, const float* __restrict__ xr, const float* zin,
                                 const float* uin, float* xout, float* zout, float* uout, size_t istride,
                                 float* __restrict__ d, const float* __restrict__ mu, const float* __restrict__ tau,
                                 int stride, int HW, int B, int write_x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float m = mu[b * stride], t = tau[b * stride];
  const float z = zin[b * istride + r], u = uin[b * istride + r], xv = xr[i];
  const float zn = subr(z, mulr(t, addr(g[i], mulr(m, subr(z, addr(xv, u))))));
  const float un = subr(addr(u, xv), zn);
  zout[b * istride + r] = zn;
  uout[b * istride + r] = un;
  d[i] = subr(zn, un);
  if (write_x) xout[b * istride + r] = xv;
}
// z = x - tau * g                                                          tasks/ct/solver.py:80
__global__ void ct_pg_step_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ d,
                                  const float* __restrict__ tau, int stride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  d[i] = subr(x[i], mulr(tau[(i / HW) * stride], g[i]));
}
__global__ void real_diff_slots_kernel(const float* __restrict__ a, const float* __restrict__ c, size_t istride,
                                       float* __restrict__ d, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  d[i] = subr(a[b * istride + r], c[b * istride + r]);
}

int radon_table(int R, int V, std::vector<float2>* cs, int* det) {
  *det = (int)std::ceil(std::sqrt(2.0) * R);
  cs->resize(V);
  const double stop = 179.0 / 180.0 * M_PI;
  for (int v = 0; v < V; ++v) {
    const double a = (V > 1) ? (double)v * (stop / (double)(V - 1)) : 0.0;   // numpy.linspace
    const float af = (float)a;                                               // .astype(float32)
    (*cs)[v] = make_float2((float)std::cos((double)af), (float)std::sin((double)af));
  }
  return PNPX_OK;
}

// scratch carve-up helper
struct Carver {
  char* p;
  template <class T>
  T* take(size_t n) {
    T* r = reinterpret_cast<T*>(p);
    p += (n * sizeof(T) + 255) & ~(size_t)255;
    return r;
  }
};

}  // namespace
}  // namespace pnpx

using namespace pnpx;

#define LOCK_CTX(ctx)                         \
  if (!(ctx)) {                               \
    pnpx::set_error("null context");          \
    return PNPX_ERR_ARG;                      \
  }                                           \
  std::lock_guard<std::mutex> _lk((ctx)->mu); \
  PNPX_HIP(hipSetDevice((ctx)->device))
#define REQUIRE(cond, msg)     \
  if (!(cond)) {               \
    pnpx::set_error(msg);      \
    return PNPX_ERR_ARG;       \
  }

extern "C" {

// ------------------------------------------------------------------------------------------- CDP
int pnpx_cdp_forward(pnpx_ctx* ctx, const float* x, const float* mask, float* out, int B, int S, int H, int W,
                     void* stream) {
  LOCK_CTX(ctx);
  REQUIRE(x && mask && out && B > 0 && S > 0, "pnpx_cdp_forward: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, B * S, H, W, false, &P));
  const int HW = H * W;
  LoadCdp ld{reinterpret_cast<const float2*>(x), (size_t)HW, reinterpret_cast<const float2*>(mask), S, W, HW};
  StoreC so{reinterpret_cast<float2*>(out), H, W};
  LoadC lo{reinterpret_cast<const float2*>(out), H, W};
  PNPX_TRY((launch_rows<false>(P, ld, so, s)));
  PNPX_TRY((launch_cols<false, false>(P, lo, MidNone(), so, s)));
  return PNPX_OK;
}

int pnpx_cdp_backward(pnpx_ctx* ctx, const float* y, const float* mask, float* out, int B, int S, int H, int W,
                      void* stream) {
  LOCK_CTX(ctx);
  REQUIRE(y && mask && out && B > 0 && S > 0, "pnpx_cdp_backward: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * S * H * W;
  void* p;
  PNPX_TRY(ctx_scratch(ctx, n * sizeof(float2) + 1024, &p));
  PNPX_TRY(fft2(ctx, y, static_cast<float*>(p), B * S, H, W, true, false, s));
  hipLaunchKernelGGL(cdp_adjoint_kernel, g1((size_t)B * H * W), dim3(256), 0, s, static_cast<const float2*>(p),
                     reinterpret_cast<const float2*>(mask), reinterpret_cast<float2*>(out), S, H * W, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

// IADMMSolver_PR.forward; `saved` != NULL (training path): per iteration the denoiser input d_i [T][n], w = F(mask_s z) before the
// residual [T][S n][2], G = g + mu (z - (x + u)) [T][n][2], q2 = z - (x + u) [T][n][2]  (n = B*H*W); activations parked
// (ticket + i).
static int pr_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const float* mask,
                      const float* sigma_d, const float* mu, const float* tau, int param_stride, int B, int S, int H, int W,
                      int T, float* saved, unsigned long long* ticket_out, hipStream_t s) {
  REQUIRE(vars_in && vars_out && y0 && mask && sigma_d && mu && tau && B > 0 && S > 0 && T >= 0 && param_stride >= T,
          "pnpx_pr_iadmm: bad argument");
  if (ticket_out) *ticket_out = 0;
  const int HW = H * W;
  const size_t is = 3 * (size_t)HW, n = (size_t)HW * B;
  const float2* vin = reinterpret_cast<const float2*>(vars_in);
  float2* vout = reinterpret_cast<float2*>(vars_out);
  if (T == 0) {
    PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    return PNPX_OK;
  }
  void* p;
  PNPX_TRY(ctx_scratch(ctx, n * S * sizeof(float2) + 2 * n * sizeof(float) + 4096, &p));
  Carver cv{static_cast<char*>(p)};
  float2* k = cv.take<float2>(n * S);
  float* d = cv.take<float>(n);
  float* xr = cv.take<float>(n);
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, B * S, H, W, false, &P));
  StoreC kst{k, H, W};
  LoadC kld{k, H, W};
  float* sv_d = saved;
  float2* sv_w = saved ? reinterpret_cast<float2*>(saved + (size_t)T * n) : nullptr;
  float2* sv_G = saved ? reinterpret_cast<float2*>(saved + (size_t)T * n * (1 + 2 * S)) : nullptr;
  float2* sv_q = saved ? reinterpret_cast<float2*>(saved + (size_t)T * n * (3 + 2 * S)) : nullptr;
  hipLaunchKernelGGL(real_of_diff_kernel, g1(n), dim3(256), 0, s, vin + HW, vin + 2 * HW, is, d, HW, B);
  PNPX_LAUNCH_CHECK();
  for (int i = 0; i < T; ++i) {
    if (saved) {
      PNPX_HIP(hipMemcpyAsync(sv_d + (size_t)i * n, d, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, d, sigma_d + i, param_stride, xr, B, H, W, s, &tk));
      if (i == 0 && ticket_out) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, d, sigma_d + i, param_stride, xr, nullptr, B, H, W, s, nullptr));
    }
    const float2* zi = (i == 0 ? vin : vout) + HW;
    const float2* ui = (i == 0 ? vin : vout) + 2 * HW;
    LoadCdp ld{zi, is, reinterpret_cast<const float2*>(mask), S, W, HW};
    PNPX_TRY((launch_rows<false>(P, ld, kst, s)));
    if (saved) {
      PNPX_TRY((launch_cols<false, true>(P, kld, MidPrResidualSave{y0, sv_w + (size_t)i * n * S, W, HW}, kst, s)));
    } else {
      PNPX_TRY((launch_cols<false, true>(P, kld, MidPrResidual{y0, W, HW}, kst, s)));
    }
    if (!saved && P.fast256_rows) {
      // inverse row pass + coded-diffraction adjoint + z / u update in one kernel (the S fields stay in registers)
      PNPX_TRY((launch_rows_group<true>(P, B, S, kld, PrUpdateAcc{reinterpret_cast<const float2*>(mask), xr, zi, ui, vout, vout + HW, vout + 2 * HW,
                                                                   is, d, mu + i, tau + i, param_stride, S, W, HW, i == T - 1}, s)));
      continue;
    }
    PNPX_TRY((launch_rows<true>(P, kld, kst, s)));
    if (saved) {
      hipLaunchKernelGGL(pr_update_save_kernel, g1(n), dim3(256), 0, s, k, reinterpret_cast<const float2*>(mask), xr, zi, ui,
                         vout, vout + HW, vout + 2 * HW, is, d, mu + i, tau + i, param_stride, S, HW, B, i == T - 1,
                         sv_G + (size_t)i * n, sv_q + (size_t)i * n);
    } else {
      hipLaunchKernelGGL(pr_update_kernel, g1(n), dim3(256), 0, s, k, reinterpret_cast<const float2*>(mask), xr, zi, ui,
                         vout, vout + HW, vout + 2 * HW, is, d, mu + i, tau + i, param_stride, S, HW, B, i == T - 1);
    }
    PNPX_LAUNCH_CHECK();
  }
  return PNPX_OK;
}

int pnpx_pr_iadmm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const float* mask,
                  const float* sigma_d, const float* mu, const float* tau, int param_stride, int B, int S, int H, int W,
                  int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return pr_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, tau, param_stride, B, S, H, W, T, nullptr, nullptr, s);
  });
}

int pnpx_pr_iadmm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const float* mask,
                        const float* sigma_d, const float* mu, const float* tau, int param_stride, int B, int S, int H,
                        int W, int T, float* saved, unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE((saved || T == 0) && ticket, "pnpx_pr_iadmm_train: saved / ticket is null");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return pr_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, tau, param_stride, B, S, H, W, T, saved, ticket, s);
  });
}

// VJP of the T-iteration PR map wrt (cat(x, z, u), sigma_d, mu, tau), iterations walked in reverse:
//   forward i:   x' = r2c(D(Re(z - u), sigma_i));  w_s = F(mask_s z);  g = mean_s conj(mask_s) F^-1((|w_s| - y0_s) / |w_s| w_s);
//                z' = z - tau (g + mu (z - (x' + u)));  u' = u + x' - z'
//   backward i:  e = gz' - gu';  I'_s = F^-1 J_s^T F(mask_s e) with the saved w_s;  pr_adjoint_kernel (above);  denoiser VJP;
//                pr_adjoint_finish_kernel
int pnpx_pr_iadmm_backward(pnpx_ctx* ctx, const float* y0, const float* mask, const float* sigma_d, const float* mu,
                           const float* tau, int param_stride, const float* saved, const float* grad_vars_out,
                           float* grad_vars_in, float* grad_sigma_d, float* grad_mu, float* grad_tau, float* work, int B,
                           int S, int H, int W, int T, unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    REQUIRE(y0 && mask && sigma_d && mu && tau && grad_vars_out && grad_vars_in && B > 0 && S > 0 && T >= 0 &&
                param_stride >= T && (T == 0 || (saved && grad_sigma_d && grad_mu && grad_tau && work)),
            "pnpx_pr_iadmm_backward: bad argument");
    const int HW = H * W;
    const size_t is = 3 * (size_t)HW, n = (size_t)HW * B;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    if (T == 0) return PNPX_OK;
    FftPlan2D P;
    PNPX_TRY(make_fft_plan(ctx, B * S, H, W, false, &P));
    float2* g = reinterpret_cast<float2*>(grad_vars_in);
    const float2* mk = reinterpret_cast<const float2*>(mask);
    float *gxr = work, *gd = work + n, *c_tau = work + 2 * n, *c_mu = work + 3 * n;
    const float* sv_d = saved;
    const float2* sv_w = reinterpret_cast<const float2*>(saved + (size_t)T * n);
    const float2* sv_G = reinterpret_cast<const float2*>(saved + (size_t)T * n * (1 + 2 * S));
    const float2* sv_q = reinterpret_cast<const float2*>(saved + (size_t)T * n * (3 + 2 * S));
    for (int i = T - 1; i >= 0; --i) {
      // the context's scratch is shared with the denoiser's VJP below, which may also GROW (= re-allocate) it: the k-space
      // buffer is carved anew every iteration and nothing in it has to survive that call
      void* p;
      PNPX_TRY(ctx_scratch(ctx, n * S * sizeof(float2) + 4096, &p));
      float2* k = static_cast<float2*>(p);
      StoreC kst{k, H, W};
      LoadC kld{k, H, W};
      PNPX_TRY((launch_rows<false>(P, LoadCdpDiff{g + HW, g + 2 * HW, is, mk, S, W, HW}, kst, s)));
      PNPX_TRY((launch_cols<false, true>(P, kld, MidPrResidualAdjoint{y0, sv_w + (size_t)i * n * S, W, HW}, kst, s)));
      PNPX_TRY((launch_rows<true>(P, kld, kst, s)));
      hipLaunchKernelGGL(pr_adjoint_kernel, g1(n), dim3(256), 0, s, k, mk, g, is, sv_G + (size_t)i * n, sv_q + (size_t)i * n,
                         mu + i, tau + i, param_stride, S, HW, B, gxr, c_tau, c_mu);
      PNPX_LAUNCH_CHECK();
      hipLaunchKernelGGL(pr_item_sum_kernel, dim3(B), dim3(256), 0, s, c_tau, grad_tau + (size_t)i * B, HW);
      hipLaunchKernelGGL(pr_item_sum_kernel, dim3(B), dim3(256), 0, s, c_mu, grad_mu + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(unet_denoise_backward_ticket(ctx, sv_d + (size_t)i * n, sigma_d + i, param_stride, gxr, gd,
                                            grad_sigma_d + (size_t)i * B, B, H, W, s, ticket ? ticket + i : 0));
      hipLaunchKernelGGL(pr_adjoint_finish_kernel, g1(n), dim3(256), 0, s, gd, g, is, HW, B);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

// ------------------------------------------------------------------------------------------- SPI
int pnpx_spi_inverse(pnpx_ctx* ctx, const float* ztilde, const float* K1, const float* K, const float* mu, float* out,
                     int B, int H, int W, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE(ztilde && K1 && K && mu && out && B > 0 && H > 0 && W > 0, "pnpx_spi_inverse: bad argument");
  hipLaunchKernelGGL(spi_inverse_kernel, g1((size_t)B * H * W), dim3(256), 0, static_cast<hipStream_t>(stream), ztilde,
                     K1, K, mu, out, H * W, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

// ADMMSolver_SPI.forward; `saved` != NULL (training path): per iteration zt = x + u [T][n] and the denoiser input [T][n]
static int spi_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* x0, const float* Kmap,
                       const float* sigma_d, const float* mu, int param_stride, int B, int H, int W, int T, float* saved,
                       unsigned long long* ticket_out, hipStream_t s) {
  REQUIRE(vars_in && vars_out && x0 && Kmap && sigma_d && mu && B > 0 && T >= 0 && param_stride >= T,
          "pnpx_spi_admm: bad argument");
  if (ticket_out) *ticket_out = 0;
  const int HW = H * W;
  const size_t is = 3 * (size_t)HW, n = (size_t)HW * B;
  if (T == 0) {
    PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float) * is * B, hipMemcpyDeviceToDevice, s));
    return PNPX_OK;
  }
  void* p;
  PNPX_TRY(ctx_scratch(ctx, 2 * n * sizeof(float) + 4096, &p));
  Carver cv{static_cast<char*>(p)};
  float* dscr = cv.take<float>(n);
  float* xr = cv.take<float>(n);
  for (int i = 0; i < T; ++i) {
    // x of iteration i: the state's x slot on the first pass, the previous denoiser output (contiguous, in xr) later
    const float* xin = (i == 0) ? vars_in : xr;
    const size_t xs = (i == 0) ? is : (size_t)HW;
    const float* uin = ((i == 0) ? vars_in : vars_out) + 2 * HW;
    if (saved) {
      float* d = saved + ((size_t)T + i) * n;
      hipLaunchKernelGGL(spi_step_save_kernel, g1(n), dim3(256), 0, s, xin, xs, uin, is, x0, Kmap, vars_out + HW,
                         vars_out + 2 * HW, d, mu + i, param_stride, HW, B, saved + (size_t)i * n);
      PNPX_LAUNCH_CHECK();
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, d, sigma_d + i, param_stride, xr, B, H, W, s, &tk));
      if (i == 0 && ticket_out) *ticket_out = tk;
    } else {
      hipLaunchKernelGGL(spi_step_kernel, g1(n), dim3(256), 0, s, xin, xs, uin, is, x0, Kmap, vars_out + HW,
                         vars_out + 2 * HW, dscr, mu + i, param_stride, HW, B);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(unet_denoise(ctx, dscr, sigma_d + i, param_stride, xr, nullptr, B, H, W, s, nullptr));
    }
  }
  // the last denoiser output becomes the state's x (the only copy of the call)
  hipLaunchKernelGGL(copy_real_slot_kernel, g1(n), dim3(256), 0, s, xr, vars_out, is, HW, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

int pnpx_spi_admm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* x0, const float* Kmap,
                  const float* sigma_d, const float* mu, int param_stride, int B, int H, int W, int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return spi_forward(ctx, vars_in, vars_out, x0, Kmap, sigma_d, mu, param_stride, B, H, W, T, nullptr, nullptr, s);
  });
}

int pnpx_spi_admm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* x0, const float* Kmap,
                        const float* sigma_d, const float* mu, int param_stride, int B, int H, int W, int T, float* saved,
                        unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE((saved || T == 0) && ticket, "pnpx_spi_admm_train: saved / ticket is null");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return spi_forward(ctx, vars_in, vars_out, x0, Kmap, sigma_d, mu, param_stride, B, H, W, T, saved, ticket, s);
  });
}

// VJP of the T-iteration SPI map wrt (cat(x, z, u), sigma_d, mu): per iteration (reverse order) the denoiser VJP of the x slot's
// cotangent, then spi_adjoint_kernel.  work = 3*B*H*W floats.
int pnpx_spi_admm_backward(pnpx_ctx* ctx, const float* x0, const float* Kmap, const float* sigma_d, const float* mu,
                           int param_stride, const float* saved, const float* grad_vars_out, float* grad_vars_in,
                           float* grad_sigma_d, float* grad_mu, float* work, int B, int H, int W, int T,
                           unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    REQUIRE(x0 && Kmap && sigma_d && mu && grad_vars_out && grad_vars_in && B > 0 && T >= 0 && param_stride >= T &&
                (T == 0 || (saved && grad_sigma_d && grad_mu && work)),
            "pnpx_spi_admm_backward: bad argument");
    const int HW = H * W;
    const size_t is = 3 * (size_t)HW, n = (size_t)HW * B;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float) * is * B, hipMemcpyDeviceToDevice, s));
    float *gxr = work, *gd = work + n, *c_mu = work + 2 * n;
    for (int i = T - 1; i >= 0; --i) {
      hipLaunchKernelGGL(slot_to_rows_kernel, g1(n), dim3(256), 0, s, grad_vars_in, is, gxr, HW, B);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved + ((size_t)T + i) * n, sigma_d + i, param_stride, gxr, gd,
                                            grad_sigma_d + (size_t)i * B, B, H, W, s, ticket ? ticket + i : 0));
      hipLaunchKernelGGL(spi_adjoint_kernel, g1(n), dim3(256), 0, s, grad_vars_in, is, gd, saved + (size_t)i * n, x0, Kmap,
                         mu + i, param_stride, HW, B, c_mu);
      PNPX_LAUNCH_CHECK();
      hipLaunchKernelGGL(pr_item_sum_kernel, dim3(B), dim3(256), 0, s, c_mu, grad_mu + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

// ------------------------------------------------------------------------------------------- PSNR
int pnpx_psnr(pnpx_ctx* ctx, const float* output, const float* gt, float* psnr, int B, int n_per_item, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE(output && gt && psnr && B > 0 && n_per_item > 0, "pnpx_psnr: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  void* p;
  PNPX_TRY(ctx_scratch(ctx, (size_t)B * PSNR_CHUNKS * sizeof(float) + 1024, &p));
  // NOTE: uses the tail of nothing else -- solver loops never run concurrently with psnr on one ctx (mutex).
  hipLaunchKernelGGL(psnr_partial_kernel, dim3(PSNR_CHUNKS, B), dim3(256), 0, s, output, gt, static_cast<float*>(p),
                     n_per_item);
  PNPX_LAUNCH_CHECK();
  hipLaunchKernelGGL(psnr_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, static_cast<const float*>(p), psnr,
                     n_per_item, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

// ------------------------------------------------------------------------------------------- CT
int pnpx_radon_det_count(int R) { return (int)std::ceil(std::sqrt(2.0) * R); }

static int upload_cs(pnpx_ctx* ctx, int R, int V, hipStream_t s, Carver* cv, const float2** cs_dev, int* det) {
  std::vector<float2> cs;
  PNPX_TRY(radon_table(R, V, &cs, det));
  float2* d = cv->take<float2>(V);
  // small synchronous upload (host table is a local): hipMemcpy waits, so the vector may die afterwards
  PNPX_HIP(hipMemcpyAsync(d, cs.data(), sizeof(float2) * V, hipMemcpyHostToDevice, s));
  PNPX_HIP(hipStreamSynchronize(s));
  (void)ctx;
  *cs_dev = d;
  return PNPX_OK;
}

int pnpx_radon_forward(pnpx_ctx* ctx, const float* img, float* sino, int B, int R, int n_view, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE(img && sino && B > 0 && R > 0 && n_view > 0, "pnpx_radon_forward: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  void* p;
  PNPX_TRY(ctx_scratch(ctx, sizeof(float2) * n_view + radon_pad_floats(B, R) * sizeof(float) + 8192, &p));
  Carver cv{static_cast<char*>(p)};
  const float2* cs;
  int det;
  PNPX_TRY(upload_cs(ctx, R, n_view, s, &cv, &cs, &det));
  float* pad = cv.take<float>(radon_pad_floats(B, R));
  launch_radon_forward(img, (size_t)R * R, nullptr, sino, cs, pad, R, n_view, det, B, s);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

int pnpx_radon_backprojection(pnpx_ctx* ctx, const float* sino, float* img, int B, int R, int n_view, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE(img && sino && B > 0 && R > 0 && n_view > 0, "pnpx_radon_backprojection: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  void* p;
  PNPX_TRY(ctx_scratch(ctx, sizeof(float2) * n_view + 4096, &p));
  Carver cv{static_cast<char*>(p)};
  const float2* cs;
  int det;
  PNPX_TRY(upload_cs(ctx, R, n_view, s, &cv, &cs, &det));
  launch_radon_backproject(sino, img, cs, R, n_view, det, B, 1.f, s);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

// scratch shared by the CT loops: (cos, sin) table, sinogram, the projector's padded copies and `extra` image-sized buffers
struct CtScratch {
  const float2* cs;
  float *sino, *pad, *img[4];
  int det;
};
// `cs_home`: where the (cos, sin) table lives.  NULL = in the scratch itself (forward loops: nothing else touches the scratch
// while they run); the backward loops pass a place in their caller-provided work buffer, because the denoiser's VJP uses --
// and may re-allocate -- the same scratch between two uses of the table, and re-carve the rest every iteration (upload = false).
static int ct_scratch(pnpx_ctx* ctx, int B, int R, int n_view, int extra, hipStream_t s, CtScratch* C,
                      float2* cs_home = nullptr, bool upload = true) {
  const size_t n = (size_t)R * R * B;
  const int det = pnpx_radon_det_count(R);
  void* p;
  PNPX_TRY(ctx_scratch(ctx, (extra * n + radon_pad_floats(B, R) + (size_t)B * n_view * det) * sizeof(float) +
                                sizeof(float2) * n_view + 16384, &p));
  Carver cv{static_cast<char*>(p)};
  float2* table = cv.take<float2>(n_view);
  if (cs_home) table = cs_home;
  if (upload) {
    std::vector<float2> cs;
    int det2;
    PNPX_TRY(radon_table(R, n_view, &cs, &det2));
    // small synchronous upload (host table is a local): the vector may die after the synchronisation
    PNPX_HIP(hipMemcpyAsync(table, cs.data(), sizeof(float2) * n_view, hipMemcpyHostToDevice, s));
    PNPX_HIP(hipStreamSynchronize(s));
  }
  C->cs = table;
  for (int k = 0; k < extra; ++k) C->img[k] = cv.take<float>(n);
  C->sino = cv.take<float>((size_t)B * n_view * det);
  C->pad = cv.take<float>(radon_pad_floats(B, R));
  C->det = det;
  return PNPX_OK;
}
// out = A^T (A img - sub) / op2 for a contiguous image batch
static int ct_normal_op(const CtScratch& C, const float* img, size_t istride, const float* sub, float* out, float op2, int R,
                        int n_view, int B, hipStream_t s) {
  launch_radon_forward(img, istride, sub, C.sino, C.cs, C.pad, R, n_view, C.det, B, s);
  PNPX_LAUNCH_CHECK();
  launch_radon_backproject(C.sino, out, C.cs, R, n_view, C.det, B, op2, s);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

// IADMMSolver_CT.forward; `saved` != NULL (training path): per iteration the denoiser input [T][n], G = g + mu (z - (x + u))
// [T][n] and q2 = z - (x + u) [T][n]
static int ct_iadmm_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view, float opnorm,
                            const float* sigma_d, const float* mu, const float* tau, int param_stride, int B, int R, int T,
                            float* saved, unsigned long long* ticket_out, hipStream_t s) {
  REQUIRE(vars_in && vars_out && y0 && sigma_d && mu && tau && B > 0 && R > 0 && n_view > 0 && T >= 0 &&
              param_stride >= T && opnorm > 0.f,
          "pnpx_ct_iadmm: bad argument");
  if (ticket_out) *ticket_out = 0;
  const int HW = R * R;
  const size_t is = 3 * (size_t)HW, n = (size_t)HW * B;
  if (T == 0) {
    PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float) * is * B, hipMemcpyDeviceToDevice, s));
    return PNPX_OK;
  }
  CtScratch C;
  PNPX_TRY(ct_scratch(ctx, B, R, n_view, 3, s, &C));
  float *d = C.img[0], *xr = C.img[1], *g = C.img[2];
  const float op2 = (float)((double)opnorm * (double)opnorm);  // backprojection / opnorm**2   transforms.py:476-477
  hipLaunchKernelGGL(real_diff_slots_kernel, g1(n), dim3(256), 0, s, vars_in + HW, vars_in + 2 * HW, is, d, HW, B);
  PNPX_LAUNCH_CHECK();
  for (int i = 0; i < T; ++i) {
    if (saved) {
      PNPX_HIP(hipMemcpyAsync(saved + (size_t)i * n, d, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, d, sigma_d + i, param_stride, xr, B, R, R, s, &tk));
      if (i == 0 && ticket_out) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, d, sigma_d + i, param_stride, xr, nullptr, B, R, R, s, nullptr));
    }
    const float* zi = ((i == 0) ? vars_in : vars_out) + HW;
    const float* ui = ((i == 0) ? vars_in : vars_out) + 2 * HW;
    PNPX_TRY(ct_normal_op(C, zi, is, y0, g, op2, R, n_view, B, s));
    if (saved) {
      hipLaunchKernelGGL(ct_update_save_kernel, g1(n), dim3(256), 0, s, g, xr, zi, ui, vars_out, vars_out + HW,
                         vars_out + 2 * HW, is, d, mu + i, tau + i, param_stride, HW, B, i == T - 1,
                         saved + ((size_t)T + i) * n, saved + (2 * (size_t)T + i) * n);
    } else {
      hipLaunchKernelGGL(ct_update_kernel, g1(n), dim3(256), 0, s, g, xr, zi, ui, vars_out, vars_out + HW,
                         vars_out + 2 * HW, is, d, mu + i, tau + i, param_stride, HW, B, i == T - 1);
    }
    PNPX_LAUNCH_CHECK();
  }
  return PNPX_OK;
}

int pnpx_ct_iadmm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view, float opnorm,
                  const float* sigma_d, const float* mu, const float* tau, int param_stride, int B, int R, int T,
                  void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return ct_iadmm_forward(ctx, vars_in, vars_out, y0, n_view, opnorm, sigma_d, mu, tau, param_stride, B, R, T, nullptr,
                            nullptr, s);
  });
}

int pnpx_ct_iadmm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view, float opnorm,
                        const float* sigma_d, const float* mu, const float* tau, int param_stride, int B, int R, int T,
                        float* saved, unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE((saved || T == 0) && ticket, "pnpx_ct_iadmm_train: saved / ticket is null");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return ct_iadmm_forward(ctx, vars_in, vars_out, y0, n_view, opnorm, sigma_d, mu, tau, param_stride, B, R, T, saved,
                            ticket, s);
  });
}

// VJP of the T-iteration CT iADMM map wrt (cat(x, z, u), sigma_d, mu, tau); the data step's adjoint is A^T A / opnorm^2 with the
// projector pair standing in for each other's transpose (as in the composed path and in torch_radon).  work = 6*B*R*R + 2*n_view floats.
int pnpx_ct_iadmm_backward(pnpx_ctx* ctx, int n_view, float opnorm, const float* sigma_d, const float* mu, const float* tau,
                           int param_stride, const float* saved, const float* grad_vars_out, float* grad_vars_in,
                           float* grad_sigma_d, float* grad_mu, float* grad_tau, float* work, int B, int R, int T,
                           unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    REQUIRE(sigma_d && mu && tau && grad_vars_out && grad_vars_in && B > 0 && R > 0 && n_view > 0 && T >= 0 &&
                param_stride >= T && opnorm > 0.f && (T == 0 || (saved && grad_sigma_d && grad_mu && grad_tau && work)),
            "pnpx_ct_iadmm_backward: bad argument");
    const int HW = R * R;
    const size_t is = 3 * (size_t)HW, n = (size_t)HW * B;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float) * is * B, hipMemcpyDeviceToDevice, s));
    if (T == 0) return PNPX_OK;
    CtScratch C;
    float2* cs_home = reinterpret_cast<float2*>(work + 6 * n);       // the table outlives the VJP calls in the work buffer
    PNPX_TRY(ct_scratch(ctx, B, R, n_view, 0, s, &C, cs_home, true));
    const float op2 = (float)((double)opnorm * (double)opnorm);
    float *e = work, *J = work + n, *gxr = work + 2 * n, *gd = work + 3 * n, *c_tau = work + 4 * n, *c_mu = work + 5 * n;
    for (int i = T - 1; i >= 0; --i) {
      PNPX_TRY(ct_scratch(ctx, B, R, n_view, 0, s, &C, cs_home, false));   // re-carve: the VJP below shares (and may grow) the scratch
      hipLaunchKernelGGL(ct_cotangent_kernel, g1(n), dim3(256), 0, s, grad_vars_in, is, e, HW, B);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(ct_normal_op(C, e, (size_t)HW, nullptr, J, op2, R, n_view, B, s));
      hipLaunchKernelGGL(ct_adjoint_kernel, g1(n), dim3(256), 0, s, e, J, grad_vars_in, is, saved + ((size_t)T + i) * n,
                         saved + (2 * (size_t)T + i) * n, mu + i, tau + i, param_stride, HW, B, gxr, c_tau, c_mu);
      PNPX_LAUNCH_CHECK();
      hipLaunchKernelGGL(pr_item_sum_kernel, dim3(B), dim3(256), 0, s, c_tau, grad_tau + (size_t)i * B, HW);
      hipLaunchKernelGGL(pr_item_sum_kernel, dim3(B), dim3(256), 0, s, c_mu, grad_mu + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved + (size_t)i * n, sigma_d + i, param_stride, gxr, gd,
                                            grad_sigma_d + (size_t)i * B, B, R, R, s, ticket ? ticket + i : 0));
      hipLaunchKernelGGL(ct_adjoint_finish_kernel, g1(n), dim3(256), 0, s, gd, grad_vars_in, is, HW, B);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

// PGSolver_CT.forward; `saved` != NULL (training path): per iteration the denoiser input [T][n] and g = A^T(A x - y0) / opnorm^2 [T][n]
static int ct_pg_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view, float opnorm,
                         const float* sigma_d, const float* tau, int param_stride, int B, int R, int T, float* saved,
                         unsigned long long* ticket_out, hipStream_t s) {
  REQUIRE(vars_in && vars_out && y0 && sigma_d && tau && B > 0 && R > 0 && n_view > 0 && T >= 0 &&
              param_stride >= T && opnorm > 0.f,
          "pnpx_ct_pg: bad argument");
  if (ticket_out) *ticket_out = 0;
  const int HW = R * R;
  const size_t n = (size_t)HW * B;
  if (T == 0) {
    PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    return PNPX_OK;
  }
  CtScratch C;
  PNPX_TRY(ct_scratch(ctx, B, R, n_view, 2, s, &C));
  const float op2 = (float)((double)opnorm * (double)opnorm);
  for (int i = 0; i < T; ++i) {
    const float* xi = (i == 0) ? vars_in : vars_out;
    float* d = saved ? saved + (size_t)i * n : C.img[0];
    float* g = saved ? saved + ((size_t)T + i) * n : C.img[1];
    PNPX_TRY(ct_normal_op(C, xi, (size_t)HW, y0, g, op2, R, n_view, B, s));
    hipLaunchKernelGGL(ct_pg_step_kernel, g1(n), dim3(256), 0, s, g, xi, d, tau + i, param_stride, HW, B);
    PNPX_LAUNCH_CHECK();
    if (saved) {
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, d, sigma_d + i, param_stride, vars_out, B, R, R, s, &tk));
      if (i == 0 && ticket_out) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, d, sigma_d + i, param_stride, vars_out, nullptr, B, R, R, s, nullptr));
    }
  }
  return PNPX_OK;
}

int pnpx_ct_pg(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view, float opnorm,
               const float* sigma_d, const float* tau, int param_stride, int B, int R, int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return ct_pg_forward(ctx, vars_in, vars_out, y0, n_view, opnorm, sigma_d, tau, param_stride, B, R, T, nullptr, nullptr, s);
  });
}

int pnpx_ct_pg_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view, float opnorm,
                     const float* sigma_d, const float* tau, int param_stride, int B, int R, int T, float* saved,
                     unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  REQUIRE((saved || T == 0) && ticket, "pnpx_ct_pg_train: saved / ticket is null");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return ct_pg_forward(ctx, vars_in, vars_out, y0, n_view, opnorm, sigma_d, tau, param_stride, B, R, T, saved, ticket, s);
  });
}

// VJP of the T-iteration CT PG map x' = D(x - tau g(x)) wrt (x, sigma_d, tau): gd = D^T gx';  d/d tau = -<gd, g_i>;
// gx = gd - tau A^T A gd / opnorm^2.  work = 3*B*R*R + 2*n_view floats.
int pnpx_ct_pg_backward(pnpx_ctx* ctx, int n_view, float opnorm, const float* sigma_d, const float* tau, int param_stride,
                        const float* saved, const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d,
                        float* grad_tau, float* work, int B, int R, int T, unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    REQUIRE(sigma_d && tau && grad_vars_out && grad_vars_in && B > 0 && R > 0 && n_view > 0 && T >= 0 &&
                param_stride >= T && opnorm > 0.f && (T == 0 || (saved && grad_sigma_d && grad_tau && work)),
            "pnpx_ct_pg_backward: bad argument");
    const int HW = R * R;
    const size_t n = (size_t)HW * B;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    if (T == 0) return PNPX_OK;
    CtScratch C;
    // (cos, sin) table behind the three n-float buffers, at an EVEN float offset: 3 * n is odd for odd B * R * R and a float2
    // array needs 8-byte alignment (work holds 3 * n + 1 + 2 * n_view floats)
    float2* cs_home = reinterpret_cast<float2*>(work + ((3 * n + 1) & ~(size_t)1));
    PNPX_TRY(ct_scratch(ctx, B, R, n_view, 0, s, &C, cs_home, true));
    const float op2 = (float)((double)opnorm * (double)opnorm);
    float *gd = work, *J = work + n, *c_tau = work + 2 * n;
    for (int i = T - 1; i >= 0; --i) {
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved + (size_t)i * n, sigma_d + i, param_stride, grad_vars_in, gd,
                                            grad_sigma_d + (size_t)i * B, B, R, R, s, ticket ? ticket + i : 0));
      PNPX_TRY(ct_scratch(ctx, B, R, n_view, 0, s, &C, cs_home, false));   // re-carve after the VJP (shared, growable scratch)
      PNPX_TRY(ct_normal_op(C, gd, (size_t)HW, nullptr, J, op2, R, n_view, B, s));
      hipLaunchKernelGGL(ct_pg_adjoint_kernel, g1(n), dim3(256), 0, s, gd, J, saved + ((size_t)T + i) * n, tau + i,
                         param_stride, HW, B, grad_vars_in, c_tau);
      PNPX_LAUNCH_CHECK();
      hipLaunchKernelGGL(pr_item_sum_kernel, dim3(B), dim3(256), 0, s, c_tau, grad_tau + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

}  // extern "C"
