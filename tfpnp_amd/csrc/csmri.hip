// CS-MRI solver loops on gfx950: ADMM / HQS / PG / APG / RED-ADMM (tasks/csmri/solver.py:24-204).
//
// Per inner iteration the data-fidelity step is three kernels (fft_lds.h):
//   row pass (load functor builds the FFT input: x+u, x, s ...)  ->
//   column pass (forward FFT along H, k-space pointwise op with y0/mask/mu, inverse FFT along H)  ->
//   inverse row pass (store functor applies the primal/dual update and emits the next denoiser input).
// The k-space image makes one HBM round trip between the row passes; fftshifts are sign flips.
// Pointwise arithmetic uses explicit round-to-nearest mul/add/div (no FMA contraction) in the reference's
// operation order, e.g. temp = ((mu * k) + y0) / (1 + mu)  (tasks/csmri/solver.py:50).
#include "common.h"
#include "fft_lds.h"

namespace pnpx {

namespace {

__device__ __forceinline__ float mulr(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float addr(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float subr(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float divr(float a, float b) { return __fdiv_rn(a, b); }

// vars tensors are [B, nvar, H, W, 2]; slot(v, s) = pointer to variable s of item 0, item stride nvar*HW
struct Slot {
  float2* p;
  size_t istride;  // float2 per item
  int W, HW;
  __device__ float2& at(int b, int y, int x) const { return p[(size_t)b * istride + (size_t)y * W + x]; }
};
struct CSlot {
  const float2* p;
  size_t istride;
  int W, HW;
  __device__ float2 at(int b, int y, int x) const { return p[(size_t)b * istride + (size_t)y * W + x]; }
};
struct RealImg {  // [B,1,H,W]
  float* p;
  int W, HW;
  __device__ float& at(int b, int y, int x) const { return p[(size_t)b * HW + (size_t)y * W + x]; }
};

// ------------------------------------------------------------------ load functors (row pass input)
struct LoadXrPlusU {  // r2c(xr) + u
  RealImg xr;
  CSlot u;
  __device__ float2 operator()(int b, int y, int x) const {
    const float2 uu = u.at(b, y, x);
    return make_float2(addr(xr.at(b, y, x), uu.x), uu.y);
  }
};
struct LoadXr {  // r2c(xr)
  RealImg xr;
  __device__ float2 operator()(int b, int y, int x) const { return make_float2(xr.at(b, y, x), 0.f); }
};
struct LoadSlot {  // complex variable
  CSlot v;
  __device__ float2 operator()(int b, int y, int x) const { return v.at(b, y, x); }
};
struct LoadSlotPlusSlot {  // x + u, both complex
  CSlot a, c;
  __device__ float2 operator()(int b, int y, int x) const {
    const float2 p = a.at(b, y, x), q = c.at(b, y, x);
    return make_float2(addr(p.x, q.x), addr(p.y, q.y));
  }
};

struct LoadSlotMinusSlot {  // a - c, both complex (backward: cotangent of z' = gz' - gu')
  CSlot a, c;
  __device__ float2 operator()(int b, int y, int x) const {
    const float2 p = a.at(b, y, x), q = c.at(b, y, x);
    return make_float2(p.x - q.x, p.y - q.y);
  }
};

// ------------------------------------------------------------------ k-space functors (column pass)
struct KSpace {
  const float2* y0;      // [B,1,H,W,2]
  const uint8_t* mask;   // [B,1,H,W]
  const float* par;      // hyper-parameter column i: par[b*stride]
  int stride, W, HW;
};
struct MidBlend {  // k[mask] = ((mu*k) + y0)[mask] / (1 + mu)      tasks/csmri/solver.py:49-51
  KSpace k;
  __device__ float2 operator()(int b, int ky, int kx, float2 v) const {
    const size_t o = (size_t)b * k.HW + (size_t)ky * k.W + kx;
    if (!k.mask[o]) return v;
    const float mu = k.par[(size_t)b * k.stride];
    const float2 y = k.y0[o];
    const float den = addr(1.f, mu);
    return make_float2(divr(addr(mulr(mu, v.x), y.x), den), divr(addr(mulr(mu, v.y), y.y), den));
  }
  // the same in two halves for the fused 256-point column kernel (fft_lds.h MidHasFetch): reads first, arithmetic later
  struct Pre {
    float2 y;
    int m;
  };
  __device__ Pre fetch(int b, int ky, int kx) const {
    const size_t o = (size_t)b * k.HW + (size_t)ky * k.W + kx;
    return Pre{k.y0[o], (int)k.mask[o]};
  }
  __device__ float2 apply(const Pre& p, int b, float2 v) const {
    if (!p.m) return v;
    const float mu = k.par[(size_t)b * k.stride];
    const float den = addr(1.f, mu);
    return make_float2(divr(addr(mulr(mu, v.x), p.y.x), den), divr(addr(mulr(mu, v.y), p.y.y), den));
  }
};
struct MidBlendSave {  // MidBlend that also keeps the k-space image BEFORE the blend (training path: d blend / d mu needs it)
  KSpace k;
  float2* save;          // [B,H,W] complex
  __device__ float2 operator()(int b, int ky, int kx, float2 v) const {
    save[(size_t)b * k.HW + (size_t)ky * k.W + kx] = v;
    return MidBlend{k}(b, ky, kx, v);
  }
};
// Adjoint of MidBlend.  With k' = mask ? (mu k + y0) / (1 + mu) : k, a cotangent g' of k' gives
//   g = mask ? mu / (1 + mu) * g' : g'          and        d<g', k'>/d mu = <g', k - y0> / (1 + mu)^2 on the mask;
// the per-pixel terms of the latter go to `contrib` and are summed per item afterwards (item_sum_kernel).
struct MidBlendAdjoint {
  KSpace k;
  const float2* ksaved;  // k of the forward iteration
  float* contrib;        // [B,H,W]
  __device__ float2 operator()(int b, int ky, int kx, float2 g) const {
    const size_t o = (size_t)b * k.HW + (size_t)ky * k.W + kx;
    if (!k.mask[o]) {
      contrib[o] = 0.f;
      return g;
    }
    const float mu = k.par[(size_t)b * k.stride];
    const float den = 1.f + mu;
    const float2 y = k.y0[o], kv = ksaved[o];
    contrib[o] = (g.x * (kv.x - y.x) + g.y * (kv.y - y.y)) / (den * den);
    const float f = mu / den;
    return make_float2(f * g.x, f * g.y);
  }
};
struct MidResidual {  // temp = k - y0; temp[~mask] = 0               tasks/csmri/solver.py:109-110
  KSpace k;
  __device__ float2 operator()(int b, int ky, int kx, float2 v) const {
    const size_t o = (size_t)b * k.HW + (size_t)ky * k.W + kx;
    if (!k.mask[o]) return make_float2(0.f, 0.f);
    const float2 y = k.y0[o];
    return make_float2(subr(v.x, y.x), subr(v.y, y.y));
  }
};

// ------------------------------------------------------------------ store functors (inverse row pass)
struct StoreAdmm {  // z = ifft2c(k); u = u + x - z; emits Re(z - u_new) (+ x as complex on the last iteration)
  Slot z, uo, xo;
  CSlot ui;
  RealImg xr, d;
  int write_x;
  __device__ void operator()(int b, int y, int x, float2 zv) const {
    const float2 uu = ui.at(b, y, x);
    const float xv = xr.at(b, y, x);
    const float2 un = make_float2(subr(addr(uu.x, xv), zv.x), subr(addr(uu.y, 0.f), zv.y));
    z.at(b, y, x) = zv;
    uo.at(b, y, x) = un;
    d.at(b, y, x) = subr(zv.x, un.x);
    if (write_x) xo.at(b, y, x) = make_float2(xv, 0.f);
  }
  // the same in two halves for the 256-point row kernel (fft_lds.h FunctorHasFetch): the reads of u and x go out with the
  // tile's own loads.  Safe: a workgroup reads and writes only its own pixels, and the outputs never alias the inputs.
  struct Pre {
    float2 uu;
    float xv;
  };
  __device__ Pre fetch(int b, int y, int x) const { return Pre{ui.at(b, y, x), xr.at(b, y, x)}; }
  __device__ void apply(const Pre& p, int b, int y, int x, float2 zv) const {
    const float2 un = make_float2(subr(addr(p.uu.x, p.xv), zv.x), subr(addr(p.uu.y, 0.f), zv.y));
    z.at(b, y, x) = zv;
    uo.at(b, y, x) = un;
    d.at(b, y, x) = subr(zv.x, un.x);
    if (write_x) xo.at(b, y, x) = make_float2(p.xv, 0.f);
  }
};
// Backward of one ADMM iteration after the data step's adjoint: gs = cotangent of (x + u).
//   cotangent of the denoiser output xr = Re(gx' + gu' + gs);   running cotangent of u = gu' + gs
struct StoreAdmmAdjoint {
  Slot gu;
  CSlot gx;
  RealImg gxr;
  __device__ void operator()(int b, int y, int x, float2 gs) const {
    const float2 a = gu.at(b, y, x);
    gxr.at(b, y, x) = gx.at(b, y, x).x + a.x + gs.x;
    gu.at(b, y, x) = make_float2(a.x + gs.x, a.y + gs.y);
  }
};
// Backward of one HQS iteration after the data step's adjoint gs: cotangent of the denoiser output xr = Re(gx' + gs)
struct StoreHqsAdjoint {
  CSlot gx;
  RealImg gxr;
  __device__ void operator()(int b, int y, int x, float2 gs) const { gxr.at(b, y, x) = gx.at(b, y, x).x + gs.x; }
};
// Backward of one RED-ADMM iteration after the data step's adjoint gs (cotangent of x' + u).  With D = mu + lamda:
//   gxt = gx' + gu' + gs;  d/d lamda += <gxt, q1> / D;  d/d mu += <gxt, q2> / D (added to the blend's term in `contrib_mu`);
//   cotangent of the denoiser output = lamda / D * Re(gxt);  gz = mu / D * gxt;  gu = gu' + gs - mu / D * gxt
struct StoreRedAdjoint {
  Slot gx, gz, gu;          // cotangent slots of the state (gx, gu hold gx', gu' on entry)
  const float2 *q1, *q2;    // [B][HW] saved by the forward
  const float *mu, *lam;
  int stride;
  RealImg gxh;
  float *contrib_mu, *contrib_lam;
  __device__ void operator()(int b, int y, int x, float2 gs) const {
    const size_t o = (size_t)b * gxh.HW + (size_t)y * gxh.W + x;
    const float m = mu[(size_t)b * stride], l = lam[(size_t)b * stride], D = m + l;
    const float2 gxp = gx.at(b, y, x), gup = gu.at(b, y, x);
    const float2 gxt = make_float2(gxp.x + gup.x + gs.x, gxp.y + gup.y + gs.y);
    const float2 a = q1[o], c = q2[o];
    contrib_lam[o] = (gxt.x * a.x + gxt.y * a.y) / D;
    contrib_mu[o] += (gxt.x * c.x + gxt.y * c.y) / D;
    gxh.at(b, y, x) = l / D * gxt.x;
    const float bm = m / D;
    gz.at(b, y, x) = make_float2(bm * gxt.x, bm * gxt.y);
    gu.at(b, y, x) = make_float2(gup.x + gs.x - bm * gxt.x, gup.y + gs.y - bm * gxt.y);
  }
};
struct StoreAdmmCx {  // same with a complex x held in a slot (RED-ADMM); no denoiser input emitted
  Slot z, uo;
  CSlot ui, xc;
  __device__ void operator()(int b, int y, int x, float2 zv) const {
    const float2 uu = ui.at(b, y, x), xv = xc.at(b, y, x);
    z.at(b, y, x) = zv;
    uo.at(b, y, x) = make_float2(subr(addr(uu.x, xv.x), zv.x), subr(addr(uu.y, xv.y), zv.y));
  }
};
struct StoreHqs {  // z = ifft2c(k); next denoiser input Re(z)
  Slot z, xo;
  RealImg xr, d;
  int write_x;
  __device__ void operator()(int b, int y, int x, float2 zv) const {
    z.at(b, y, x) = zv;
    d.at(b, y, x) = zv.x;
    if (write_x) xo.at(b, y, x) = make_float2(xr.at(b, y, x), 0.f);
  }
};
struct StoreGrad {  // d = Re(base - tau * g)                             tasks/csmri/solver.py:111-112,146
  CSlot base;       // complex base point (x or s) ...
  RealImg base_r;   // ... or a real one (use_real)
  int use_real;
  const float* tau;
  int stride;
  RealImg d;
  float* gsave = nullptr;   // training path: keeps Re(g) [B][HW] (d d / d tau = -Re(g))
  __device__ void operator()(int b, int y, int x, float2 g) const {
    const float bx = use_real ? base_r.at(b, y, x) : base.at(b, y, x).x;
    d.at(b, y, x) = subr(bx, mulr(tau[(size_t)b * stride], g.x));
    if (gsave) gsave[(size_t)b * d.HW + (size_t)y * d.W + x] = g.x;
  }
};
struct MidMask {  // k[~mask] = 0 (adjoint / Jacobian of the masked residual wrt its k-space input)
  KSpace k;
  __device__ float2 operator()(int b, int ky, int kx, float2 v) const {
    return k.mask[(size_t)b * k.HW + (size_t)ky * k.W + kx] ? v : make_float2(0.f, 0.f);
  }
};
// PG backward, inverse row pass: with r = F^-1 M F r2c(gd):  cotangent of x_i = r2c(gd) - tau * r; also the per-pixel term of
// d/d tau = -<gd, Re(w_i)> (w_i = the forward's masked-residual image, saved).  Iterations > 0 only need the real part.
struct StorePgAdjoint {
  const float* gd;      // [B][HW]
  const float* wre;     // [B][HW] saved Re(w_i)
  const float* tau;
  int stride;
  RealImg gxr;          // next cotangent of the denoiser output (real) ...
  Slot gx;              // ... or the complex input cotangent (first iteration)
  int first;
  float* contrib;       // [B][HW]
  __device__ void operator()(int b, int y, int x, float2 r) const {
    const size_t o = (size_t)b * gxr.HW + (size_t)y * gxr.W + x;
    const float t = tau[(size_t)b * stride], v = gd[o];
    contrib[o] = -v * wre[o];
    if (first) gx.at(b, y, x) = make_float2(v - t * r.x, -t * r.y);
    else gxr.at(b, y, x) = v - t * r.x;
  }
};

// ------------------------------------------------------------------ pointwise kernels
__global__ void real_of_diff_kernel(const float2* __restrict__ a, const float2* __restrict__ c, size_t istride,
                                    float* __restrict__ d, int HW, int B) {  // d = Re(a - c) (c may be null)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float av = a[b * istride + r].x;
  d[i] = c ? subr(av, c[b * istride + r].x) : av;
}
__global__ void real_to_slot_kernel(const float* __restrict__ xr, float2* __restrict__ dst, size_t istride, int HW,
                                    int B) {  // dst = r2c(xr)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  dst[b * istride + r] = make_float2(xr[i], 0.f);
}
__global__ void copy_slot_kernel(const float2* __restrict__ src, float2* __restrict__ dst, size_t istride, int HW,
                                 int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  dst[b * istride + r] = src[b * istride + r];
}
// APG extrapolation: s = x + beta*(x - x_prev), x = r2c(xr)                tasks/csmri/solver.py:149-159
__global__ void apg_update_kernel(const float* __restrict__ xr, const float2* xprev, float2* xout,
                                  float2* __restrict__ sout, size_t istride,
                                  const float* __restrict__ beta, int stride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float be = beta[b * stride];
  const float2 xp = xprev[b * istride + r];
  const float xv = xr[i];
  sout[b * istride + r] = make_float2(addr(xv, mulr(be, subr(xv, xp.x))), addr(0.f, mulr(be, subr(0.f, xp.y))));
  xout[b * istride + r] = make_float2(xv, 0.f);
}
// training forward: the same update that also keeps x' - x_prev (complex) for d/d beta
__global__ void apg_update_save_kernel(const float* __restrict__ xr, const float2* xprev, float2* xout,
                                       float2* __restrict__ sout, size_t istride, const float* __restrict__ beta,
                                       int stride, int HW, int B, float2* __restrict__ diff) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float be = beta[b * stride];
  const float2 xp = xprev[b * istride + r];
  const float xv = xr[i];
  const float dx = subr(xv, xp.x), dy = subr(0.f, xp.y);
  diff[i] = make_float2(dx, dy);
  sout[b * istride + r] = make_float2(addr(xv, mulr(be, dx)), addr(0.f, mulr(be, dy)));
  xout[b * istride + r] = make_float2(xv, 0.f);
}
// APG backward, first half of an iteration: with s' = x' + beta (x' - x_prev):
//   d/d beta term <gs', x' - x_prev>;  cotangent of the denoiser output Re(gx' + (1 + beta) gs');  cotangent of x_prev = -beta gs'
__global__ void apg_adjoint_pre_kernel(float2* g, size_t istride, const float* __restrict__ beta, int stride,
                                       const float2* __restrict__ diff, float* __restrict__ gxr,
                                       float* __restrict__ contrib, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  float2* gi = g + b * istride + r;
  const float be = beta[b * stride];
  const float2 gx = gi[0], gs = gi[HW], df = diff[i];
  contrib[i] = gs.x * df.x + gs.y * df.y;
  gxr[i] = gx.x + (1.f + be) * gs.x;
  gi[0] = make_float2(-be * gs.x, -be * gs.y);
}
// RED x-update: x = (lamda*x_half + mu*(z-u)) / (mu + lamda)               tasks/csmri/solver.py:188-190
__global__ void red_update_kernel(const float* __restrict__ xh, const float2* __restrict__ z,
                                  const float2* __restrict__ u, float2* xout, size_t istride,
                                  const float* __restrict__ mu, const float* __restrict__ lam, int stride, int HW,
                                  int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float m = mu[b * stride], l = lam[b * stride];
  const float2 zv = z[b * istride + r], uv = u[b * istride + r];
  const float den = addr(m, l);
  xout[b * istride + r] = make_float2(divr(addr(mulr(l, xh[i]), mulr(m, subr(zv.x, uv.x))), den),
                                      divr(addr(mulr(l, 0.f), mulr(m, subr(zv.y, uv.y))), den));
}

// training forward of RED-ADMM: the same x-update that also keeps q1 = r2c(xh) - x' and q2 = (z - u) - x' (d x' / d lamda and
// d x' / d mu up to the factor 1 / (mu + lamda))
__global__ void red_update_save_kernel(const float* __restrict__ xh, const float2* __restrict__ z,
                                       const float2* __restrict__ u, float2* xout, size_t istride,
                                       const float* __restrict__ mu, const float* __restrict__ lam, int stride, int HW,
                                       int B, float2* __restrict__ q1, float2* __restrict__ q2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  const float m = mu[b * stride], l = lam[b * stride];
  const float2 zv = z[b * istride + r], uv = u[b * istride + r];
  const float den = addr(m, l);
  const float2 zu = make_float2(subr(zv.x, uv.x), subr(zv.y, uv.y));
  const float2 xn = make_float2(divr(addr(mulr(l, xh[i]), mulr(m, zu.x)), den), divr(addr(mulr(l, 0.f), mulr(m, zu.y)), den));
  xout[b * istride + r] = xn;
  q1[i] = make_float2(xh[i] - xn.x, -xn.y);
  q2[i] = make_float2(zu.x - xn.x, zu.y - xn.y);
}
// RED-ADMM backward: the x slot takes the denoiser's input cotangent (d = Re(x))
__global__ void red_adjoint_finish_kernel(const float* __restrict__ gd, float2* g, size_t istride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  g[b * istride + r] = make_float2(gd[i], 0.f);
}
// backward of d = Re(z - u): gz = r2c(gd), gu -= r2c(gd); x of the previous state is not read by an iteration: gx = 0
__global__ void admm_adjoint_finish_kernel(const float* __restrict__ gd, float2* g, size_t istride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  float2* gi = g + b * istride + r;
  const float v = gd[i];
  const float2 gu = gi[2 * (size_t)HW];
  gi[0] = make_float2(0.f, 0.f);
  gi[HW] = make_float2(v, 0.f);
  gi[2 * (size_t)HW] = make_float2(gu.x - v, gu.y);
}
// HQS: backward of d = Re(z): gz = r2c(gd); x of the previous state is not read by an iteration: gx = 0
__global__ void hqs_adjoint_finish_kernel(const float* __restrict__ gd, float2* g, size_t istride, int HW, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)HW * B) return;
  const size_t b = i / HW, r = i - b * HW;
  float2* gi = g + b * istride + r;
  gi[0] = make_float2(0.f, 0.f);
  gi[HW] = make_float2(gd[i], 0.f);
}
// out[b] = sum of the HW values of item b, fixed summation order (deterministic)
__global__ void __launch_bounds__(256) item_sum_kernel(const float* __restrict__ c, float* __restrict__ out, int HW) {
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

struct Scratch {
  float* d;    // [B,1,H,W] denoiser input
  float* xr;   // [B,1,H,W] denoiser output
  float2* k;   // [B,H,W] complex k-space / row-pass intermediate
};

int get_scratch(pnpx_ctx* ctx, int B, int H, int W, Scratch* S) {
  const size_t hw = (size_t)H * W * B;
  void* p;
  PNPX_TRY(ctx_scratch(ctx, hw * 16 + 1024, &p));
  S->k = static_cast<float2*>(p);
  S->d = reinterpret_cast<float*>(S->k + hw);
  S->xr = S->d + hw;
  return PNPX_OK;
}

inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

int check_common(const void* a, const void* b, const void* c, const void* d, const void* e, int B, int H, int W,
                 int T, int stride) {
  if (!a || !b || !c || !d || !e || B <= 0 || T < 0 || stride < T) {
    set_error("csmri: bad argument (null pointer, B<=0 or param_stride < T)");
    return PNPX_ERR_ARG;
  }
  (void)H;
  (void)W;
  return PNPX_OK;
}

}  // namespace

}  // namespace pnpx

using namespace pnpx;

#define LOCK_CTX(ctx)                                  \
  if (!(ctx)) {                                        \
    pnpx::set_error("null context");                   \
    return PNPX_ERR_ARG;                               \
  }                                                    \
  std::lock_guard<std::mutex> _lk((ctx)->mu);          \
  PNPX_HIP(hipSetDevice((ctx)->device))

// ADMM forward; `saved` != NULL (training path) keeps, per iteration, the denoiser input d_i [T][B][HW] floats followed
// by the k-space image before the blend k_i [T][B][HW] complex, and parks the denoiser activations of every iteration
// in the context's training ring (unet_denoise_train): iteration i holds ticket *ticket_out + i (tickets are handed
// out consecutively under the context lock; 0: the first iteration was not parked).
static int admm_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const uint8_t* mask,
                        const float* sigma_d, const float* mu, int param_stride, int B, int H, int W, int T,
                        float* saved, hipStream_t s, unsigned long long* ticket_out = nullptr) {
  PNPX_TRY(check_common(vars_in, vars_out, y0, mask, sigma_d, B, H, W, T, param_stride));
  if (ticket_out) *ticket_out = 0;
  const bool park = saved && ticket_out;
  const int HW = H * W;
  const size_t is = 3 * (size_t)HW;
  Scratch S;
  PNPX_TRY(get_scratch(ctx, B, H, W, &S));
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
  const float2* vin = reinterpret_cast<const float2*>(vars_in);
  float2* vout = reinterpret_cast<float2*>(vars_out);
  if (T == 0) {
    PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    return PNPX_OK;
  }
  // d0 = Re(z - u)                                                   tasks/csmri/solver.py:45
  hipLaunchKernelGGL(real_of_diff_kernel, g1((size_t)HW * B), dim3(256), 0, s, vin + HW, vin + 2 * HW, is, S.d, HW, B);
  PNPX_LAUNCH_CHECK();
  RealImg xr{S.xr, W, HW}, d{S.d, W, HW};
  Slot xo{vout, is, W, HW}, zo{vout + HW, is, W, HW}, uo{vout + 2 * HW, is, W, HW};
  StoreC kst{S.k, H, W};
  LoadC kld{S.k, H, W};
  for (int i = 0; i < T; ++i) {
    if (saved)
      PNPX_HIP(hipMemcpyAsync(saved + (size_t)i * B * HW, S.d, sizeof(float) * B * HW, hipMemcpyDeviceToDevice, s));
    if (park) {
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, S.d, sigma_d + i, param_stride, S.xr, B, H, W, s, &tk));
      if (i == 0) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, S.d, sigma_d + i, param_stride, S.xr, nullptr, B, H, W, s, nullptr));
    }
    CSlot ui{(i == 0 ? vin : vout) + 2 * HW, is, W, HW};
    PNPX_TRY((launch_rows<false>(P, LoadXrPlusU{xr, ui}, kst, s)));
    KSpace ks{reinterpret_cast<const float2*>(y0), mask, mu + i, param_stride, W, HW};
    if (saved) {
      float2* ksave = reinterpret_cast<float2*>(saved + (size_t)T * B * HW) + (size_t)i * B * HW;
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlendSave{ks, ksave}, kst, s)));
    } else {
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlend{ks}, kst, s)));
    }
    PNPX_TRY((launch_rows<true>(P, kld, StoreAdmm{zo, uo, xo, ui, xr, d, i == T - 1}, s)));
  }
  return PNPX_OK;
}

extern "C" int pnpx_csmri_admm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                               const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride, int B,
                               int H, int W, int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return admm_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, param_stride, B, H, W, T, nullptr, s);
  });
}

extern "C" int pnpx_csmri_admm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                                     const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride,
                                     int B, int H, int W, int T, float* saved, unsigned long long* ticket,
                                     void* stream) {
  LOCK_CTX(ctx);
  if ((!saved && T > 0) || !ticket) {
    pnpx::set_error("csmri_admm_train: saved / ticket is null");
    return PNPX_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return admm_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, param_stride, B, H, W, T, saved, s, ticket);
  });
}

// Vector-Jacobian product of the T-iteration ADMM map wrt (variables, sigma_d, mu), iterations walked in reverse:
//   forward i:   x = r2c(D(d_i, sigma_i)),  d_i = Re(z - u);   z' = F^-1 blend_mu_i(F(x + u));   u' = u + x - z'
//   backward i:  gz~ = gz' - gu';  gs = F^-1 blend^T(F gz~)  [F unitary: adjoint = inverse];  g_mu_i = sum contrib
//                gxr = Re(gx' + gu' + gs);  gu~ = gu' + gs;  (gd, g_sigma_i) = D^T(gxr);  gz = r2c(gd), gu = gu~ - r2c(gd), gx = 0
extern "C" int pnpx_csmri_admm_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                                        const float* mu, int param_stride, const float* saved,
                                        const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d,
                                        float* grad_mu, float* work, int B, int H, int W, int T,
                                        unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    PNPX_TRY(check_common(grad_vars_out, grad_vars_in, y0, mask, sigma_d, B, H, W, T, param_stride));
    if (T > 0 && (!saved || !grad_sigma_d || !grad_mu || !work || !mu)) {
      set_error("csmri_admm_backward: null pointer");
      return PNPX_ERR_ARG;
    }
    const int HW = H * W;
    const size_t is = 3 * (size_t)HW, n = (size_t)B * HW;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    if (T == 0) return PNPX_OK;
    FftPlan2D P;
    PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
    float2* g = reinterpret_cast<float2*>(grad_vars_in);
    float *gxr = work, *gd = work + n, *contrib = work + 2 * n;
    const float* saved_d = saved;
    const float2* saved_k = reinterpret_cast<const float2*>(saved + (size_t)T * n);

    for (int i = T - 1; i >= 0; --i) {
      Scratch S;   // the k-space transit buffer lives in the context scratch, which the denoiser VJP may re-allocate
      PNPX_TRY(get_scratch(ctx, B, H, W, &S));
      StoreC kst{S.k, H, W};
      LoadC kld{S.k, H, W};
      CSlot gx{g, is, W, HW}, gz{g + HW, is, W, HW}, guc{g + 2 * HW, is, W, HW};
      Slot gu{g + 2 * HW, is, W, HW};
      PNPX_TRY((launch_rows<false>(P, LoadSlotMinusSlot{gz, guc}, kst, s)));
      KSpace ks{reinterpret_cast<const float2*>(y0), mask, mu + i, param_stride, W, HW};
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlendAdjoint{ks, saved_k + (size_t)i * n, contrib}, kst, s)));
      PNPX_TRY((launch_rows<true>(P, kld, StoreAdmmAdjoint{gu, gx, RealImg{gxr, W, HW}}, s)));
      hipLaunchKernelGGL(item_sum_kernel, dim3(B), dim3(256), 0, s, contrib, grad_mu + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
      // iteration i parked its activations under ticket + i (if the ring still holds them; else re-computation)
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved_d + (size_t)i * n, sigma_d + i, param_stride, gxr, gd,
                                            grad_sigma_d + (size_t)i * B, B, H, W, s, ticket ? ticket + i : 0));
      hipLaunchKernelGGL(admm_adjoint_finish_kernel, g1(n), dim3(256), 0, s, gd, g, is, HW, B);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

// HQS forward; `saved` != NULL (training path): as admm_forward -- per iteration the denoiser input d_i [T][B][HW] and the
// k-space image before the blend [T][B][HW] complex are kept, activations parked in the training ring (ticket + i).
static int hqs_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const uint8_t* mask,
                       const float* sigma_d, const float* mu, int param_stride, int B, int H, int W, int T, float* saved,
                       hipStream_t s, unsigned long long* ticket_out = nullptr) {
  PNPX_TRY(check_common(vars_in, vars_out, y0, mask, sigma_d, B, H, W, T, param_stride));
  if (ticket_out) *ticket_out = 0;
  const bool park = saved && ticket_out;
  const int HW = H * W;
  const size_t is = 2 * (size_t)HW;
  Scratch S;
  PNPX_TRY(get_scratch(ctx, B, H, W, &S));
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
  const float2* vin = reinterpret_cast<const float2*>(vars_in);
  float2* vout = reinterpret_cast<float2*>(vars_out);
  if (T == 0) {
    PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    return PNPX_OK;
  }
  hipLaunchKernelGGL(real_of_diff_kernel, g1((size_t)HW * B), dim3(256), 0, s, vin + HW, (const float2*)nullptr, is,
                     S.d, HW, B);
  PNPX_LAUNCH_CHECK();
  RealImg xr{S.xr, W, HW}, d{S.d, W, HW};
  Slot xo{vout, is, W, HW}, zo{vout + HW, is, W, HW};
  StoreC kst{S.k, H, W};
  LoadC kld{S.k, H, W};
  for (int i = 0; i < T; ++i) {
    if (saved)
      PNPX_HIP(hipMemcpyAsync(saved + (size_t)i * B * HW, S.d, sizeof(float) * B * HW, hipMemcpyDeviceToDevice, s));
    if (park) {
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, S.d, sigma_d + i, param_stride, S.xr, B, H, W, s, &tk));
      if (i == 0) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, S.d, sigma_d + i, param_stride, S.xr, nullptr, B, H, W, s, nullptr));
    }
    PNPX_TRY((launch_rows<false>(P, LoadXr{xr}, kst, s)));
    KSpace ks{reinterpret_cast<const float2*>(y0), mask, mu + i, param_stride, W, HW};
    if (saved) {
      float2* ksave = reinterpret_cast<float2*>(saved + (size_t)T * B * HW) + (size_t)i * B * HW;
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlendSave{ks, ksave}, kst, s)));
    } else {
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlend{ks}, kst, s)));
    }
    PNPX_TRY((launch_rows<true>(P, kld, StoreHqs{zo, xo, xr, d, i == T - 1}, s)));
  }
  return PNPX_OK;
}

extern "C" int pnpx_csmri_hqs(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                              const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride, int B,
                              int H, int W, int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return hqs_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, param_stride, B, H, W, T, nullptr, s);
  });
}

extern "C" int pnpx_csmri_hqs_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                                    const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride, int B,
                                    int H, int W, int T, float* saved, unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  if ((!saved && T > 0) || !ticket) {
    pnpx::set_error("csmri_hqs_train: saved / ticket is null");
    return PNPX_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return hqs_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, param_stride, B, H, W, T, saved, s, ticket);
  });
}

// VJP of the T-iteration HQS map wrt (variables, sigma_d, mu), iterations walked in reverse:
//   forward i:   x = r2c(D(d_i, sigma_i)),  d_i = Re(z);   z' = F^-1 blend_mu_i(F(x))
//   backward i:  gs = F^-1 blend^T(F gz');  g_mu_i = sum contrib;  gxr = Re(gx' + gs);  (gd, g_sigma_i) = D^T(gxr);
//                gz = r2c(gd), gx = 0
extern "C" int pnpx_csmri_hqs_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                                       const float* mu, int param_stride, const float* saved,
                                       const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d,
                                       float* grad_mu, float* work, int B, int H, int W, int T,
                                       unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    PNPX_TRY(check_common(grad_vars_out, grad_vars_in, y0, mask, sigma_d, B, H, W, T, param_stride));
    if (T > 0 && (!saved || !grad_sigma_d || !grad_mu || !work || !mu)) {
      set_error("csmri_hqs_backward: null pointer");
      return PNPX_ERR_ARG;
    }
    const int HW = H * W;
    const size_t is = 2 * (size_t)HW, n = (size_t)B * HW;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    if (T == 0) return PNPX_OK;
    FftPlan2D P;
    PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
    float2* g = reinterpret_cast<float2*>(grad_vars_in);
    float *gxr = work, *gd = work + n, *contrib = work + 2 * n;
    const float* saved_d = saved;
    const float2* saved_k = reinterpret_cast<const float2*>(saved + (size_t)T * n);
    for (int i = T - 1; i >= 0; --i) {
      Scratch S;
      PNPX_TRY(get_scratch(ctx, B, H, W, &S));
      StoreC kst{S.k, H, W};
      LoadC kld{S.k, H, W};
      CSlot gx{g, is, W, HW}, gz{g + HW, is, W, HW};
      PNPX_TRY((launch_rows<false>(P, LoadSlot{gz}, kst, s)));
      KSpace ks{reinterpret_cast<const float2*>(y0), mask, mu + i, param_stride, W, HW};
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlendAdjoint{ks, saved_k + (size_t)i * n, contrib}, kst, s)));
      PNPX_TRY((launch_rows<true>(P, kld, StoreHqsAdjoint{gx, RealImg{gxr, W, HW}}, s)));
      hipLaunchKernelGGL(item_sum_kernel, dim3(B), dim3(256), 0, s, contrib, grad_mu + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved_d + (size_t)i * n, sigma_d + i, param_stride, gxr, gd,
                                            grad_sigma_d + (size_t)i * B, B, H, W, s, ticket ? ticket + i : 0));
      hipLaunchKernelGGL(hqs_adjoint_finish_kernel, g1(n), dim3(256), 0, s, gd, g, is, HW, B);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

// PG forward; `saved` != NULL (training path): per iteration the denoiser input d_i [T][B][HW] followed by Re(w_i)
// [T][B][HW] (w_i = ifft2c(mask * (fft2c(x_i) - y0))), activations parked in the training ring (ticket + i).
static int pg_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const uint8_t* mask,
                      const float* sigma_d, const float* tau, int param_stride, int B, int H, int W, int T, float* saved,
                      hipStream_t s, unsigned long long* ticket_out = nullptr) {
  PNPX_TRY(check_common(vars_in, vars_out, y0, mask, sigma_d, B, H, W, T, param_stride));
  if (ticket_out) *ticket_out = 0;
  const bool park = saved && ticket_out;
  const int HW = H * W;
  const size_t is = (size_t)HW;
  Scratch S;
  PNPX_TRY(get_scratch(ctx, B, H, W, &S));
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
  const float2* vin = reinterpret_cast<const float2*>(vars_in);
  float2* vout = reinterpret_cast<float2*>(vars_out);
  if (T == 0) {
    PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    return PNPX_OK;
  }
  RealImg xr{S.xr, W, HW}, d{S.d, W, HW};
  StoreC kst{S.k, H, W};
  LoadC kld{S.k, H, W};
  CSlot x0{vin, is, W, HW};
  for (int i = 0; i < T; ++i) {
    // gradient step on x (complex x0 on the first iteration, r2c(denoised) afterwards)
    if (i == 0) PNPX_TRY((launch_rows<false>(P, LoadSlot{x0}, kst, s)));
    else PNPX_TRY((launch_rows<false>(P, LoadXr{xr}, kst, s)));
    KSpace ks{reinterpret_cast<const float2*>(y0), mask, tau + i, param_stride, W, HW};
    PNPX_TRY((launch_cols<false, true>(P, kld, MidResidual{ks}, kst, s)));
    float* wsave = saved ? saved + ((size_t)T + i) * B * HW : nullptr;
    PNPX_TRY((launch_rows<true>(P, kld, StoreGrad{x0, xr, i != 0, tau + i, param_stride, d, wsave}, s)));
    if (saved)
      PNPX_HIP(hipMemcpyAsync(saved + (size_t)i * B * HW, S.d, sizeof(float) * B * HW, hipMemcpyDeviceToDevice, s));
    if (park) {
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, S.d, sigma_d + i, param_stride, S.xr, B, H, W, s, &tk));
      if (i == 0) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, S.d, sigma_d + i, param_stride, S.xr, nullptr, B, H, W, s, nullptr));
    }
  }
  hipLaunchKernelGGL(real_to_slot_kernel, g1((size_t)HW * B), dim3(256), 0, s, S.xr, vout, is, HW, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

extern "C" int pnpx_csmri_pg(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                             const uint8_t* mask, const float* sigma_d, const float* tau, int param_stride, int B,
                             int H, int W, int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return pg_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, tau, param_stride, B, H, W, T, nullptr, s);
  });
}

extern "C" int pnpx_csmri_pg_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                                   const uint8_t* mask, const float* sigma_d, const float* tau, int param_stride, int B,
                                   int H, int W, int T, float* saved, unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  if ((!saved && T > 0) || !ticket) {
    pnpx::set_error("csmri_pg_train: saved / ticket is null");
    return PNPX_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return pg_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, tau, param_stride, B, H, W, T, saved, s, ticket);
  });
}

// VJP of the T-iteration PG map wrt (x, sigma_d, tau), iterations walked in reverse:
//   forward i:   w = F^-1 M (F x - y0);  d_i = Re(x - tau_i w);  x' = r2c(D(d_i, sigma_i))
//   backward i:  (gd, g_sigma_i) = D^T(Re gx');  g_tau_i = -<gd, Re w>;  gx = r2c(gd) - tau_i F^-1 M F r2c(gd)
//                (F^-1 M F is self-adjoint: F unitary, M a real mask); iterations > 0 only pass Re(gx) on.
extern "C" int pnpx_csmri_pg_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                                      const float* tau, int param_stride, const float* saved,
                                      const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d,
                                      float* grad_tau, float* work, int B, int H, int W, int T,
                                      unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    PNPX_TRY(check_common(grad_vars_out, grad_vars_in, y0, mask, sigma_d, B, H, W, T, param_stride));
    if (T > 0 && (!saved || !grad_sigma_d || !grad_tau || !work || !tau)) {
      set_error("csmri_pg_backward: null pointer");
      return PNPX_ERR_ARG;
    }
    const int HW = H * W;
    const size_t is = (size_t)HW, n = (size_t)B * HW;
    if (T == 0) {
      PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
      return PNPX_OK;
    }
    FftPlan2D P;
    PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
    float *gxr = work, *gd = work + n, *contrib = work + 2 * n;
    // cotangent of the last denoiser output = Re(grad of x_T)
    hipLaunchKernelGGL(real_of_diff_kernel, g1(n), dim3(256), 0, s, reinterpret_cast<const float2*>(grad_vars_out),
                       (const float2*)nullptr, is, gxr, HW, B);
    PNPX_LAUNCH_CHECK();
    for (int i = T - 1; i >= 0; --i) {
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved + (size_t)i * n, sigma_d + i, param_stride, gxr, gd,
                                            grad_sigma_d + (size_t)i * B, B, H, W, s, ticket ? ticket + i : 0));
      Scratch S;
      PNPX_TRY(get_scratch(ctx, B, H, W, &S));
      StoreC kst{S.k, H, W};
      LoadC kld{S.k, H, W};
      PNPX_TRY((launch_rows<false>(P, LoadXr{RealImg{gd, W, HW}}, kst, s)));
      KSpace ks{reinterpret_cast<const float2*>(y0), mask, tau + i, param_stride, W, HW};
      PNPX_TRY((launch_cols<false, true>(P, kld, MidMask{ks}, kst, s)));
      PNPX_TRY((launch_rows<true>(P, kld,
                                  StorePgAdjoint{gd, saved + ((size_t)T + i) * n, tau + i, param_stride, RealImg{gxr, W, HW},
                                                 Slot{reinterpret_cast<float2*>(grad_vars_in), is, W, HW}, i == 0, contrib},
                                  s)));
      hipLaunchKernelGGL(item_sum_kernel, dim3(B), dim3(256), 0, s, contrib, grad_tau + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

// APG forward; `saved` != NULL (training path): per iteration the denoiser input d_i [T][B][HW], Re(w_i) [T][B][HW]
// (w_i = ifft2c(mask * (fft2c(s_i) - y0))) and x' - x_prev [T][B][HW] complex; activations parked (ticket + i).
static int apg_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const uint8_t* mask,
                       const float* sigma_d, const float* tau, const float* beta, int param_stride, int B, int H, int W,
                       int T, float* saved, hipStream_t s, unsigned long long* ticket_out = nullptr) {
  PNPX_TRY(check_common(vars_in, vars_out, y0, mask, sigma_d, B, H, W, T, param_stride));
  if (!beta || !tau) {
    set_error("csmri_apg: null tau/beta");
    return PNPX_ERR_ARG;
  }
  if (ticket_out) *ticket_out = 0;
  const bool park = saved && ticket_out;
  const int HW = H * W;
  const size_t is = 2 * (size_t)HW, n = (size_t)B * HW;
  Scratch S;
  PNPX_TRY(get_scratch(ctx, B, H, W, &S));
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
  float2* vout = reinterpret_cast<float2*>(vars_out);
  PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
  RealImg xr{S.xr, W, HW}, d{S.d, W, HW};
  StoreC kst{S.k, H, W};
  LoadC kld{S.k, H, W};
  CSlot sc{vout + HW, is, W, HW};
  for (int i = 0; i < T; ++i) {
    PNPX_TRY((launch_rows<false>(P, LoadSlot{sc}, kst, s)));
    KSpace ks{reinterpret_cast<const float2*>(y0), mask, tau + i, param_stride, W, HW};
    PNPX_TRY((launch_cols<false, true>(P, kld, MidResidual{ks}, kst, s)));
    float* wsave = saved ? saved + ((size_t)T + i) * n : nullptr;
    PNPX_TRY((launch_rows<true>(P, kld, StoreGrad{sc, xr, 0, tau + i, param_stride, d, wsave}, s)));
    if (saved) PNPX_HIP(hipMemcpyAsync(saved + (size_t)i * n, S.d, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    if (park) {
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, S.d, sigma_d + i, param_stride, S.xr, B, H, W, s, &tk));
      if (i == 0) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, S.d, sigma_d + i, param_stride, S.xr, nullptr, B, H, W, s, nullptr));
    }
    if (saved) {
      float2* diff = reinterpret_cast<float2*>(saved + 2 * (size_t)T * n) + (size_t)i * n;
      hipLaunchKernelGGL(apg_update_save_kernel, g1(n), dim3(256), 0, s, S.xr, vout, vout, vout + HW, is, beta + i,
                         param_stride, HW, B, diff);
    } else {
      hipLaunchKernelGGL(apg_update_kernel, g1(n), dim3(256), 0, s, S.xr, vout, vout, vout + HW, is, beta + i,
                         param_stride, HW, B);
    }
    PNPX_LAUNCH_CHECK();
  }
  return PNPX_OK;
}

extern "C" int pnpx_csmri_apg(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                              const uint8_t* mask, const float* sigma_d, const float* tau, const float* beta,
                              int param_stride, int B, int H, int W, int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return apg_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, tau, beta, param_stride, B, H, W, T, nullptr, s);
  });
}

extern "C" int pnpx_csmri_apg_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                                    const uint8_t* mask, const float* sigma_d, const float* tau, const float* beta,
                                    int param_stride, int B, int H, int W, int T, float* saved,
                                    unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  if ((!saved && T > 0) || !ticket) {
    pnpx::set_error("csmri_apg_train: saved / ticket is null");
    return PNPX_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return apg_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, tau, beta, param_stride, B, H, W, T, saved, s, ticket);
  });
}

// VJP of the T-iteration APG map wrt (cat(x, s), sigma_d, tau, beta), iterations walked in reverse:
//   forward i:   w = F^-1 M (F s - y0);  d_i = Re(s - tau_i w);  x' = r2c(D(d_i, sigma_i));  s' = x' + beta_i (x' - x)
//   backward i:  g_beta_i = <gs', x' - x>;  gxr = Re(gx' + (1 + beta_i) gs');  gx = -beta_i gs';  (gd, g_sigma_i) = D^T(gxr);
//                g_tau_i = -<gd, Re w>;  gs = r2c(gd) - tau_i F^-1 M F r2c(gd)
extern "C" int pnpx_csmri_apg_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                                       const float* tau, const float* beta, int param_stride, const float* saved,
                                       const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d,
                                       float* grad_tau, float* grad_beta, float* work, int B, int H, int W, int T,
                                       unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    PNPX_TRY(check_common(grad_vars_out, grad_vars_in, y0, mask, sigma_d, B, H, W, T, param_stride));
    if (T > 0 && (!saved || !grad_sigma_d || !grad_tau || !grad_beta || !work || !tau || !beta)) {
      set_error("csmri_apg_backward: null pointer");
      return PNPX_ERR_ARG;
    }
    const int HW = H * W;
    const size_t is = 2 * (size_t)HW, n = (size_t)B * HW;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    if (T == 0) return PNPX_OK;
    FftPlan2D P;
    PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
    float2* g = reinterpret_cast<float2*>(grad_vars_in);
    float *gxr = work, *gd = work + n, *contrib = work + 2 * n, *contrib_b = work + 3 * n;
    const float2* diff = reinterpret_cast<const float2*>(saved + 2 * (size_t)T * n);
    for (int i = T - 1; i >= 0; --i) {
      hipLaunchKernelGGL(apg_adjoint_pre_kernel, g1(n), dim3(256), 0, s, g, is, beta + i, param_stride, diff + (size_t)i * n,
                         gxr, contrib_b, HW, B);
      PNPX_LAUNCH_CHECK();
      hipLaunchKernelGGL(item_sum_kernel, dim3(B), dim3(256), 0, s, contrib_b, grad_beta + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved + (size_t)i * n, sigma_d + i, param_stride, gxr, gd,
                                            grad_sigma_d + (size_t)i * B, B, H, W, s, ticket ? ticket + i : 0));
      Scratch S;
      PNPX_TRY(get_scratch(ctx, B, H, W, &S));
      StoreC kst{S.k, H, W};
      LoadC kld{S.k, H, W};
      PNPX_TRY((launch_rows<false>(P, LoadXr{RealImg{gd, W, HW}}, kst, s)));
      KSpace ks{reinterpret_cast<const float2*>(y0), mask, tau + i, param_stride, W, HW};
      PNPX_TRY((launch_cols<false, true>(P, kld, MidMask{ks}, kst, s)));
      // (the "first iteration" form of the PG store functor writes the complex cotangent -- here into the s slot, always)
      PNPX_TRY((launch_rows<true>(P, kld,
                                  StorePgAdjoint{gd, saved + ((size_t)T + i) * n, tau + i, param_stride, RealImg{gxr, W, HW},
                                                 Slot{g + HW, is, W, HW}, 1, contrib},
                                  s)));
      hipLaunchKernelGGL(item_sum_kernel, dim3(B), dim3(256), 0, s, contrib, grad_tau + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}

// RED-ADMM forward; `saved` != NULL (training path): per iteration d_i = Re(x) [T][n], the k-space image before the blend
// [T][n][2], q1 = r2c(xh) - x' [T][n][2], q2 = (z - u) - x' [T][n][2] (n = B*H*W); activations parked (ticket + i).
static int red_forward(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const uint8_t* mask,
                       const float* sigma_d, const float* mu, const float* lamda, int param_stride, int B, int H, int W,
                       int T, float* saved, hipStream_t s, unsigned long long* ticket_out = nullptr) {
  PNPX_TRY(check_common(vars_in, vars_out, y0, mask, sigma_d, B, H, W, T, param_stride));
  if (!mu || !lamda) {
    set_error("csmri_redadmm: null mu/lamda");
    return PNPX_ERR_ARG;
  }
  if (ticket_out) *ticket_out = 0;
  const bool park = saved && ticket_out;
  const int HW = H * W;
  const size_t is = 3 * (size_t)HW, n = (size_t)B * HW;
  Scratch S;
  PNPX_TRY(get_scratch(ctx, B, H, W, &S));
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
  float2* vout = reinterpret_cast<float2*>(vars_out);
  PNPX_HIP(hipMemcpyAsync(vars_out, vars_in, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
  StoreC kst{S.k, H, W};
  LoadC kld{S.k, H, W};
  Slot zo{vout + HW, is, W, HW}, uo{vout + 2 * HW, is, W, HW};
  CSlot xc{vout, is, W, HW}, uc{vout + 2 * HW, is, W, HW};
  for (int i = 0; i < T; ++i) {
    hipLaunchKernelGGL(real_of_diff_kernel, g1(n), dim3(256), 0, s, vout, (const float2*)nullptr, is, S.d, HW, B);
    PNPX_LAUNCH_CHECK();
    if (saved) PNPX_HIP(hipMemcpyAsync(saved + (size_t)i * n, S.d, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    if (park) {
      unsigned long long tk = 0;
      PNPX_TRY(unet_denoise_train(ctx, S.d, sigma_d + i, param_stride, S.xr, B, H, W, s, &tk));
      if (i == 0) *ticket_out = tk;
    } else {
      PNPX_TRY(unet_denoise(ctx, S.d, sigma_d + i, param_stride, S.xr, nullptr, B, H, W, s, nullptr));
    }
    if (saved) {
      float2* q1 = reinterpret_cast<float2*>(saved + 3 * (size_t)T * n) + (size_t)i * n;
      float2* q2 = reinterpret_cast<float2*>(saved + 5 * (size_t)T * n) + (size_t)i * n;
      hipLaunchKernelGGL(red_update_save_kernel, g1(n), dim3(256), 0, s, S.xr, vout + HW, vout + 2 * HW, vout, is, mu + i,
                         lamda + i, param_stride, HW, B, q1, q2);
    } else {
      hipLaunchKernelGGL(red_update_kernel, g1(n), dim3(256), 0, s, S.xr, vout + HW, vout + 2 * HW, vout, is, mu + i,
                         lamda + i, param_stride, HW, B);
    }
    PNPX_LAUNCH_CHECK();
    PNPX_TRY((launch_rows<false>(P, LoadSlotPlusSlot{xc, uc}, kst, s)));
    KSpace ks{reinterpret_cast<const float2*>(y0), mask, mu + i, param_stride, W, HW};
    if (saved) {
      float2* ksave = reinterpret_cast<float2*>(saved + (size_t)T * n) + (size_t)i * n;
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlendSave{ks, ksave}, kst, s)));
    } else {
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlend{ks}, kst, s)));
    }
    PNPX_TRY((launch_rows<true>(P, kld, StoreAdmmCx{zo, uo, uc, xc}, s)));
  }
  return PNPX_OK;
}

extern "C" int pnpx_csmri_redadmm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                                  const uint8_t* mask, const float* sigma_d, const float* mu, const float* lamda,
                                  int param_stride, int B, int H, int W, int T, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return red_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, lamda, param_stride, B, H, W, T, nullptr, s);
  });
}

extern "C" int pnpx_csmri_redadmm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                                        const uint8_t* mask, const float* sigma_d, const float* mu, const float* lamda,
                                        int param_stride, int B, int H, int W, int T, float* saved,
                                        unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  if ((!saved && T > 0) || !ticket) {
    pnpx::set_error("csmri_redadmm_train: saved / ticket is null");
    return PNPX_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    return red_forward(ctx, vars_in, vars_out, y0, mask, sigma_d, mu, lamda, param_stride, B, H, W, T, saved, s, ticket);
  });
}

// VJP of the T-iteration RED-ADMM map wrt (cat(x, z, u), sigma_d, mu, lamda), iterations walked in reverse:
//   forward i:   xh = D(Re x, sigma_i);  x' = (lamda r2c(xh) + mu (z - u)) / (mu + lamda);  z' = F^-1 blend_mu(F(x' + u));
//                u' = u + x' - z'
//   backward i:  gs = F^-1 blend^T F (gz' - gu');  StoreRedAdjoint (above);  (gd, g_sigma_i) = D^T(lamda / D Re gxt);  gx = r2c(gd)
extern "C" int pnpx_csmri_redadmm_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                                           const float* mu, const float* lamda, int param_stride, const float* saved,
                                           const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d,
                                           float* grad_mu, float* grad_lamda, float* work, int B, int H, int W, int T,
                                           unsigned long long ticket, void* stream) {
  LOCK_CTX(ctx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return pnpx::guarded(ctx, s, [&]() -> int {
    PNPX_TRY(check_common(grad_vars_out, grad_vars_in, y0, mask, sigma_d, B, H, W, T, param_stride));
    if (T > 0 && (!saved || !grad_sigma_d || !grad_mu || !grad_lamda || !work || !mu || !lamda)) {
      set_error("csmri_redadmm_backward: null pointer");
      return PNPX_ERR_ARG;
    }
    const int HW = H * W;
    const size_t is = 3 * (size_t)HW, n = (size_t)B * HW;
    PNPX_HIP(hipMemcpyAsync(grad_vars_in, grad_vars_out, sizeof(float2) * is * B, hipMemcpyDeviceToDevice, s));
    if (T == 0) return PNPX_OK;
    FftPlan2D P;
    PNPX_TRY(make_fft_plan(ctx, B, H, W, true, &P));
    float2* g = reinterpret_cast<float2*>(grad_vars_in);
    float *gxh = work, *gd = work + n, *contrib = work + 2 * n, *contrib_l = work + 3 * n;
    const float2* saved_k = reinterpret_cast<const float2*>(saved + (size_t)T * n);
    const float2* saved_q1 = reinterpret_cast<const float2*>(saved + 3 * (size_t)T * n);
    const float2* saved_q2 = reinterpret_cast<const float2*>(saved + 5 * (size_t)T * n);
    for (int i = T - 1; i >= 0; --i) {
      Scratch S;
      PNPX_TRY(get_scratch(ctx, B, H, W, &S));
      StoreC kst{S.k, H, W};
      LoadC kld{S.k, H, W};
      CSlot gzc{g + HW, is, W, HW}, guc{g + 2 * HW, is, W, HW};
      PNPX_TRY((launch_rows<false>(P, LoadSlotMinusSlot{gzc, guc}, kst, s)));
      KSpace ks{reinterpret_cast<const float2*>(y0), mask, mu + i, param_stride, W, HW};
      PNPX_TRY((launch_cols<false, true>(P, kld, MidBlendAdjoint{ks, saved_k + (size_t)i * n, contrib}, kst, s)));
      PNPX_TRY((launch_rows<true>(P, kld,
                                  StoreRedAdjoint{Slot{g, is, W, HW}, Slot{g + HW, is, W, HW}, Slot{g + 2 * HW, is, W, HW},
                                                  saved_q1 + (size_t)i * n, saved_q2 + (size_t)i * n, mu + i, lamda + i,
                                                  param_stride, RealImg{gxh, W, HW}, contrib, contrib_l},
                                  s)));
      hipLaunchKernelGGL(item_sum_kernel, dim3(B), dim3(256), 0, s, contrib, grad_mu + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
      hipLaunchKernelGGL(item_sum_kernel, dim3(B), dim3(256), 0, s, contrib_l, grad_lamda + (size_t)i * B, HW);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(unet_denoise_backward_ticket(ctx, saved + (size_t)i * n, sigma_d + i, param_stride, gxh, gd,
                                            grad_sigma_d + (size_t)i * B, B, H, W, s, ticket ? ticket + i : 0));
      hipLaunchKernelGGL(red_adjoint_finish_kernel, g1(n), dim3(256), 0, s, gd, g, is, HW, B);
      PNPX_LAUNCH_CHECK();
    }
    return PNPX_OK;
  });
}
