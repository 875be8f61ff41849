// Vector-Jacobian product of the UNet denoiser wrt its input image and its noise level.
//
// the reference trains its policy THROUGH the solver (PnPEnv.forward under autograd,
// tfpnp/env/base.py:193-206, called from tfpnp/trainer/mddpg/trainer.py:171-192): gradients of the reward flow
// through UNetDenoiser2D.forward (tfpnp/pnp/denoiser/base.py:23-32; weights frozen) into sigma_d and into the
// image argument.  This file is that backward pass:
//     (gx, gsigma) = J^T g,   out = clamp(in[:, :1] + UNet(cat[x, sigma*1]), 0, 1)
// Strategy: gradient checkpointing per call -- the forward pass is RE-COMPUTED here (same kernel family and arena as
// the inference forward, all activations kept) and then back-propagated layer by layer, so nothing has to be kept
// alive between the forward and the backward of the autograd graph.  Gradients are fp32 planar tensors in a third arena with the same zero-border
// layout, so every input-gradient convolution is the SAME MFMA kernel run on transposed, tap-flipped weights
// (pack_conv_weights_transposed) with the LeakyReLU derivative of the saved activation fused in its epilogue.
#include "common.h"
#include "conv3x3.h"
#include "conv_hs.h"
#include "hs_rec.h"
#include "unet_plan.h"
#include "grad_common.h"

namespace pnpx {
namespace {

inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

__device__ __forceinline__ float dlrelu(float a) { return a > 0.f ? 1.f : 0.2f; }

// A saved forward activation: fp32 padded planar (conv_mode 0) or half-split HS8 records (conv_hs.hip: [B][C/8][H+2][W+2]
// records of hi[8] | lo[8] f16, value * 16 = hi + lo).  Only signs and orderings are needed here, so the scale drops out.
struct SavedAct {
  const void* p;
  int hs;
};
__device__ __forceinline__ float act_at(const SavedAct& A, size_t b, int C, int c, int H, int W, int y, int x) {
  if (A.hs) {
    const size_t rec = ((b * (C >> 3) + (c >> 3)) * (H + 2) + (y + 1)) * (size_t)(W + 2) + (x + 1);
    const _Float16* r = reinterpret_cast<const _Float16*>(static_cast<const char*>(A.p) + rec * 32);
    return (float)r[c & 7] + (float)r[8 + (c & 7)];
  }
  return static_cast<const float*>(A.p)[((b * C + c) * (size_t)padded_h(H) + (y + 1)) * padded_w(W) + x + PADL];
}

// d/d(feat) of  out = clamp(x + sum_c w[c] feat[c] + b):  g_feat[c] = w[c] * g_out * 1[0 <= pre <= 1] * lrelu'(feat[c])
// (feat = y[0], the last conv's activation; the result is the gradient wrt that conv's PRE-activation).
// Also emits g_res = g_out * 1[0 <= pre <= 1], the gradient reaching the residual path in[:, :1].
__global__ __launch_bounds__(256) void outc_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ pre,
                                                       const float* __restrict__ w, SavedAct feat,
                                                       float* __restrict__ g_feat, float* __restrict__ g_res, int H,
                                                       int W, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  const size_t t = i / W;
  const int y = (int)(t % H);
  const size_t b = t / H;
  const float p = pre[i];
  const float g = (p >= 0.f && p <= 1.f) ? g_out[i] : 0.f;   // torch.clamp's backward mask is inclusive (min <= x <= max)
  g_res[i] = g;
  const int Hp = padded_h(H), Wp = padded_w(W);
  const size_t o = b * 32 * (size_t)Hp * Wp + (size_t)(y + 1) * Wp + x + PADL;
#pragma unroll 4
  for (int c = 0; c < 32; ++c) {
    const size_t oc = o + (size_t)c * Hp * Wp;
    g_feat[oc] = w[c] * g * dlrelu(act_at(feat, b, 32, c, H, W, y, x));
  }
}

// Adjoint of bilinear x2 (align_corners) restricted to the channel range [c_off, c_off + C) of the concat gradient,
// as a gather: every source pixel sums the destination pixels that interpolate from it; then x lrelu'(saved source).
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ gcat, int Ccat, int c_off,
                                                           SavedAct src_act, float* __restrict__ g_src,
                                                           int C, int h, int w, int Ht, int Wt, float sy, float sx,
                                                           size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xs = (int)(i % w);
  size_t t = i / w;
  const int ys = (int)(t % h);
  t /= h;
  const int c = (int)(t % C);
  const size_t b = t / C;
  const int H = 2 * h, W = 2 * w;
  const int Hp = padded_h(Ht), Wp = padded_w(Wt), hp = padded_h(h), wp = padded_w(w);
  const float* g = gcat + (b * Ccat + c_off + c) * (size_t)Hp * Wp;
  // candidate destination rows/cols: those whose floor(s*dst) is ys-1 or ys  (s ~ 0.5 => at most 5 candidates)
  const int ylo = max(0, 2 * ys - 3), yhi = min(H - 1, 2 * ys + 3);
  const int xlo = max(0, 2 * xs - 3), xhi = min(W - 1, 2 * xs + 3);
  float acc = 0.f;
  for (int yd = ylo; yd <= yhi; ++yd) {
    const float fy = sy * yd;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly = fy - y0;
    float wy = 0.f;
    if (y0 == ys) wy += 1.f - ly;
    if (y1 == ys) wy += ly;
    if (wy == 0.f) continue;
    for (int xd = xlo; xd <= xhi; ++xd) {
      const float fx = sx * xd;
      const int x0 = (int)fx;
      const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float lx = fx - x0;
      float wx = 0.f;
      if (x0 == xs) wx += 1.f - lx;
      if (x1 == xs) wx += lx;
      if (wx != 0.f) acc += wy * wx * g[(size_t)(yd + 1) * Wp + xd + PADL];
    }
  }
  const size_t o = (b * C + c) * (size_t)hp * wp + (size_t)(ys + 1) * wp + xs + PADL;
  g_src[o] = acc * dlrelu(act_at(src_act, b, C, c, h, w, ys, xs));
}

// r6: the same adjoint from an LDS window (fp32 family; at 48 x 256^2 the plain kernel above spent 1.4 ms on the full-resolution level --
// 25 scattered L1 reads and 49 loop iterations per source pixel -- 2.4 ms of the 13 ms backward pass per iteration).  A workgroup = one
// 16 x 16 tile of source pixels of one (image, channel): the 38 x 38 window of destination pixels is read once, coalesced, into two
// column-parity planes (the 16 lanes of a tile row read every second window column: consecutive words of one plane); the seven candidate
// row / column weights are computed once per thread with the plain kernel's float arithmetic and the 7 x 7 loop keeps its accumulation
// order, so the result is bit-identical to upsample_bwd_kernel.
// Tile = 16 rows x TX columns of source pixels (TX = 64 where the level is that wide: 536-byte window rows, 1.24x the useful bytes read;
// 16 x 16 tiles read 1.41x in 152-byte pieces), 256 threads x TX / 16 pixels each.
constexpr int UBF_T = 16;
template <int TX>
__global__ __launch_bounds__(256) void upsample_bwd_lds_kernel(const float* __restrict__ gcat, int Ccat, int c_off, const float* __restrict__ src_act,
                                                               float* __restrict__ g_src, int C, int h, int w, int Ht, int Wt, float sy, float sx) {
  constexpr int WH = 2 * UBF_T + 6, WW = 2 * TX + 6, WP = WW / 2 + 1;      // window rows / columns, plane row stride (words)
  __shared__ float win[2][WH][WP];
  const int c = blockIdx.z % C;
  const size_t b = blockIdx.z / C;
  const int xs0 = blockIdx.x * TX, ys0 = blockIdx.y * UBF_T;
  const int H = 2 * h, W = 2 * w;
  const int Hp = padded_h(Ht), Wp = padded_w(Wt), hp = padded_h(h), wp = padded_w(w);
  const float* g = gcat + (b * Ccat + c_off + c) * (size_t)Hp * Wp;
  const int wy0 = 2 * ys0 - 3, wx0 = 2 * xs0 - 3;            // window origin (may be negative: clamped entries carry weight 0)
  for (int k = threadIdx.x; k < WH * WW; k += 256) {
    const int ry = k / WW, rx = k - ry * WW;
    const int yd = min(max(wy0 + ry, 0), H - 1), xd = min(max(wx0 + rx, 0), W - 1);
    win[rx & 1][ry][rx >> 1] = g[(size_t)(yd + 1) * Wp + xd + PADL];
  }
  __syncthreads();
  const int ys = ys0 + (threadIdx.x / 16);
  if (ys >= h) return;
  float wyv[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int yd = 2 * ys - 3 + i;
    wyv[i] = 0.f;
    if (yd >= 0 && yd < H) {
      const float fy = sy * yd;
      const int y0 = (int)fy;
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
      const float ly = fy - y0;
      if (y0 == ys) wyv[i] += 1.f - ly;
      if (y1 == ys) wyv[i] += ly;
    }
  }
  const int oy = 2 * (ys - ys0);
#pragma unroll
  for (int kx = 0; kx < TX / 16; ++kx) {
    const int xs = xs0 + (threadIdx.x % 16) + 16 * kx;
    if (xs >= w) continue;
    float wxv[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int xd = 2 * xs - 3 + i;
      wxv[i] = 0.f;
      if (xd >= 0 && xd < W) {
        const float fx = sx * xd;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float lx = fx - x0;
        if (x0 == xs) wxv[i] += 1.f - lx;
        if (x1 == xs) wxv[i] += lx;
      }
    }
    const int ox = 2 * (xs - xs0);       // window position of candidate (0, 0)
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (wyv[i] == 0.f) continue;
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (wxv[j] != 0.f) acc += wyv[i] * wxv[j] * win[(ox + j) & 1][oy + i][(ox + j) >> 1];
    }
    const size_t o = (b * C + c) * (size_t)hp * wp + (size_t)(ys + 1) * wp + xs + PADL;
    g_src[o] = acc * dlrelu(src_act[o]);
  }
}

// Gradient reaching an encoder output x[l] (H x W, C channels): the skip part of the decoder's concat gradient
// (channels [0, C) of gcat) plus, for l < 4, the max-pool routing of g_pool (first maximum in scan order, like ATen),
// all times lrelu'(x[l]).
__global__ __launch_bounds__(256) void skip_pool_merge_kernel(const float* __restrict__ gcat, int Ccat,
                                                              const float* __restrict__ g_pool,
                                                              SavedAct xact, float* __restrict__ g_x,
                                                              int C, int H, int W, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  size_t t = i / W;
  const int y = (int)(t % H);
  t /= H;
  const int c = (int)(t % C);
  const size_t b = t / C;
  const int Hp = padded_h(H), Wp = padded_w(W);
  const size_t plane = (size_t)Hp * Wp;
  const size_t o = (b * C + c) * plane + (size_t)(y + 1) * Wp + x + PADL;
  float g = gcat ? gcat[(b * Ccat + c) * plane + (size_t)(y + 1) * Wp + x + PADL] : 0.f;
  if (g_pool) {
    const int Ho = H / 2, Wo = W / 2;
    const int yo = y >> 1, xo = x >> 1;
    if (yo < Ho && xo < Wo) {
      const float v00 = act_at(xact, b, C, c, H, W, 2 * yo, 2 * xo), v01 = act_at(xact, b, C, c, H, W, 2 * yo, 2 * xo + 1);
      const float v10 = act_at(xact, b, C, c, H, W, 2 * yo + 1, 2 * xo);
      const float v11 = act_at(xact, b, C, c, H, W, 2 * yo + 1, 2 * xo + 1);
      int arg = 0;
      float m = v00;
      if (v01 > m) { m = v01; arg = 1; }
      if (v10 > m) { m = v10; arg = 2; }
      if (v11 > m) { m = v11; arg = 3; }
      if (arg == ((y & 1) * 2 + (x & 1))) {
        const int Hpo = padded_h(Ho), Wpo = padded_w(Wo);
        g += g_pool[(b * C + c) * (size_t)Hpo * Wpo + (size_t)(yo + 1) * Wpo + xo + PADL];
      }
    }
  }
  g_x[o] = g * dlrelu(act_at(xact, b, C, c, H, W, y, x));
}

// r6: the same for fp32 planar tensors with even H, W, one thread per 2 x 2 block (the max-pool window): the four saved activations, the
// two gradient rows and the pooled gradient are read once per block as 8-byte pieces instead of once per pixel (1.1 ms per backward pass
// at 48 x 256^2 before); same arithmetic per pixel, bit-identical.
__global__ __launch_bounds__(256) void skip_pool_merge_f32_kernel(const float* __restrict__ gcat, int Ccat, const float* __restrict__ g_pool,
                                                                  const float* __restrict__ xact, float* __restrict__ g_x, int C, int H, int W,
                                                                  size_t nblk) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  const int Wo = W / 2, Ho = H / 2;
  const int xo = (int)(i % Wo);
  size_t t = i / Wo;
  const int yo = (int)(t % Ho);
  t /= Ho;
  const int c = (int)(t % C);
  const size_t b = t / C;
  const int Hp = padded_h(H), Wp = padded_w(W);
  const size_t plane = (size_t)Hp * Wp;
  const size_t o = (b * C + c) * plane + (size_t)(2 * yo + 1) * Wp + 2 * xo + PADL;
  const f2 a0 = *reinterpret_cast<const f2*>(xact + o), a1 = *reinterpret_cast<const f2*>(xact + o + Wp);
  f2 g0 = {0.f, 0.f}, g1v = {0.f, 0.f};
  if (gcat) {
    const size_t oc = (b * Ccat + c) * plane + (size_t)(2 * yo + 1) * Wp + 2 * xo + PADL;
    g0 = *reinterpret_cast<const f2*>(gcat + oc);
    g1v = *reinterpret_cast<const f2*>(gcat + oc + Wp);
  }
  if (g_pool) {
    int arg = 0;
    float m = a0[0];
    if (a0[1] > m) { m = a0[1]; arg = 1; }
    if (a1[0] > m) { m = a1[0]; arg = 2; }
    if (a1[1] > m) { m = a1[1]; arg = 3; }
    const int Hpo = padded_h(Ho), Wpo = padded_w(Wo);
    const float gpv = g_pool[(b * C + c) * (size_t)Hpo * Wpo + (size_t)(yo + 1) * Wpo + xo + PADL];
    if (arg == 0) g0[0] += gpv;
    else if (arg == 1) g0[1] += gpv;
    else if (arg == 2) g1v[0] += gpv;
    else g1v[1] += gpv;
  }
  *reinterpret_cast<f2*>(g_x + o) = (f2){g0[0] * dlrelu(a0[0]), g0[1] * dlrelu(a0[1])};
  *reinterpret_cast<f2*>(g_x + o + Wp) = (f2){g1v[0] * dlrelu(a1[0]), g1v[1] * dlrelu(a1[1])};
}

// gx = g_in0[:, 0] + g_res;  gsigma[b] = sum over pixels of g_in0[:, 1]   (two-stage, deterministic)
__global__ __launch_bounds__(256) void input_grad_kernel(const float* __restrict__ g_in0, int Cg,
                                                         const float* __restrict__ g_res, float* __restrict__ gx,
                                                         float* __restrict__ part, int H, int W) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int n = H * W, per = (n + SIG_CHUNKS - 1) / SIG_CHUNKS;
  const int lo = chunk * per, hi = min(n, lo + per);
  const int Hp = padded_h(H), Wp = padded_w(W);
  const float* g0 = g_in0 + (size_t)b * Cg * Hp * Wp;
  const float* g1c = g0 + (size_t)Hp * Wp;
  float acc = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const int y = i / W, x = i - y * W;
    const size_t o = (size_t)(y + 1) * Wp + x + PADL;
    gx[(size_t)b * n + i] = g0[o] + g_res[(size_t)b * n + i];
    acc += g1c[o];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[b * SIG_CHUNKS + chunk] = (w[0] + w[1]) + (w[2] + w[3]);
}


// ------------------------------------------------------------------------------------ half-split (HS8) variants
// conv_mode 1: gradients live in HS8 tensors too, so that the 27 adjoint convolutions run on the f16x3 MFMA kernel
// (conv_hs.hip, LeakyReLU' from the saved activation's sign in its epilogue).  f16 has a narrow exponent range, so the
// whole backward pass runs on grad_out / m with m = the power of two >= max|grad_out| and the two results are
// multiplied by m at the end (exact: power-of-two scaling commutes with every linear step).  One thread per record.
__global__ __launch_bounds__(256) void outc_bwd_hs_kernel(const float* __restrict__ g_out, const float* __restrict__ pre,
                                                          const float* __restrict__ w, const HsRec* __restrict__ feat,
                                                          HsRec* __restrict__ g_feat, float* __restrict__ g_res,
                                                          const float2* __restrict__ sc, int H, int W, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = (int)(i & 3);
  const size_t pix = i >> 2;
  const int x = (int)(pix % W);
  const size_t t = pix / W;
  const int y = (int)(t % H);
  const size_t b = t / H;
  const float p = pre[pix];
  const float gn = (p >= 0.f && p <= 1.f) ? g_out[pix] * sc->x : 0.f;
  if (g == 0) g_res[pix] = gn;
  const size_t r = ((b * 4 + g) * (H + 2) + (y + 1)) * (size_t)(W + 2) + (x + 1);
  float f[8], o[8];
  hs_unpack(feat[r], f);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = w[g * 8 + k] * gn * dlrelu(f[k]) * HS_ASCALE;
  g_feat[r] = hs_pack(o);
}

// Adjoint of the bilinear x2 (align_corners) up-sampling, gather form (deterministic, no atomics), times LeakyReLU' of
// the saved low-resolution activation.  A workgroup = one 16 x 16 tile of low-resolution pixels of one (image, group); the
// (2*16 + 6)^2 high-resolution gradient records every one of them can touch are staged ONCE in LDS (coalesced 16-byte
// pieces) instead of ~28 scattered 32-byte global reads per thread (r2: 0.94 ms per VJP, ~7x read amplification); every
// thread then runs the same (yd, xd) double loop in the same order as before, so results are bit-identical.
constexpr int UB_T = 16, UB_W = 2 * UB_T + 6;
__global__ __launch_bounds__(256) void upsample_bwd_hs_kernel(const HsRec* __restrict__ gcat, int Gcat, int g_off,
                                                              const HsRec* __restrict__ src_act,
                                                              HsRec* __restrict__ g_src, int Gs, int h, int w, int Ht,
                                                              int Wt, float sy, float sx) {
  __shared__ uint4 win[UB_W * UB_W * 2];
  const int g = blockIdx.z % Gs;
  const size_t b = blockIdx.z / Gs;
  const int xs0 = blockIdx.x * UB_T, ys0 = blockIdx.y * UB_T;
  const int H = 2 * h, W = 2 * w;
  const HsRec* gc = gcat + (b * Gcat + g_off + g) * (size_t)(Ht + 2) * (Wt + 2);
  const int wy0 = 2 * ys0 - 3, wx0 = 2 * xs0 - 3;            // window origin (may be negative: clamped rows are never used)
  const uint4* gsrc = reinterpret_cast<const uint4*>(gc);
  for (int k = threadIdx.x; k < UB_W * UB_W * 2; k += 256) {
    const int rec = k >> 1, piece = k & 1;
    const int ry = rec / UB_W, rx = rec - ry * UB_W;
    const int yd = min(max(wy0 + ry, 0), H - 1), xd = min(max(wx0 + rx, 0), W - 1);
    // four planes [hi | lo][even | odd window column] of [UB_W][UB_W / 2] 16-byte pieces: the 16 lanes of a row (consecutive
    // low-resolution pixels = every second window column) then read consecutive pieces -- with whole 32-byte records at a
    // 64-byte lane stride the same reads were four-way bank-conflicted
    win[((piece * 2 + (rx & 1)) * UB_W + ry) * (UB_W / 2) + (rx >> 1)] = gsrc[((size_t)(yd + 1) * (Wt + 2) + xd + 1) * 2 + piece];
  }
  __syncthreads();
  const int xs = xs0 + (threadIdx.x % UB_T), ys = ys0 + (threadIdx.x / UB_T);
  if (xs >= w || ys >= h) return;
  auto window = [&](int ry, int rx, float v[8]) {
    HsRec r;
    r.hi = *reinterpret_cast<const h8v*>(&win[((rx & 1) * UB_W + ry) * (UB_W / 2) + (rx >> 1)]);
    r.lo = *reinterpret_cast<const h8v*>(&win[((2 + (rx & 1)) * UB_W + ry) * (UB_W / 2) + (rx >> 1)]);
    hs_unpack(r, v);
  };
  // interpolation weights of the 7 candidate high-resolution rows / columns 2*ys-3 .. 2*ys+3 towards THIS low-resolution
  // pixel, computed once (exactly the forward kernel's float arithmetic); the 7 x 7 loop below is fully unrolled and a
  // wave skips the (row, column) pairs none of its lanes needs -- same accumulation order as the plain double loop.
  float wyv[7], wxv[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int yd = 2 * ys - 3 + i, xd = 2 * xs - 3 + i;
    wyv[i] = wxv[i] = 0.f;
    if (yd >= 0 && yd < H) {
      const float fy = sy * yd;
      const int y0 = (int)fy;
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
      const float ly = fy - y0;
      if (y0 == ys) wyv[i] += 1.f - ly;
      if (y1 == ys) wyv[i] += ly;
    }
    if (xd >= 0 && xd < W) {
      const float fx = sx * xd;
      const int x0 = (int)fx;
      const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float lx = fx - x0;
      if (x0 == xs) wxv[i] += 1.f - lx;
      if (x1 == xs) wxv[i] += lx;
    }
  }
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const int oy = 2 * (ys - ys0), ox = 2 * (xs - xs0);       // window position of candidate (0, 0)
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (wyv[i] != 0.f && wxv[j] != 0.f) {
        float v[8];
        window(oy + i, ox + j, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += wyv[i] * wxv[j] * v[k];
      }
    }
  }
  const size_t r = ((b * Gs + g) * (h + 2) + (ys + 1)) * (size_t)(w + 2) + (xs + 1);
  float a[8];
  hs_unpack(src_act[r], a);
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] *= dlrelu(a[k]);
  g_src[r] = hs_pack(acc);
}

__global__ __launch_bounds__(256) void skip_pool_merge_hs_kernel(const HsRec* __restrict__ gcat, int Gcat,
                                                                 const HsRec* __restrict__ g_pool,
                                                                 const HsRec* __restrict__ xact, HsRec* __restrict__ g_x,
                                                                 int G, int H, int W, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  size_t t = i / W;
  const int y = (int)(t % H);
  t /= H;
  const int g = (int)(t % G);
  const size_t b = t / G;
  const size_t plane = (size_t)(H + 2) * (W + 2);
  const size_t r = (b * G + g) * plane + (size_t)(y + 1) * (W + 2) + (x + 1);
  float gv[8], a[8];
  if (gcat) {
    hs_unpack(gcat[(b * Gcat + g) * plane + (size_t)(y + 1) * (W + 2) + (x + 1)], gv);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) gv[k] = 0.f;
  }
  hs_unpack(xact[r], a);
  if (g_pool) {
    const int Ho = H / 2, Wo = W / 2, yo = y >> 1, xo = x >> 1;
    if (yo < Ho && xo < Wo) {
      const HsRec* w0 = xact + (b * G + g) * plane + (size_t)(2 * yo + 1) * (W + 2) + (2 * xo + 1);
      float v00[8], v01[8], v10[8], v11[8], gp[8];
      hs_unpack(w0[0], v00);
      hs_unpack(w0[1], v01);
      hs_unpack(w0[W + 2], v10);
      hs_unpack(w0[W + 3], v11);
      hs_unpack(g_pool[((b * G + g) * (Ho + 2) + (yo + 1)) * (size_t)(Wo + 2) + (xo + 1)], gp);
      const int me = (y & 1) * 2 + (x & 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int arg = 0;
        float m = v00[k];
        if (v01[k] > m) { m = v01[k]; arg = 1; }
        if (v10[k] > m) { m = v10[k]; arg = 2; }
        if (v11[k] > m) { m = v11[k]; arg = 3; }
        if (arg == me) gv[k] += gp[k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) gv[k] *= dlrelu(a[k]);
  g_x[r] = hs_pack(gv);
}

// gradient arena: one tensor per forward activation + the concat gradients of the four decoder blocks
struct GradPlan {
  Act in0;        // 32 channels (the adjoint of the first conv is padded from 2 to 32 output channels)
  Act a[5], b[5], x[5], p[5], y[4], cat[4];
  size_t total = 0;
};
GradPlan make_grad_plan(int mode, int capB, int H, int W) {
  GradPlan P;
  size_t off = 0;
  auto add = [&](Act& d, int C, int h, int w) {
    d.off = off;
    d.C = C;
    d.H = h;
    d.W = w;
    off += act_bytes_per_image(mode, C, h, w) * (size_t)capB;
    off = (off + 255) & ~(size_t)255;
  };
  add(P.in0, 32, H, W);
  for (int l = 0; l < 5; ++l) {
    const int h = H >> l, w = W >> l, c = 32 << l;
    add(P.a[l], c, h, w);
    add(P.b[l], c, h, w);
    add(P.x[l], c, h, w);
    if (l >= 1) add(P.p[l], c / 2, h, w);
    if (l <= 3) {
      add(P.y[l], c, h, w);
      add(P.cat[l], 3 * c, h, w);
    }
  }
  P.total = off + (1u << 20);
  return P;
}

}  // namespace

int unet_denoise_backward(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, const float* grad_out,
                          float* grad_x, float* grad_sigma, int B, int H, int W, hipStream_t s, UNetArena* cached,
                          const float* cached_pre) {
  if (ctx->drunet.loaded) {
    if (B <= 0 || !grad_out || !grad_x || !grad_sigma) {
      set_error("denoiser backward: need B > 0 and non-null gradients");
      return PNPX_ERR_ARG;
    }
    return drunet_denoise_backward(ctx, x, sigma, sigma_stride, grad_out, grad_x, grad_sigma, B, H, W, s);
  }
  if (!ctx->has_weights) {
    set_error("denoiser backward called before pnpx_unet_load");
    return PNPX_ERR_NO_WEIGHTS;
  }
  if (B <= 0 || H < 16 || W < 16) {
    set_error("denoiser backward: need B > 0 and H, W >= 16");
    return PNPX_ERR_SHAPE;
  }
  const size_t npix = (size_t)B * H * W;
  // scratch: recomputed clamped output (unused), pre-clamp output, residual gradient, sigma partials
  void* sp;
  PNPX_TRY(ctx_scratch(ctx, (3 * npix + (size_t)B * SIG_CHUNKS) * sizeof(float) + 8192, &sp));
  float* out_tmp = static_cast<float*>(sp);
  float* pre = out_tmp + npix;
  float* g_res = pre + npix;
  float* part = g_res + npix;
  float2* gscale = reinterpret_cast<float2*>(part + (((size_t)B * SIG_CHUNKS + 63) & ~(size_t)63));   // {1/m, m}
  unsigned* gmax_bits = reinterpret_cast<unsigned*>(gscale + 1);

  // 1. recompute the forward pass with the context's own kernel family into the context's activation arena, keeping
  //    every activation (no fused network tail): the backward needs signs (LeakyReLU'), orderings (max-pool routing)
  //    and the pre-clamp output only, which the half-split activations give exactly as well as fp32 ones.
  //    (skipped when the caller kept them: `cached` is the arena of a keep_all forward of this very input)
  const int fmode = cached ? cached->mode : ctx->conv_mode;
  const bool hs = (fmode == CONV_HS);
  UNetArena& fa = cached ? *cached : ctx->arena;
  if (cached) {
    if (!(B <= cached->capB && H == cached->capH && W == cached->capW && cached_pre)) {
      set_error("denoiser backward: the cached activations do not belong to a %dx%dx%d forward", B, H, W);
      return PNPX_ERR_ARG;
    }
    pre = const_cast<float*>(cached_pre);
  } else {
    PNPX_TRY(unet_denoise(ctx, x, sigma, sigma_stride, out_tmp, pre, B, H, W, s, nullptr, &ctx->arena, fmode, true));
  }
  const UNetPlan F = make_plan(fmode, fa.capB, H, W);
  char* FA = static_cast<char*>(fa.buf.p);

  // 2. gradient arena (zero borders: gradients are convolution INPUTS of the adjoint convs)
  {
    UNetArena& ga = ctx->arena_grad;
    if (!(B <= ga.capB && H == ga.capH && W == ga.capW && ga.mode == fmode)) {
      const bool same = (H == ga.capH && W == ga.capW);
      const int nb = same ? (B > ga.capB ? B : ga.capB) : B;
      const GradPlan GP = make_grad_plan(fmode, nb, H, W);
      PNPX_HIP(hipDeviceSynchronize());
      if (ga.buf.bytes < GP.total) {
        if (ga.buf.p) PNPX_HIP(hipFree(ga.buf.p));
        ga = UNetArena();
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, GP.total);
        if (e != hipSuccess) {
          set_error("gradient arena allocation of %zu bytes failed: %s", GP.total, hipGetErrorString(e));
          return PNPX_ERR_ALLOC;
        }
        ga.buf.p = p;
        ga.buf.bytes = GP.total;
      }
      PNPX_HIP(hipMemset(ga.buf.p, 0, GP.total));
      PNPX_HIP(hipDeviceSynchronize());
      ga.capB = nb;
      ga.capH = H;
      ga.capW = W;
      ga.mode = fmode;
    }
  }
  const GradPlan G = make_grad_plan(fmode, ctx->arena_grad.capB, H, W);
  char* GA = static_cast<char*>(ctx->arena_grad.buf.p);

  if (hs) {
    // ---- half-split backward: HS8 gradients, f16x3 MFMA adjoint convolutions
    PNPX_HIP(hipMemsetAsync(gmax_bits, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(absmax_kernel, dim3(512), dim3(256), 0, s, grad_out, npix, gmax_bits);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, s, gmax_bits, gscale);
    // the adjoint chain over images lo .. lo + B - 1 on stream s: the whole batch, or one of two launch chains (r5: like the forward's,
    // unet.hip launch_chains; the launch table plans for the chain beside it: ConvHsFuse::share).  The gradient scale above is the whole
    // batch's, so per-image results do not depend on the slicing.
    auto run_hs = [&](int lo, int B, hipStream_t s, int share) -> int {
    const size_t npix = (size_t)B * H * W, px0 = (size_t)lo * H * W;
    auto grec = [&](const Act& d) { return reinterpret_cast<HsRec*>(GA + d.off + (size_t)lo * act_bytes_per_image(CONV_HS, d.C, d.H, d.W)); };
    auto frec = [&](const Act& d) { return reinterpret_cast<const HsRec*>(FA + d.off + (size_t)lo * act_bytes_per_image(CONV_HS, d.C, d.H, d.W)); };
    hipLaunchKernelGGL(outc_bwd_hs_kernel, g1(npix * 4), dim3(256), 0, s, grad_out + px0, pre + px0, ctx->outc_w, frec(F.y[0]),
                       grec(G.y[0]), g_res + px0, gscale, H, W, npix * 4);
    PNPX_LAUNCH_CHECK();
    auto convH = [&](int li, const Act& gin, const Act& gout, const Act* saved) -> int {
      const ConvLayerHsDev& D = ctx->conv_hs_bwd[li];
      ConvLayerHs Lh;
      Lh.cin = D.cin;
      Lh.cout = D.cout;
      Lh.cin_pad = D.cin_pad;
      Lh.mt = D.mt;
      Lh.w = D.w;
      Lh.b = ctx->zero_bias;
      Lh.inv_scale = D.inv_scale;
      if (gin.C != D.cin || gout.C != D.cout) {
        set_error("backward: gradient tensor channels %d -> %d do not match adjoint layer %d (%d -> %d)", gin.C, gout.C,
                  li, D.cin, D.cout);
        return PNPX_ERR_SHAPE;
      }
      ConvHsFuse f;
      f.dmask = saved ? FA + saved->off + (size_t)lo * act_bytes_per_image(CONV_HS, saved->C, saved->H, saved->W) : nullptr;
      f.share = share;
      f.slope = saved ? 0.2f : 1.0f;
      f.range_flag = ctx->opt_range_guard ? ctx->range_flag_dev : nullptr;
      return launch_conv_hs(Lh, reinterpret_cast<const char*>(grec(gin)), gin.C / 8, nullptr, 0, reinterpret_cast<char*>(grec(gout)), B, gout.H,
                            gout.W, f, s);
    };
    for (int l = 0; l <= 3; ++l) {
      const int li = 15 + 3 * (3 - l);
      PNPX_TRY(convH(li + 2, G.y[l], G.b[l], &F.db[l]));
      PNPX_TRY(convH(li + 1, G.b[l], G.a[l], &F.da[l]));
      PNPX_TRY(convH(li, G.a[l], G.cat[l], nullptr));
      const Act& below_f = (l == 3) ? F.x[4] : F.y[l + 1];
      const Act& below_g = (l == 3) ? G.x[4] : G.y[l + 1];
      const int h = below_f.H, w = below_f.W, Gs = below_f.C / 8;
      const float sy = (2 * h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
      const float sx = (2 * w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
      hipLaunchKernelGGL(upsample_bwd_hs_kernel, dim3((w + UB_T - 1) / UB_T, (h + UB_T - 1) / UB_T, (unsigned)(B * Gs)), dim3(256),
                         0, s, grec(G.cat[l]), G.cat[l].C / 8, F.x[l].C / 8, frec(below_f), grec(below_g), Gs, h, w,
                         G.cat[l].H, G.cat[l].W, sy, sx);
      PNPX_LAUNCH_CHECK();
    }
    for (int l = 4; l >= 0; --l) {
      if (l < 4) {
        const size_t n = (size_t)B * (F.x[l].C / 8) * F.x[l].H * F.x[l].W;
        hipLaunchKernelGGL(skip_pool_merge_hs_kernel, g1(n), dim3(256), 0, s, grec(G.cat[l]), G.cat[l].C / 8,
                           grec(G.p[l + 1]), frec(F.x[l]), grec(G.x[l]), F.x[l].C / 8, F.x[l].H, F.x[l].W, n);
        PNPX_LAUNCH_CHECK();
      }
      PNPX_TRY(convH(3 * l + 2, G.x[l], G.b[l], &F.b[l]));
      PNPX_TRY(convH(3 * l + 1, G.b[l], G.a[l], &F.a[l]));
      PNPX_TRY(convH(3 * l, G.a[l], l == 0 ? G.in0 : G.p[l], nullptr));
    }
    hipLaunchKernelGGL(input_grad_hs_kernel, dim3(SIG_CHUNKS, B), dim3(256), 0, s, grec(G.in0), g_res + px0, grad_x + px0,
                       part + (size_t)lo * SIG_CHUNKS, gscale, H, W);
    PNPX_LAUNCH_CHECK();
    hipLaunchKernelGGL(sigma_grad_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part + (size_t)lo * SIG_CHUNKS, grad_sigma + lo, B);
    PNPX_LAUNCH_CHECK();
    return PNPX_OK;
    };
    const int chains = launch_chains(ctx, B, H, W);
    if (chains <= 1) return run_hs(0, B, s, 1);
    return fan_out_chains(ctx, chains, B, s, [&](int lo, int hi, hipStream_t st) -> int { return run_hs(lo, hi - lo, st, chains); });
  }

  // 3..6 over images lo .. lo + B - 1 on stream s: the whole batch, or one of two launch chains (r5, option fp32_chains; the fp32 family is
  // not power-capped, a second chain fills the first one's tails and partial rounds; per-image results do not depend on the slicing)
  auto run = [&](int lo, int B, hipStream_t s) -> int {
  const size_t npix = (size_t)B * H * W, px0 = (size_t)lo * H * W;
  const float* const grad_out_s = grad_out + px0;
  const float* const pre_s = pre + px0;
  float* const g_res_s = g_res + px0;
  float* const grad_x_s = grad_x ? grad_x + px0 : nullptr;
  float* const part_s = part + (size_t)lo * SIG_CHUNKS;
  float* const grad_sigma_s = grad_sigma ? grad_sigma + lo : nullptr;
  auto gp = [&](const Act& d) { return reinterpret_cast<float*>(GA + d.off + (size_t)lo * act_bytes_per_image(fmode, d.C, d.H, d.W)); };
  auto sa = [&](const Act& d) { return SavedAct{FA + d.off + (size_t)lo * act_bytes_per_image(fmode, d.C, d.H, d.W), 0}; };
  // 3. tail: clamp + residual + 1x1 out-conv -> gradient wrt the pre-activation of the last conv (y[0])
  hipLaunchKernelGGL(outc_bwd_kernel, g1(npix), dim3(256), 0, s, grad_out_s, pre_s, ctx->outc_w, sa(F.y[0]), gp(G.y[0]),
                     g_res_s, H, W, npix);
  PNPX_LAUNCH_CHECK();

  auto convT = [&](int li, const Act& gin, const Act& gout, const Act* saved) -> int {
    const float* dm = saved ? reinterpret_cast<const float*>(FA + saved->off + (size_t)lo * act_bytes_per_image(fmode, saved->C, saved->H, saved->W)) : nullptr;
    const char* dmh = nullptr;
    // r5: Winograd F(2x2,3x3) for the adjoint convolutions too (8-wave kernel, LeakyReLU' mask in its epilogue)
    if (!hs && ctx->opt_fp32_winograd && ((ctx->opt_fp32_wino8 >> li) & 1) && ctx->conv_wino_u_bwd[li] && gout.C == ctx->conv_bwd[li].cout &&
        conv3x3_wino8_ok(gin.C, 0, gout.C, gout.H, gout.W))
      return launch_conv3x3_wino8_grad(ctx->conv_wino_u_bwd[li], ctx->zero_bias, gout.C, gp(gin), gin.C, gp(gout), dm, 0.2f, B, gout.H, gout.W, s);
    return launch_conv3x3_grad(ctx->conv_bwd[li], gp(gin), gp(gout), dm, B, gout.H, gout.W, s, dmh);
  };

  // 4. decoder blocks, top (level 0) to bottom (level 3)
  for (int l = 0; l <= 3; ++l) {
    const int li = 15 + 3 * (3 - l);
    PNPX_TRY(convT(li + 2, G.y[l], G.b[l], &F.db[l]));  // through conv2, x lrelu'(decoder b)
    PNPX_TRY(convT(li + 1, G.b[l], G.a[l], &F.da[l]));  // through conv1, x lrelu'(decoder a)
    PNPX_TRY(convT(li, G.a[l], G.cat[l], nullptr));          // through conv0 -> gradient of cat[skip, up]
    // up part -> through the bilinear upsample -> pre-activation gradient of the tensor below
    const Act& below_f = (l == 3) ? F.x[4] : F.y[l + 1];
    const Act& below_g = (l == 3) ? G.x[4] : G.y[l + 1];
    const int h = below_f.H, w = below_f.W, Cb = below_f.C;
    const float sy = (2 * h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    const float sx = (2 * w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    const size_t n = (size_t)B * Cb * h * w;
    if (!hs && (size_t)B * Cb <= 65535) {      // r6: LDS-window form (bit-identical; fp32 planar activations)
      if (w >= 64)
        hipLaunchKernelGGL(upsample_bwd_lds_kernel<64>, dim3((w + 63) / 64, (h + UBF_T - 1) / UBF_T, (unsigned)(B * Cb)), dim3(256), 0, s, gp(G.cat[l]),
                           G.cat[l].C, F.x[l].C, static_cast<const float*>(sa(below_f).p), gp(below_g), Cb, h, w, G.cat[l].H, G.cat[l].W, sy, sx);
      else
        hipLaunchKernelGGL(upsample_bwd_lds_kernel<16>, dim3((w + 15) / 16, (h + UBF_T - 1) / UBF_T, (unsigned)(B * Cb)), dim3(256), 0, s, gp(G.cat[l]),
                           G.cat[l].C, F.x[l].C, static_cast<const float*>(sa(below_f).p), gp(below_g), Cb, h, w, G.cat[l].H, G.cat[l].W, sy, sx);
    } else {
      hipLaunchKernelGGL(upsample_bwd_kernel, g1(n), dim3(256), 0, s, gp(G.cat[l]), G.cat[l].C, F.x[l].C, sa(below_f),
                         gp(below_g), Cb, h, w, G.cat[l].H, G.cat[l].W, sy, sx, n);
    }
    PNPX_LAUNCH_CHECK();
  }
  // 5. encoder blocks, bottom (level 4) to top
  for (int l = 4; l >= 0; --l) {
    if (l < 4) {   // gradient reaching x[l]: skip part of the decoder concat + max-pool routing from level l+1
      const size_t n = (size_t)B * F.x[l].C * F.x[l].H * F.x[l].W;
      if (F.x[l].H % 2 == 0 && F.x[l].W % 2 == 0) {      // r6: one thread per max-pool window (bit-identical)
        hipLaunchKernelGGL(skip_pool_merge_f32_kernel, g1(n / 4), dim3(256), 0, s, gp(G.cat[l]), G.cat[l].C, gp(G.p[l + 1]),
                           static_cast<const float*>(sa(F.x[l]).p), gp(G.x[l]), F.x[l].C, F.x[l].H, F.x[l].W, n / 4);
      } else {
        hipLaunchKernelGGL(skip_pool_merge_kernel, g1(n), dim3(256), 0, s, gp(G.cat[l]), G.cat[l].C, gp(G.p[l + 1]),
                           sa(F.x[l]), gp(G.x[l]), F.x[l].C, F.x[l].H, F.x[l].W, n);
      }
      PNPX_LAUNCH_CHECK();
    }
    PNPX_TRY(convT(3 * l + 2, G.x[l], G.b[l], &F.b[l]));
    PNPX_TRY(convT(3 * l + 1, G.b[l], G.a[l], &F.a[l]));
    PNPX_TRY(convT(3 * l, G.a[l], l == 0 ? G.in0 : G.p[l], nullptr));
  }
  // 6. input gradients
  hipLaunchKernelGGL(input_grad_kernel, dim3(SIG_CHUNKS, B), dim3(256), 0, s, gp(G.in0), G.in0.C, g_res_s, grad_x_s, part_s, H,
                     W);
  PNPX_LAUNCH_CHECK();
  hipLaunchKernelGGL(sigma_grad_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part_s, grad_sigma_s, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
  };
  // the adjoint chain always forks exactly TWO launch chains when fp32_chains >= 2 (measured best, profiles/r5_wino8.md; the forward
  // honours larger n): documented in include/pnpx.h
  if (ctx->opt_fp32_chains >= 2 && B >= 2)
    return fan_out_chains(ctx, 2, B, s, [&](int lo, int hi, hipStream_t st) -> int { return run(lo, hi - lo, st); });
  return run(0, B, s);
}

}  // namespace pnpx
