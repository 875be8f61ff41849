// UNet(2,1) denoiser forward on gfx950: orchestration + the non-GEMM kernels.
//
// Replaces UNetDenoiser2D.forward (tfpnp/pnp/denoiser/base.py:23-32) and UNet.forward
// (tfpnp/pnp/denoiser/models/unet.py:52-66):
//   in = cat[x, sigma*1]; x1 = inc(in); x2..x5 = down(.) = ConvBlock(MaxPool2d(2)(.));
//   y = up(y_below, skip) = ConvBlock(cat[skip, bilinear_x2_align_corners(y_below)]);
//   out = clamp(in[:, :1] + Conv1x1(y), 0, 1).
// Activations live in the context's arena in the padded planar layout of common.h.  The channel concat is
// never materialised (the conv kernel reads two source tensors).
#include <cstdlib>

#include "common.h"
#include "conv3x3.h"
#include "conv_hs.h"
#include "conv_first.h"
#include "hs_rec.h"
#include "unet_plan.h"

namespace pnpx {

// ----------------------------------------------------------------------------------------- kernels
// x [B,1,H,W] + sigma [B] -> padded 2-channel tensor (channel 1 = sigma inside the image, 0 in the border:
// exactly what zero-padding the reference's concatenated noise map gives, denoiser/base.py:29-30).
__global__ void prep_input_kernel(const float* __restrict__ x, const float* __restrict__ sigma, int sigma_stride,
                                  float* __restrict__ dst, int H, int W, int Hp, int Wp) {
  const int b = blockIdx.z;
  const int y = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= W) return;
  const float v = x[((size_t)b * H + y) * W + xx];
  float* d = dst + (size_t)b * 2 * Hp * Wp + (size_t)(y + 1) * Wp + xx + PADL;
  d[0] = v;
  d[(size_t)Hp * Wp] = sigma[(size_t)b * sigma_stride];
}

// (conv_first_f32_kernel: conv_first.h, shared with drunet_f32.hip)
// MaxPool2d(2) (models/unet.py:82-85), floor semantics.  One thread per output pixel, flat index.
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       size_t n_out, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const int Hp = padded_h(H), Wp = padded_w(W), Hpo = padded_h(Ho), Wpo = padded_w(Wo);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int xo = (int)(i % Wo);
  const size_t t = i / Wo;
  const int yo = (int)(t % Ho);
  const size_t bc = t / Ho;
  const float* s = src + bc * Hp * Wp + (size_t)(2 * yo + 1) * Wp + 2 * xo + PADL;
  const float2 r0 = *reinterpret_cast<const float2*>(s);
  const float2 r1 = *reinterpret_cast<const float2*>(s + Wp);
  dst[bc * Hpo * Wpo + (size_t)(yo + 1) * Wpo + xo + PADL] = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (models/unet.py:99): src = dst*(in-1)/(out-1),
// i0 = floor(src), i1 = i0 + (i0 < in-1), weights (1-l, l); same association as ATen's CPU kernel.
// The target tensor has the skip connection's size (Ht x Wt >= 2h x 2w): the reference zero-pads the upsampled map
// on the bottom/right when a level's size is odd (F.pad, models/unet.py:109-113; the offset diff//2 is always 0
// because diff is 0 or 1); those cells are part of the never-written zero margin here.
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         size_t n_out, int h, int w, int Ht, int Wt, float sy, float sx) {
  const int H = 2 * h, W = 2 * w;
  const int hp = padded_h(h), wp = padded_w(w), Hp = padded_h(Ht), Wp = padded_w(Wt);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int x = (int)(i % W);
  const size_t t = i / W;
  const int y = (int)(t % H);
  const size_t bc = t / H;
  const float fy = sy * y, fx = sx * x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - y0, lx = fx - x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* s = src + bc * hp * wp + PADL;
  const float v00 = s[(size_t)(y0 + 1) * wp + x0], v01 = s[(size_t)(y0 + 1) * wp + x1];
  const float v10 = s[(size_t)(y1 + 1) * wp + x0], v11 = s[(size_t)(y1 + 1) * wp + x1];
  dst[bc * Hp * Wp + (size_t)(y + 1) * Wp + x + PADL] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// The same up-sampling, four outputs along x per thread (one 16-byte store; fp32 family, r4).  The four outputs' sources lie in a
// window of 4 consecutive input pixels per row (scale < 1/2: x0 advances by at most 2 over three steps, x1 <= x0 + 1), loaded once
// -- 8 loads per 4 outputs instead of 16 -- and picked by position.  Same expression per element as the scalar kernel.
// Blocks of (x quads <= 64) x (256 / that many rows), grid.z = B * C.  Used when the output width is a multiple of 4.
// (The pooling kernel stays scalar: four outputs per thread read 32-byte-strided 16-byte pieces and measured 0.129 vs 0.096 ms.)
constexpr int UPS_ROWS = 4;   // output rows per thread of upsample2x_v4_kernel (the x-dependent terms are computed once per thread)
__global__ __launch_bounds__(256) void upsample2x_v4_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w,
                                                            int Ht, int Wt, float sy, float sx) {
  const int W = 2 * w;
  const int hp = padded_h(h), wp = padded_w(w), Hp = padded_h(Ht), Wp = padded_w(Wt);
  const int xq = blockIdx.x * blockDim.x + threadIdx.x;
  const int yb = (blockIdx.y * blockDim.y + threadIdx.y) * UPS_ROWS;
  if (4 * xq >= W || yb >= 2 * h) return;
  const size_t bc = blockIdx.z;
  const int xs = (int)(sx * (4 * xq));
  int i0[4], i1[4], xi[4];
  float lx[4], hx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = 4 * xq + j;
    const float fx = sx * x;
    const int x0 = (int)fx;
    const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
    lx[j] = fx - x0;
    hx[j] = 1.f - lx[j];
    i0[j] = x0 - xs;
    i1[j] = x1 - xs;
    xi[j] = xs + j < w ? xs + j : w - 1;
  }
  auto pick = [](const float* v, int i) { return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3]; };
  const float* sb = src + bc * hp * wp + PADL;
  float* db = dst + bc * Hp * Wp + 4 * xq + PADL;
#pragma unroll
  for (int r = 0; r < UPS_ROWS; ++r) {
    const int y = yb + r;
    if (y >= 2 * h) break;
    const float fy = sy * y;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly = fy - y0, hy = 1.f - ly;
    const float* s0 = sb + (size_t)(y0 + 1) * wp;
    const float* s1 = sb + (size_t)(y1 + 1) * wp;
    float a0[4], a1[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0[j] = s0[xi[j]];
      a1[j] = s1[xi[j]];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = hy * (hx[j] * pick(a0, i0[j]) + lx[j] * pick(a0, i1[j])) + ly * (hx[j] * pick(a1, i0[j]) + lx[j] * pick(a1, i1[j]));
    *reinterpret_cast<float4*>(db + (size_t)(y + 1) * Wp) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// outconv 1x1 (32 -> 1) + residual on channel 0 + clamp (models/unet.py:63-66,124-131; denoiser/base.py:32).
__global__ void outc_residual_kernel(const float* __restrict__ feat, const float* __restrict__ x,
                                     const float* __restrict__ w, const float* __restrict__ bias,
                                     float* __restrict__ out, float* __restrict__ out_pre, int H, int W) {
  const int Hp = padded_h(H), Wp = padded_w(W);
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  if (xx >= W) return;
  const float* f = feat + (size_t)b * 32 * Hp * Wp + (size_t)(y + 1) * Wp + xx + PADL;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) acc = fmaf(w[c], f[(size_t)c * Hp * Wp], acc);
  const size_t o = ((size_t)b * H + y) * W + xx;
  const float v = x[o] + (acc + bias[0]);
  if (out_pre) out_pre[o] = v;
  out[o] = fminf(fmaxf(v, 0.f), 1.f);
}

// ----------------------------------------------------------------------------------------- HS8 kernels
// Same four ops on the half-split layout of conv_hs.hip: records of 32 B = hi[8] | lo[8] f16 per (group, pixel),
// values scaled by HS_ASCALE, tensor [B][G][H+2][W+2] records with a zero border.  One thread per record.
// (record type and pack / unpack helpers: hs_rec.h)
__global__ __launch_bounds__(256) void prep_input_hs_kernel(const float* __restrict__ x, const float* __restrict__ sigma,
                                                            int sigma_stride, HsRec* __restrict__ dst, int H, int W,
                                                            size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xx = (int)(i % W);
  const size_t t = i / W;
  const int y = (int)(t % H);
  const size_t b = t / H;
  float v[8] = {x[i] * HS_ASCALE, sigma[b * sigma_stride] * HS_ASCALE, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  dst[(b * 2 * (H + 2) + (y + 1)) * (W + 2) + xx + 1] = hs_pack(v);   // group 0 of 2; group 1 stays zero
}

// (conv_first_hs_kernel: conv_first.h, shared with drunet.hip)
__global__ __launch_bounds__(256) void maxpool2_hs_kernel(const HsRec* __restrict__ src, HsRec* __restrict__ dst,
                                                          size_t n_out, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int xo = (int)(i % Wo);
  const size_t t = i / Wo;
  const int yo = (int)(t % Ho);
  const size_t bg = t / Ho;
  const HsRec* s = src + (bg * (H + 2) + (2 * yo + 1)) * (W + 2) + 2 * xo + 1;
  float a[8], c[8], d[8], e[8], o[8];
  hs_unpack(s[0], a);
  hs_unpack(s[1], c);
  hs_unpack(s[W + 2], d);
  hs_unpack(s[W + 3], e);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = fmaxf(fmaxf(a[k], c[k]), fmaxf(d[k], e[k]));
  dst[(bg * (Ho + 2) + (yo + 1)) * (Wo + 2) + xo + 1] = hs_pack(o);
}

// Bilinear x2 (align_corners) on HS8 records through LDS.  A workgroup produces a TX x TY tile of output records of one
// channel group (TX * TY = 256, one record per thread); the (TY/2 + 2) x (TX/2 + 2) source records it interpolates from
// are staged once in LDS with dense, coalesced 32-byte loads, so every source record is fetched once per tile instead
// of once per output pixel through the L1 (4 x), and the index arithmetic is per tile, not per pixel.  Same
// association as ATen's kernel: (1-ly) * ((1-lx) * v00 + lx * v01) + ly * ((1-lx) * v10 + lx * v11).
template <int TX, int R>
__global__ __launch_bounds__(256) void upsample2x_hs_kernel(const HsRec* __restrict__ src, HsRec* __restrict__ dst, int h,
                                                            int w, int Ht, int Wt, float sy, float sx) {
  // one workgroup = TX x TY output records, TY = R passes of 256 / TX rows: the taller the tile, the fewer times a source
  // row is fetched (a one-row tile reads two source rows per output row = 4x the source tensor; 16 rows: 1.3x)
  constexpr int TYP = 256 / TX, TY = TYP * R;
  constexpr int SW = TX / 2 + 2, SH = TY / 2 + 2;
  // planar LDS image: hi halves [SH][SW] then lo halves (neighbouring source records are 16 bytes apart, so the ds_read_b128 of
  // a 16-lane group -- 8 distinct source columns, pairs of lanes sharing one -- covers distinct banks; with whole 32-byte
  // records the same reads were two-way conflicted: r3 PMC 46 % conflict cycles)
  constexpr int NREC = SH * SW;
  __shared__ uint4 tile[NREC * 2];
  const int H = 2 * h, W = 2 * w;
  const int X0 = blockIdx.x * TX, Y0 = blockIdx.y * TY;
  const size_t bg = blockIdx.z;
  const int c_lo = (int)(sx * X0), r_lo = (int)(sy * Y0);
  const uint4* s = reinterpret_cast<const uint4*>(src + bg * (size_t)(h + 2) * (w + 2) + 1);
  for (int k = threadIdx.x; k < NREC * 2; k += 256) {     // 16-byte pieces: consecutive lanes = consecutive bytes
    const int rec = k >> 1, piece = k & 1;
    const int rr = rec / SW, cc = rec - rr * SW;
    const int yy = min(r_lo + rr, h - 1), xx = min(c_lo + cc, w - 1);
    tile[piece * NREC + rec] = s[((size_t)(yy + 1) * (w + 2) + xx) * 2 + piece];
  }
  __syncthreads();
  const int tx = threadIdx.x % TX, ty0 = threadIdx.x / TX;
  const int x = X0 + tx;
  if (x >= W) return;
  const float fx = sx * x;
  const int x0 = (int)fx;
  const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float lx = fx - x0;
  const float hx = 1.f - lx;
  auto unpack = [&](int idx, float v[8]) {
    HsRec r;
    r.hi = *reinterpret_cast<const h8v*>(&tile[idx]);
    r.lo = *reinterpret_cast<const h8v*>(&tile[NREC + idx]);
    hs_unpack(r, v);
  };
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int y = Y0 + ty0 + r * TYP;
    if (y >= H) return;
    const float fy = sy * y;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly = fy - y0;
    const float hy = 1.f - ly;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    unpack((y0 - r_lo) * SW + (x0 - c_lo), v00);
    unpack((y0 - r_lo) * SW + (x1 - c_lo), v01);
    unpack((y1 - r_lo) * SW + (x0 - c_lo), v10);
    unpack((y1 - r_lo) * SW + (x1 - c_lo), v11);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = hy * (hx * v00[k] + lx * v01[k]) + ly * (hx * v10[k] + lx * v11[k]);
    const HsRec ro = hs_pack(o);
    uint4* dp = reinterpret_cast<uint4*>(dst + (bg * (Ht + 2) + (y + 1)) * (Wt + 2) + x + 1);
    dp[0] = __builtin_bit_cast(uint4, ro.hi);   // (nontemporal stores here: 0.67 instead of 0.42 ms per forward -- r4 A/B)
    dp[1] = __builtin_bit_cast(uint4, ro.lo);
  }
}

__global__ __launch_bounds__(256) void outc_residual_hs_kernel(const HsRec* __restrict__ feat, const float* __restrict__ x,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, float* __restrict__ out_pre, int H,
                                                               int W, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xx = (int)(i % W);
  const size_t t = i / W;
  const int y = (int)(t % H);
  const size_t b = t / H;
  const size_t plane = (size_t)(H + 2) * (W + 2);
  const HsRec* f = feat + b * 4 * plane + (size_t)(y + 1) * (W + 2) + xx + 1;
  float acc = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v[8];
    hs_unpack(f[g * plane], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fmaf(w[g * 8 + k], v[k], acc);
  }
  const float r = x[i] + (acc * (1.f / HS_ASCALE) + bias[0]);
  if (out_pre) out_pre[i] = r;
  out[i] = fminf(fmaxf(r, 0.f), 1.f);
}

int reserve_arena(pnpx_ctx* ctx, UNetArena& ar, int mode, int B, int H, int W, size_t extra_bytes) {
  if (B <= ar.capB && H == ar.capH && W == ar.capW && mode == ar.mode) return PNPX_OK;
  const bool same_geom = (H == ar.capH && W == ar.capW);
  const int nb = same_geom ? (B > ar.capB ? B : ar.capB) : B;
  UNetPlan P = make_plan(mode, nb, H, W);
  const size_t total = P.total + extra_bytes * (size_t)nb;
  PNPX_HIP(hipSetDevice(ctx->device));
  PNPX_HIP(hipDeviceSynchronize());
  if (ar.buf.bytes < total) {
    if (ar.buf.p) PNPX_HIP(hipFree(ar.buf.p));
    ar = UNetArena();
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, total);
    if (e != hipSuccess) {
      set_error("arena allocation of %zu bytes failed: %s", total, hipGetErrorString(e));
      return PNPX_ERR_ALLOC;
    }
    ar.buf.p = p;
    ar.buf.bytes = total;
  }
  PNPX_HIP(hipMemset(ar.buf.p, 0, total));  // borders stay zero until the layout changes
  PNPX_HIP(hipDeviceSynchronize());
  ar.capB = nb;
  ar.capH = H;
  ar.capW = W;
  ar.mode = mode;
  return PNPX_OK;
}

size_t unet_arena_bytes(int mode, int B, int H, int W) { return make_plan(mode, B, H, W).total; }

void train_cache_free(pnpx_ctx* ctx) {
  (void)hipDeviceSynchronize();
  for (auto& sl : ctx->train_ring) {
    if (sl.arena.buf.p) (void)hipFree(sl.arena.buf.p);
    if (sl.pre.p) (void)hipFree(sl.pre.p);
  }
  ctx->train_ring.clear();
  ctx->train_H = ctx->train_W = 0;
  ctx->train_mode = -1;
}

// Next slot of the training ring for a B x H x W forward, or nullptr (ring disabled, budget below one arena, or out of
// memory).  The ring serves one geometry / kernel family at a time and is re-laid out when that changes; its length is
// what the "train_cache_gb" budget buys at the largest batch seen (at most 64 slots).
static pnpx_ctx::TrainSlot* train_slot_acquire(pnpx_ctx* ctx, int B, int H, int W) {
  if (ctx->opt_train_cache_gb == 0 || ctx->train_alloc_failed) return nullptr;
  const int mode = ctx->conv_mode;
  int capB = B;
  for (const auto& sl : ctx->train_ring) capB = sl.arena.capB > capB ? sl.arena.capB : capB;
  const size_t per = unet_arena_bytes(mode, capB, H, W) + sizeof(float) * (size_t)capB * H * W;
  if (ctx->opt_train_cache_gb < 0 && ctx->train_budget_bytes == 0) {
    // automatic budget, fixed when the ring is first laid out: a quarter of what is free on the device right now (the
    // ring is raw hipMalloc memory PyTorch's caching allocator cannot reclaim), at most 96 GiB
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      free_b = 0;
    }
    size_t held = 0;
    for (const auto& sl : ctx->train_ring) held += sl.arena.buf.bytes + sl.pre.bytes;
    ctx->train_budget_bytes = std::min<size_t>((free_b + held) / 4, (size_t)96 << 30);
    if (ctx->train_budget_bytes == 0) ctx->train_budget_bytes = 1;   // "measured, nothing to spare"
  }
  const size_t budget = ctx->opt_train_cache_gb < 0 ? ctx->train_budget_bytes : ((size_t)ctx->opt_train_cache_gb << 30);
  size_t n = budget / per;
  if (n == 0) return nullptr;
  if (n > 64) n = 64;
  if (H != ctx->train_H || W != ctx->train_W || mode != ctx->train_mode || n != ctx->train_ring.size()) {
    train_cache_free(ctx);
    ctx->train_ring.resize(n);
    ctx->train_H = H;
    ctx->train_W = W;
    ctx->train_mode = mode;
  }
  pnpx_ctx::TrainSlot& sl = ctx->train_ring[ctx->train_counter % ctx->train_ring.size()];
  sl.ticket = 0;
  const size_t need = sizeof(float) * (size_t)B * H * W;
  if (sl.pre.bytes < need) {
    (void)hipDeviceSynchronize();
    if (sl.pre.p) (void)hipFree(sl.pre.p);
    sl.pre = DeviceBuf();
    void* q = nullptr;
    if (hipMalloc(&q, need) != hipSuccess) {
      (void)hipGetLastError();
      ctx->train_alloc_failed = true;
      train_cache_free(ctx);
      return nullptr;
    }
    sl.pre.p = q;
    sl.pre.bytes = need;
  }
  return &sl;
}

int unet_denoise_train(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, int B, int H,
                       int W, hipStream_t s, unsigned long long* ticket) {
  *ticket = 0;
  if (ctx->drunet.loaded)     // DRUNet: no activation ring; ticket 0 = its VJP re-computes the forward (drunet_denoise_backward)
    return unet_denoise(ctx, x, sigma, sigma_stride, out, nullptr, B, H, W, s, nullptr);
  if (pnpx_ctx::TrainSlot* sl = train_slot_acquire(ctx, B, H, W)) {
    const int st = unet_denoise(ctx, x, sigma, sigma_stride, out, static_cast<float*>(sl->pre.p), B, H, W, s, nullptr,
                                &sl->arena, ctx->conv_mode, true);
    if (st == PNPX_OK) {
      sl->ticket = *ticket = ++ctx->train_counter;
      sl->B = B;
      return PNPX_OK;
    }
    if (st != PNPX_ERR_ALLOC) return st;
    ctx->train_alloc_failed = true;   // out of memory for the ring: give it back and carry on without
    train_cache_free(ctx);
  }
  return unet_denoise(ctx, x, sigma, sigma_stride, out, nullptr, B, H, W, s, nullptr);
}

int unet_denoise_backward_ticket(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride,
                                 const float* grad_out, float* grad_x, float* grad_sigma, int B, int H, int W,
                                 hipStream_t s, unsigned long long ticket) {
  if (ticket != 0 && !ctx->train_ring.empty() && H == ctx->train_H && W == ctx->train_W) {
    pnpx_ctx::TrainSlot& sl = ctx->train_ring[(ticket - 1) % ctx->train_ring.size()];
    if (sl.ticket == ticket && sl.B == B)
      return unet_denoise_backward(ctx, x, sigma, sigma_stride, grad_out, grad_x, grad_sigma, B, H, W, s, &sl.arena,
                                   static_cast<const float*>(sl.pre.p));
  }
  return unet_denoise_backward(ctx, x, sigma, sigma_stride, grad_out, grad_x, grad_sigma, B, H, W, s);
}

int ctx_reserve_unet(pnpx_ctx* ctx, int B, int H, int W) {
  return reserve_arena(ctx, ctx->arena, ctx->conv_mode, B, H, W, 0);
}

// ----------------------------------------------------------------------------------------- forward
namespace {

struct Recorder {
  ProfileSink* sink;
  hipStream_t s;
  int mark(const char* name, double flops) {
    if (!sink) return PNPX_OK;
    if ((size_t)sink->n + 1 >= sink->events->size()) return PNPX_OK;
    PNPX_HIP(hipEventRecord((*sink->events)[sink->n + 1], s));
    sink->names.push_back(name);
    sink->flops.push_back(flops);
    sink->n++;
    return PNPX_OK;
  }
  void credit(double flops) {   // extra algorithmic FLOPs of the launch marked last (work folded into it)
    if (sink && !sink->flops.empty()) sink->flops.back() += flops;
  }
};

inline dim3 g1d(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

// Half-split forward pass.  Levels 0 and 1 (the memory-bound, shallow-K layers) can be processed in sub-batches so
// that a ConvBlock's temporaries are re-read soon after they were written (PNPX_SUBBATCH images at level 0, twice
// that at level 1; 0 = whole batch).  Measured (B=48, 256^2): r1 6.58 ms whole batch, 6.38 ms at 24; with the r3 kernels
// (tools/ab_wall.py "subbatch=..."): 5.72 ms at 24, 5.68 whole batch, 5.83 / 5.86 / 6.05 at 16 / 12 / 8 (launch count, thinner
// grids), and no difference beyond the run-to-run spread in the hot bench -- the Infinity Cache does not turn these layers
// around.  24 stays the default.  Deeper levels always run the whole batch.
static int unet_forward_hs(pnpx_ctx* ctx, UNetArena& ar, const UNetPlan& P, const float* x, const float* sigma,
                           int sigma_stride, float* out, float* out_pre, int B, int H, int W, hipStream_t s,
                           Recorder& rec, bool keep_all, int b_base = 0, int share = 1) {
  // share: launch chains running side by side (this call is one of them): passed to the launch table.
  // b_base: the B images of this call are images b_base .. b_base + B - 1 of the arena (x / sigma / out already point at
  // the first of them): two halves of a batch can run as independent launch chains on two streams (dual_stream)
  char* A = static_cast<char*>(ar.buf.p);
  auto bpi = [&](const Act& d) { return act_bytes_per_image(CONV_HS, d.C, d.H, d.W); };
  auto at = [&](const Act& d, int b0) { return A + d.off + (size_t)(b0 + b_base) * bpi(d); };
  auto rat = [&](const Act& d, int b0) { return reinterpret_cast<HsRec*>(at(d, b0)); };
  const bool no_pool_fuse = !ctx->opt_fuse_pool, no_outc_fuse = keep_all || !ctx->opt_fuse_outc;
  unsigned* const range_flag = ctx->opt_range_guard ? ctx->range_flag_dev : nullptr;
  // (A variant that interpolated the bilinear x2 upsample inside the conv loader was built and parity-tested in r1; it cost
  // ~190 VALU ops per 32-byte record in the MFMA waves, was slower than "separate upsample kernel + DMA loader" (6.48 vs
  // 6.31 ms per forward) and was removed again -- it also cost every instance registers.)
  const int sub_default = ctx->opt_subbatch;
  auto sub_of = [&](int level) {   // images per sub-batch at this level
    if (sub_default <= 0 || level > 1) return B;
    // level-1 tensors are 2x smaller per image: twice the images per sub-batch
    const int sb = sub_default * (level == 1 ? 2 : 1) * (256 * 256) / ((H * W) > 0 ? (H * W) : 1);
    return sb < 1 ? 1 : (sb > B ? B : sb);
  };
  auto pool_fused = [&](int l) { return !no_pool_fuse && l < 4 && conv_hs_can_pool(P.x[l].H, P.x[l].W); };

  const size_t npix = (size_t)B * H * W;
  const bool first_fused = ctx->opt_fuse_first != 0;   // the VALU first convolution replaces prep_input + conv 0
  if (!first_fused) {
    hipLaunchKernelGGL(prep_input_hs_kernel, g1d(npix), dim3(256), 0, s, x, sigma, sigma_stride, rat(P.in0, 0), H, W, npix);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(rec.mark("prep_input", 0));
  }

  auto conv = [&](int li, const Act& i0, const Act* i1, const Act& o, int b0, int nb, const ConvHsFuse& fuse) -> int {
    const ConvLayer& L = ctx->conv[li];
    const ConvLayerHsDev& D = ctx->conv_hs[li];
    ConvLayerHs Lh;
    Lh.cin = D.cin;
    Lh.cout = D.cout;
    Lh.cin_pad = D.cin_pad;
    Lh.mt = D.mt;
    Lh.w = D.w;
    Lh.b = L.b;
    Lh.inv_scale = D.inv_scale;
    ConvHsFuse fz = fuse;
    fz.range_flag = range_flag;
    fz.wreg = ctx->opt_wreg;
    fz.share = share;
    PNPX_TRY(launch_conv_hs(Lh, at(i0, b0), i0.C / 8, i1 ? at(*i1, b0) : nullptr, i1 ? i1->C / 8 : 0, at(o, b0), nb, o.H,
                            o.W, fz, s));
    return rec.mark("conv3x3", 2.0 * 9.0 * L.cin * L.cout * (double)o.H * o.W * nb);
  };
  auto block = [&](int li, const Act& i0, const Act* i1, int lvl, const Act& o, int b0, int nb,
                   const ConvHsFuse& fuse, const ConvHsFuse& first_fuse = ConvHsFuse()) -> int {
    const Act& ta = i1 ? P.da[lvl] : P.a[lvl];   // decoder blocks (two sources) have their own temporaries
    const Act& tb = i1 ? P.db[lvl] : P.b[lvl];
    bool folded = false;
    if (li == 0 && first_fused && ctx->opt_fold_first && !keep_all) {
      // the first convolution (2 -> 32 channels, vector ALU) folded into the loader of the block's second convolution
      // (weights-in-registers instance): its output tensor `ta` is neither written nor read
      const ConvLayerHsDev& D1 = ctx->conv_hs[1];
      ConvLayerHs L1;
      L1.cin = D1.cin;
      L1.cout = D1.cout;
      L1.mt = D1.mt;
      ConvHsFuse f1;
      f1.wreg = ctx->opt_wreg;
      if (conv_hs_can_fold_first(L1, ta.C / 8, nb, H, W, f1)) {
        f1.first_x = x + (size_t)b0 * H * W;
        f1.first_sigma = sigma + (size_t)b0 * sigma_stride;
        f1.first_sigma_stride = sigma_stride;
        f1.first_w = ctx->conv0_w;
        f1.first_b = ctx->conv[0].b;
        f1.first_zero = ctx->zero_bias;
        f1.first_slope = 0.2f;
        PNPX_TRY(conv(1, ta, nullptr, tb, b0, nb, f1));
        rec.credit(2.0 * 9.0 * 2 * 32 * (double)H * W * nb);   // the folded first convolution's FLOPs
        folded = true;
      }
    }
    if (folded) {
    } else if (li == 0 && first_fused) {
      hipLaunchKernelGGL(conv_first_hs_kernel, dim3((H * W + 255) / 256, 4, nb), dim3(256), 0, s, x + (size_t)b0 * H * W,
                         sigma + (size_t)b0 * sigma_stride, sigma_stride, ctx->conv0_w, ctx->conv[0].b, rat(ta, b0), H, W,
                         0.2f, HS_ASCALE);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(rec.mark("conv3x3", 2.0 * 9.0 * 2 * 32 * (double)H * W * nb));
    } else
    PNPX_TRY(conv(li, i0, i1, ta, b0, nb, first_fuse));
    if (!folded) PNPX_TRY(conv(li + 1, ta, nullptr, tb, b0, nb, ConvHsFuse()));
    return conv(li + 2, tb, nullptr, o, b0, nb, fuse);
  };

  // encoder: the last conv of a block also writes the 2x2 max-pooled tensor (fused epilogue) when the level is wide
  // enough for 32-pixel tiles; otherwise a separate pool kernel runs.
  for (int l = 0; l < 5; ++l) {
    const Act& in = (l == 0) ? P.in0 : P.p[l];
    const int sb = sub_of(l);
    for (int b0 = 0; b0 < B; b0 += sb) {
      const int nb = (B - b0 < sb) ? (B - b0) : sb;
      if (l > 0 && !pool_fused(l - 1)) {
        const Act& src = P.x[l - 1];
        const size_t n_pool = (size_t)nb * (src.C / 8) * (src.H / 2) * (src.W / 2);
        hipLaunchKernelGGL(maxpool2_hs_kernel, g1d(n_pool), dim3(256), 0, s, rat(src, b0), rat(P.p[l], b0), n_pool, src.H,
                           src.W);
        PNPX_LAUNCH_CHECK();
        PNPX_TRY(rec.mark("maxpool2", 0));
      }
      ConvHsFuse f;
      if (pool_fused(l)) f.pool_out = at(P.p[l + 1], b0);
      PNPX_TRY(block(3 * l, in, nullptr, l, P.x[l], b0, nb, f));
    }
  }
  // decoder
  const Act* below = &P.x[4];
  for (int l = 3; l >= 0; --l) {
    const int h = below->H, w = below->W;
    const float sy = (2 * h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    const float sx = (2 * w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    const int sb = sub_of(l);
    for (int b0 = 0; b0 < B; b0 += sb) {
      const int nb = (B - b0 < sb) ? (B - b0) : sb;
      ConvHsFuse f;
      // the decoder entry with 32 output channels up-samples its second source on the fly (producer waves in the conv
      // kernel, conv_hs_kernel.h UPS): no up-sampled tensor in HBM
      ConvHsFuse f_first;
      bool ups_fused = false;
      if (ctx->opt_fuse_up >= (l == 0 ? 1 : 2)) {      // 1: the full-resolution entry only, 2: every decoder entry
        const ConvLayerHsDev& D0 = ctx->conv_hs[15 + 3 * (3 - l)];
        ConvLayerHs L0;
        L0.cout = D0.cout;
        L0.mt = D0.mt;
        // (odd sizes: the up-sampled 2h x 2w image sits inside a larger zero-padded tensor -- separate kernel)
        ups_fused = P.u[l].H == 2 * h && P.u[l].W == 2 * w && P.x[l].H == 2 * h && P.x[l].W == 2 * w &&
                    conv_hs_can_fuse_upsample(L0, P.x[l].C / 8, below->C / 8, 2 * h, 2 * w);
      }
      if (ups_fused) {
        f_first.ups_h = h;
        f_first.ups_w = w;
      } else {
        const unsigned nbg = (unsigned)(nb * (below->C / 8));
        const int Wo = 2 * w, Ho = 2 * h;
        if (Wo > 32) {   // 64 x 16 output tiles
          hipLaunchKernelGGL((upsample2x_hs_kernel<64, 4>), dim3((Wo + 63) / 64, (Ho + 15) / 16, nbg), dim3(256), 0, s,
                             rat(*below, b0), rat(P.u[l], b0), h, w, P.u[l].H, P.u[l].W, sy, sx);
        } else {         // 32 x 16
          hipLaunchKernelGGL((upsample2x_hs_kernel<32, 2>), dim3((Wo + 31) / 32, (Ho + 15) / 16, nbg), dim3(256), 0, s,
                             rat(*below, b0), rat(P.u[l], b0), h, w, P.u[l].H, P.u[l].W, sy, sx);
        }
        PNPX_LAUNCH_CHECK();
        PNPX_TRY(rec.mark("upsample2x", 0));
      }
      if (l == 0 && !no_outc_fuse) {   // the network tail (1x1 conv + residual + clamp) rides on the last conv's epilogue
        f.outc_w = ctx->outc_w;
        f.outc_b = ctx->outc_b;
        f.x_in = x + (size_t)b0 * H * W;
        f.out_img = out + (size_t)b0 * H * W;
        f.out_pre = out_pre ? out_pre + (size_t)b0 * H * W : nullptr;
      }
      PNPX_TRY(block(15 + 3 * (3 - l), P.x[l], ups_fused ? below : &P.u[l], l, P.y[l], b0, nb, f, f_first));
    }
    below = &P.y[l];
  }
  if (no_outc_fuse) {
    hipLaunchKernelGGL(outc_residual_hs_kernel, g1d(npix), dim3(256), 0, s, rat(P.y[0], 0), x, ctx->outc_w, ctx->outc_b,
                       out, out_pre, H, W, npix);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(rec.mark("outc_residual_clamp", 2.0 * 32 * (double)H * W * B));
  }
  return PNPX_OK;
}

// fp32 forward pass (conv_mode 0) over B images that are images b_base .. b_base + B - 1 of the arena (x / sigma / out already point at the
// first of them): the whole batch, or one of two launch chains (unet_denoise).  level_chains: the bottom level may fork two chains itself.
static int unet_forward_f32(pnpx_ctx* ctx, UNetArena& ar, const UNetPlan& P, const float* x, const float* sigma, int sigma_stride, float* out,
                            float* out_pre, int B, int H, int W, hipStream_t s, Recorder& rec, bool keep_all, bool level_chains, int b_base = 0, int B_call = 0) {
  if (B_call <= 0) B_call = B;      // the whole call's batch (this may be one launch chain's slice of it)
  // ---- plain-fp32 path (conv_mode 0): padded planar fp32 activations, whole batch per launch
  char* A = static_cast<char*>(ar.buf.p);
  auto fptr = [&](const Act& d) { return reinterpret_cast<float*>(A + d.off + (size_t)b_base * act_bytes_per_image(CONV_F32, d.C, d.H, d.W)); };
  // r6: scratch of the K-split launches (this slice's images).  Option fp32_ksplit: 0 = never; 1 (default) = the layers whose unsplit
  // tiles cannot fill the chip for THIS call's batch (tiles per image x B_call < 256: launch chains beside each other count as one call);
  // 2 = every layer the geometry rule names, at every batch size (per-image results then bit-identical across ALL batch sizes; costs
  // 4 % at 48 x 256^2, where the extra epilogues and the slab round trip buy nothing: profiles/r6_ksplit.txt)
  float* const ks_part0 = reinterpret_cast<float*>(A + P.ks_part + (size_t)b_base * P.ks_part_per_image);
  auto ks_on = [&](int cout, int h, int w) {
    if (ctx->opt_fp32_ksplit == 2) return true;
    return ctx->opt_fp32_ksplit == 1 && (long long)(h / 16) * (w / 16) * (cout / 64) * B_call < 256;
  };
  // the first convolution (2 -> 32 channels) straight from the fp32 image on the vector ALU (training forwards too: the VJP needs the
  // layer's output, not its padded input tensor)
  const bool first_valu = ctx->opt_fuse_first && W % 4 == 0 && ctx->conv[0].cout == 32;
  if (!first_valu) {
    hipLaunchKernelGGL(prep_input_kernel, dim3((W + 63) / 64, H, B), dim3(64), 0, s, x, sigma, sigma_stride, fptr(P.in0), H,
                       W, padded_h(H), padded_w(W));
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(rec.mark("prep_input", 0));
  }
  // pooled: the encoder's MaxPool2d(2) of this layer's output, written by the same launch when the Winograd kernel runs the layer
  // (one 2x2 output tile = one lane's four values); *pooled_done tells the caller whether the separate pooling kernel is still needed
  auto conv = [&](int li, const Act& i0, const Act* i1, const Act& o, const Act* pooled = nullptr, bool* pooled_done = nullptr) -> int {
    const ConvLayer& L = ctx->conv[li];
    const int C1 = i1 ? i1->C : 0;
    if (pooled_done) *pooled_done = false;
    if (ctx->opt_fp32_winograd && ctx->conv_wino_u[li] && conv3x3_wino_ok(i0.C, C1, L.cout, o.H, o.W)) {
      const bool w8 = ((ctx->opt_fp32_wino8 >> li) & 1) && conv3x3_wino8_ok(i0.C, C1, L.cout, o.H, o.W);
      if (w8) {
        const bool ks = ks_on(L.cout, o.H, o.W);
        PNPX_TRY(launch_conv3x3_wino8(ctx->conv_wino_u[li], L.b, L.cout, fptr(i0), i0.C, i1 ? fptr(*i1) : nullptr, C1, fptr(o), B, o.H, o.W, s, 0.2f,
                                      nullptr, pooled ? fptr(*pooled) : nullptr, ks ? ks_part0 : nullptr, ctx->opt_ksplit_rule));
      }
      else
        PNPX_TRY(launch_conv3x3_wino(ctx->conv_wino_u[li], L.b, L.cout, fptr(i0), i0.C, i1 ? fptr(*i1) : nullptr, C1, fptr(o), B, o.H, o.W, s, 0.2f,
                                     nullptr, pooled ? fptr(*pooled) : nullptr));
      if (pooled_done) *pooled_done = pooled != nullptr;
      return rec.mark("conv3x3_wino", 2.0 * 9.0 * L.cin * L.cout * (double)o.H * o.W * B);   // algorithmic FLOPs; 4/9 of them executed
    } else {
      PNPX_TRY(launch_conv3x3(L, fptr(i0), i0.C, i1 ? fptr(*i1) : nullptr, C1, fptr(o), B, o.H, o.W, s));
    }
    return rec.mark("conv3x3", 2.0 * 9.0 * L.cin * L.cout * (double)o.H * o.W * B);
  };
  auto block = [&](int li, const Act& i0, const Act* i1, int lvl, const Act& o, const Act* pooled = nullptr, bool* pooled_done = nullptr) -> int {
    const Act& ta = i1 ? P.da[lvl] : P.a[lvl];
    const Act& tb = i1 ? P.db[lvl] : P.b[lvl];
    if (li == 0 && first_valu) {
      hipLaunchKernelGGL(conv_first_f32_kernel, dim3((H * (W / 4) + 255) / 256, 4, B), dim3(256), 0, s, x, sigma, sigma_stride, ctx->conv0_w,
                         ctx->conv[0].b, fptr(ta), H, W, 0.2f);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(rec.mark("conv3x3", 2.0 * 9.0 * 2 * 32 * (double)H * W * B));
    } else
    PNPX_TRY(conv(li, i0, i1, ta));
    PNPX_TRY(conv(li + 1, ta, nullptr, tb));
    return conv(li + 2, tb, nullptr, o, pooled, pooled_done);
  };
  bool pooled = false;
  PNPX_TRY(block(0, P.in0, nullptr, 0, P.x[0], &P.p[1], &pooled));
  for (int l = 1; l < 5; ++l) {
    const Act& src = P.x[l - 1];
    if (!pooled) {
      const size_t n_pool = (size_t)B * src.C * (src.H / 2) * (src.W / 2);
      hipLaunchKernelGGL(maxpool2_kernel, g1d(n_pool), dim3(256), 0, s, fptr(src), fptr(P.p[l]), n_pool, src.H, src.W);
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(rec.mark("maxpool2", 0));
    }
    // r5: the bottom level's three launches as TWO launch chains over halves of the batch when their tile count sits between round
    // boundaries (B = 48 at 256^2: 384 tiles of 64 couts x 16 x 16 px on 256 one-per-CU workgroups = two rounds for 1.5 rounds of work):
    // the second half's first layer fills the CUs the first half's leaves idle, and so on down the three layers.  Per-image results do
    // not depend on the slicing (same kernel, independent tiles).
    if (l == 4 && level_chains && B >= 2) {
      const Act& o4 = P.x[4];
      bool all8 = ctx->opt_fp32_winograd != 0;
      for (int li = 12; li < 15; ++li)
        all8 = all8 && ctx->conv_wino_u[li] && ((ctx->opt_fp32_wino8 >> li) & 1) && conv3x3_wino8_ok(ctx->conv[li].cin, 0, ctx->conv[li].cout, o4.H, o4.W);
      const long long ntiles = (long long)(o4.H / 16) * (o4.W / 16) * B * (ctx->conv[13].cout / 64);
      if (all8 && ntiles > 256 && ntiles < 768 && ntiles % 256 != 0) {
        PNPX_TRY(fan_out_chains(ctx, 2, B, s, [&](int lo, int hi, hipStream_t st) -> int {
          auto slice = [&](int li, const Act& i, const Act& o) -> int {
            const ConvLayer& L = ctx->conv[li];
            return launch_conv3x3_wino8(ctx->conv_wino_u[li], L.b, L.cout, fptr(i) + (size_t)lo * i.C * padded_h(i.H) * padded_w(i.W), i.C, nullptr, 0,
                                        fptr(o) + (size_t)lo * o.C * padded_h(o.H) * padded_w(o.W), hi - lo, o.H, o.W, st, 0.2f, nullptr, nullptr,
                                        ks_on(o.C, o.H, o.W) ? ks_part0 + (size_t)lo * (P.ks_part_per_image / sizeof(float)) : nullptr,
                                        ctx->opt_ksplit_rule);
          };
          PNPX_TRY(slice(12, P.p[4], P.a[4]));
          PNPX_TRY(slice(13, P.a[4], P.b[4]));
          return slice(14, P.b[4], P.x[4]);
        }));
        pooled = false;
        continue;
      }
    }
    PNPX_TRY(block(3 * l, P.p[l], nullptr, l, P.x[l], l < 4 ? &P.p[l + 1] : nullptr, &pooled));
  }
  const Act* below = &P.x[4];
  for (int l = 3; l >= 0; --l) {
    const int h = below->H, w = below->W;
    const float sy = (2 * h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    const float sx = (2 * w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    const size_t n_up = (size_t)B * below->C * (2 * h) * (2 * w);
    // r5: the decoder entry interpolates its second source itself (conv3x3_wino8.hip UPS instances: the low-resolution window of a
    // chunk's halo staged in LDS, bilinear x2 into the halo buffer) -- no up-sampled tensor in HBM (models/unet.py:92-121)
    const int li0 = 15 + 3 * (3 - l);
    // (training forwards too: the VJP of the up-sampling is linear in the gradient and reads no up-sampled activation)
    // r6: a decoder entry that the K-split would take in this call (the 32 x 32 level's 768 -> 256 entry when the call cannot fill the chip:
    // its 48-chunk tiles are the longest of the forward) runs UNFUSED -- the separate up-sampling kernel + the plain two-source instance,
    // which splits; the fused instance's interpolation pipeline assumes a tile starts in its first, full-resolution source.  (The two
    // forms agree to 2-5e-7; like the K-split itself this is a property of the call's class, not of the batch.)
    const bool entry_splits = ks_on(ctx->conv[li0].cout, 2 * h, 2 * w) &&
                              conv3x3_wino8_ksplit(P.x[l].C, below->C, ctx->conv[li0].cout, 2 * h, 2 * w, ctx->opt_ksplit_rule) > 1;
    const bool ups_fused = !entry_splits && ctx->opt_fp32_fuse_up && ctx->opt_fp32_winograd && ctx->conv_wino_u[li0] && ((ctx->opt_fp32_wino8 >> li0) & 1) &&
                           P.x[l].H == 2 * h && P.x[l].W == 2 * w && conv3x3_wino8_ups_ok(P.x[l].C, below->C, ctx->conv[li0].cout, 2 * h, 2 * w);
    if (ups_fused) {
      // nothing to launch
    } else if ((2 * w) % 4 == 0 && (size_t)B * below->C <= 65535) {
      const int quads = w / 2, bx = quads >= 64 ? 64 : quads, by = 256 / bx;
      hipLaunchKernelGGL(upsample2x_v4_kernel, dim3((quads + bx - 1) / bx, (2 * h + by * UPS_ROWS - 1) / (by * UPS_ROWS), B * below->C), dim3(bx, by), 0, s,
                         fptr(*below), fptr(P.u[l]), h, w, P.u[l].H, P.u[l].W, sy, sx);
    } else {
      hipLaunchKernelGGL(upsample2x_kernel, g1d(n_up), dim3(256), 0, s, fptr(*below), fptr(P.u[l]), n_up, h, w, P.u[l].H,
                         P.u[l].W, sy, sx);
    }
    if (!ups_fused) {
      PNPX_LAUNCH_CHECK();
      PNPX_TRY(rec.mark("upsample2x", 0));
    }
    // first convolution of a decoder block: the fused instance reads the low-resolution tensor
    auto entry = [&](const Act& ta) -> int {
      if (!ups_fused) return conv(li0, P.x[l], &P.u[l], ta);
      const ConvLayer& L = ctx->conv[li0];
      PNPX_TRY(launch_conv3x3_wino8_ups(ctx->conv_wino_u[li0], L.b, L.cout, fptr(P.x[l]), P.x[l].C, fptr(*below), below->C, fptr(ta), B, 2 * h, 2 * w, s));
      return rec.mark("conv3x3_wino", 2.0 * 9.0 * L.cin * L.cout * (double)(2 * h) * (2 * w) * B);
    };
    // the network's last 3x3 layer takes the 1x1 out-conv + residual + clamp into its epilogue when the Winograd kernel runs it
    // (inference passes only: a training forward keeps the layer's output for the VJP)
    const bool fuse_outc = l == 0 && !keep_all && ctx->opt_fuse_outc && ctx->opt_fp32_winograd && ctx->conv_wino_u[26] &&
                           conv3x3_wino_outc_ok(ctx->conv[26].cin, ctx->conv[26].cout, H, W);
    if (fuse_outc) {
      PNPX_TRY(entry(P.da[0]));
      PNPX_TRY(conv(25, P.da[0], nullptr, P.db[0]));
      const bool w8 = ((ctx->opt_fp32_wino8 >> 26) & 1) && conv3x3_wino8_ok(ctx->conv[26].cin, 0, ctx->conv[26].cout, H, W);
      PNPX_TRY((w8 ? launch_conv3x3_wino8_outc : launch_conv3x3_wino_outc)(ctx->conv_wino_u[26], ctx->conv[26].b, fptr(P.db[0]), ctx->conv[26].cin,
                                                                          ctx->outc_w, ctx->outc_b, x, out, out_pre, B, H, W, s));
      return rec.mark("conv3x3_wino", 2.0 * 9.0 * ctx->conv[26].cin * ctx->conv[26].cout * (double)H * W * B);
    }
    PNPX_TRY(entry(P.da[l]));
    PNPX_TRY(conv(li0 + 1, P.da[l], nullptr, P.db[l]));
    PNPX_TRY(conv(li0 + 2, P.db[l], nullptr, P.y[l]));
    below = &P.y[l];
  }
  hipLaunchKernelGGL(outc_residual_kernel, dim3((W + 63) / 64, H, B), dim3(64), 0, s, fptr(P.y[0]), x, ctx->outc_w,
                     ctx->outc_b, out, out_pre, H, W);
  PNPX_LAUNCH_CHECK();
  PNPX_TRY(rec.mark("outc_residual_clamp", 2.0 * 32 * (double)H * W * B));
  return PNPX_OK;
}

// Number of independent launch chains for a B-image forward: option "chains" (0 = automatic, n = exactly n when B >= n).
int launch_chains(const pnpx_ctx* ctx, int B, int H, int W) {
  int n = ctx->opt_chains;
  if (n == 0) {
    // automatic: two chains where they were measured to pay (tools/chains_table.py, every batch size 1..48 at 256^2).  r3 (lone-launch
    // tile shapes, profiles/r3_chains_table.txt): only where a launch's last round was partly empty, and nothing from B = 47 up.  r5: the
    // launch table plans for the chain running beside it (conv_hs.hip hs_choose, ConvHsFuse::share), and two chains pay from B = 5 up
    // (profiles/r5_chains_table_hs.txt: -4 % at B = 6, -13 % at 9-11 / 17-19 / 33-35, -5.5 % at B = 48, -5 % at 64, -3 % at 96; B = 8
    // +1 %, B <= 4 +2-3 % except 3).  q = batch in 256 x 256 images.
    const long long q = (long long)B * H * W / (256 * 256);
    n = (q >= 5 && q != 8) ? 2 : 1;
  }
  if (n > 8) n = 8;
  return n > B ? B : (n < 1 ? 1 : n);
}

void join_side_streams_after_failure(pnpx_ctx* ctx) {
  for (hipStream_t st : ctx->side_streams) (void)hipStreamSynchronize(st);
}
int chain_streams(pnpx_ctx* ctx, int chains) {
  if (chains > 16) {
    set_error("launch chains: at most 16 slices");
    return PNPX_ERR_ARG;
  }
  while ((int)ctx->side_streams.size() < chains - 1) {
    hipStream_t st = nullptr;
    PNPX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    ctx->side_streams.push_back(st);
  }
  return PNPX_OK;
}
int chain_event(pnpx_ctx* ctx, hipEvent_t* ev) {
  constexpr size_t POOL = 256;     // a call uses `chains` events: an event comes round again >= 32 calls later
  if (ctx->ev_pool.size() < POOL) {
    hipEvent_t e = nullptr;
    PNPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->ev_pool.push_back(e);
    *ev = e;
    return PNPX_OK;
  }
  *ev = ctx->ev_pool[ctx->ev_next++ % POOL];
  return PNPX_OK;
}

int unet_denoise(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, float* out_pre,
                 int B, int H, int W, hipStream_t s, ProfileSink* prof, UNetArena* arena, int mode, bool keep_all) {
  if (ctx->drunet.loaded) {   // the context's denoiser is a DRUNet (drunet.hip)
    if (arena || keep_all || prof) {
      set_error("the DRUNet denoiser keeps its activations in its own arena (drunet_denoise_backward re-computes them) and has no per-launch profile");
      return PNPX_ERR_ARG;
    }
    return drunet_denoise(ctx, x, sigma, sigma_stride, out, out_pre, B, H, W, s);
  }
  if (!ctx->has_weights) {
    set_error("denoiser called before pnpx_unet_load");
    return PNPX_ERR_NO_WEIGHTS;
  }
  if (B <= 0 || H < 16 || W < 16) {
    set_error("denoiser: need B > 0 and H, W >= 16 (four 2x2 poolings; got B=%d H=%d W=%d)", B, H, W);
    return PNPX_ERR_SHAPE;
  }
  UNetArena& ar = arena ? *arena : ctx->arena;
  if (mode < 0) mode = ctx->conv_mode;
  PNPX_TRY(reserve_arena(ctx, ar, mode, B, H, W, 0));
  const bool hs = (mode == CONV_HS);
  const UNetPlan P = make_plan(mode, ar.capB, H, W);
  Recorder rec{prof, s};
  if (prof) PNPX_HIP(hipEventRecord((*prof->events)[0], s));
  const int chains = hs && !prof ? launch_chains(ctx, B, H, W) : 1;
  if (chains > 1) {
    // Independent launch chains (contiguous slices of the batch) on the caller's stream + side streams.  Every kernel is
    // a persistent grid of <= 256 one-per-CU workgroups; for SMALL batches most launches cannot fill the chip (48-200
    // tiles at the deep levels) and each costs ~10 us of fill / drain, so slices that run side by side keep more CUs
    // busy.  (At B = 48 the chip sits at its power cap for the whole forward and overlap buys nothing: measured 5.90 vs
    // 5.90 ms)  Per-image results do not depend on the slicing (bit-identical, tested).
    const size_t px = (size_t)H * W;
    return fan_out_chains(ctx, chains, B, s, [&](int lo, int hi, hipStream_t st) -> int {
      Recorder none{nullptr, st};
      return unet_forward_hs(ctx, ar, P, x + lo * px, sigma + (size_t)lo * sigma_stride, sigma_stride, out + lo * px,
                             out_pre ? out_pre + lo * px : nullptr, hi - lo, H, W, st, none, keep_all, lo, chains);
    });
  }
  if (hs) return unet_forward_hs(ctx, ar, P, x, sigma, sigma_stride, out, out_pre, B, H, W, s, rec, keep_all);

  // ---- plain-fp32 path (conv_mode 0): padded planar fp32 activations
  // r5: two launch chains over halves of the batch (option fp32_chains = 2; bit-identical per image).  Unlike the half-split family this
  // one is not power-capped, so what one chain's launch leaves idle at its tail / between round boundaries the other chain's launches
  // fill.  fp32_chains = 1: only the bottom level (16 x 16 at 256^2: 384 tiles on 256 workgroups) forks two chains.
  if (!prof && ctx->opt_fp32_chains >= 2 && B >= ctx->opt_fp32_chains) {
    const size_t px = (size_t)H * W;
    return fan_out_chains(ctx, ctx->opt_fp32_chains, B, s, [&](int lo, int hi, hipStream_t st) -> int {
      Recorder none{nullptr, st};
      return unet_forward_f32(ctx, ar, P, x + lo * px, sigma + (size_t)lo * sigma_stride, sigma_stride, out + lo * px,
                              out_pre ? out_pre + lo * px : nullptr, hi - lo, H, W, st, none, keep_all, false, lo, B);
    });
  }
  return unet_forward_f32(ctx, ar, P, x, sigma, sigma_stride, out, out_pre, B, H, W, s, rec, keep_all, !prof && ctx->opt_fp32_chains == 1);
}

}  // namespace pnpx
