// UNet(2,1) denoiser forward on gfx950: orchestration + the non-GEMM kernels.
//
// Replaces UNetDenoiser2D.forward (tfpnp/pnp/denoiser/base.py:23-32) and UNet.forward
// (tfpnp/pnp/denoiser/models/unet.py:52-66):
//   in = cat[x, sigma*1]; x1 = inc(in); x2..x5 = down(.) = ConvBlock(MaxPool2d(2)(.));
//   y = up(y_below, skip) = ConvBlock(cat[skip, bilinear_x2_align_corners(y_below)]);
//   out = clamp(in[:, :1] + Conv1x1(y), 0, 1).
// Activations live in the context's arena in the padded planar layout of common.h.  The channel concat is
// never materialised (the conv kernel reads two source tensors).
#include "common.h"
#include "conv3x3.h"

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
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         size_t n_out, int h, int w, float sy, float sx) {
  const int H = 2 * h, W = 2 * w;
  const int hp = padded_h(h), wp = padded_w(w), Hp = padded_h(H), Wp = padded_w(W);
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

// ----------------------------------------------------------------------------------------- arena plan
struct UNetPlan {
  // per level l (resolution H>>l, W>>l, channels 32<<l)
  ActDesc in0;            // 2 ch @ level 0
  ActDesc a[5], b[5];     // ConvBlock temporaries
  ActDesc x[5];           // encoder outputs x1..x5 (skips)
  ActDesc p[5];           // pooled inputs of level l (l >= 1): channels 16<<l
  ActDesc u[4];           // upsampled decoder inputs at level l (l <= 3): channels 64<<l
  ActDesc y[4];           // decoder outputs at level l (l <= 3)
  size_t total = 0;       // floats, for capB images
  int capB = 0;
};

static UNetPlan make_plan(int capB, int H, int W) {
  UNetPlan P;
  P.capB = capB;
  size_t off = 0;
  auto add = [&](ActDesc& d, int C, int h, int w) {
    d.off = off;
    d.C = C;
    d.H = h;
    d.W = w;
    off += d.per_image() * (size_t)capB;
    off = (off + 63) & ~(size_t)63;
  };
  add(P.in0, 2, H, W);
  for (int l = 0; l < 5; ++l) {
    const int h = H >> l, w = W >> l, c = 32 << l;
    add(P.a[l], c, h, w);
    add(P.b[l], c, h, w);
    add(P.x[l], c, h, w);
    if (l >= 1) add(P.p[l], c / 2, h, w);
    if (l <= 3) {
      add(P.u[l], 2 * c, h, w);
      add(P.y[l], c, h, w);
    }
  }
  P.total = off + (1u << 18);  // 1 MiB slack: overhanging tiles read (never write) past their tensor
  return P;
}

int ctx_reserve_unet(pnpx_ctx* ctx, int B, int H, int W) {
  if (B <= ctx->capB && H == ctx->capH && W == ctx->capW) return PNPX_OK;
  const int nb = (H == ctx->capH && W == ctx->capW) ? (B > ctx->capB ? B : ctx->capB) : B;
  UNetPlan P = make_plan(nb, H, W);
  PNPX_HIP(hipSetDevice(ctx->device));
  PNPX_HIP(hipDeviceSynchronize());
  if (ctx->arena.p) PNPX_HIP(hipFree(ctx->arena.p));
  ctx->arena = DeviceBuf();
  ctx->capB = ctx->capH = ctx->capW = 0;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, P.total * sizeof(float));
  if (e != hipSuccess) {
    set_error("arena allocation of %zu bytes failed: %s", P.total * sizeof(float), hipGetErrorString(e));
    return PNPX_ERR_ALLOC;
  }
  PNPX_HIP(hipMemset(p, 0, P.total * sizeof(float)));  // borders stay zero for the arena's lifetime
  PNPX_HIP(hipDeviceSynchronize());
  ctx->arena.p = p;
  ctx->arena.bytes = P.total * sizeof(float);
  ctx->capB = nb;
  ctx->capH = H;
  ctx->capW = W;
  return PNPX_OK;
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
};

inline dim3 grid2d(int W, int H, size_t Z, int bx) { return dim3((W + bx - 1) / bx, H, (unsigned)Z); }

}  // namespace

int unet_denoise(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, float* out_pre,
                 int B, int H, int W, hipStream_t s, ProfileSink* prof) {
  if (!ctx->has_weights) {
    set_error("denoiser called before pnpx_unet_load");
    return PNPX_ERR_NO_WEIGHTS;
  }
  if (B <= 0 || H <= 0 || W <= 0 || (H % 16) != 0 || (W % 16) != 0) {
    set_error("denoiser: H and W must be positive multiples of 16 (got B=%d H=%d W=%d)", B, H, W);
    return PNPX_ERR_SHAPE;
  }
  PNPX_TRY(ctx_reserve_unet(ctx, B, H, W));
  const UNetPlan P = make_plan(ctx->capB, H, W);
  float* A = static_cast<float*>(ctx->arena.p);
  auto ptr = [&](const ActDesc& d) { return A + d.off; };
  Recorder rec{prof, s};
  if (prof) PNPX_HIP(hipEventRecord((*prof->events)[0], s));

  const int bx = 64;
  hipLaunchKernelGGL(prep_input_kernel, grid2d(W, H, B, bx), dim3(bx), 0, s, x, sigma, sigma_stride, ptr(P.in0), H, W,
                     padded_h(H), padded_w(W));
  PNPX_LAUNCH_CHECK();
  PNPX_TRY(rec.mark("prep_input", 0));

  auto conv = [&](int li, const ActDesc& i0, const ActDesc* i1, const ActDesc& o) -> int {
    const ConvLayer& L = ctx->conv[li];
    PNPX_TRY(launch_conv3x3(L, ptr(i0), i0.C, i1 ? ptr(*i1) : nullptr, i1 ? i1->C : 0, ptr(o), B, o.H, o.W, s));
    return rec.mark("conv3x3", 2.0 * 9.0 * L.cin * L.cout * (double)o.H * o.W * B);
  };
  auto block = [&](int li, const ActDesc& i0, const ActDesc* i1, int lvl, const ActDesc& o) -> int {
    PNPX_TRY(conv(li, i0, i1, P.a[lvl]));
    PNPX_TRY(conv(li + 1, P.a[lvl], nullptr, P.b[lvl]));
    return conv(li + 2, P.b[lvl], nullptr, o);
  };

  // encoder
  PNPX_TRY(block(0, P.in0, nullptr, 0, P.x[0]));
  for (int l = 1; l < 5; ++l) {
    const ActDesc& src = P.x[l - 1];
    const size_t n_pool = (size_t)B * src.C * (src.H / 2) * (src.W / 2);
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((n_pool + 255) / 256)), dim3(256), 0, s, ptr(src),
                       ptr(P.p[l]), n_pool, src.H, src.W);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(rec.mark("maxpool2", 0));
    PNPX_TRY(block(3 * l, P.p[l], nullptr, l, P.x[l]));
  }
  // decoder
  const ActDesc* below = &P.x[4];
  for (int l = 3; l >= 0; --l) {
    const int h = below->H, w = below->W;
    const float sy = (2 * h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    const float sx = (2 * w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    const size_t n_up = (size_t)B * below->C * (2 * h) * (2 * w);
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((n_up + 255) / 256)), dim3(256), 0, s, ptr(*below),
                       ptr(P.u[l]), n_up, h, w, sy, sx);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(rec.mark("upsample2x", 0));
    PNPX_TRY(block(15 + 3 * (3 - l), P.x[l], &P.u[l], l, P.y[l]));
    below = &P.y[l];
  }
  hipLaunchKernelGGL(outc_residual_kernel, grid2d(W, H, B, bx), dim3(bx), 0, s, ptr(P.y[0]), x, ctx->outc_w,
                     ctx->outc_b, out, out_pre, H, W);
  PNPX_LAUNCH_CHECK();
  PNPX_TRY(rec.mark("outc_residual_clamp", 2.0 * 32 * (double)H * W * B));
  return PNPX_OK;
}

}  // namespace pnpx
