// DRUNet denoiser forward on the fp32 kernel family (conv_mode 0): the precision-matched mode of BASELINE config #5's prior.
//
// Same network and same denoiser contract as drunet.hip (KAIR UNetRes assembled from the reference's building blocks,
// tfpnp/pnp/denoiser/models/basicblock.py: conv :61-101, ResBlock :211-227, upsample_convtranspose :413-419,
// downsample_strideconv :437-446; denoiser/base.py:23-32), in fp32 arithmetic throughout like the reference's:
//   * 3x3 layers: Winograd F(2x2,3x3) on the fp32 MFMA (conv3x3_wino.hip) where the level's size is a multiple of 16, the direct
//     fp32 MFMA kernel (conv3x3.hip) elsewhere; ReLU = negative slope 0, the ResBlock skip = the kernels' residual operand;
//   * strided 2x2 conv = space-to-depth re-layout + 1x1 conv over 4*Cin phase-major channels, transposed 2x2 conv = 1x1 conv to
//     4*Cout phase-major channels + depth-to-space; the 1x1 convolutions run on the direct kernel's one-tap instance
//     (conv3x3.hip: launch_conv1x1_act; as centre-tap 3x3 launches they took 23 of 89 ms);
//   * head (2 -> 64) on the direct kernel's 2-channel path from a padded [x | sigma] tensor, tail (64 -> 1) as a 32-cout launch whose
//     channel 0 is clamped into the output image.
// Activations are padded planar fp32 tensors (common.h), one arena per context.  Weights are packed on the first conv_mode-0 call
// from the host copy drunet_load keeps (the fp64 Winograd transform of 32 M weights takes seconds; a context that never leaves
// conv_mode 1 does not pay for it).  r5: the VJP in the same arithmetic (drunet_denoise_backward_f32, bottom of this file): the forward is
// re-computed keeping every ResBlock's ReLU output, then the adjoint chain of drunet.hip runs on the fp32 kernels (Winograd adjoints
// with the ReLU' mask / the skip's add in their epilogues).
#include <cstring>
#include <vector>

#include "common.h"
#include "conv3x3.h"
#include "conv_first.h"
#include "grad_common.h"

namespace pnpx {

namespace {

constexpr int NC[4] = {64, 128, 256, 512};

inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// [x | sigma] -> padded 2-channel tensor (sigma inside the image, zero border: what zero-padding the concatenated map gives)
__global__ void f32_prep_kernel(const float* __restrict__ x, const float* __restrict__ sigma, int sigma_stride, float* __restrict__ dst,
                                int H, int W, int Hp, int Wp) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= W) return;
  float* d = dst + (size_t)b * 2 * Hp * Wp + (size_t)(y + 1) * Wp + xx + PADL;
  d[0] = x[((size_t)b * H + y) * W + xx];
  d[(size_t)Hp * Wp] = sigma[(size_t)b * sigma_stride];
}

// space-to-depth: in [B][C][h][w] -> out [B][4C][h/2][w/2], channel (2 dy + dx) * C + c = in[c][2y + dy][2x + dx]
__global__ __launch_bounds__(256) void f32_s2d_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int h, int w, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over out interior: B * 4C * (h/2) * (w/2)
  if (i >= n) return;
  const int wo = w / 2, ho = h / 2;
  const int x = (int)(i % wo);
  size_t t = i / wo;
  const int y = (int)(t % ho);
  t /= ho;
  const int k = (int)(t % (4 * C));
  const size_t b = t / (4 * C);
  const int ph = k / C, c = k - ph * C;
  const int dy = ph >> 1, dx = ph & 1;
  const float v = in[((b * C + c) * padded_h(h) + (2 * y + dy + 1)) * padded_w(w) + 2 * x + dx + PADL];
  out[((b * 4 * C + k) * padded_h(ho) + (y + 1)) * padded_w(wo) + x + PADL] = v;
}

// depth-to-space: in [B][4C][h][w] -> out [B][C][2h][2w], out[c][2y + dy][2x + dx] = in[(2 dy + dx) * C + c][y][x]
__global__ __launch_bounds__(256) void f32_d2s_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int h, int w, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over out interior: B * C * 2h * 2w
  if (i >= n) return;
  const int W2 = 2 * w, H2 = 2 * h;
  const int X = (int)(i % W2);
  size_t t = i / W2;
  const int Y = (int)(t % H2);
  t /= H2;
  const int c = (int)(t % C);
  const size_t b = t / C;
  const int ph = (Y & 1) * 2 + (X & 1);
  const float v = in[((b * 4 * C + ph * C + c) * padded_h(h) + (Y / 2 + 1)) * padded_w(w) + X / 2 + PADL];
  out[((b * C + c) * padded_h(H2) + (Y + 1)) * padded_w(W2) + X + PADL] = v;
}

// o = a + b on the interior (borders stay zero)
__global__ __launch_bounds__(256) void f32_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int h,
                                                      int w, size_t n) {   // n = B * C * h * w
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % w);
  const size_t t = i / w;
  const int y = (int)(t % h);
  const size_t bc = t / h;
  const size_t r = (bc * padded_h(h) + (y + 1)) * padded_w(w) + x + PADL;
  o[r] = a[r] + b[r];
}

// channel 0 of the tail's 32-channel output -> pre-clamp image and clamped image
__global__ void f32_out_kernel(const float* __restrict__ t32, float* __restrict__ out, float* __restrict__ out_pre, int H, int W) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= W) return;
  const float v = t32[((size_t)b * 32 * padded_h(H) + (y + 1)) * padded_w(W) + xx + PADL];
  const size_t o = ((size_t)b * H + y) * W + xx;
  if (out_pre) out_pre[o] = v;
  out[o] = fminf(fmaxf(v, 0.f), 1.f);
}

struct LayerDesc {
  int kind;   // 0 head, 1 res conv a (ReLU), 2 res conv b (+res), 3 strided 2x2, 4 transposed 2x2, 5 tail   (= drunet.hip)
  int cin, cout;
};
std::vector<LayerDesc> layers_of(int nb) {
  std::vector<LayerDesc> L;
  L.push_back({0, 2, NC[0]});
  auto res = [&](int c) {
    for (int i = 0; i < nb; ++i) {
      L.push_back({1, c, c});
      L.push_back({2, c, c});
    }
  };
  for (int l = 0; l < 3; ++l) {
    res(NC[l]);
    L.push_back({3, NC[l], NC[l + 1]});
  }
  res(NC[3]);
  for (int l = 2; l >= 0; --l) {
    L.push_back({4, NC[l + 1], NC[l]});
    res(NC[l]);
  }
  L.push_back({5, NC[0], 1});
  return L;
}

size_t plane(int C, int h, int w) { return (size_t)C * padded_h(h) * padded_w(w); }   // floats per image

struct Plan {
  size_t in2, S[4], P[4], Q[4], M[4], U[4], DT[4], T32, total;   // float offsets; DT[l] holds 2 * C_l channels at level l (l >= 1)
  std::vector<size_t> MK[4];   // keep plans (nb_keep > 0): ReLU output of every ResBlock, [encoder nb | decoder nb] per level (level 3: nb)
};
Plan plan_of(int B, int H, int W, int nb_keep = 0) {
  Plan P{};
  size_t off = 0;
  auto add = [&](size_t& o, size_t floats_per_image) {
    o = off;
    off += floats_per_image * B;
    off = (off + 63) & ~(size_t)63;
  };
  add(P.in2, plane(2, H, W));
  for (int l = 0; l < 4; ++l) {
    const int h = H >> l, w = W >> l, c = NC[l];
    add(P.S[l], plane(c, h, w));
    add(P.P[l], plane(c, h, w));
    add(P.Q[l], plane(c, h, w));
    add(P.M[l], plane(c, h, w));
    if (l <= 2) add(P.U[l], plane(c, h, w));
    if (l >= 1) add(P.DT[l], plane(2 * c, h, w));
  }
  add(P.T32, plane(32, H, W));
  for (int l = 0; l < 4 && nb_keep > 0; ++l) {
    P.MK[l].resize(l == 3 ? nb_keep : 2 * nb_keep);
    for (size_t& o : P.MK[l]) add(o, plane(NC[l], H >> l, W >> l));
  }
  P.total = off + 4096;   // slack for the LDS-DMA gathers' over-read lanes
  return P;
}

// the weight tensor w3[cout][cin][3][3] a layer runs with (1x1 layers: centre tap of a phase-major channel stack; tail: 32 couts)
void layer_w3(const LayerDesc& d, const float* w, int* cin_o, int* cout_o, std::vector<float>& w3) {
  int cin = d.cin, cout = d.cout;
  if (d.kind == 3) {          // Conv2d k2 s2 [cout][cin][2][2]
    cin = 4 * d.cin;
    w3.assign((size_t)cout * cin * 9, 0.f);
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int ph = 0; ph < 4; ++ph) w3[((size_t)co * cin + ph * d.cin + ci) * 9 + 4] = w[((size_t)co * d.cin + ci) * 4 + ph];
  } else if (d.kind == 4) {   // ConvTranspose2d k2 s2 [cin][cout][2][2]
    cout = 4 * d.cout;
    w3.assign((size_t)cout * cin * 9, 0.f);
    for (int ci = 0; ci < cin; ++ci)
      for (int co = 0; co < d.cout; ++co)
        for (int ph = 0; ph < 4; ++ph) w3[((size_t)(ph * d.cout + co) * cin + ci) * 9 + 4] = w[((size_t)ci * d.cout + co) * 4 + ph];
  } else if (d.kind == 5) {   // tail [1][64][3][3] -> 32 couts, rows 1..31 zero
    cout = 32;
    w3.assign((size_t)cout * cin * 9, 0.f);
    std::memcpy(w3.data(), w, sizeof(float) * (size_t)cin * 9);
  } else {
    w3.assign(w, w + (size_t)cout * cin * 9);
  }
  *cin_o = cin;
  *cout_o = cout;
}

int prepare_weights(pnpx_ctx* ctx) {
  DruNet& N = ctx->drunet;
  if (N.f32_ready) return PNPX_OK;
  const std::vector<LayerDesc> L = layers_of(N.nb);
  std::vector<float> host, w3;
  auto align = [&]() { host.resize((host.size() + 63) & ~(size_t)63, 0.f); };
  std::vector<size_t> woff(L.size()), uoff(L.size(), 0);
  std::vector<ConvLayer> lay(L.size());
  const float* src = N.params_host.data();
  for (size_t i = 0; i < L.size(); ++i) {
    const LayerDesc& d = L[i];
    const size_t np = (size_t)d.cin * d.cout * ((d.kind == 3 || d.kind == 4) ? 4 : 9);
    int cin, cout;
    layer_w3(d, src, &cin, &cout, w3);
    src += np;
    const int cc = conv_pack_cc(cin), mt = (cc == 2) ? 32 : conv_pack_mt(cout);   // (the 2-channel path exists for 32-cout tiles)
    align();
    woff[i] = host.size();
    if (d.kind == 3 || d.kind == 4) {       // 1x1 over phase-major channels: the centre taps of w3, packed for the one-tap instance
      std::vector<float> w1((size_t)cout * cin);
      for (size_t q = 0; q < w1.size(); ++q) w1[q] = w3[q * 9 + 4];
      host.resize(host.size() + w1.size());
      pack_conv_weights_1x1(w1.data(), cout, cin, host.data() + woff[i]);
    } else {
      host.resize(host.size() + (size_t)cout * cin * 9);
      pack_conv_weights(w3.data(), cout, cin, mt, cc, host.data() + woff[i]);
    }
    lay[i].cin = cin;
    lay[i].cout = cout;
    lay[i].mt = mt;
    lay[i].cc = cc;
    if ((d.kind == 1 || d.kind == 2) && conv3x3_wino_packs(cout, cin)) {
      align();
      uoff[i] = host.size();
      host.resize(host.size() + conv3x3_wino_floats(cout, cin));
      pack_conv_weights_wino(w3.data(), cout, cin, host.data() + uoff[i]);
    }
  }
  align();
  const size_t zoff = host.size();
  host.resize(host.size() + 1024 + 4096, 0.f);   // zero bias of the bias-free network (largest cout: 1024) + DMA over-read slack
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, host.size() * sizeof(float));
  if (e != hipSuccess) {
    set_error("DRUNet fp32 weight allocation of %zu bytes failed: %s", host.size() * sizeof(float), hipGetErrorString(e));
    return PNPX_ERR_ALLOC;
  }
  PNPX_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  float* d = static_cast<float*>(p);
  N.f32_wino.assign(L.size(), nullptr);
  for (size_t i = 0; i < L.size(); ++i) {
    lay[i].w = d + woff[i];
    lay[i].b = d + zoff;
    if (uoff[i]) N.f32_wino[i] = d + uoff[i];
  }
  N.f32_layers = lay;
  N.f32_weights.p = p;
  N.f32_weights.bytes = host.size() * sizeof(float);
  N.f32_ready = true;
  return PNPX_OK;
}

// ---- VJP (r5)
// g_pre = grad_out on pixels whose pre-clamp output lies in [0, 1] (torch.clamp's backward mask is inclusive), else 0
__global__ __launch_bounds__(256) void f32_tail_mask_kernel(const float* __restrict__ g_out, const float* __restrict__ pre, float* __restrict__ g_pre,
                                                            size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p = pre[i];
  g_pre[i] = (p >= 0.f && p <= 1.f) ? g_out[i] : 0.f;
}
// gx = g_in0[:, 0];  gsigma[b] = sum over pixels of g_in0[:, 1]   (two-stage, deterministic; the DRUNet has no input residual)
__global__ __launch_bounds__(256) void f32_input_grad_kernel(const float* __restrict__ g_in0, int Cg, float* __restrict__ gx,
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
    gx[(size_t)b * n + i] = g0[o];
    acc += g1c[o];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[b * SIG_CHUNKS + chunk] = (w[0] + w[1]) + (w[2] + w[3]);
}

// adjoint packings, index-parallel to the forward layers: a convolution's adjoint is the convolution with the transposed, tap-flipped
// weights (3x3: direct packing + Winograd; the 1x1 layers of the strided / transposed 2x2 convolutions: transposed 1x1; head: 64 -> 2,
// zero-padded to 32 couts; the tail's adjoint runs on the vector ALU with the half-split context's [64][2][9] table)
int prepare_weights_bwd(pnpx_ctx* ctx) {
  DruNet& N = ctx->drunet;
  if (N.f32_bwd_ready) return PNPX_OK;
  const std::vector<LayerDesc> L = layers_of(N.nb);
  std::vector<float> host, w3, wa;
  auto align = [&]() { host.resize((host.size() + 63) & ~(size_t)63, 0.f); };
  std::vector<size_t> woff(L.size(), 0), uoff(L.size(), 0);
  std::vector<ConvLayer> lay(L.size());
  const float* src = N.params_host.data();
  for (size_t i = 0; i < L.size(); ++i) {
    const LayerDesc& d = L[i];
    const size_t np = (size_t)d.cin * d.cout * ((d.kind == 3 || d.kind == 4) ? 4 : 9);
    int cin, cout;
    layer_w3(d, src, &cin, &cout, w3);      // forward tensor [cout][cin][9]
    src += np;
    if (d.kind == 5) continue;
    const int ca_in = cout, ca_out = (cin < 32) ? 32 : cin;      // adjoint: cout -> cin (head: 2, zero-padded to 32)
    wa.assign((size_t)ca_out * ca_in * 9, 0.f);
    for (int ci = 0; ci < cin; ++ci)
      for (int co = 0; co < cout; ++co)
        for (int tap = 0; tap < 9; ++tap) wa[((size_t)ci * ca_in + co) * 9 + tap] = w3[((size_t)co * cin + ci) * 9 + (8 - tap)];
    align();
    woff[i] = host.size();
    const int mt = conv_pack_mt(ca_out);
    if (d.kind == 3 || d.kind == 4) {
      std::vector<float> w1((size_t)ca_out * ca_in);
      for (size_t q = 0; q < w1.size(); ++q) w1[q] = wa[q * 9 + 4];
      host.resize(host.size() + w1.size());
      pack_conv_weights_1x1(w1.data(), ca_out, ca_in, host.data() + woff[i]);
    } else {
      host.resize(host.size() + (size_t)ca_out * ca_in * 9);
      pack_conv_weights(wa.data(), ca_out, ca_in, mt, 8, host.data() + woff[i]);
    }
    lay[i].cin = ca_in;
    lay[i].cout = ca_out;
    lay[i].mt = (d.kind == 3 || d.kind == 4) ? 64 : mt;
    lay[i].cc = 8;
    if ((d.kind == 1 || d.kind == 2) && conv3x3_wino_packs(ca_out, ca_in)) {
      align();
      uoff[i] = host.size();
      host.resize(host.size() + conv3x3_wino_floats(ca_out, ca_in));
      pack_conv_weights_wino(wa.data(), ca_out, ca_in, host.data() + uoff[i]);
    }
  }
  align();
  const size_t zoff = host.size();
  host.resize(host.size() + 1024 + 4096, 0.f);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, host.size() * sizeof(float));
  if (e != hipSuccess) {
    set_error("DRUNet fp32 adjoint-weight allocation of %zu bytes failed: %s", host.size() * sizeof(float), hipGetErrorString(e));
    return PNPX_ERR_ALLOC;
  }
  PNPX_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  float* d = static_cast<float*>(p);
  N.f32_wino_bwd.assign(L.size(), nullptr);
  for (size_t i = 0; i < L.size(); ++i) {
    lay[i].w = d + woff[i];
    lay[i].b = d + zoff;
    if (uoff[i]) N.f32_wino_bwd[i] = d + uoff[i];
  }
  N.f32_layers_bwd = lay;
  N.f32_weights_bwd.p = p;
  N.f32_weights_bwd.bytes = host.size() * sizeof(float);
  N.f32_bwd_ready = true;
  return PNPX_OK;
}

}  // namespace

void drunet_f32_free(pnpx_ctx* ctx) {
  DruNet& N = ctx->drunet;
  if (N.f32_weights.p) (void)hipFree(N.f32_weights.p);
  if (N.f32_arena.p) (void)hipFree(N.f32_arena.p);
  N.f32_weights = DeviceBuf();
  N.f32_arena = DeviceBuf();
  N.f32_ready = false;
  N.f32_capB = N.f32_capH = N.f32_capW = 0;
  N.f32_arena_keeps = false;
  if (N.f32_weights_bwd.p) (void)hipFree(N.f32_weights_bwd.p);
  if (N.f32_arena_grad.p) (void)hipFree(N.f32_arena_grad.p);
  N.f32_weights_bwd = DeviceBuf();
  N.f32_arena_grad = DeviceBuf();
  N.f32_bwd_ready = false;
  N.f32_gcapB = N.f32_gcapH = N.f32_gcapW = 0;
}

int drunet_denoise_f32(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, float* out_pre, int B, int H,
                       int W, hipStream_t s, bool keep_mids) {
  DruNet& N = ctx->drunet;
  if (B <= 0 || H < 8 || W < 8 || (H & 7) || (W & 7)) {
    set_error("DRUNet: need B > 0 and H, W positive multiples of 8 (three 2x2 strided convolutions; got B=%d H=%d W=%d)", B, H, W);
    return PNPX_ERR_SHAPE;
  }
  PNPX_TRY(prepare_weights(ctx));
  if (B > N.f32_capB || H != N.f32_capH || W != N.f32_capW || (keep_mids && !N.f32_arena_keeps)) {   // (re)lay the arena out; the zero borders are written here, once
    // the keep planes (every ResBlock's middle activation: ~1 GB per 512 x 512 image at nb = 4) are sized for the batch of
    // the call that NEEDS them, not for the largest inference batch the arena has seen; a re-layout for an inference call drops them
    // (the next VJP lays them out again for its own batch).  Without keeps the arena keeps its largest batch, as before.
    const bool keeps = keep_mids;
    const int nb_img = (!keeps && H == N.f32_capH && W == N.f32_capW && N.f32_capB > B) ? N.f32_capB : B;
    const Plan Pl = plan_of(nb_img, H, W, keeps ? N.nb : 0);
    PNPX_HIP(hipDeviceSynchronize());
    if (N.f32_arena.bytes < Pl.total * sizeof(float)) {
      if (N.f32_arena.p) PNPX_HIP(hipFree(N.f32_arena.p));
      N.f32_arena = DeviceBuf();
      N.f32_capB = N.f32_capH = N.f32_capW = 0;
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, Pl.total * sizeof(float));
      if (e != hipSuccess) {
        set_error("DRUNet fp32 arena allocation of %zu bytes failed: %s", Pl.total * sizeof(float), hipGetErrorString(e));
        return PNPX_ERR_ALLOC;
      }
      N.f32_arena.p = p;
      N.f32_arena.bytes = Pl.total * sizeof(float);
    }
    PNPX_HIP(hipMemset(N.f32_arena.p, 0, N.f32_arena.bytes));
    N.f32_capB = nb_img;
    N.f32_capH = H;
    N.f32_capW = W;
    N.f32_arena_keeps = keeps;
  }
  const Plan Pl = plan_of(N.f32_capB, H, W, N.f32_arena_keeps ? N.nb : 0);
  float* const A = static_cast<float*>(N.f32_arena.p);
  const std::vector<LayerDesc> L = layers_of(N.nb);
  size_t li = 0;

  // layer li: `in` -> `outp` at h x w, negative slope (0 ReLU / 1 linear), optional residual
  auto conv = [&](const float* in, float* outp, int h, int w, float slope, const float* res) -> int {
    const ConvLayer& Lc = N.f32_layers[li];
    const float* u = N.f32_wino[li];
    const int kind = L[li].kind;
    ++li;
    if (kind == 3 || kind == 4) return launch_conv1x1_act(Lc, in, outp, B, h, w, slope, res, s);
    if (u && ctx->opt_fp32_winograd && conv3x3_wino_ok(Lc.cin, 0, Lc.cout, h, w))
    {
      if (ctx->opt_fp32_wino8 && conv3x3_wino8_ok(Lc.cin, 0, Lc.cout, h, w))
        return launch_conv3x3_wino8(u, Lc.b, Lc.cout, in, Lc.cin, nullptr, 0, outp, B, h, w, s, slope, res, nullptr);
      return launch_conv3x3_wino(u, Lc.b, Lc.cout, in, Lc.cin, nullptr, 0, outp, B, h, w, s, slope, res, nullptr);
    }
    return launch_conv3x3_act(Lc, in, Lc.cin, nullptr, 0, outp, B, h, w, slope, res, s);
  };
  auto resblocks = [&](int l, int dec, float* cur, float** result) -> int {
    const int h = H >> l, w = W >> l;
    float* pq[2] = {A + Pl.P[l], A + Pl.Q[l]};
    int k = 0;
    for (int i = 0; i < N.nb; ++i) {
      float* mid = keep_mids ? A + Pl.MK[l][dec * N.nb + i] : A + Pl.M[l];
      PNPX_TRY(conv(cur, mid, h, w, 0.f, nullptr));       // conv + ReLU
      float* dst = pq[k];
      if (dst == cur) dst = pq[k ^= 1];
      PNPX_TRY(conv(mid, dst, h, w, 1.f, cur));            // conv + x
      cur = dst;
      k ^= 1;
    }
    *result = cur;
    return PNPX_OK;
  };

  hipLaunchKernelGGL(f32_prep_kernel, dim3((W + 63) / 64, H, B), dim3(64), 0, s, x, sigma, sigma_stride, A + Pl.in2, H, W, padded_h(H),
                     padded_w(W));
  PNPX_LAUNCH_CHECK();
  PNPX_TRY(conv(A + Pl.in2, A + Pl.S[0], H, W, 1.f, nullptr));   // head, linear
  float* cur = A + Pl.S[0];
  for (int l = 0; l < 3; ++l) {
    PNPX_TRY(resblocks(l, 0, cur, &cur));
    const int h = H >> l, w = W >> l;
    const size_t n = (size_t)B * 4 * NC[l] * (h / 2) * (w / 2);
    hipLaunchKernelGGL(f32_s2d_kernel, g1(n), dim3(256), 0, s, cur, A + Pl.DT[l + 1], NC[l], h, w, n);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(conv(A + Pl.DT[l + 1], A + Pl.S[l + 1], h / 2, w / 2, 1.f, nullptr));
    cur = A + Pl.S[l + 1];
  }
  PNPX_TRY(resblocks(3, 0, cur, &cur));
  for (int l = 2; l >= 0; --l) {
    const int h = H >> (l + 1), w = W >> (l + 1);
    const size_t na = (size_t)B * NC[l + 1] * h * w;
    hipLaunchKernelGGL(f32_add_kernel, g1(na), dim3(256), 0, s, cur, A + Pl.S[l + 1], A + Pl.M[l + 1], h, w, na);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(conv(A + Pl.M[l + 1], A + Pl.DT[l + 1], h, w, 1.f, nullptr));   // 1x1 to 4 * C_l phase-major channels
    const size_t nd = (size_t)B * NC[l] * (2 * h) * (2 * w);
    hipLaunchKernelGGL(f32_d2s_kernel, g1(nd), dim3(256), 0, s, A + Pl.DT[l + 1], A + Pl.U[l], NC[l], h, w, nd);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(resblocks(l, 1, A + Pl.U[l], &cur));
  }
  {
    const size_t na = (size_t)B * NC[0] * H * W;
    hipLaunchKernelGGL(f32_add_kernel, g1(na), dim3(256), 0, s, cur, A + Pl.S[0], A + Pl.M[0], H, W, na);
    PNPX_LAUNCH_CHECK();
  }
  PNPX_TRY(conv(A + Pl.M[0], A + Pl.T32, H, W, 1.f, nullptr));   // tail, linear, 32 couts (channel 0 real)
  hipLaunchKernelGGL(f32_out_kernel, dim3((W + 63) / 64, H, B), dim3(64), 0, s, A + Pl.T32, out, out_pre, H, W);
  PNPX_LAUNCH_CHECK();
  if (li != L.size()) {
    set_error("DRUNet fp32: internal layer walk mismatch (%zu of %zu)", li, L.size());
    return PNPX_ERR_ARG;
  }
  return PNPX_OK;
}

// Back-propagation through  out = clamp(net(cat[x, sigma 1]), 0, 1)  in fp32 arithmetic: the dataflow of drunet.hip::drunet_denoise_backward on
// the fp32 kernels.  A ResBlock  out = B(relu(A(in))) + in  back-propagates as  g_mid = (B^T g_out) * 1[mid > 0],  g_in = A^T g_mid + g_out
// -- the mask and the add sit in the epilogues of the (Winograd) adjoint convolutions.
int drunet_denoise_backward_f32(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, const float* grad_out, float* grad_x,
                                float* grad_sigma, int B, int H, int W, hipStream_t s) {
  DruNet& N = ctx->drunet;
  const size_t npix = (size_t)B * H * W;
  void* sp;
  PNPX_TRY(ctx_scratch(ctx, (3 * npix + (size_t)B * SIG_CHUNKS) * sizeof(float) + 8192, &sp));
  float* out_tmp = static_cast<float*>(sp);
  float* pre = out_tmp + npix;
  float* g_pre = pre + npix;
  float* part = g_pre + npix;
  // 1. forward, keeping the ReLU output of every ResBlock
  PNPX_TRY(drunet_denoise_f32(ctx, x, sigma, sigma_stride, out_tmp, pre, B, H, W, s, true));
  PNPX_TRY(prepare_weights_bwd(ctx));
  const Plan F = plan_of(N.f32_capB, H, W, N.nb);
  const float* const FA = static_cast<const float*>(N.f32_arena.p);
  // 2. gradient arena (zero borders: gradients are convolution inputs of the adjoint convolutions)
  if (B > N.f32_gcapB || H != N.f32_gcapH || W != N.f32_gcapW) {
    const int nb_img = (H == N.f32_gcapH && W == N.f32_gcapW && N.f32_gcapB > B) ? N.f32_gcapB : B;
    const Plan Gp = plan_of(nb_img, H, W);
    PNPX_HIP(hipDeviceSynchronize());
    if (N.f32_arena_grad.bytes < Gp.total * sizeof(float)) {
      if (N.f32_arena_grad.p) PNPX_HIP(hipFree(N.f32_arena_grad.p));
      N.f32_arena_grad = DeviceBuf();
      N.f32_gcapB = N.f32_gcapH = N.f32_gcapW = 0;
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, Gp.total * sizeof(float));
      if (e != hipSuccess) {
        set_error("DRUNet fp32 gradient arena allocation of %zu bytes failed: %s", Gp.total * sizeof(float), hipGetErrorString(e));
        return PNPX_ERR_ALLOC;
      }
      N.f32_arena_grad.p = p;
      N.f32_arena_grad.bytes = Gp.total * sizeof(float);
    }
    PNPX_HIP(hipMemset(N.f32_arena_grad.p, 0, N.f32_arena_grad.bytes));
    N.f32_gcapB = nb_img;
    N.f32_gcapH = H;
    N.f32_gcapW = W;
  }
  const Plan G = plan_of(N.f32_gcapB, H, W);
  float* const GA = static_cast<float*>(N.f32_arena_grad.p);
  const int nb = N.nb;
  const std::vector<LayerDesc> L = layers_of(nb);
  auto idx_down = [&](int l) { return 1 + l * (2 * nb + 1); };
  auto idx_strided = [&](int l) { return idx_down(l) + 2 * nb; };
  const int idx_body = 1 + 3 * (2 * nb + 1);
  auto idx_convT = [&](int l) { return idx_body + 2 * nb + (2 - l) * (2 * nb + 1); };
  auto idx_up = [&](int l) { return idx_convT(l) + 1; };

  // adjoint of layer li: `in` -> `outp` at h x w; dmask (ReLU' of the saved activation) or res (added), not both
  auto conv = [&](int li, const float* in, float* outp, int h, int w, const float* dmask, const float* res) -> int {
    const ConvLayer& Lc = N.f32_layers_bwd[li];
    const int kind = L[li].kind;
    if (kind == 3 || kind == 4) return launch_conv1x1_act(Lc, in, outp, B, h, w, 1.f, res, s);
    const float* u = N.f32_wino_bwd[li];
    if (u && ctx->opt_fp32_winograd && ctx->opt_fp32_wino8 && conv3x3_wino8_ok(Lc.cin, 0, Lc.cout, h, w)) {
      if (dmask) return launch_conv3x3_wino8_grad(u, Lc.b, Lc.cout, in, Lc.cin, outp, dmask, 0.f, B, h, w, s);
      return launch_conv3x3_wino8(u, Lc.b, Lc.cout, in, Lc.cin, nullptr, 0, outp, B, h, w, s, 1.f, res, nullptr);
    }
    if (dmask) return launch_conv3x3_grad(Lc, in, outp, dmask, B, h, w, s, nullptr, 0.f);
    return launch_conv3x3_act(Lc, in, Lc.cin, nullptr, 0, outp, B, h, w, 1.f, res, s);
  };
  auto resblocks_bwd = [&](int l, int dec, int idx0, float* cur, float** result) -> int {
    const int h = H >> l, w = W >> l;
    float* pq[2] = {GA + G.P[l], GA + G.Q[l]};
    int k = 0;
    for (int i = nb - 1; i >= 0; --i) {
      const float* mid = FA + F.MK[l][dec * nb + i];
      PNPX_TRY(conv(idx0 + 2 * i + 1, cur, GA + G.M[l], h, w, mid, nullptr));     // B^T g_out * relu'(mid)
      float* dst = pq[k];
      if (dst == cur) dst = pq[k ^= 1];
      PNPX_TRY(conv(idx0 + 2 * i, GA + G.M[l], dst, h, w, nullptr, cur));         // A^T g_mid + g_out
      cur = dst;
      k ^= 1;
    }
    *result = cur;
    return PNPX_OK;
  };
  auto add = [&](const float* a, const float* b, float* o, int l) -> int {
    const int h = H >> l, w = W >> l;
    const size_t n = (size_t)B * NC[l] * h * w;
    hipLaunchKernelGGL(f32_add_kernel, g1(n), dim3(256), 0, s, a, b, o, h, w, n);
    PNPX_LAUNCH_CHECK();
    return PNPX_OK;
  };

  // 3. clamp + tail (64 <- 1 channel: vector ALU, the [64][2][9] table of the half-split context; its second input channel is zero)
  hipLaunchKernelGGL(f32_tail_mask_kernel, g1(npix), dim3(256), 0, s, grad_out, pre, g_pre, npix);
  PNPX_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_first_f32_kernel, dim3((H * (W / 4) + 255) / 256, NC[0] / 8, B), dim3(256), 0, s, g_pre, N.zero, 0, N.tail_bwd_w,
                     N.zero, GA + G.S[0], H, W, 1.0f);
  PNPX_LAUNCH_CHECK();
  // 4. decoder, level 0 up to level 2: G.S[l] = gradient of m_l = c_l + S[l] (kept: it is also the skip's gradient)
  float* cur = GA + G.S[0];
  for (int l = 0; l <= 2; ++l) {
    PNPX_TRY(resblocks_bwd(l, 1, idx_up(l), cur, &cur));                       // -> gradient of U[l]
    const int h = H >> l, w = W >> l;
    const size_t n = (size_t)B * 4 * NC[l] * (h / 2) * (w / 2);
    hipLaunchKernelGGL(f32_s2d_kernel, g1(n), dim3(256), 0, s, cur, GA + G.DT[l + 1], NC[l], h, w, n);   // adjoint of depth-to-space
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(conv(idx_convT(l), GA + G.DT[l + 1], GA + G.S[l + 1], h / 2, w / 2, nullptr, nullptr));
    cur = GA + G.S[l + 1];
  }
  // 5. body and encoder, level 3 down to level 0: G.M[l] = total gradient of S[l]
  PNPX_TRY(resblocks_bwd(3, 0, idx_body, cur, &cur));
  PNPX_TRY(add(cur, GA + G.S[3], GA + G.M[3], 3));
  for (int l = 2; l >= 0; --l) {
    const int h = H >> (l + 1), w = W >> (l + 1);
    PNPX_TRY(conv(idx_strided(l), GA + G.M[l + 1], GA + G.DT[l + 1], h, w, nullptr, nullptr));
    const size_t n = (size_t)B * NC[l] * (2 * h) * (2 * w);
    hipLaunchKernelGGL(f32_d2s_kernel, g1(n), dim3(256), 0, s, GA + G.DT[l + 1], GA + G.U[l], NC[l], h, w, n);   // adjoint of space-to-depth
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(resblocks_bwd(l, 0, idx_down(l), GA + G.U[l], &cur));
    PNPX_TRY(add(cur, GA + G.S[l], GA + G.M[l], l));
  }
  // 6. head: 64 -> (image, noise map) gradients
  PNPX_TRY(conv(0, GA + G.M[0], GA + G.T32, H, W, nullptr, nullptr));
  hipLaunchKernelGGL(f32_input_grad_kernel, dim3(SIG_CHUNKS, B), dim3(256), 0, s, GA + G.T32, 32, grad_x, part, H, W);
  PNPX_LAUNCH_CHECK();
  hipLaunchKernelGGL(sigma_grad_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part, grad_sigma, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

}  // namespace pnpx
