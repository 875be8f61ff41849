// DRUNet denoiser forward on gfx950 (BASELINE config #5: "Poisson prox + DRUNet denoiser").
//
// The reference ships the KAIR building blocks but no model (tfpnp/pnp/denoiser/models/basicblock.py: conv :61-101,
// ResBlock :211-227, upsample_convtranspose :413-419, downsample_strideconv :437-446); the topology is KAIR's UNetRes
// (DRUNet, Zhang et al. 2021): bias-free, 4 scales of 64-128-256-512 channels, nb ResBlocks per scale each way,
//   x1 = head(cat[x, sigma]);  x2 = down1(x1) = strided2x2(ResBlocks(x1));  x3, x4 likewise;  v = body(x4);
//   v = up3(v + x4) = ResBlocks(convT2x2(v + x4));  up2(v + x3);  up1(v + x2);  out = tail(v + x1),
// behind the denoiser contract of UNetDenoiser2D.forward (denoiser/base.py:23-32): noise-level map as second input
// channel, output clamped to [0, 1].
//
// Everything runs on the half-split f16x3 MFMA convolution of the UNet (conv_hs.hip) over HS8 tensors:
//   * ResBlock = conv3x3 + ReLU (EPI_ACT, slope 0), then conv3x3 + residual add in the epilogue (EPI_RES, linear);
//   * strided 2x2 conv = space-to-depth re-layout (record copies) + a 1x1 convolution over 4*Cin channels;
//   * transposed 2x2 conv = a 1x1 convolution to 4*Cout channels + depth-to-space re-layout;
//     (1x1 convolutions are sparse-tap launches, conv_hs_taps.hip: only the centre tap is stored, copied and multiplied);
//   * head (K = 18) on the vector ALU straight from the fp32 image (conv_first_hs_kernel, as the UNet's first layer);
//   * tail (64 -> 1) as a 32-cout launch with the fused "1x1 + residual + clamp" epilogue (EPI_OUTC) selecting channel 0.
#include <cstring>
#include <vector>

#include "common.h"
#include "conv_first.h"
#include "conv_hs.h"
#include "grad_common.h"
#include "hs_rec.h"
#include "hs_relayout.h"

namespace pnpx {

namespace {

constexpr int DRU_NC[4] = {64, 128, 256, 512};

// out = a + b on interior records (fp32 add of the recombined halves, re-split)
__global__ __launch_bounds__(256) void dru_add_kernel(const HsRec* __restrict__ a, const HsRec* __restrict__ b,
                                                      HsRec* __restrict__ out, int H, int W, size_t n) {   // n = B*G*H*W
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  size_t t = i / W;
  const int y = (int)(t % H);
  const size_t bg = t / H;
  const size_t r = (bg * (H + 2) + (y + 1)) * (W + 2) + x + 1;
  float va[8], vb[8];
  hs_unpack(a[r], va);
  hs_unpack(b[r], vb);
#pragma unroll
  for (int k = 0; k < 8; ++k) va[k] += vb[k];
  out[r] = hs_pack(va);
}

inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

struct LayerDesc {
  int kind;   // 0 head, 1 res conv a (ReLU), 2 res conv b (+res), 3 strided 2x2, 4 transposed 2x2, 5 tail
  int cin, cout;
};
// state_dict order (KAIR UNetRes): head; down1..3 = nb x (res.0, res.2) + strided conv; body; up3..1 = convT + nb x (..); tail
std::vector<LayerDesc> dru_layers(int nb) {
  std::vector<LayerDesc> L;
  L.push_back({0, 2, DRU_NC[0]});
  auto res = [&](int c) {
    for (int i = 0; i < nb; ++i) {
      L.push_back({1, c, c});
      L.push_back({2, c, c});
    }
  };
  for (int l = 0; l < 3; ++l) {
    res(DRU_NC[l]);
    L.push_back({3, DRU_NC[l], DRU_NC[l + 1]});
  }
  res(DRU_NC[3]);
  for (int l = 2; l >= 0; --l) {
    L.push_back({4, DRU_NC[l + 1], DRU_NC[l]});
    res(DRU_NC[l]);
  }
  L.push_back({5, DRU_NC[0], 1});
  return L;
}
size_t layer_params(const LayerDesc& d) {
  const int k = (d.kind == 3 || d.kind == 4) ? 4 : 9;
  return (size_t)d.cin * d.cout * k;
}

}  // namespace

size_t drunet_num_params(int nb) {
  size_t n = 0;
  for (const LayerDesc& d : dru_layers(nb)) n += layer_params(d);
  return n;
}

void drunet_free(pnpx_ctx* ctx) {
  DruNet& N = ctx->drunet;
  (void)hipDeviceSynchronize();
  if (N.weights.p) (void)hipFree(N.weights.p);
  if (N.arena.p) (void)hipFree(N.arena.p);
  if (N.arena_grad.p) (void)hipFree(N.arena_grad.p);
  drunet_f32_free(ctx);
  N = DruNet();
}

int drunet_load(pnpx_ctx* ctx, const float* params, size_t n, int nb) {
  if (nb < 1 || nb > 8 || !params || n != drunet_num_params(nb)) {
    set_error("pnpx_drunet_load: expected %zu parameters for nb=%d (1..8), got %zu", nb >= 1 && nb <= 8 ? drunet_num_params(nb) : (size_t)0, nb, n);
    return PNPX_ERR_ARG;
  }
  const std::vector<LayerDesc> L = dru_layers(nb);
  std::vector<float> host;   // blob: packed HS weights per MFMA layer, head weights (native), zero bias, e0, 0
  auto align = [&]() { host.resize((host.size() + 255) & ~(size_t)255, 0.f); };
  std::vector<size_t> off(L.size(), 0);
  std::vector<ConvLayerHsDev> dev(L.size()), devb(L.size());
  std::vector<size_t> offb(L.size(), 0);
  std::vector<int> taps(L.size(), 0x1FF);
  std::vector<float> wt;
  size_t tailb_off = 0;
  // adjoint of a layer given as w[cout][cin][9]: wt[ci][co][tap] = w[co][ci][8 - tap], packed like a forward layer with
  // cin / cout swapped (output channels padded to a multiple of 32 with zero rows)
  auto pack_adjoint = [&](size_t i, const float* w, int cout, int cin, int tapmask) {
    const int cout_b = (cin + 31) / 32 * 32, cin_b = cout, cin_pad_b = (cin_b + 15) / 16 * 16;
    wt.assign((size_t)cout_b * cin_b * 9, 0.f);
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci)
        for (int t = 0; t < 9; ++t) wt[((size_t)ci * cin_b + co) * 9 + t] = w[((size_t)co * cin + ci) * 9 + (8 - t)];
    int ntap = 0;
    for (int t = 0; t < 9; ++t) ntap += (tapmask >> t) & 1;
    const int mt = conv_hs_mt(cout_b);
    align();
    offb[i] = host.size();
    const size_t n16 = (size_t)cout_b * cin_pad_b * ntap * 2;
    host.resize(host.size() + (n16 + 1) / 2, 0.f);
    const float scale = pack_conv_weights_hs_taps(wt.data(), cout_b, cin_b, mt, tapmask, reinterpret_cast<uint16_t*>(host.data() + offb[i]));
    devb[i].cin = cin_b;
    devb[i].cout = cout_b;
    devb[i].cin_pad = cin_pad_b;
    devb[i].mt = mt;
    devb[i].inv_scale = 1.0f / (scale * HS_ASCALE);
  };
  const float* src = params;
  std::vector<float> w3;
  size_t head_off = 0;
  for (size_t i = 0; i < L.size(); ++i) {
    const LayerDesc& d = L[i];
    const float* w = src;
    src += layer_params(d);
    if (d.kind == 0) {   // head: native [64][2][3][3] for the VALU kernel; its adjoint (64 -> 2, padded to 32) on the MFMA kernel
      align();
      head_off = host.size();
      host.insert(host.end(), w, w + layer_params(d));
      pack_adjoint(i, w, d.cout, d.cin, 0x1FF);
      continue;
    }
    int cin = d.cin, cout = d.cout;
    const float* wp = w;
    if (d.kind == 3) {          // Conv2d k2 s2 [cout][cin][2][2] -> 1x1 over the space-to-depth input (phase-major channels)
      cin = 4 * d.cin;
      w3.assign((size_t)cout * cin * 9, 0.f);
      for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < d.cin; ++ci)
          for (int ph = 0; ph < 4; ++ph) w3[((size_t)co * cin + ph * d.cin + ci) * 9 + 4] = w[((size_t)co * d.cin + ci) * 4 + ph];
      wp = w3.data();
    } else if (d.kind == 4) {   // ConvTranspose2d k2 s2 [cin][cout][2][2] -> 1x1 to 4*cout phase-major channels
      cout = 4 * d.cout;
      w3.assign((size_t)cout * cin * 9, 0.f);
      for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < d.cout; ++co)
          for (int ph = 0; ph < 4; ++ph) w3[((size_t)(ph * d.cout + co) * cin + ci) * 9 + 4] = w[((size_t)ci * d.cout + co) * 4 + ph];
      wp = w3.data();
    } else if (d.kind == 5) {   // tail [1][64][3][3] -> 32 couts, rows 1..31 zero
      cout = 32;
      w3.assign((size_t)cout * cin * 9, 0.f);
      std::memcpy(w3.data(), w, sizeof(float) * (size_t)cin * 9);
      wp = w3.data();
    }
    const int mt = conv_hs_mt(cout);
    const int cin_pad = (cin + 15) / 16 * 16;
    const int tapmask = (d.kind == 3 || d.kind == 4) ? 0x010 : 0x1FF;   // 1x1 layers store / multiply the centre tap only
    int ntap = 0;
    for (int t = 0; t < 9; ++t) ntap += (tapmask >> t) & 1;
    align();
    off[i] = host.size();
    const size_t n16 = (size_t)cout * cin_pad * ntap * 2;
    host.resize(host.size() + (n16 + 1) / 2, 0.f);
    const float scale = pack_conv_weights_hs_taps(wp, cout, cin, mt, tapmask, reinterpret_cast<uint16_t*>(host.data() + off[i]));
    taps[i] = tapmask;
    if (d.kind == 5) {          // tail adjoint (1 -> 64 channels) on the vector ALU: conv_first weights [64][2][9], noise-map half zero
      align();
      tailb_off = host.size();
      host.resize(host.size() + 64 * 18, 0.f);
      for (int c = 0; c < 64; ++c)
        for (int t = 0; t < 9; ++t) host[tailb_off + (size_t)c * 18 + t] = w[(size_t)c * 9 + (8 - t)];
    } else {
      pack_adjoint(i, wp, cout, cin, tapmask);
    }
    dev[i].cin = cin;
    dev[i].cout = cout;
    dev[i].cin_pad = cin_pad;
    dev[i].mt = mt;
    dev[i].inv_scale = 1.0f / (scale * HS_ASCALE);
  }
  align();
  const size_t zoff = host.size();
  host.resize(host.size() + 1024, 0.f);          // zero bias (largest cout: 1024)
  align();
  const size_t eoff = host.size();
  host.resize(host.size() + 5 * 32, 0.f);        // e0 table: the fused tail epilogue's 1x1 weights select channel 0 (x 2^(4k): DruNet::shift)
  for (int k = 0; k < 5; ++k) host[eoff + 32 * k] = std::ldexp(1.0f, 4 * k);
  host.resize(host.size() + 1024, 0.f);          // + a zero scalar (outc bias) and DMA over-read slack
  drunet_free(ctx);
  if (ctx->weights.p) {           // a context holds ONE denoiser: loading a DRUNet unloads the UNet
    (void)hipFree(ctx->weights.p);
    ctx->weights = DeviceBuf();
    ctx->has_weights = false;
    train_cache_free(ctx);
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, host.size() * sizeof(float));
  if (e != hipSuccess) {
    set_error("DRUNet weight allocation of %zu bytes failed: %s", host.size() * sizeof(float), hipGetErrorString(e));
    return PNPX_ERR_ALLOC;
  }
  PNPX_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  DruNet& N = ctx->drunet;
  float* d = static_cast<float*>(p);
  for (size_t i = 0; i < L.size(); ++i) {
    dev[i].w = reinterpret_cast<char*>(d + off[i]);
    devb[i].w = reinterpret_cast<char*>(d + offb[i]);
  }
  N.layers = dev;
  N.layers_bwd = devb;
  N.tail_bwd_w = d + tailb_off;
  N.taps = taps;
  N.head_w = d + head_off;
  N.zero = d + zoff;
  N.e0 = d + eoff;
  N.nb = nb;
  N.weights.p = p;
  N.weights.bytes = host.size() * sizeof(float);
  N.params_host.assign(params, params + n);   // conv_mode 0 packs its own layouts from these on first use (drunet_f32.hip)
  N.loaded = true;
  return PNPX_OK;
}

namespace {

struct DruPlan {
  // per level l: S (skip / level input), P, Q (ResBlock outputs, alternating), M (ResBlock middle; skip sums), U (decoder
  // level input, l <= 2), DT (space-to-depth input of the strided conv / output of the transposed conv, l >= 1: 2*C_l ch)
  size_t S[4], P[4], Q[4], M[4], U[4], DT[4];
  size_t MK[4][16];   // backward pass only: one M per ResBlock (level l: encoder blocks 0..nb-1, decoder blocks nb..2nb-1)
  size_t IN0;         // gradient arena only: 32-channel output of the head's adjoint
  size_t zimg;     // [B][H][W] fp32 zeros (residual operand of the fused tail epilogue)
  size_t total;
};
size_t rec_bytes(int C, int h, int w) { return (size_t)(C / 8) * (h + 2) * (w + 2) * 32; }
DruPlan dru_plan(int capB, int H, int W, int nb_keep = 0, bool grad = false) {
  DruPlan P{};
  size_t off = 0;
  auto add = [&](size_t& o, size_t bytes_per_image) {
    o = off;
    off += bytes_per_image * capB;
    off = (off + 255) & ~(size_t)255;
  };
  for (int l = 0; l < 4; ++l) {
    const int h = H >> l, w = W >> l, c = DRU_NC[l];
    add(P.S[l], rec_bytes(c, h, w));
    add(P.P[l], rec_bytes(c, h, w));
    add(P.Q[l], rec_bytes(c, h, w));
    add(P.M[l], rec_bytes(c, h, w));
    if (l <= 2) add(P.U[l], rec_bytes(c, h, w));
    if (l >= 1) add(P.DT[l], rec_bytes(2 * c, h, w));
    for (int j = 0; j < (l == 3 ? 1 : 2) * nb_keep; ++j) add(P.MK[l][j], rec_bytes(c, h, w));
  }
  if (grad) add(P.IN0, rec_bytes(32, H, W));
  add(P.zimg, sizeof(float) * (size_t)H * W);
  P.total = off + (1u << 20);
  return P;
}

}  // namespace

int drunet_denoise(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, float* out_pre,
                   int B, int H, int W, hipStream_t s, bool keep_mids) {
  DruNet& N = ctx->drunet;
  if (!N.loaded) {
    set_error("DRUNet denoiser called before pnpx_drunet_load");
    return PNPX_ERR_NO_WEIGHTS;
  }
  if (ctx->conv_mode != CONV_HS) {
    return drunet_denoise_f32(ctx, x, sigma, sigma_stride, out, out_pre, B, H, W, s, keep_mids);   // fp32 arithmetic throughout (drunet_f32.hip)
  }
  if (B <= 0 || H < 8 || W < 8 || (H & 7) || (W & 7)) {
    set_error("DRUNet: need B > 0 and H, W positive multiples of 8 (three 2x2 strided convolutions; got B=%d H=%d W=%d)", B, H, W);
    return PNPX_ERR_SHAPE;
  }
  if (B > N.capB || H != N.capH || W != N.capW || (keep_mids && !N.arena_keeps)) {     // (re)lay the arena out: zero borders are written here, once
    const int nb_img = (H == N.capH && W == N.capW && N.capB > B) ? N.capB : B;
    const bool keeps = keep_mids || (N.arena_keeps && H == N.capH && W == N.capW);
    const DruPlan Pl = dru_plan(nb_img, H, W, keeps ? N.nb : 0);
    PNPX_HIP(hipDeviceSynchronize());
    if (N.arena.bytes < Pl.total) {
      if (N.arena.p) PNPX_HIP(hipFree(N.arena.p));
      N.arena = DeviceBuf();
      N.capB = N.capH = N.capW = 0;
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, Pl.total);
      if (e != hipSuccess) {
        set_error("DRUNet arena allocation of %zu bytes failed: %s", Pl.total, hipGetErrorString(e));
        return PNPX_ERR_ALLOC;
      }
      N.arena.p = p;
      N.arena.bytes = Pl.total;
    }
    PNPX_HIP(hipMemset(N.arena.p, 0, N.arena.bytes));
    N.capB = nb_img;
    N.capH = H;
    N.capW = W;
    N.arena_keeps = keeps;
  }
  const DruPlan Pl = dru_plan(N.capB, H, W, N.arena_keeps ? N.nb : 0);
  const int sh = N.shift;      // this pass runs on inputs scaled by 2^-sh (see DruNet::shift)
  char* const A0 = static_cast<char*>(N.arena.p);
  unsigned* const range_flag = ctx->opt_range_guard ? ctx->range_flag_dev : nullptr;
  const std::vector<LayerDesc> L = dru_layers(N.nb);

  const int chains = keep_mids ? 1 : launch_chains(ctx, B, H, W);      // (the launch table plans for the chains side by side: f.share)
  // The forward over images b0 .. b0 + B - 1 of the arena on stream s (x / sigma / out already point at the first of them): every
  // tensor is [image][group][h + 2][w + 2] records, so a slice of the batch is a contiguous piece of each -- independent launch
  // chains over slices run side by side on side streams exactly as for the UNet (unet.hip: launch_chains; bit-identical per image).
  auto run = [&](int b0, int B, const float* x, const float* sigma, float* out, float* out_pre, hipStream_t s) -> int {
  struct Shifted {   // arena offsets of this slice
    size_t S[4], P[4], Q[4], M[4], U[4], DT[4], MK[4][16], zimg;
  } Sl{};
  for (int l = 0; l < 4; ++l) {
    const size_t per = rec_bytes(DRU_NC[l], H >> l, W >> l) * (size_t)b0;
    Sl.S[l] = Pl.S[l] + per;
    Sl.P[l] = Pl.P[l] + per;
    Sl.Q[l] = Pl.Q[l] + per;
    Sl.M[l] = Pl.M[l] + per;
    Sl.U[l] = Pl.U[l] + per;
    Sl.DT[l] = Pl.DT[l] + rec_bytes(2 * DRU_NC[l], H >> l, W >> l) * (size_t)b0;
    for (int j = 0; j < 16; ++j) Sl.MK[l][j] = Pl.MK[l][j] + per;
  }
  Sl.zimg = Pl.zimg + sizeof(float) * (size_t)H * W * b0;
  char* const A = A0;
  const Shifted& Pl = Sl;      // the body below addresses the slice through the same names
  size_t li = 0;

  // one MFMA convolution launch: layer li, `in` (cin channels) -> `outp`
  auto conv = [&](const char* in, char* outp, int lvl_h, int lvl_w, float slope, const char* res, bool tail) -> int {
    const ConvLayerHsDev& D = N.layers[li];
    ConvLayerHs Lh;
    Lh.cin = D.cin;
    Lh.cout = D.cout;
    Lh.cin_pad = D.cin_pad;
    Lh.mt = D.mt;
    Lh.w = D.w;
    Lh.b = N.zero;
    Lh.inv_scale = D.inv_scale;
    ConvHsFuse f;
    f.slope = slope;
    f.res = res;
    f.range_flag = range_flag;
    f.wreg = 0;
    f.taps = N.taps[li];
    f.share = chains;
    if (tail) {
      f.outc_w = N.e0 + 32 * (sh / 4);   // selects channel 0 and multiplies the 2^-sh of the head back
      f.outc_b = N.e0 + 5 * 32;          // a zero
      f.x_in = reinterpret_cast<const float*>(A + Pl.zimg);
      f.out_img = out;
      f.out_pre = out_pre;
    }
    ++li;
    return launch_conv_hs(Lh, in, D.cin_pad / 8, nullptr, 0, outp, B, lvl_h, lvl_w, f, s);
  };
  auto resblocks = [&](int l, int dec, char* cur, char** result) -> int {
    const int h = H >> l, w = W >> l;
    char* pq[2] = {A + Pl.P[l], A + Pl.Q[l]};
    int k = 0;
    for (int i = 0; i < N.nb; ++i) {
      char* mid = keep_mids ? A + Pl.MK[l][dec * N.nb + i] : A + Pl.M[l];
      PNPX_TRY(conv(cur, mid, h, w, 0.f, nullptr, false));       // conv + ReLU
      char* dst = pq[k];
      if (dst == cur) dst = pq[k ^= 1];
      PNPX_TRY(conv(mid, dst, h, w, 1.f, cur, false));            // conv + x
      cur = dst;
      k ^= 1;
    }
    *result = cur;
    return PNPX_OK;
  };

  // head: cat[x, sigma] -> 64 channels, linear (VALU, exact fp32 FMA chains)
  hipLaunchKernelGGL(conv_first_hs_kernel, dim3((H * W + 255) / 256, DRU_NC[0] / 8, B), dim3(256), 0, s, x, sigma,
                     sigma_stride, N.head_w, N.zero, reinterpret_cast<HsRec*>(A + Pl.S[0]), H, W, 1.0f, std::ldexp(HS_ASCALE, -sh));
  PNPX_LAUNCH_CHECK();
  li = 1;
  char* cur = A + Pl.S[0];
  for (int l = 0; l < 3; ++l) {
    PNPX_TRY(resblocks(l, 0, cur, &cur));
    const int h = H >> l, w = W >> l, G = DRU_NC[l] / 8;
    const size_t n = (size_t)B * 4 * G * (h / 2) * (w / 2) * 2;
    hipLaunchKernelGGL(hs_s2d_kernel, g1(n), dim3(256), 0, s, reinterpret_cast<const uint4*>(cur),
                       reinterpret_cast<uint4*>(A + Pl.DT[l + 1]), G, h, w, n);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(conv(A + Pl.DT[l + 1], A + Pl.S[l + 1], h / 2, w / 2, 1.f, nullptr, false));
    cur = A + Pl.S[l + 1];
  }
  PNPX_TRY(resblocks(3, 0, cur, &cur));
  for (int l = 2; l >= 0; --l) {
    const int h = H >> (l + 1), w = W >> (l + 1), Gin = DRU_NC[l + 1] / 8, Gout = DRU_NC[l] / 8;
    const size_t na = (size_t)B * Gin * h * w;
    hipLaunchKernelGGL(dru_add_kernel, g1(na), dim3(256), 0, s, reinterpret_cast<const HsRec*>(cur),
                       reinterpret_cast<const HsRec*>(A + Pl.S[l + 1]), reinterpret_cast<HsRec*>(A + Pl.M[l + 1]), h, w, na);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(conv(A + Pl.M[l + 1], A + Pl.DT[l + 1], h, w, 1.f, nullptr, false));
    const size_t nd = (size_t)B * Gout * (2 * h) * (2 * w) * 2;
    hipLaunchKernelGGL(hs_d2s_kernel, g1(nd), dim3(256), 0, s, reinterpret_cast<const uint4*>(A + Pl.DT[l + 1]),
                       reinterpret_cast<uint4*>(A + Pl.U[l]), Gout, h, w, nd);
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(resblocks(l, 1, A + Pl.U[l], &cur));
  }
  {
    const size_t na = (size_t)B * (DRU_NC[0] / 8) * H * W;
    hipLaunchKernelGGL(dru_add_kernel, g1(na), dim3(256), 0, s, reinterpret_cast<const HsRec*>(cur),
                       reinterpret_cast<const HsRec*>(A + Pl.S[0]), reinterpret_cast<HsRec*>(A + Pl.M[0]), H, W, na);
    PNPX_LAUNCH_CHECK();
  }
  PNPX_TRY(conv(A + Pl.M[0], A + Pl.P[0], H, W, 1.f, nullptr, true));   // tail -> out / out_pre (P[0] is not written)
  if (li != L.size()) {
    set_error("DRUNet: internal layer walk mismatch (%zu of %zu)", li, L.size());
    return PNPX_ERR_ARG;
  }
  return PNPX_OK;
  };

  if (chains <= 1) return run(0, B, x, sigma, out, out_pre, s);
  const size_t px = (size_t)H * W;
  return fan_out_chains(ctx, chains, B, s, [&](int lo, int hi, hipStream_t st) -> int {
    return run(lo, hi - lo, x + lo * px, sigma + (size_t)lo * sigma_stride, out + lo * px, out_pre ? out_pre + lo * px : nullptr, st);
  });
}


// ------------------------------------------------------------------------------------------------ VJP
namespace {

// g_pre = grad_out / m on pixels whose pre-clamp output lies in [0, 1] (torch.clamp's backward mask is inclusive), else 0
__global__ __launch_bounds__(256) void dru_tail_mask_kernel(const float* __restrict__ g_out, const float* __restrict__ pre,
                                                            const float2* __restrict__ sc, float* __restrict__ g_pre, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p = pre[i];
  g_pre[i] = (p >= 0.f && p <= 1.f) ? g_out[i] * sc->x : 0.f;
}

}  // namespace

// Back-propagation through  out = clamp(tail(m0), 0, 1)  down to (x, sigma); see the dataflow at the top of this file.
// Every step is the adjoint of a forward step on the SAME kernels: a 3x3 convolution's adjoint is the convolution with the
// transposed, tap-flipped weights (layers_bwd); a ResBlock  out = B(relu(A(in))) + in  back-propagates as
//     g_mid = B^T g_out * 1[mid > 0]   (EPI_DMASK, the saved ReLU output as mask, slope 0)
//     g_in  = A^T g_mid + g_out        (EPI_RES, linear)
// the strided 2x2 convolution (s2d + 1x1) as 1x1^T + d2s, the transposed 2x2 convolution (1x1 + d2s) as s2d + 1x1^T; skip
// sums pass their gradient to both operands.  Gradients are HS8 tensors scaled to max |grad_out| <= 1 (unet_bwd.hip).
int drunet_denoise_backward(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, const float* grad_out,
                            float* grad_x, float* grad_sigma, int B, int H, int W, hipStream_t s) {
  DruNet& N = ctx->drunet;
  if (ctx->conv_mode != CONV_HS)      // fp32 arithmetic throughout (drunet_f32.hip, r5)
    return drunet_denoise_backward_f32(ctx, x, sigma, sigma_stride, grad_out, grad_x, grad_sigma, B, H, W, s);
  const size_t npix = (size_t)B * H * W;
  void* sp;
  PNPX_TRY(ctx_scratch(ctx, (3 * npix + (size_t)B * SIG_CHUNKS) * sizeof(float) + 8192, &sp));
  float* out_tmp = static_cast<float*>(sp);
  float* pre = out_tmp + npix;
  float* g_pre = pre + npix;
  float* part = g_pre + npix;
  float2* gscale = reinterpret_cast<float2*>(part + (((size_t)B * SIG_CHUNKS + 63) & ~(size_t)63));   // {1/m, m}
  unsigned* gmax_bits = reinterpret_cast<unsigned*>(gscale + 1);

  // 1. forward, keeping the ReLU output of every ResBlock
  PNPX_TRY(drunet_denoise(ctx, x, sigma, sigma_stride, out_tmp, pre, B, H, W, s, true));
  const DruPlan F = dru_plan(N.capB, H, W, N.nb);
  char* const FA = static_cast<char*>(N.arena.p);

  // 2. gradient arena (zero borders: gradients are convolution inputs of the adjoint convolutions)
  if (B > N.gcapB || H != N.gcapH || W != N.gcapW) {
    const int nb_img = (H == N.gcapH && W == N.gcapW && N.gcapB > B) ? N.gcapB : B;
    const DruPlan Gp = dru_plan(nb_img, H, W, 0, true);
    PNPX_HIP(hipDeviceSynchronize());
    if (N.arena_grad.bytes < Gp.total) {
      if (N.arena_grad.p) PNPX_HIP(hipFree(N.arena_grad.p));
      N.arena_grad = DeviceBuf();
      N.gcapB = N.gcapH = N.gcapW = 0;
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, Gp.total);
      if (e != hipSuccess) {
        set_error("DRUNet gradient arena allocation of %zu bytes failed: %s", Gp.total, hipGetErrorString(e));
        return PNPX_ERR_ALLOC;
      }
      N.arena_grad.p = p;
      N.arena_grad.bytes = Gp.total;
    }
    PNPX_HIP(hipMemset(N.arena_grad.p, 0, N.arena_grad.bytes));
    N.gcapB = nb_img;
    N.gcapH = H;
    N.gcapW = W;
  }
  const DruPlan G = dru_plan(N.gcapB, H, W, 0, true);
  char* const GA = static_cast<char*>(N.arena_grad.p);
  unsigned* const range_flag = ctx->opt_range_guard ? ctx->range_flag_dev : nullptr;
  const int nb = N.nb;

  // layer indices (state_dict order, dru_layers)
  auto idx_down = [&](int l) { return 1 + l * (2 * nb + 1); };            // first conv of the encoder ResBlocks of level l
  auto idx_strided = [&](int l) { return idx_down(l) + 2 * nb; };         // level l -> l + 1
  const int idx_body = 1 + 3 * (2 * nb + 1);
  auto idx_convT = [&](int l) { return idx_body + 2 * nb + (2 - l) * (2 * nb + 1); };   // level l + 1 -> l
  auto idx_up = [&](int l) { return idx_convT(l) + 1; };

  auto conv = [&](int li, const char* in, char* outp, int h, int w, float slope, const char* dmask, const char* res) -> int {
    const ConvLayerHsDev& D = N.layers_bwd[li];
    ConvLayerHs Lh;
    Lh.cin = D.cin;
    Lh.cout = D.cout;
    Lh.cin_pad = D.cin_pad;
    Lh.mt = D.mt;
    Lh.w = D.w;
    Lh.b = N.zero;
    Lh.inv_scale = D.inv_scale;
    ConvHsFuse f;
    f.slope = slope;
    f.dmask = dmask;
    f.res = res;
    f.range_flag = range_flag;
    f.wreg = 0;
    f.taps = N.taps[li];
    return launch_conv_hs(Lh, in, D.cin_pad / 8, nullptr, 0, outp, B, h, w, f, s);
  };
  // adjoint of the nb ResBlocks of (level l, encoder / decoder side) applied to the gradient in `cur`; result pointer returned
  auto resblocks_bwd = [&](int l, int dec, int idx0, char* cur, char** result) -> int {
    const int h = H >> l, w = W >> l;
    char* pq[2] = {GA + G.P[l], GA + G.Q[l]};
    int k = 0;
    for (int i = nb - 1; i >= 0; --i) {
      const char* mid = FA + F.MK[l][dec * nb + i];
      PNPX_TRY(conv(idx0 + 2 * i + 1, cur, GA + G.M[l], h, w, 0.f, mid, nullptr));     // B^T g_out * relu'(mid)
      char* dst = pq[k];
      if (dst == cur) dst = pq[k ^= 1];
      PNPX_TRY(conv(idx0 + 2 * i, GA + G.M[l], dst, h, w, 1.f, nullptr, cur));         // A^T g_mid + g_out
      cur = dst;
      k ^= 1;
    }
    *result = cur;
    return PNPX_OK;
  };
  auto add = [&](const char* a, const char* b, char* o, int l) -> int {
    const int h = H >> l, w = W >> l;
    const size_t n = (size_t)B * (DRU_NC[l] / 8) * h * w;
    hipLaunchKernelGGL(dru_add_kernel, g1(n), dim3(256), 0, s, reinterpret_cast<const HsRec*>(a), reinterpret_cast<const HsRec*>(b),
                       reinterpret_cast<HsRec*>(o), h, w, n);
    PNPX_LAUNCH_CHECK();
    return PNPX_OK;
  };

  // 3. clamp + tail
  PNPX_HIP(hipMemsetAsync(gmax_bits, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(absmax_kernel, dim3(512), dim3(256), 0, s, grad_out, npix, gmax_bits);
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, s, gmax_bits, gscale);
  hipLaunchKernelGGL(dru_tail_mask_kernel, g1(npix), dim3(256), 0, s, grad_out, pre, gscale, g_pre, npix);
  PNPX_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_first_hs_kernel, dim3((H * W + 255) / 256, DRU_NC[0] / 8, B), dim3(256), 0, s, g_pre, N.zero, 0,
                     N.tail_bwd_w, N.zero, reinterpret_cast<HsRec*>(GA + G.S[0]), H, W, 1.0f, HS_ASCALE);
  PNPX_LAUNCH_CHECK();
  // 4. decoder, level 0 up to level 2: G.S[l] = gradient of m_l = c_l + S[l] (kept: it is also the skip's gradient)
  char* cur = GA + G.S[0];
  for (int l = 0; l <= 2; ++l) {
    PNPX_TRY(resblocks_bwd(l, 1, idx_up(l), cur, &cur));                       // -> gradient of U[l]
    const int h = H >> l, w = W >> l, Gl = DRU_NC[l] / 8;
    const size_t n = (size_t)B * 4 * Gl * (h / 2) * (w / 2) * 2;
    hipLaunchKernelGGL(hs_s2d_kernel, g1(n), dim3(256), 0, s, reinterpret_cast<const uint4*>(cur),
                       reinterpret_cast<uint4*>(GA + G.DT[l + 1]), Gl, h, w, n);   // adjoint of depth-to-space
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(conv(idx_convT(l), GA + G.DT[l + 1], GA + G.S[l + 1], h / 2, w / 2, 1.f, nullptr, nullptr));
    cur = GA + G.S[l + 1];
  }
  // 5. body and encoder, level 3 down to level 0: G.M[l] = total gradient of S[l]
  PNPX_TRY(resblocks_bwd(3, 0, idx_body, cur, &cur));
  PNPX_TRY(add(cur, GA + G.S[3], GA + G.M[3], 3));
  for (int l = 2; l >= 0; --l) {
    const int h = H >> (l + 1), w = W >> (l + 1), Gl = DRU_NC[l] / 8;
    PNPX_TRY(conv(idx_strided(l), GA + G.M[l + 1], GA + G.DT[l + 1], h, w, 1.f, nullptr, nullptr));
    const size_t n = (size_t)B * Gl * (2 * h) * (2 * w) * 2;
    hipLaunchKernelGGL(hs_d2s_kernel, g1(n), dim3(256), 0, s, reinterpret_cast<const uint4*>(GA + G.DT[l + 1]),
                       reinterpret_cast<uint4*>(GA + G.U[l]), Gl, h, w, n);        // adjoint of space-to-depth
    PNPX_LAUNCH_CHECK();
    PNPX_TRY(resblocks_bwd(l, 0, idx_down(l), GA + G.U[l], &cur));
    PNPX_TRY(add(cur, GA + G.S[l], GA + G.M[l], l));
  }
  // 6. head: 64 -> (image, noise map) gradients; no residual path (g_res = zeros)
  PNPX_TRY(conv(0, GA + G.M[0], GA + G.IN0, H, W, 1.f, nullptr, nullptr));
  hipLaunchKernelGGL(input_grad_hs_kernel, dim3(SIG_CHUNKS, B), dim3(256), 0, s, reinterpret_cast<const HsRec*>(GA + G.IN0),
                     reinterpret_cast<const float*>(FA + F.zimg), grad_x, part, gscale, H, W);
  PNPX_LAUNCH_CHECK();
  hipLaunchKernelGGL(sigma_grad_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part, grad_sigma, B);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

}  // namespace pnpx
