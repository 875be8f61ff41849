// C-ABI entry points of libpnpx.so: context, weights, denoiser, generic transforms.  (Solver loops live next to
// their kernels in csmri.hip / tasks.hip.)
#include <cstring>

#include "common.h"
#include "conv3x3.h"
#include "conv_hs.h"

namespace pnpx {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
  return PNPX_ERR_HIP;
}

int ctx_scratch(pnpx_ctx* ctx, size_t bytes, void** out) {
  if (ctx->scratch.bytes < bytes) {
    PNPX_HIP(hipDeviceSynchronize());
    if (ctx->scratch.p) PNPX_HIP(hipFree(ctx->scratch.p));
    ctx->scratch = DeviceBuf();
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      set_error("scratch allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
      return PNPX_ERR_ALLOC;
    }
    ctx->scratch.p = p;
    ctx->scratch.bytes = bytes;
  }
  *out = ctx->scratch.p;
  return PNPX_OK;
}

// UNet(2,1) layer table in state_dict order (tfpnp/pnp/denoiser/models/unet.py:37-46).
struct LayerSpec {
  int cin, cout;
};
static void unet_layers(LayerSpec out[27]) {
  const int blocks[9][2] = {{2, 32},    {32, 64},   {64, 128}, {128, 256}, {256, 512},
                            {768, 256}, {384, 128}, {192, 64}, {96, 32}};
  for (int b = 0; b < 9; ++b)
    for (int j = 0; j < 3; ++j) out[3 * b + j] = LayerSpec{j == 0 ? blocks[b][0] : blocks[b][1], blocks[b][1]};
}

// A DRUNet context answers a tripped guard in steps (r5): the bias-free ReLU network is positively homogeneous, so the
// first two trips move its passes 16x further inside the range each (DruNet::shift 4, then 8: exact up to f16 subnormals, 4e-6 at
// 2^-8); a third trip latches the context to conv_mode 0 like a UNet context (drunet_f32.hip: exact fp32, forward and -- since r5 --
// VJP).  Larger shifts are not used: at 2^-12 the lo halves sit in the f16 subnormals (1e-4 class errors, silently).
// One trip = one step: the device is drained before the flag is cleared, so that kernels of the offending call still in flight
// cannot set it again behind the host's back, and range_guard_enter steps only once per acknowledged trip.
constexpr int DRUNET_MAX_SHIFT = 8;
static int drunet_rescale(pnpx_ctx* ctx) {
  PNPX_HIP(hipDeviceSynchronize());
  if (ctx->drunet.shift < DRUNET_MAX_SHIFT) ctx->drunet.shift += 4;
  else ctx->conv_mode = CONV_F32;
  *ctx->range_flag_host = 0;
  return PNPX_OK;
}

void range_guard_enter(pnpx_ctx* ctx) {
  if (!ctx->opt_range_guard || !ctx->range_flag_host) return;
  if (*static_cast<volatile unsigned*>(ctx->range_flag_host) == 0 || ctx->range_tripped) return;
  ctx->range_tripped = true;          // an EARLIER call overflowed: its output was invalid (pnpx_ctx_status reports it)
  // later calls: 16x further inside the range / exact.  ADVICE r5: if the rescale step itself fails (its device synchronisation
  // returned an error) neither the shift nor the flag clear happened and, the trip being latched, no later call would step either:
  // fall back to the exact family right away so that nothing invalid is produced from here on (the HIP error resurfaces at the
  // call's own first launch).
  if (ctx->drunet.loaded && ctx->conv_mode == CONV_HS) {
    if (drunet_rescale(ctx) != PNPX_OK) ctx->conv_mode = CONV_F32;
  } else ctx->conv_mode = CONV_F32;     // every later call is exact
}

int range_guard_strict(pnpx_ctx* ctx, hipStream_t s, bool* rerun) {
  *rerun = false;
  PNPX_HIP(hipStreamSynchronize(s));
  if (*static_cast<volatile unsigned*>(ctx->range_flag_host) == 0) return PNPX_OK;
  if (ctx->conv_mode != CONV_HS || ctx->range_tripped) {
    // nothing left to fall back to (the flag was set although the exact family ran, or an unacknowledged trip is pending): strict
    // mode's contract is that nothing invalid leaves the call as a success
    set_error("half-split range guard: the call's result is invalid and no fallback is left (conv_mode %d)", ctx->conv_mode);
    return PNPX_ERR_RANGE;
  }
  if (ctx->drunet.loaded) PNPX_TRY(drunet_rescale(ctx));   // this call is repeated 16x further inside the range (or in exact fp32) before it returns
  else {
    ctx->conv_mode = CONV_F32;        // this call is repeated in exact fp32 before it returns: nothing invalid escapes
    *ctx->range_flag_host = 0;
  }
  *rerun = true;
  return PNPX_OK;
}

}  // namespace pnpx

using namespace pnpx;

#define LOCK_CTX(ctx)                         \
  if (!(ctx)) {                               \
    pnpx::set_error("null context");          \
    return PNPX_ERR_ARG;                      \
  }                                           \
  std::lock_guard<std::mutex> _lk((ctx)->mu); \
  PNPX_HIP(hipSetDevice((ctx)->device))

extern "C" {

const char* pnpx_version(void) { return "pnpx 0.1 (gfx950)"; }
const char* pnpx_last_error(void) { return g_err.c_str(); }

int pnpx_ctx_create(int device, pnpx_ctx** out) {
  if (!out) {
    set_error("pnpx_ctx_create: null out");
    return PNPX_ERR_ARG;
  }
  int n = 0;
  PNPX_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) {
    set_error("pnpx_ctx_create: device %d out of range (%d visible)", device, n);
    return PNPX_ERR_ARG;
  }
  PNPX_HIP(hipSetDevice(device));
  pnpx_ctx* c = new pnpx_ctx();
  c->device = device;
  // range-guard word: pinned, host-mapped, written by the device only in the (rare) overflow branch
  void* h = nullptr;
  hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped);
  if (e != hipSuccess) {
    delete c;
    return hip_fail(e, "hipHostMalloc(range flag)", __FILE__, __LINE__);
  }
  c->range_flag_host = static_cast<unsigned*>(h);
  *c->range_flag_host = 0;
  void* d = nullptr;
  e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) {
    (void)hipHostFree(h);
    delete c;
    return hip_fail(e, "hipHostGetDevicePointer(range flag)", __FILE__, __LINE__);
  }
  c->range_flag_dev = static_cast<unsigned*>(d);
  *out = c;
  return PNPX_OK;
}

int pnpx_ctx_destroy(pnpx_ctx* ctx) {
  if (!ctx) return PNPX_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  if (ctx->weights.p) (void)hipFree(ctx->weights.p);
  for (pnpx::UNetArena* a : {&ctx->arena, &ctx->arena_grad})
    if (a->buf.p) (void)hipFree(a->buf.p);
  policy_free(ctx);
  pnpx::drunet_free(ctx);
  pnpx::train_cache_free(ctx);
  if (ctx->scratch.p) (void)hipFree(ctx->scratch.p);
  for (auto& t : ctx->twiddle)
    if (t.second) (void)hipFree(t.second);
  for (auto e : ctx->events) (void)hipEventDestroy(e);
  if (ctx->range_flag_host) (void)hipHostFree(ctx->range_flag_host);
  for (hipStream_t st : ctx->side_streams) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
  delete ctx;
  return PNPX_OK;
}

size_t pnpx_policy_num_params(int num_inputs, int n_det, int spi_head) {
  return policy_num_params(num_inputs, n_det, spi_head);
}

int pnpx_policy_load(pnpx_ctx* ctx, const float* params_host, size_t n_params, int num_inputs, int n_det,
                     int spi_head) {
  LOCK_CTX(ctx);
  return policy_load(ctx, params_host, n_params, num_inputs, n_det, spi_head);
}

int pnpx_policy_forward(pnpx_ctx* ctx, const float* ob, float* probs, float* det, int B, int H, int W,
                        void* stream) {
  LOCK_CTX(ctx);
  if (!ob || !probs || !det) {
    set_error("pnpx_policy_forward: null pointer");
    return PNPX_ERR_ARG;
  }
  return policy_forward(ctx, ob, probs, det, B, H, W, static_cast<hipStream_t>(stream));
}

int pnpx_ctx_reserve(pnpx_ctx* ctx, int B, int H, int W) {
  LOCK_CTX(ctx);
  if (B <= 0 || H < 16 || W < 16) {
    set_error("pnpx_ctx_reserve: need B > 0 and H, W >= 16");
    return PNPX_ERR_SHAPE;
  }
  return ctx_reserve_unet(ctx, B, H, W);
}

int pnpx_ctx_set_option(pnpx_ctx* ctx, const char* key, int value) {
  LOCK_CTX(ctx);
  auto is = [&](const char* k) { return key && !std::strcmp(key, k); };
  if (is("conv_mode") && (value == CONV_F32 || value == CONV_HS)) {
    ctx->conv_mode = value;
    return PNPX_OK;
  }
  if (is("drunet_shift") && value >= 0 && value <= 8 && (value & 3) == 0) {   // DRUNet passes run on inputs scaled by 2^-value
    ctx->drunet.shift = value;                                                 // (raised by 4 whenever the range guard trips)
    return PNPX_OK;
  }
  if (is("subbatch") && value >= 0) {
    ctx->opt_subbatch = value;
    return PNPX_OK;
  }
  if (is("fuse_pool") && (value == 0 || value == 1)) {
    ctx->opt_fuse_pool = value;
    return PNPX_OK;
  }
  if (is("fuse_outc") && (value == 0 || value == 1)) {
    ctx->opt_fuse_outc = value;
    return PNPX_OK;
  }
  if (is("fuse_first") && (value == 0 || value == 1)) {
    ctx->opt_fuse_first = value;
    return PNPX_OK;
  }
  if (is("policy_s2_hs") && (value == 0 || value == 1)) {   // (the arena is re-zeroed: the two layouts share a buffer)
    PNPX_HIP(hipDeviceSynchronize());
    ctx->opt_policy_s2_hs = value;
    ctx->policy.capB = ctx->policy.capH = ctx->policy.capW = 0;
    return PNPX_OK;
  }
  if (is("fold_first") && (value == 0 || value == 1)) {
    ctx->opt_fold_first = value;
    return PNPX_OK;
  }
  if (is("fft_tile") && value >= 0 && value <= 8192) {
    ctx->opt_fft_tile = value;
    return PNPX_OK;
  }
  if (is("fp32_winograd") && (value == 0 || value == 1)) {
    ctx->opt_fp32_winograd = value;
    return PNPX_OK;
  }
  if (is("fp32_chains") && value >= 0 && value <= 8) {
    ctx->opt_fp32_chains = value;
    return PNPX_OK;
  }
  if (is("fp32_fuse_up") && (value == 0 || value == 1)) {
    ctx->opt_fp32_fuse_up = value;
    return PNPX_OK;
  }
  if (is("fp32_ksplit") && value >= 0 && value <= 2) {
    ctx->opt_fp32_ksplit = value;
    return PNPX_OK;
  }
  if (is("fp32_ksplit_rule") && value >= 1 && value < 256) {      // tuning: 1 the default rule, else pieces per tile class (conv3x3_wino8_ksplit)
    ctx->opt_ksplit_rule = value;
    return PNPX_OK;
  }
  if (is("fp32_wino8_layers") && value >= 0 && value < (1 << 27)) {     // bit li: layer li on the 8-wave Winograd kernel
    ctx->opt_fp32_wino8 = value;
    return PNPX_OK;
  }
  if (is("fft_fast") && (value == 0 || value == 1)) {
    ctx->opt_fft_fast = value;
    return PNPX_OK;
  }
  if (is("fft_affine") && (value == 0 || value == 1)) {
    ctx->opt_fft_affine = value;
    return PNPX_OK;
  }
  if (is("chains") && value >= 0 && value <= 8) {
    ctx->opt_chains = value;
    return PNPX_OK;
  }
  if (is("wreg") && value >= 0 && value <= 2) {
    ctx->opt_wreg = value;
    return PNPX_OK;
  }
  if (is("fuse_up") && (value >= 0 && value <= 2)) {
    ctx->opt_fuse_up = value;
    return PNPX_OK;
  }
  if (is("range_guard") && value >= 0 && value <= 2) {   // also re-arms a tripped guard
    PNPX_HIP(hipDeviceSynchronize());
    ctx->opt_range_guard = value;
    ctx->range_tripped = false;
    *ctx->range_flag_host = 0;
    return PNPX_OK;
  }
  if (is("train_cache_gb") && value >= -1) {   // -1 = automatic budget; shrinking to 0 also releases what is held
    ctx->opt_train_cache_gb = value;
    ctx->train_alloc_failed = false;
    ctx->train_budget_bytes = 0;
    if (value == 0) pnpx::train_cache_free(ctx);
    return PNPX_OK;
  }
  if (is("train_cache_release")) {   // give the ring's memory back now (budget unchanged; it re-grows on the next
    pnpx::train_cache_free(ctx);     // forward under autograd).  Outstanding tickets fall back to re-computation.
    ctx->train_alloc_failed = false;
    ctx->train_budget_bytes = 0;
    return PNPX_OK;
  }
  set_error("pnpx_ctx_set_option: unknown option '%s' or bad value %d", key ? key : "(null)", value);
  return PNPX_ERR_ARG;
}

int pnpx_ctx_get_option(pnpx_ctx* ctx, const char* key, int* value) {
  LOCK_CTX(ctx);
  auto is = [&](const char* k) { return key && !std::strcmp(key, k); };
  if (!value) {
    set_error("pnpx_ctx_get_option: null value pointer");
    return PNPX_ERR_ARG;
  }
  if (is("conv_mode")) *value = ctx->conv_mode;
  else if (is("drunet_shift")) *value = ctx->drunet.shift;
  else if (is("subbatch")) *value = ctx->opt_subbatch;
  else if (is("fuse_pool")) *value = ctx->opt_fuse_pool;
  else if (is("fuse_outc")) *value = ctx->opt_fuse_outc;
  else if (is("fuse_first")) *value = ctx->opt_fuse_first;
  else if (is("fuse_up")) *value = ctx->opt_fuse_up;
  else if (is("wreg")) *value = ctx->opt_wreg;
  else if (is("chains")) *value = ctx->opt_chains;
  else if (is("fold_first")) *value = ctx->opt_fold_first;
  else if (is("policy_s2_hs")) *value = ctx->opt_policy_s2_hs;
  else if (is("fft_affine")) *value = ctx->opt_fft_affine;
  else if (is("fft_fast")) *value = ctx->opt_fft_fast;
  else if (is("fp32_winograd")) *value = ctx->opt_fp32_winograd;
  else if (is("fp32_wino8_layers")) *value = ctx->opt_fp32_wino8;
  else if (is("fp32_fuse_up")) *value = ctx->opt_fp32_fuse_up;
  else if (is("fp32_ksplit")) *value = ctx->opt_fp32_ksplit;
  else if (is("fp32_ksplit_rule")) *value = ctx->opt_ksplit_rule;
  else if (is("fp32_chains")) *value = ctx->opt_fp32_chains;
  else if (is("fft_tile")) *value = ctx->opt_fft_tile;
  else if (is("range_guard")) *value = ctx->opt_range_guard;
  else if (is("train_cache_gb")) *value = ctx->opt_train_cache_gb;
  else {
    set_error("pnpx_ctx_get_option: unknown option '%s'", key ? key : "(null)");
    return PNPX_ERR_ARG;
  }
  return PNPX_OK;
}

int pnpx_ctx_status(pnpx_ctx* ctx) {
  LOCK_CTX(ctx);
  range_guard_enter(ctx);
  if (ctx->range_tripped) {
    if (ctx->drunet.loaded)
      set_error("half-split range guard tripped: an activation of an earlier DRUNet call left the f16 hi/lo range (|v| >= 4095 "
                "or NaN); that call's output is invalid.  The context now runs %s (the bias-free network is positively "
                "homogeneous: inputs scaled by 2^-%d, the tail multiplies back; a trip at 2^-8 selects exact fp32); acknowledge "
                "with pnpx_ctx_set_option(ctx, \"range_guard\", 1)", ctx->conv_mode == CONV_HS ? "re-scaled passes" : "conv_mode 0",
                ctx->drunet.shift);
    else
    set_error("half-split range guard tripped: an activation of an earlier call left the f16 hi/lo range (|v| >= 4095 or "
              "NaN); that call's output is invalid.  The context now runs conv_mode 0 (exact fp32 MFMA); re-arm with "
              "pnpx_ctx_set_option(ctx, \"range_guard\", 1)");
    return PNPX_ERR_RANGE;
  }
  return PNPX_OK;
}

size_t pnpx_ctx_bytes(const pnpx_ctx* ctx) {
  if (!ctx) return 0;
  size_t n = ctx->weights.bytes + ctx->arena.buf.bytes + ctx->arena_grad.buf.bytes + ctx->scratch.bytes +
             ctx->drunet.weights.bytes + ctx->drunet.arena.bytes + ctx->drunet.arena_grad.bytes + ctx->drunet.f32_weights.bytes +
             ctx->drunet.f32_arena.bytes + ctx->drunet.f32_weights_bwd.bytes + ctx->drunet.f32_arena_grad.bytes +
             ctx->policy.weights.bytes + ctx->policy.arena.bytes;
  for (const auto& sl : ctx->train_ring) n += sl.arena.buf.bytes + sl.pre.bytes;
  return n;
}

size_t pnpx_unet_num_params(void) {
  LayerSpec L[27];
  unet_layers(L);
  size_t n = 0;
  for (int i = 0; i < 27; ++i) n += (size_t)L[i].cin * L[i].cout * 9 + L[i].cout;
  return n + 32 + 1;
}

size_t pnpx_drunet_num_params(int nb) { return (nb >= 1 && nb <= 8) ? pnpx::drunet_num_params(nb) : 0; }

int pnpx_drunet_load(pnpx_ctx* ctx, const float* params_host, size_t n_params, int nb) {
  LOCK_CTX(ctx);
  return pnpx::drunet_load(ctx, params_host, n_params, nb);
}

int pnpx_unet_load(pnpx_ctx* ctx, const float* params_host, size_t n_params) {
  LOCK_CTX(ctx);
  pnpx::drunet_free(ctx);   // a context holds ONE denoiser
  if (!params_host || n_params != pnpx_unet_num_params()) {
    set_error("pnpx_unet_load: expected %zu parameters, got %zu", pnpx_unet_num_params(), n_params);
    return PNPX_ERR_ARG;
  }
  LayerSpec L[27];
  unet_layers(L);
  // device blob: packed conv weights (each 256-float aligned) + biases + outc, + slack for DMA over-read
  std::vector<float> host;
  size_t woff[27], boff[27];
  auto align = [&]() { host.resize((host.size() + 255) & ~(size_t)255, 0.f); };
  const float* src = params_host;
  for (int i = 0; i < 27; ++i) {
    const int mt = conv_pack_mt(L[i].cout), cc = conv_pack_cc(L[i].cin);
    align();
    woff[i] = host.size();
    host.resize(host.size() + (size_t)L[i].cin * L[i].cout * 9);
    pack_conv_weights(src, L[i].cout, L[i].cin, mt, cc, host.data() + woff[i]);
    src += (size_t)L[i].cin * L[i].cout * 9;
    align();
    boff[i] = host.size();
    host.insert(host.end(), src, src + L[i].cout);
    src += L[i].cout;
    ctx->conv[i].cin = L[i].cin;
    ctx->conv[i].cout = L[i].cout;
    ctx->conv[i].mt = mt;
    ctx->conv[i].cc = cc;
  }
  // Winograd F(2x2,3x3) transformed weights for the fp32 family (conv3x3_wino.hip), layers conv3x3_wino_packs() accepts
  size_t uoff[27];
  {
    const float* wsrc = params_host;
    for (int i = 0; i < 27; ++i) {
      uoff[i] = 0;
      if (conv3x3_wino_packs(L[i].cout, L[i].cin)) {
        align();
        uoff[i] = host.size();
        host.resize(host.size() + conv3x3_wino_floats(L[i].cout, L[i].cin));
        pack_conv_weights_wino(wsrc, L[i].cout, L[i].cin, host.data() + uoff[i]);
      }
      wsrc += (size_t)L[i].cin * L[i].cout * 9 + L[i].cout;
    }
  }
  // half-split (f16 hi/lo) packing of the same convolutions for conv_hs.hip
  size_t hoff[27];
  float hscale[27];
  for (int i = 0; i < 27; ++i) {
    const int mt = conv_hs_mt(L[i].cout);
    const int cin_pad = (L[i].cin + 15) / 16 * 16;
    align();
    hoff[i] = host.size();
    const size_t n16 = (size_t)L[i].cout * cin_pad * 9 * 2;       // f16 elements (hi + lo)
    host.resize(host.size() + (n16 + 1) / 2, 0.f);
    const float* wsrc = params_host;
    for (int k = 0; k < i; ++k) wsrc += (size_t)L[k].cin * L[k].cout * 9 + L[k].cout;
    hscale[i] = pack_conv_weights_hs(wsrc, L[i].cout, L[i].cin, mt, reinterpret_cast<uint16_t*>(host.data() + hoff[i]));
    ctx->conv_hs[i].cin = L[i].cin;
    ctx->conv_hs[i].cout = L[i].cout;
    ctx->conv_hs[i].cin_pad = cin_pad;
    ctx->conv_hs[i].mt = mt;
    ctx->conv_hs[i].inv_scale = 1.0f / (hscale[i] * HS_ASCALE);
  }
  // adjoint (input-gradient) convolutions for the backward pass: transposed, tap-flipped, fp32 kernel family
  size_t toff[27];
  for (int i = 0; i < 27; ++i) {
    const int cout_pad = (L[i].cin + 31) / 32 * 32;            // adjoint output channels = forward input channels
    const int mt = (cout_pad % 64 == 0) ? 64 : 32;
    align();
    toff[i] = host.size();
    host.resize(host.size() + (size_t)cout_pad * L[i].cout * 9);
    const float* wsrc = params_host;
    for (int k = 0; k < i; ++k) wsrc += (size_t)L[k].cin * L[k].cout * 9 + L[k].cout;
    pack_conv_weights_transposed(wsrc, L[i].cout, L[i].cin, cout_pad, mt, 8, host.data() + toff[i]);
    ctx->conv_bwd[i].cin = L[i].cout;
    ctx->conv_bwd[i].cout = cout_pad;
    ctx->conv_bwd[i].mt = mt;
    ctx->conv_bwd[i].cc = 8;
  }
  // ... and the same adjoint convolutions for the half-split kernel family (backward pass of conv_mode 1)
  size_t thoff[27], tuoff[27];      // tuoff: the same adjoint layers as Winograd weights of the fp32 family (conv3x3_wino8.hip, r5)
  float thscale[27];
  {
    std::vector<float> wt;
    for (int i = 0; i < 27; ++i) {
      const int cin_t = L[i].cout;                                 // adjoint input channels (multiple of 32)
      const int cout_t = (L[i].cin + 31) / 32 * 32;                // adjoint output channels, zero-padded
      const int mt = (cout_t % 64 == 0) ? 64 : 32;
      const float* wsrc = params_host;
      for (int k = 0; k < i; ++k) wsrc += (size_t)L[k].cin * L[k].cout * 9 + L[k].cout;
      wt.assign((size_t)cout_t * cin_t * 9, 0.f);
      for (int ci = 0; ci < L[i].cin; ++ci)
        for (int co = 0; co < cin_t; ++co)
          for (int tap = 0; tap < 9; ++tap)
            wt[((size_t)ci * cin_t + co) * 9 + tap] = wsrc[((size_t)co * L[i].cin + ci) * 9 + (8 - tap)];
      align();
      thoff[i] = host.size();
      const size_t n16 = (size_t)cout_t * cin_t * 9 * 2;
      host.resize(host.size() + (n16 + 1) / 2, 0.f);
      thscale[i] = pack_conv_weights_hs(wt.data(), cout_t, cin_t, mt, reinterpret_cast<uint16_t*>(host.data() + thoff[i]));
      tuoff[i] = 0;
      if (conv3x3_wino_packs(cout_t, cin_t)) {
        align();
        tuoff[i] = host.size();
        host.resize(host.size() + conv3x3_wino_floats(cout_t, cin_t));
        pack_conv_weights_wino(wt.data(), cout_t, cin_t, host.data() + tuoff[i]);
      }
      ctx->conv_hs_bwd[i].cin = cin_t;
      ctx->conv_hs_bwd[i].cout = cout_t;
      ctx->conv_hs_bwd[i].cin_pad = cin_t;
      ctx->conv_hs_bwd[i].mt = mt;
      ctx->conv_hs_bwd[i].inv_scale = 1.0f / (thscale[i] * HS_ASCALE);
    }
  }
  align();
  const size_t oc0 = host.size();          // first convolution, native [32][2][3][3] layout
  host.insert(host.end(), params_host, params_host + 32 * 2 * 9);
  align();
  const size_t ozero = host.size();
  host.resize(host.size() + 768, 0.f);
  align();
  const size_t ow = host.size();
  host.insert(host.end(), src, src + 32);
  src += 32;
  align();
  const size_t ob = host.size();
  host.push_back(*src);
  host.resize(host.size() + 1024, 0.f);  // slack: the last weight block's 16-byte DMA may over-read
  PNPX_HIP(hipDeviceSynchronize());
  if (ctx->weights.p) PNPX_HIP(hipFree(ctx->weights.p));
  ctx->weights = DeviceBuf();
  ctx->has_weights = false;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, host.size() * sizeof(float));
  if (e != hipSuccess) {
    set_error("weight allocation of %zu bytes failed: %s", host.size() * sizeof(float), hipGetErrorString(e));
    return PNPX_ERR_ALLOC;
  }
  PNPX_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  float* d = static_cast<float*>(p);
  for (int i = 0; i < 27; ++i) {
    ctx->conv[i].w = d + woff[i];
    ctx->conv[i].b = d + boff[i];
    ctx->conv_wino_u[i] = uoff[i] ? d + uoff[i] : nullptr;
    ctx->conv_wino_u_bwd[i] = tuoff[i] ? d + tuoff[i] : nullptr;
    ctx->conv_hs[i].w = reinterpret_cast<char*>(d + hoff[i]);
    ctx->conv_bwd[i].w = d + toff[i];
    ctx->conv_bwd[i].b = nullptr;
    ctx->conv_hs_bwd[i].w = reinterpret_cast<char*>(d + thoff[i]);
  }
  ctx->conv0_w = d + oc0;
  ctx->zero_bias = d + ozero;
  ctx->outc_w = d + ow;
  ctx->outc_b = d + ob;
  ctx->weights.p = p;
  ctx->weights.bytes = host.size() * sizeof(float);
  ctx->has_weights = true;
  pnpx::train_cache_free(ctx);   // activations parked by training forwards belong to the previous weights
  return PNPX_OK;
}

int pnpx_unet_denoise(pnpx_ctx* ctx, const float* x, const float* sigma, float* out, float* out_preclamp, int B,
                      int H, int W, void* stream) {
  LOCK_CTX(ctx);
  return pnpx::guarded(ctx, static_cast<hipStream_t>(stream), [&]() -> int {
  if (!x || !sigma || !out) {
    set_error("pnpx_unet_denoise: null pointer");
    return PNPX_ERR_ARG;
  }
  return unet_denoise(ctx, x, sigma, 1, out, out_preclamp, B, H, W, static_cast<hipStream_t>(stream), nullptr);
  });
}

int pnpx_unet_denoise_backward(pnpx_ctx* ctx, const float* x, const float* sigma, const float* grad_out, float* grad_x,
                               float* grad_sigma, int B, int H, int W, void* stream) {
  LOCK_CTX(ctx);
  return pnpx::guarded(ctx, static_cast<hipStream_t>(stream), [&]() -> int {
  if (!x || !sigma || !grad_out || !grad_x || !grad_sigma) {
    set_error("pnpx_unet_denoise_backward: null pointer");
    return PNPX_ERR_ARG;
  }
  return unet_denoise_backward(ctx, x, sigma, 1, grad_out, grad_x, grad_sigma, B, H, W, static_cast<hipStream_t>(stream));
  });
}

int pnpx_unet_denoise_train(pnpx_ctx* ctx, const float* x, const float* sigma, float* out, int B, int H, int W,
                            unsigned long long* ticket, void* stream) {
  LOCK_CTX(ctx);
  return pnpx::guarded(ctx, static_cast<hipStream_t>(stream), [&]() -> int {
    if (!x || !sigma || !out || !ticket) {
      set_error("pnpx_unet_denoise_train: null pointer");
      return PNPX_ERR_ARG;
    }
    return unet_denoise_train(ctx, x, sigma, 1, out, B, H, W, static_cast<hipStream_t>(stream), ticket);
  });
}

int pnpx_unet_denoise_backward_ticket(pnpx_ctx* ctx, const float* x, const float* sigma, const float* grad_out,
                                      float* grad_x, float* grad_sigma, int B, int H, int W, unsigned long long ticket,
                                      void* stream) {
  LOCK_CTX(ctx);
  return pnpx::guarded(ctx, static_cast<hipStream_t>(stream), [&]() -> int {
    if (!x || !sigma || !grad_out || !grad_x || !grad_sigma) {
      set_error("pnpx_unet_denoise_backward_ticket: null pointer");
      return PNPX_ERR_ARG;
    }
    return unet_denoise_backward_ticket(ctx, x, sigma, 1, grad_out, grad_x, grad_sigma, B, H, W,
                                        static_cast<hipStream_t>(stream), ticket);
  });
}

int pnpx_unet_profile(pnpx_ctx* ctx, const float* x, const float* sigma, float* out, int B, int H, int W,
                      void* stream, int cap, float* ms_out, double* flops_out, const char** names_out, int* n_out) {
  LOCK_CTX(ctx);
  if (!x || !sigma || !out || !ms_out || !flops_out || !names_out || !n_out || cap <= 0) {
    set_error("pnpx_unet_profile: null pointer");
    return PNPX_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // make sure the arena exists before timing (growth synchronises)
  PNPX_TRY(unet_denoise(ctx, x, sigma, 1, out, nullptr, B, H, W, s, nullptr));
  while (ctx->events.size() < 1024) {
    hipEvent_t e;
    PNPX_HIP(hipEventCreate(&e));
    ctx->events.push_back(e);
  }
  ProfileSink sink;
  sink.events = &ctx->events;
  PNPX_TRY(unet_denoise(ctx, x, sigma, 1, out, nullptr, B, H, W, s, &sink));
  PNPX_HIP(hipStreamSynchronize(s));
  int n = sink.n < cap ? sink.n : cap;
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    PNPX_HIP(hipEventElapsedTime(&ms, ctx->events[i], ctx->events[i + 1]));
    ms_out[i] = ms;
    flops_out[i] = sink.flops[i];
    names_out[i] = sink.names[i];
  }
  *n_out = n;
  return PNPX_OK;
}

int pnpx_fft2(pnpx_ctx* ctx, const float* in, float* out, int n_img, int H, int W, int inverse, int centered,
              void* stream) {
  LOCK_CTX(ctx);
  if (!in || !out) {
    set_error("pnpx_fft2: null pointer");
    return PNPX_ERR_ARG;
  }
  return fft2(ctx, in, out, n_img, H, W, inverse != 0, centered != 0, static_cast<hipStream_t>(stream));
}

}  // extern "C"
