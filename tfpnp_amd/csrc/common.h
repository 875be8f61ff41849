// Internal declarations shared by the translation units of libpnpx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pnpx.h"

namespace pnpx {

// ---------------------------------------------------------------------------------------------------
// Error plumbing: nothing throws across the C ABI.
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define PNPX_HIP(expr)                                                       \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return ::pnpx::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)
#define PNPX_TRY(expr)          \
  do {                          \
    int _s = (expr);            \
    if (_s != PNPX_OK) return _s; \
  } while (0)
#define PNPX_LAUNCH_CHECK() PNPX_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------------------------
// Padded planar activation layout used INSIDE the denoiser (never visible at the C ABI):
//   tensor [B][C][Hp][Wp] fp32, interior pixel (y,x) at row y+1, column x+PADL.  All border cells are
//   zero for the lifetime of the workspace (memset once, producers write interiors only), so 3x3
//   convolutions read their zero padding straight from memory and need no bounds checks.
constexpr int PADL = 4;
__host__ __device__ inline int padded_w(int W) { return W + 2 * PADL; }
__host__ __device__ inline int padded_h(int H) { return H + 2; }

struct ActDesc {      // one activation tensor in the arena
  size_t off = 0;     // float offset into the arena
  int C = 0, H = 0, W = 0;
  size_t per_image() const { return (size_t)C * padded_h(H) * padded_w(W); }
};

// ---------------------------------------------------------------------------------------------------
struct ConvLayer {          // one 3x3 conv of the UNet, weights repacked for the MFMA kernel
  int cin = 0, cout = 0;
  int mt = 0, cc = 0;       // cout tile and cin chunk the packing was made for
  float* w = nullptr;       // device: [cout/mt][cin/cc][9][cc][mt]
  float* b = nullptr;       // device: [cout]
};

struct DeviceBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct UNetArena {
  DeviceBuf buf;
  int capB = 0, capH = 0, capW = 0, mode = -1;
};

struct ConvLayerHsDev {     // the same conv packed for the half-split f16 kernel (conv_hs.hip)
  int cin = 0, cout = 0, cin_pad = 0, mt = 0;
  char* w = nullptr;        // device: [cout/mt][cin_pad/16][9][hi,lo][2][mt][8] f16
  float inv_scale = 1.f;    // 1 / (weight scale * activation scale)
};

enum ConvMode { CONV_F32 = 0, CONV_HS = 1 };

// ---------------------------------------------------------------------------------------------------
// Policy actor (ResNet-18 encoder + heads, eval mode) -- policy.hip.  One PolicyConv per launch: BatchNorm is folded
// into weights/bias at load time; stride-2 convolutions run as stride-1 convolutions over a space-to-depth input
// with per-(cout tile, K-chunk) tap masks; the 1x1 stride-2 shortcut rides in the same launch as extra cout tiles.
struct PolStep;
struct PolicyConv {
  const float* w = nullptr;              // device: present tap slices [8][64], in PolStep order
  const float* bias = nullptr;           // device: [cout]
  const PolStep* steps = nullptr;        // device: [cout/64][cin/8] (first nsteps[ct] entries of each row valid)
  const int* nsteps = nullptr;           // device: [cout/64]
  int cin = 0;      // K channels of the (possibly space-to-depth) input, multiple of 8
  int cout = 0;     // output channels over both outputs, multiple of 64
  int split_c = 0;  // channels written to the first output (ReLU); the rest go to the second output (linear)
};
struct PolicyNet {
  bool loaded = false;
  int num_inputs = 0, cin_pad = 0, n_det = 0, spi_head = 0;
  PolicyConv conv[17];       // stem, then per stage: conv1+shortcut, conv2, block-2 conv1, block-2 conv2
  // the twelve stride-1 convolutions (per stage: conv2, block-2 conv1, block-2 conv2) packed for the half-split
  // f16x3 MFMA kernel (conv_hs.hip); conv[] keeps their fp32 packing only for the stride-2 launches and the stem
  ConvLayerHsDev conv_hs[12];
  const float* bias_hs[12] = {};
  // stage entries (stride 2) on the sparse-tap half-split instances over the HS8 space-to-depth input:
  // [st][0] = conv1 as a 2x2-window convolution (tap mask 0x01B), [st][1] = the 1x1 shortcut (0x010, first Cin channels)
  ConvLayerHsDev s2_hs[4][2];
  const float* s2_bias[4][2] = {};
  ConvLayerHsDev stem_hs;          // the stem (3x3 stride 2) as a 2x2-window sparse-tap launch over the HS8 space-to-depth observation
  const float* stem_hs_bias = nullptr;
  const float* fc_sm_w = nullptr;   // [2][512], [2]
  const float* fc_sm_b = nullptr;
  const float* fc_det_w = nullptr;  // [n_det][512] ([64][512] with the SPI head)
  const float* fc_det_b = nullptr;
  const float* fc_det2_w = nullptr; // SPI head only: [n_det][64]
  const float* fc_det2_b = nullptr;
  DeviceBuf weights;
  DeviceBuf arena;           // activations for capB observations of capH x capW, zero borders
  int capB = 0, capH = 0, capW = 0;
};

// number of independent launch chains for a B-image denoiser forward (unet.hip; option "chains", 0 = automatic)
int launch_chains(const pnpx_ctx* ctx, int B, int H, int W);
// a launch-chain fan-out failed half way: drain the side streams before the error is returned, so that nothing queued on them
// is still writing the shared arena / output buffers when the caller (or the next call) re-uses them
void join_side_streams_after_failure(pnpx_ctx* ctx);
// DRUNet denoiser (drunet.hip): packed weights per MFMA launch in state_dict order + its own activation arena
struct DruNet {
  bool loaded = false;
  int nb = 0;
  std::vector<ConvLayerHsDev> layers;
  std::vector<int> taps;           // per layer: 3x3 tap mask its weights are packed with (0x010 for the 1x1 layers)
  const float* head_w = nullptr;   // [64][2][3][3] native (VALU head convolution)
  const float* zero = nullptr;     // [1024] zeros: the bias operand of the bias-free network
  const float* e0 = nullptr;       // [5][32]: row k = (2^(4k), 0, ...): channel selector of the fused tail epilogue that also undoes a
                                   // 2^-(4k) down-scaling of the pass (shift, below); followed by zeros
  // Range handling.  The network is bias-free with ReLU activations only, i.e. positively homogeneous: f(a * in) = a * f(in) for
  // a > 0, and a power-of-two `a` commutes exactly with every step (the hi/lo split included, subnormals aside).  When the
  // half-split range guard trips (an activation left |v| < 4095) the pass is repeated / later passes run on inputs scaled by
  // 2^-shift (in the head convolution's store) with the tail multiplying back.  0, 4 or 8; a trip at 8 latches the context to
  // conv_mode 0 (drunet_f32.hip) like a UNet context.
  int shift = 0;
  // adjoint (input-gradient) layers, index-parallel to `layers` (transposed, tap-flipped; the head's adjoint is a 64 -> 32
  // launch whose channels 0 / 1 are the image / noise-map gradients; the tail's adjoint runs on the vector ALU)
  std::vector<ConvLayerHsDev> layers_bwd;
  const float* tail_bwd_w = nullptr;   // [64][2][3][3]: conv_first_hs_kernel weights of the tail's adjoint (1 -> 64 channels)
  DeviceBuf weights, arena, arena_grad;
  // conv_mode 0 (drunet_f32.hip): fp32 packings of the same layers, made on first use from the host copy of the parameters
  std::vector<float> params_host;
  bool f32_ready = false;
  std::vector<ConvLayer> f32_layers;       // direct fp32 MFMA kernel
  std::vector<const float*> f32_wino;      // Winograd weights of the 3x3 ResBlock layers (else null)
  DeviceBuf f32_weights, f32_arena;
  int f32_capB = 0, f32_capH = 0, f32_capW = 0;
  bool f32_arena_keeps = false;            // ... with room for every ResBlock's middle activation (the fp32 VJP, r5)
  bool f32_bwd_ready = false;              // adjoint packings of the fp32 family (made on the first fp32 VJP)
  std::vector<ConvLayer> f32_layers_bwd;
  std::vector<const float*> f32_wino_bwd;
  DeviceBuf f32_weights_bwd, f32_arena_grad;
  int f32_gcapB = 0, f32_gcapH = 0, f32_gcapW = 0;
  int capB = 0, capH = 0, capW = 0;
  bool arena_keeps = false;            // the arena has room for every ResBlock's middle activation (backward pass)
  int gcapB = 0, gcapH = 0, gcapW = 0;
};

}  // namespace pnpx

struct pnpx_ctx {
  int device = 0;
  std::mutex mu;
  // --- denoiser
  bool has_weights = false;
  pnpx::ConvLayer conv[27];
  pnpx::ConvLayerHsDev conv_hs[27];
  int conv_mode = pnpx::CONV_F32;  // which conv kernel family the denoiser runs (pnpx_ctx_set_option).  r6: the DEFAULT is the fp32 family (the
                                   // reference's own arithmetic, inside 1e-4 of the fp32 oracle on every drift seed); CONV_HS is the opt-in fast mode
  // --- options (pnpx_ctx_set_option; no environment variables are read on any launch path)
  int opt_subbatch = 24;           // images per level-0 sub-batch of the half-split forward (0 = whole batch)
  int opt_fuse_pool = 1;           // fused 2x2 max-pool epilogue
  int opt_fuse_outc = 1;           // fused 1x1 out-conv + residual + clamp epilogue
  int opt_fuse_first = 1;          // VALU first convolution straight from the fp32 image (no padded-input tensor)
  int opt_fuse_up = 1;             // bilinear x2 of the full-resolution decoder entry inside the conv kernel (producer waves): r5 default --
                                   // bit-identical to the separate kernel (tools/ab_fuse_up.py), forward 5.82 -> 5.76 ms at 48 x 256^2
  int opt_policy_s2_hs = 1;        // policy actor: the stride-2 stage entries on the sparse-tap half-split instances
  int opt_fold_first = 0;          // opt-in: first convolution folded into the loader of the second one (conv_hs WREG == 2;
                                   // bit-identical, 400 MB less HBM traffic per forward, time-neutral: 5.835 vs 5.841 ms)
  int opt_fft_tile = 0;            // complex points per FFT workgroup tile (0 = FFT_TILE_POINTS)
  int opt_fft_fast = 1;            // N = 256 lines on the register-radix-16 FFT kernels (fft_lds.h fft256_*)
  int opt_fft_affine = 1;          // XCD-affine block -> image mapping of the FFT passes (fft_lds.h)
  int opt_chains = 0;              // denoiser forward as n independent launch chains over slices of the batch (0 = auto)
  std::vector<hipStream_t> side_streams;
  // fork / join events of the launch chains: a ROTATING pool.  An event must not be re-recorded while a hipStreamWaitEvent on
  // its previous record may still sit un-submitted in another stream's host-side queue: with one fork event and one join event
  // per side stream re-recorded every call, two host threads driving two contexts on one GPU lost joins (a slice's output read
  // before it was written: one item of a batch wrong, 15 of 25 stress runs; r4, tools/attic/stress_threads.py).
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
  int opt_wreg = 2;                // weights-in-registers instances for the 32 -> 32 channel layers (0 off, 1 / 2 = shape)
  int opt_range_guard = 1;         // 0 off, 1 sticky flag + latch to conv_mode 0, 2 strict (sync + transparent re-run)
  int opt_train_cache_gb = -1;     // training path: keep the activations of up to this many GiB of denoiser forwards for
                                   // the backward pass instead of re-computing them (0 = always re-compute; -1 = auto:
                                   // a quarter of the device memory free when the ring is first laid out, <= 96 GiB)
  size_t train_budget_bytes = 0;   // the automatic budget once measured (reset by set_option / train_cache_release)
  // --- half-split range guard: host-mapped word the conv_hs epilogues set when a stored value leaves the f16 range
  unsigned* range_flag_host = nullptr;   // pinned host allocation
  unsigned* range_flag_dev = nullptr;    // its device address
  bool range_tripped = false;            // latched by the host once the flag was seen set (cleared by set_option)
  float* conv_wino_u[27] = {};     // Winograd-transformed fp32 weights of the layers conv3x3_wino.hip can run (else null)
  float* conv_wino_u_bwd[27] = {}; // ... of the adjoint (transposed, tap-flipped) convolutions of the fp32 backward pass (r5)
  int opt_fp32_winograd = 1;       // conv_mode 0: run those layers as F(2x2,3x3) (fp32 arithmetic, 2.25x fewer MFMAs; 0 = the direct kernel everywhere)
  int opt_fp32_wino8 = (1 << 27) - 1;   // bit li: layer li runs on the 8-wave Winograd kernel (conv3x3_wino8.hip) where its geometry allows; a DRUNet
                                   // context: any bit = its ResBlock layers do
  int opt_fp32_chains = 2;         // conv_mode 0: n >= 2 = the forward as n launch chains over slices of the batch (default 2: this family is not
                                   // power-capped, a second chain fills the other's tails and partial rounds: 8.76 -> 8.44 ms at 48 x 256^2, -7 % at
                                   // B = 24; 3 and 4 chains measured slower); 1 = only the bottom level forks two chains; 0 = one chain
  int opt_fp32_ksplit = 1;         // conv_mode 0, r6: the deep levels' layers split their channel chunks over 2 / 4 work items (conv3x3_wino8.hip KSPLIT):
                                   // 1 = in calls whose unsplit tiles cannot fill the chip, 2 = always (bit-identical across all batch sizes), 0 = never
  int opt_ksplit_rule = 1;         // tuning: pieces per tile class (conv3x3_wino8_ksplit; 1 = the default rule)
  int opt_fp32_fuse_up = 1;        // conv_mode 0: the decoder entries up-sample their second source inside the 8-wave Winograd kernel (no up-sampled tensor)
  pnpx::ConvLayer conv_bwd[27];    // adjoint (input-gradient) convolutions, fp32 kernel family
  pnpx::ConvLayerHsDev conv_hs_bwd[27];  // ... and packed for the half-split kernel family
  float* zero_bias = nullptr;      // [768] zeros (bias operand of the adjoint convolutions)
  float* conv0_w = nullptr;  // [32][2][9] fp32: the first convolution in its native layout (VALU kernel, unet.hip)
  float* outc_w = nullptr;   // [32]
  float* outc_b = nullptr;   // [1]
  pnpx::DeviceBuf weights;   // single allocation holding all of the above
  // --- UNet activation arenas (capacity capB images of capH x capW, zero borders valid for one layout `mode`):
  //     arena = forward pass in the ctx's conv_mode (also the re-computation inside the backward pass);
  //     arena_grad = gradients of every activation (fp32 planar)
  pnpx::UNetArena arena, arena_grad;
  // training path: a ring of activation arenas.  Every denoiser forward run for autograd (pnpx_unet_denoise_train, the
  // iterations of pnpx_csmri_admm_train) takes the next slot, keeps every activation + the pre-clamp output there and
  // hands out a ticket; a VJP that presents a ticket still held by its slot skips the re-computation of the forward.
  struct TrainSlot {
    pnpx::UNetArena arena;
    pnpx::DeviceBuf pre;
    unsigned long long ticket = 0;
    int B = 0;
  };
  std::vector<TrainSlot> train_ring;
  unsigned long long train_counter = 0;
  int train_H = 0, train_W = 0, train_mode = -1;   // geometry the ring is laid out for
  bool train_alloc_failed = false;                 // stop retrying after an out-of-memory until the option is set again
  // --- policy actor
  pnpx::PolicyNet policy;
  // --- DRUNet denoiser (when loaded it IS the context's denoiser: every prox call of every solver runs it)
  pnpx::DruNet drunet;
  // --- solver scratch (complex fields etc.), grown on demand
  pnpx::DeviceBuf scratch;
  // --- FFT twiddle tables e^{-2 pi i m / N}, one per transform length: device float2[N]
  std::map<int, float2*> twiddle;
  // --- events for pnpx_unet_profile
  std::vector<hipEvent_t> events;
};

namespace pnpx {

int chain_event(pnpx_ctx* ctx, hipEvent_t* ev);        // next event of the rotating pool (unet.hip)
int chain_streams(pnpx_ctx* ctx, int chains);          // make sure chains - 1 side streams exist
// Run `chains` contiguous slices of a B-item batch as independent launch chains: slice 0 on the caller's stream `s`, the others
// on the context's side streams, forked from / joined back into `s` by events.  run_slice(lo, hi, stream) -> status.
template <class F>
int fan_out_chains(pnpx_ctx* ctx, int chains, int B, hipStream_t s, F&& run_slice) {
  PNPX_TRY(chain_streams(ctx, chains));
  hipEvent_t fork = nullptr, joins[16] = {};
  PNPX_TRY(chain_event(ctx, &fork));
  PNPX_HIP(hipEventRecord(fork, s));
  // from here on work may sit on the side streams: EVERY error exit drains them first (nothing queued there may outlive the
  // failing call and write the shared arena / output buffers while the caller re-uses them)
  auto fail = [&](int rc) {
    join_side_streams_after_failure(ctx);
    return rc;
  };
  auto hip = [&](hipError_t e, const char* what) { return e == hipSuccess ? PNPX_OK : hip_fail(e, what, __FILE__, __LINE__); };
  for (int c = chains - 1; c >= 0; --c) {       // the caller's stream takes slice 0 last: its host-side issue overlaps
    const int lo = (int)((long long)B * c / chains), hi = (int)((long long)B * (c + 1) / chains);
    hipStream_t st = c ? ctx->side_streams[c - 1] : s;
    int rc = c ? hip(hipStreamWaitEvent(st, fork, 0), "hipStreamWaitEvent(fork)") : PNPX_OK;
    if (rc == PNPX_OK) rc = run_slice(lo, hi, st);
    if (rc == PNPX_OK && c) rc = chain_event(ctx, &joins[c]);
    if (rc == PNPX_OK && c) rc = hip(hipEventRecord(joins[c], st), "hipEventRecord(join)");
    if (rc != PNPX_OK) return fail(rc);
  }
  for (int c = 1; c < chains; ++c) {
    const int rc = hip(hipStreamWaitEvent(s, joins[c], 0), "hipStreamWaitEvent(join)");
    if (rc != PNPX_OK) return fail(rc);
  }
  return PNPX_OK;
}

int ctx_reserve_unet(pnpx_ctx* ctx, int B, int H, int W);   // main arena, ctx->conv_mode
int reserve_arena(pnpx_ctx* ctx, UNetArena& ar, int mode, int B, int H, int W, size_t extra_bytes);
int ctx_scratch(pnpx_ctx* ctx, size_t bytes, void** out);
int ctx_twiddle(pnpx_ctx* ctx, int N, const float2** out);
// Half-split range guard (api.hip).  range_guard_enter: called at the top of every entry that runs the denoiser; if the
// flag of an earlier call is set, latches the context to conv_mode 0.  range_guard_strict: option value 2 -- synchronise
// `s`, and if the flag was set by THIS call return true after switching to conv_mode 0 so the caller re-runs the call.
void range_guard_enter(pnpx_ctx* ctx);
int range_guard_strict(pnpx_ctx* ctx, hipStream_t s, bool* rerun);
// Wraps the body of a C-ABI entry that runs the denoiser (entries are functional: inputs are never modified, so a body
// can simply be executed again).
template <class F>
int guarded(pnpx_ctx* ctx, hipStream_t s, F&& body) {
  range_guard_enter(ctx);
  int st = body();
  // strict mode: the UNet repeats the call once in exact fp32; a DRUNet context repeats it on inputs scaled down by another
  // factor 16 per attempt (api.hip::range_guard_strict) until the guard stays quiet
  // (a DRUNet context: 2^-4, 2^-8, then exact fp32 -- api.hip::drunet_rescale)
  for (int attempt = 0; attempt < 4 && st == PNPX_OK && ctx->opt_range_guard == 2 && ctx->conv_mode == CONV_HS; ++attempt) {
    bool rerun = false;
    PNPX_TRY(range_guard_strict(ctx, s, &rerun));
    if (!rerun) break;
    st = body();
  }
  return st;
}

struct ProfileSink {      // optional per-launch event recording for pnpx_unet_profile
  std::vector<hipEvent_t>* events = nullptr;
  std::vector<const char*> names;
  std::vector<double> flops;
  int n = 0;
};

// Denoiser forward on padded input already resident in the arena is internal; these are the pieces the
// solver loops call.  x/sigma/out are unpadded [B,1,H,W] / [B] tensors (C-ABI layout).
int unet_denoise(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, float* out_pre,
                 int B, int H, int W, hipStream_t s, ProfileSink* prof, UNetArena* arena = nullptr, int mode = -1,
                 bool keep_all = false);   // keep_all: store every activation (no fused network tail) -- backward pass
// VJP of the denoiser wrt x and sigma (unet_bwd.hip): recomputes the forward pass in fp32 and back-propagates.
// cached != NULL: the activations (a keep_all forward of exactly this input in `cached`) and its pre-clamp output are
// reused instead of re-computed.
int unet_denoise_backward(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, const float* grad_out,
                          float* grad_x, float* grad_sigma, int B, int H, int W, hipStream_t s,
                          UNetArena* cached = nullptr, const float* cached_pre = nullptr);
size_t unet_arena_bytes(int mode, int B, int H, int W);
void train_cache_free(pnpx_ctx* ctx);
// Denoiser forward for autograd: like unet_denoise, additionally parks the activations in the training ring; *ticket = 0
// when the ring is disabled / out of budget (the VJP then re-computes).
int unet_denoise_train(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, int B, int H,
                       int W, hipStream_t s, unsigned long long* ticket);
// VJP that looks the ticket up in the ring first (ticket 0 / overwritten slot: re-computation)
int unet_denoise_backward_ticket(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride,
                                 const float* grad_out, float* grad_x, float* grad_sigma, int B, int H, int W,
                                 hipStream_t s, unsigned long long ticket);

// DRUNet denoiser (drunet.hip)
size_t drunet_num_params(int nb);
int drunet_load(pnpx_ctx* ctx, const float* params, size_t n, int nb);
int drunet_denoise(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, float* out_pre, int B,
                   int H, int W, hipStream_t s, bool keep_mids = false);
// VJP wrt x and sigma: forward re-computed keeping every ResBlock's ReLU output, then the adjoint chain on the same kernels
int drunet_denoise_backward(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, const float* grad_out,
                            float* grad_x, float* grad_sigma, int B, int H, int W, hipStream_t s);
void drunet_free(pnpx_ctx* ctx);
int drunet_denoise_f32(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, float* out, float* out_pre, int B, int H,
                       int W, hipStream_t s, bool keep_mids = false);
int drunet_denoise_backward_f32(pnpx_ctx* ctx, const float* x, const float* sigma, int sigma_stride, const float* grad_out, float* grad_x,
                                float* grad_sigma, int B, int H, int W, hipStream_t s);
void drunet_f32_free(pnpx_ctx* ctx);

// Policy actor (policy.hip)
size_t policy_num_params(int num_inputs, int n_det, int spi_head);
int policy_load(pnpx_ctx* ctx, const float* params, size_t n, int num_inputs, int n_det, int spi_head);
int policy_forward(pnpx_ctx* ctx, const float* ob, float* probs, float* det, int B, int H, int W, hipStream_t s);
void policy_free(pnpx_ctx* ctx);

// FFT building blocks (fft.hip)
int fft2(pnpx_ctx* ctx, const float* in, float* out, int n_img, int H, int W, bool inverse, bool centered,
         hipStream_t s);

}  // namespace pnpx
