/* pnpx.h -- C ABI of the MI355X-native PnP proximal-solver inner loop (libpnpx.so).
 *
 * Plain C: pointers, sizes, ints.  No torch / HIP types in any signature (a HIP stream is passed as
 * void*, a null stream is the device's default stream).  Every function returns 0 on success and a
 * non-zero pnpx_status otherwise; nothing throws across this boundary.  pnpx_last_error() returns a
 * thread-local description of the last failure.
 *
 * Each entry point replaces one piece of the reference (Vandermode/TFPnP, /root/reference); the
 * file:line it stands in for is cited on the declaration.  Tensor layouts at this boundary are the
 * reference's own: contiguous fp32, NCHW, complex numbers as a trailing dimension of 2 (re, im),
 * masks as one byte per pixel (torch.bool storage).  All pointers are DEVICE pointers unless a
 * parameter name ends in _host.
 *
 * Threading: a pnpx_ctx belongs to one device; calls on the same ctx are serialised by an internal
 * mutex; different ctxs (one per GPU / per thread, as torch.nn.DataParallel would drive the reference,
 * tfpnp/policy/sync_batchnorm/replicate.py:50-75) are independent.  Work is enqueued on the caller's
 * stream; no call synchronises the device except ctx creation, weight upload and workspace growth.
 */
#ifndef PNPX_H
#define PNPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pnpx_ctx pnpx_ctx; /* opaque: device id, packed UNet weights, workspaces, FFT tables */

enum pnpx_status {
  PNPX_OK = 0,
  PNPX_ERR_ARG = 1,         /* null pointer / non-positive size                                  */
  PNPX_ERR_SHAPE = 2,       /* unsupported geometry (e.g. FFT length > 2048, image side < 16)    */
  PNPX_ERR_NO_WEIGHTS = 3,  /* denoiser used before pnpx_unet_load                               */
  PNPX_ERR_ALLOC = 4,       /* device allocation failed                                          */
  PNPX_ERR_HIP = 5,         /* a HIP runtime call failed; see pnpx_last_error()                  */
  PNPX_ERR_RANGE = 6        /* half-split range guard tripped (pnpx_ctx_status)                  */
};

const char* pnpx_version(void);
const char* pnpx_last_error(void);

/* ---- context ------------------------------------------------------------------------------------- */
int pnpx_ctx_create(int device, pnpx_ctx** out);
int pnpx_ctx_destroy(pnpx_ctx* ctx);
/* Pre-size the internal workspaces for batches up to B of H x W images (optional; otherwise they
 * grow on first use, which synchronises the device once). */
int pnpx_ctx_reserve(pnpx_ctx* ctx, int B, int H, int W);
/* Options (the library reads no environment variables).
 *  "conv_mode": 0 (default since r6) = fp32 MFMA convolutions, the reference's own arithmetic (csrc/conv3x3_wino8.hip /
 *      conv3x3_wino.hip for the layers "fp32_winograd" covers, csrc/conv3x3.hip otherwise): inside 1e-4 of the fp32 oracle over a
 *      whole 30-iteration episode on every seed tested, also on expansive weights (profiles/r5_drift_seeds.md).
 *      1 = the FAST mode: half-split f16 MFMA convolutions (every value an f16 hi + lo pair = 22-bit significand, 3 MFMAs per
 *      product, csrc/conv_hs.hip): 1e-6-class per call and ~1.55x the episode rate, but narrower than fp32 -- on the chaotic
 *      expansive-weight episode 2 of 12 seeds end at 1.0e-4 / 1.24e-4 from the fp32 oracle (default-scale weights: 1e-6 both
 *      modes), and |v| < 4095 behind "range_guard".
 *  "fp32_winograd" (default 1): in conv_mode 0, layers with cout % 64 == 0 (sources % 16, H and W % 16) or cout % 32 == 0
 *      (sources % 8, H % 16, W % 32) run as Winograd F(2x2,3x3) in fp32 (1.3-1.8x per layer; 7e-7 from the fp64 oracle where
 *      the direct kernel is 1.1e-6 -- same accuracy class, different summation order).  0 = the direct kernel on every layer.
 *  "fp32_wino8_layers" (default: all 27 bits set): bit i = layer i of the UNet (state_dict order) runs on the 8-wave Winograd kernel
 *      (csrc/conv3x3_wino8.hip: the 16 positions of a block split over the two waves of a SIMD; forward and adjoint convolutions) where
 *      its geometry allows; 0 = the round-4 4-wave kernel (csrc/conv3x3_wino.hip) everywhere.  A DRUNet context: any bit.
 *  "fp32_chains" (default 2): in conv_mode 0 the denoiser forward runs as n independent launch chains over contiguous slices of the batch
 *      (caller's stream + side streams, joined before the entry returns; bit-identical per image): 2 measured best (-3.6 % at 48 x 256^2,
 *      -7 % at B = 24); 1 = only the bottom level's three launches fork two chains; 0 = one chain.  The VJP's adjoint chain forks exactly
 *      two chains for every n >= 2.
 *  "fp32_fuse_up" (default 1): in conv_mode 0 the decoder-entry convolutions interpolate the bilinear x2 up-sampling of their second
 *      source inside the 8-wave kernel (tfpnp/pnp/denoiser/models/unet.py:92-121; no up-sampled tensor); 0 = the separate kernel
 *      (results agree to 5e-7).
 *  "fp32_ksplit" (default 1, r6): in conv_mode 0 the layers of the two deepest levels (<= 16 tiles of 64 couts x 16 x 16 px per image)
 *      can split their input-channel chunks over 4 / 2 workgroups of the 8-wave kernel; the pieces' partial sums go to a scratch slab and
 *      a second small launch adds them in piece order (deterministic), then bias + activation.
 *      1 = in calls whose unsplit tiles cannot fill the chip (tiles per image x batch < 256: below 32 images of 256 x 256 at the
 *      16 x 16 level, below 16 at 32 x 32; the 32 x 32 decoder entry then runs unfused so that it splits too): -20 % per forward at B = 6,
 *      -9 % at 12, nothing changes from 32 images up.  A given image
 *      gets the same bits in every call of the same class; across the class boundary results differ in the summation order
 *      (5e-7).  2 = split at every batch size: bit-identical per image across ALL batch sizes, +5 % at 48 x 256^2.  0 = never.
 *      "fp32_ksplit_rule" (tuning): pieces per tile class.
 *  "range_guard": the half-split kernels carry activations as f16 hi+lo pairs of 16*v, i.e. |v| < 4095.  Their
 *      epilogues set a sticky flag when a stored value leaves that range or is NaN.
 *      1 (default): the flag is looked at (no synchronisation) at the top of the next call; once seen, the context
 *         switches to conv_mode 0 for good and pnpx_ctx_status() returns PNPX_ERR_RANGE (the call that tripped it
 *         returned invalid numbers).  Setting the option again re-arms the guard.
 *      2 (strict): every denoiser / solver entry synchronises its stream before returning and, if the flag was set,
 *         repeats itself in conv_mode 0 -- the caller always receives valid output, at the price of a sync per call.
 *      0: off.
 *  "train_cache_gb" (default -1): device-memory budget of the training path's activation cache (see
 *      pnpx_csmri_admm_train); -1 = a quarter of the device memory free when the ring is first laid out, at most 96 GiB
 *      (the ring is raw device memory outside any framework allocator); 0 releases it and makes every backward
 *      re-compute.  "train_cache_release" (any value): give the ring's memory back now, budget unchanged.
 *  "wreg" (default 2): weights-in-registers instances of the 32 -> 32 channel convolutions (0 = generic kernel, 1 = four
 *      waves x four pixel blocks, 2 = eight waves x two pixel blocks); bit-identical results.
 *  "chains" (default 0 = automatic): run a denoiser forward (and the half-split VJP's adjoint chain) as n independent launch chains over
 *      slices of the batch on side streams; the convolution launch table picks its tile shapes for the chains running side by side.
 *      Automatic (r5): two chains from 5 images of 256 x 256 up (-4 % at B = 6, -13 % at 9-11, -6 % at 48; profiles/r5_chains_table_hs.txt).
 *      Bit-identical per image for every n.
 *  "fft_affine" (default 1), "fft_tile" (default 0 = 1024 points): XCD-affine image mapping and tile size of the FFT passes.
 *  "fft_fast" (default 1): N = 256 lines on the register-radix-16 kernels (0 = the generic Stockham passes; same results to rounding).
 *  "fuse_first" (default 1): the half-split family's first convolution reads the fp32 image and the noise level directly (no padded
 *      two-channel input tensor); 0 = separate input preparation + the generic kernel (bit-identical).
 *  "policy_s2_hs" (default 1): the policy actor's stem and stride-2 stage entries on the sparse-tap half-split instances over space-to-depth
 *      tensors (0 = the fp32 space-to-depth convolution kernel of policy_conv.hip for those layers).
 *  "fold_first" (default 0): 1 = the network's first convolution is evaluated inside the tile loader of the second one
 *      (its output tensor is neither written nor read; bit-identical, time-neutral).
 *  "fuse_up" (default 1 since r5): 1 = the full-resolution decoder entry (96 -> 32 channels) up-samples its low-resolution source
 *      inside the convolution kernel (four producer waves per workgroup interpolate each K-chunk's halo into LDS), so the
 *      largest up-sampled tensor never exists in HBM.  Same arithmetic (bit-identical on the r5 build at every size tried,
 *      tools/ab_fuse_up.py; the test bound is 1e-6), forward 5.82 -> 5.76 ms at 48 x 256^2.  (conv_mode 0 has its own switch,
 *      "fp32_fuse_up", default 1: all four decoder entries interpolate inside the 8-wave Winograd kernel.)
 *      2 = opt-in: the 64-cout decoder entries too (8-row-tile instance; bit-identical, measured 3 % SLOWER than their separate
 *      up-sampling launches: profiles/r5_hs_fuse_up.md).
 *  "subbatch" (images per level-0 sub-batch, 0 = whole batch), "fuse_pool", "fuse_outc" (0/1): diagnostics. */
int pnpx_ctx_set_option(pnpx_ctx* ctx, const char* key, int value);
int pnpx_ctx_get_option(pnpx_ctx* ctx, const char* key, int* value);
/* PNPX_OK, or PNPX_ERR_RANGE once the range guard has tripped.  Does not synchronise: call it after a point where
 * the stream is known to have drained (e.g. after reading a result back). */
int pnpx_ctx_status(pnpx_ctx* ctx);
/* Bytes of device memory currently held by the context (weights + workspaces). */
size_t pnpx_ctx_bytes(const pnpx_ctx* ctx);

/* ---- denoiser prox: UNetDenoiser2D (tfpnp/pnp/denoiser/base.py:7-32, models/unet.py:34-66) ------- */
/* Number of fp32 parameters of UNet(2,1) (11 773 857) = length of the blob below. */
size_t pnpx_unet_num_params(void);
/* params_host: the reference state_dict's 56 tensors concatenated in state_dict order
 * (inc.conv.conv-0.conv2d.weight, ...bias, ..., outc.conv.weight, outc.conv.bias), each in its native
 * [Cout,Cin,kh,kw] layout -- i.e. what torch.load('unet-nm.pt') holds (denoiser/base.py:15-16).
 * Repacked on the host into the MFMA tile layout and uploaded. */
int pnpx_unet_load(pnpx_ctx* ctx, const float* params_host, size_t n_params);
/* out = clamp(UNet(cat[x, sigma*1]), 0, 1)   (denoiser/base.py:23-32).
 * x, out: [B,1,H,W]; sigma: [B]; out_preclamp (nullable): UNet output before the clamp.
 * Any H, W >= 16; odd level sizes follow the reference's floor-pool / zero-pad rule (models/unet.py:82-85,109-113). */
int pnpx_unet_denoise(pnpx_ctx* ctx, const float* x, const float* sigma, float* out, float* out_preclamp,
                      int B, int H, int W, void* stream);
/* ---- DRUNet denoiser (BASELINE config #5).  The reference ships only the KAIR building blocks
 * (tfpnp/pnp/denoiser/models/basicblock.py: conv :61-101, ResBlock :211-227, upsample_convtranspose :413-419,
 * downsample_strideconv :437-446); the topology is KAIR's UNetRes: bias-free, channels 64-128-256-512, nb ResBlocks per
 * scale each way, strided / transposed 2x2 convolutions, additive skips, input = cat[x, sigma*1], 1 output channel.
 * params_host: the state_dict's tensors concatenated in state_dict order (m_head.weight, m_down1.0.res.0.weight, ...,
 * m_tail.weight; key list: tfpnp_amd/synth.py::drunet_param_specs), native PyTorch layouts.
 * A context holds ONE denoiser: after pnpx_drunet_load every denoiser prox -- pnpx_unet_denoise and the denoiser call
 * inside every solver entry below -- runs the DRUNet (out = clamp(net(cat[x, sigma*1]), 0, 1), H and W multiples of 8);
 * pnpx_unet_load switches back.  The *_backward / *_train entries work too: a DRUNet context has no activation ring
 * (tickets are 0), its VJP re-computes the forward keeping every ResBlock's ReLU output and back-propagates on the same
 * kernel family.  conv_mode 0 runs the DRUNet in fp32 arithmetic throughout (csrc/drunet_f32.hip; forward and, since r5,
 * the *_backward / *_train entries).  Range guard: the bias-free ReLU network is positively homogeneous, so the first two trips
 * make later passes (range_guard 1) or the very call (range_guard 2, repeated) run on inputs scaled by 2^-4, then 2^-8, with the
 * tail multiplying back (option "drunet_shift" reads / sets the exponent: 0, 4 or 8; acknowledging a trip keeps it); a trip at
 * 2^-8 latches the context to conv_mode 0 like a UNet context. */
size_t pnpx_drunet_num_params(int nb);
int pnpx_drunet_load(pnpx_ctx* ctx, const float* params_host, size_t n_params, int nb);
/* Vector-Jacobian product of pnpx_unet_denoise wrt x and sigma (weights are frozen): given grad_out [B,1,H,W] returns
 * grad_x [B,1,H,W] and grad_sigma [B].  This is what autograd computes through UNetDenoiser2D.forward when the
 * reference differentiates the solver for policy training (tfpnp/env/base.py:193-206, trainer/mddpg/trainer.py:171-192).
 * The forward pass is re-computed internally in exact fp32 (gradient checkpointing); nothing is kept between calls. */
int pnpx_unet_denoise_backward(pnpx_ctx* ctx, const float* x, const float* sigma, const float* grad_out,
                               float* grad_x, float* grad_sigma, int B, int H, int W, void* stream);
/* The same pair for autograd graphs that hold many denoiser calls (the reference differentiates T solver iterations,
 * PnPEnv.forward -> solver.forward, tfpnp/env/base.py:193-206): pnpx_unet_denoise_train = pnpx_unet_denoise that also
 * parks every activation of this forward in the context's training ring and names it with *ticket (host memory; 0 when
 * the ring is off or out of budget); pnpx_unet_denoise_backward_ticket skips the re-computation when the ring still
 * holds that ticket and re-computes otherwise (ticket 0, slot re-used by a later forward, other weights) -- identical
 * result either way.  Ring length = "train_cache_gb" budget / arena size (5 GiB at 48 x 256^2), at most 64 slots. */
int pnpx_unet_denoise_train(pnpx_ctx* ctx, const float* x, const float* sigma, float* out, int B, int H, int W,
                            unsigned long long* ticket, void* stream);
int pnpx_unet_denoise_backward_ticket(pnpx_ctx* ctx, const float* x, const float* sigma, const float* grad_out,
                                      float* grad_x, float* grad_sigma, int B, int H, int W, unsigned long long ticket,
                                      void* stream);
/* Per-layer timing of one denoise call with HIP events on `stream` (synchronises).  ms_out[i] for the
 * i-th kernel launch of the forward pass, flops_out[i] its algorithmic FLOPs (0 for non-conv launches),
 * names_out[i] a static string.  Returns the number of entries written (<= cap) in *n_out. */
int pnpx_unet_profile(pnpx_ctx* ctx, const float* x, const float* sigma, float* out, int B, int H, int W,
                      void* stream, int cap, float* ms_out, double* flops_out, const char** names_out,
                      int* n_out);

/* ---- policy actor (tfpnp/policy/network.py) ------------------------------------------------------- */
/* ResNetActorBase (network.py:129-147) in eval mode: ResNet-18 encoder (network.py:87-125; BasicBlock :33-58;
 * SynchronizedBatchNorm2d on running statistics, sync_batchnorm/batchnorm.py:63-68) -> adaptive_avg_pool2d(1) ->
 * fc_softmax (Linear(512,2)+Softmax) and fc_deterministic (Linear(512,n_det)+Sigmoid; spi_head=1: the
 * Linear(512,64)-ReLU-Linear(64,n_det)-Sigmoid head of ResNetActor_SPI, network.py:262-268).
 * `params_host`: fp32 values of the module's state_dict in this order (integer num_batches_tracked entries skipped):
 *   actor_encoder.conv1.weight, actor_encoder.bn1.{weight,bias,running_mean,running_var},
 *   for L in layer1..layer4:  L.0.{conv1.weight, bn1.*, conv2.weight, bn2.*, shortcut.0.weight, shortcut.1.*},
 *                             L.1.{conv1.weight, bn1.*, conv2.weight, bn2.*},
 *   fc_softmax.0.{weight,bias}, fc_deterministic.0.{weight,bias} [, fc_deterministic.2.{weight,bias}]
 * where bn.* = weight, bias, running_mean, running_var.  num_inputs = channels of the policy observation
 * (env.get_policy_ob, e.g. tasks/csmri/env.py:14-23 -> 9), n_det = action_bundle * num_actions. */
size_t pnpx_policy_num_params(int num_inputs, int n_det, int spi_head);
int pnpx_policy_load(pnpx_ctx* ctx, const float* params_host, size_t n_params, int num_inputs, int n_det,
                     int spi_head);
/* ob [B,num_inputs,H,W] (device, contiguous; H, W multiples of 32) -> probs [B,2] (softmax over {continue, stop}),
 * det [B,n_det] (sigmoid outputs, before the action-range mapping of network.py:163-175). */
int pnpx_policy_forward(pnpx_ctx* ctx, const float* ob, float* probs, float* det, int B, int H, int W,
                        void* stream);

/* ---- transforms (tfpnp/utils/transforms.py) ------------------------------------------------------ */
/* fft2 / ifft2 (transforms.py:68-103): centered (ifftshift -> FFT -> fftshift), orthonormal, over the
 * last two image dims of [n_img, H, W, 2].  centered=0 gives the plain torch.fft(x, 2, normalized=True)
 * used by cdp_forward/backward (transforms.py:300,318).  H, W in [1, 2048], any factorisation. */
int pnpx_fft2(pnpx_ctx* ctx, const float* in, float* out, int n_img, int H, int W, int inverse,
              int centered, void* stream);
/* cdp_forward (transforms.py:282-301): x [B,1,H,W,2], mask [B,S,H,W,2] -> out [B,S,H,W,2]. */
int pnpx_cdp_forward(pnpx_ctx* ctx, const float* x, const float* mask, float* out, int B, int S, int H,
                     int W, void* stream);
/* cdp_backward (transforms.py:304-320): y [B,S,H,W,2], mask [B,S,H,W,2] -> out [B,1,H,W,2] (mean over S). */
int pnpx_cdp_backward(pnpx_ctx* ctx, const float* y, const float* mask, float* out, int B, int S, int H,
                      int W, void* stream);
/* spi_inverse (transforms.py:404-439): ztilde, K1 [B,1,H,W]; K, mu [B] -> out [B,1,H,W]. */
int pnpx_spi_inverse(pnpx_ctx* ctx, const float* ztilde, const float* K1, const float* K, const float* mu,
                     float* out, int B, int H, int W, void* stream);
/* torch_psnr (tfpnp/env/base.py:237-242): output, gt [B,1,H,W] -> psnr [B]. */
int pnpx_psnr(pnpx_ctx* ctx, const float* output, const float* gt, float* psnr, int B, int n_per_item,
              void* stream);

/* ---- episode orchestration: the caller contract of PnPEnv.step (tfpnp/env/base.py:157-191) -------- */
/* Live-row gather `x[self.idx_left, ...]` (base.py:162-166) of n_tensors state tensors in ONE launch:
 * dst_t[r] = src_t[idx[r]] for r < n_rows, rows of row_bytes[t] bytes.  src_host / dst_host / row_bytes_host are HOST
 * arrays (of device pointers / sizes); idx is a device int64 array. */
int pnpx_rows_gather(pnpx_ctx* ctx, int n_tensors, const void* const* src_host, void* const* dst_host,
                     const size_t* row_bytes_host, const int64_t* idx, int n_rows, void* stream);
/* Write-back `state[...][self.idx_left, ...] = value` (base.py:171-172): dst_t[idx[r]] = src_t[r]. */
int pnpx_rows_scatter(pnpx_ctx* ctx, int n_tensors, const void* const* src_host, void* const* dst_host,
                      const size_t* row_bytes_host, const int64_t* idx, int n_rows, void* stream);
/* `self.idx_left = self.idx_left[idx_stop == 0]; all_done = len(self.idx_left) == 0` (base.py:180-182) as a
 * device-side stream compaction: idx_out[0..n_live) = the idx_left[i] with idx_stop[i] == 0 (int64, in order).
 * *n_live_host receives the count: this call SYNCHRONISES `stream` -- it is the one host read of an env step
 * (`all_done` is a Python bool in the reference's contract). */
int pnpx_live_compact(pnpx_ctx* ctx, const int64_t* idx_left, const int64_t* idx_stop, int n, int64_t* idx_out,
                      int* n_live_host, void* stream);
/* Policy observation (get_policy_ob: tasks/csmri/env.py:14-23, tasks/pr/env.py:14-21, tasks/ct/env.py:13-20,
 * tasks/spi/env.py:12-19): channel-concatenation of n_entries (<= 12) state tensors, each viewed as
 *   kind 0: fp32 [B,c,H,W] as is;  1: complex2real of [B,c,H,W,2] (transforms.py:16-17);
 *   kind 2: complex2channel of [B,c,H,W,2] (re, im of channel j -> channels 2j, 2j+1; transforms.py:20-26);
 *   kind 3: uint8 / bool [B,c,H,W] cast to float
 * into out [n_rows, sum(channels), H, W].  Row r reads source row idx[r] (idx == NULL: row r). */
int pnpx_policy_ob_pack(pnpx_ctx* ctx, int n_entries, const void* const* src_host, const int* kind_host,
                        const int* channels_host, const int64_t* idx, int n_rows, int H, int W, float* out,
                        void* stream);

/* ---- solver loops: T inner iterations per call ---------------------------------------------------- */
/* Hyper-parameter arrays are [B, param_stride] row-major (the policy's [B, action_pack] tensors,
 * tfpnp/policy/network.py:164-175); iteration i reads column i.  vars_in is never modified; vars_out
 * may not alias it. */

/* ADMMSolver_CSMRI.forward (tasks/csmri/solver.py:29-57).
 * vars [B,3,H,W,2] = cat(x,z,u); y0 [B,1,H,W,2]; mask [B,1,H,W] bytes. */
int pnpx_csmri_admm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                    const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride, int B,
                    int H, int W, int T, void* stream);
/* Training path of the same call: what autograd records and replays when the reference differentiates
 * ADMMSolver_CSMRI.forward for policy training (PnPEnv.forward, tfpnp/env/base.py:193-206, called from
 * trainer/mddpg/trainer.py:171-192).
 * pnpx_csmri_admm_train = pnpx_csmri_admm (bit-identical vars_out) that also fills `saved`, 3*T*B*H*W floats: the
 *   denoiser input of every iteration [T][B][H*W] followed by the k-space image before the blend [T][B][H*W][2].
 * pnpx_csmri_admm_backward: given grad_vars_out [B,3,H,W,2] returns grad_vars_in [B,3,H,W,2] (may not alias),
 *   grad_sigma_d [T][B] and grad_mu [T][B] (iteration-major, dense); `work` is 3*B*H*W floats of caller workspace.
 *   The denoiser weights are constants (frozen, as in the reference); y0 / mask get no gradient.  Per iteration: one
 *   masked-FFT adjoint (same three fused passes as the forward), one deterministic per-item reduction (d/d mu) and the
 *   denoiser VJP (pnpx_unet_denoise_backward's kernels).
 * Activation ring: every iteration's denoiser forward is a pnpx_unet_denoise_train call (above), so within the
 *   "train_cache_gb" budget (default 96 GiB; ~5 GiB per 48 x 256^2 iteration -- the 288 GB of an MI355X are what make
 *   this affordable) the backward pass does not re-compute the denoiser forwards (8 instead of 14.5 ms per iteration at
 *   48 x 256^2).  *ticket (host memory, written before the call returns) is iteration 0's ticket, iteration i holds
 *   ticket + i; slots that a later training forward has re-used are answered by re-computation from `saved` -- same
 *   gradients, never stale ones. */
int pnpx_csmri_admm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                          const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride, int B,
                          int H, int W, int T, float* saved, unsigned long long* ticket, void* stream);
int pnpx_csmri_admm_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                             const float* mu, int param_stride, const float* saved, const float* grad_vars_out,
                             float* grad_vars_in, float* grad_sigma_d, float* grad_mu, float* work, int B, int H,
                             int W, int T, unsigned long long ticket, void* stream);
/* HQSSolver_CSMRI.forward (tasks/csmri/solver.py:64-89).  vars [B,2,H,W,2] = cat(x,z). */
int pnpx_csmri_hqs(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                   const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride, int B,
                   int H, int W, int T, void* stream);
/* Training path of HQSSolver_CSMRI.forward, same contract as pnpx_csmri_admm_train / _backward above with vars
 * [B,2,H,W,2]: `saved` = 3*T*B*H*W floats (denoiser inputs, then the k-space images before the blend), activations parked
 * under *ticket + i; the backward returns grad_vars_in [B,2,H,W,2], grad_sigma_d and grad_mu [T][B], work = 3*B*H*W floats. */
int pnpx_csmri_hqs_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                         const uint8_t* mask, const float* sigma_d, const float* mu, int param_stride, int B,
                         int H, int W, int T, float* saved, unsigned long long* ticket, void* stream);
int pnpx_csmri_hqs_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                            const float* mu, int param_stride, const float* saved, const float* grad_vars_out,
                            float* grad_vars_in, float* grad_sigma_d, float* grad_mu, float* work, int B, int H,
                            int W, int T, unsigned long long ticket, void* stream);
/* PGSolver_CSMRI.forward (tasks/csmri/solver.py:96-120).  vars [B,1,H,W,2] = x. */
int pnpx_csmri_pg(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                  const uint8_t* mask, const float* sigma_d, const float* tau, int param_stride, int B,
                  int H, int W, int T, void* stream);
/* Training path of PGSolver_CSMRI.forward (same contract; vars [B,1,H,W,2]): `saved` = 2*T*B*H*W floats (the denoiser inputs,
 * then the real parts of the masked-residual images ifft2c(mask * (fft2c(x_i) - y0))); grads wrt (x, sigma_d, tau). */
int pnpx_csmri_pg_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                        const uint8_t* mask, const float* sigma_d, const float* tau, int param_stride, int B,
                        int H, int W, int T, float* saved, unsigned long long* ticket, void* stream);
int pnpx_csmri_pg_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                           const float* tau, int param_stride, const float* saved, const float* grad_vars_out,
                           float* grad_vars_in, float* grad_sigma_d, float* grad_tau, float* work, int B, int H,
                           int W, int T, unsigned long long ticket, void* stream);
/* APGSolver_CSMRI.forward (tasks/csmri/solver.py:127-165).  vars [B,2,H,W,2] = cat(x,s). */
int pnpx_csmri_apg(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                   const uint8_t* mask, const float* sigma_d, const float* tau, const float* beta,
                   int param_stride, int B, int H, int W, int T, void* stream);
/* Training path of APGSolver_CSMRI.forward (same contract; vars [B,2,H,W,2]): `saved` = 4*T*B*H*W floats (denoiser inputs,
 * real parts of the masked-residual images, x' - x_prev as complex); grads wrt (cat(x, s), sigma_d, tau, beta); work =
 * 4*B*H*W floats. */
int pnpx_csmri_apg_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                         const uint8_t* mask, const float* sigma_d, const float* tau, const float* beta,
                         int param_stride, int B, int H, int W, int T, float* saved, unsigned long long* ticket,
                         void* stream);
int pnpx_csmri_apg_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                            const float* tau, const float* beta, int param_stride, const float* saved,
                            const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d, float* grad_tau,
                            float* grad_beta, float* work, int B, int H, int W, int T, unsigned long long ticket,
                            void* stream);
/* REDADMMSolver_CSMRI.forward (tasks/csmri/solver.py:172-204).  vars [B,3,H,W,2]. */
int pnpx_csmri_redadmm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                       const uint8_t* mask, const float* sigma_d, const float* mu, const float* lamda,
                       int param_stride, int B, int H, int W, int T, void* stream);
/* Training path of REDADMMSolver_CSMRI.forward (same contract; vars [B,3,H,W,2]): `saved` = 7*T*B*H*W floats (denoiser
 * inputs Re x, the k-space images before the blend, r2c(xh) - x' and (z - u) - x' as complex); grads wrt (cat(x, z, u),
 * sigma_d, mu, lamda); work = 4*B*H*W floats. */
int pnpx_csmri_redadmm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                             const uint8_t* mask, const float* sigma_d, const float* mu, const float* lamda,
                             int param_stride, int B, int H, int W, int T, float* saved, unsigned long long* ticket,
                             void* stream);
int pnpx_csmri_redadmm_backward(pnpx_ctx* ctx, const float* y0, const uint8_t* mask, const float* sigma_d,
                                const float* mu, const float* lamda, int param_stride, const float* saved,
                                const float* grad_vars_out, float* grad_vars_in, float* grad_sigma_d, float* grad_mu,
                                float* grad_lamda, float* work, int B, int H, int W, int T, unsigned long long ticket,
                                void* stream);
/* IADMMSolver_PR.forward (tasks/pr/solver.py:37-76).
 * vars [B,3,H,W,2]; y0 [B,S,H,W]; mask [B,S,H,W,2]. */
int pnpx_pr_iadmm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0,
                  const float* mask, const float* sigma_d, const float* mu, const float* tau,
                  int param_stride, int B, int S, int H, int W, int T, void* stream);
/* Training path of IADMMSolver_PR.forward (same contract): `saved` = (2*S + 5)*T*B*H*W floats (denoiser inputs, the S
 * k-space images before the residual, g + mu (z - (x + u)) and z - (x + u) as complex); grads wrt (cat(x, z, u), sigma_d,
 * mu, tau), each hyper-parameter gradient [T][B]; work = 4*B*H*W floats.  ticket as in pnpx_csmri_admm_train. */
int pnpx_pr_iadmm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, const float* mask,
                        const float* sigma_d, const float* mu, const float* tau, int param_stride, int B, int S,
                        int H, int W, int T, float* saved, unsigned long long* ticket, void* stream);
int pnpx_pr_iadmm_backward(pnpx_ctx* ctx, const float* y0, const float* mask, const float* sigma_d, const float* mu,
                           const float* tau, int param_stride, const float* saved, const float* grad_vars_out,
                           float* grad_vars_in, float* grad_sigma_d, float* grad_mu, float* grad_tau, float* work,
                           int B, int S, int H, int W, int T, unsigned long long ticket, void* stream);
/* ADMMSolver_SPI.forward (tasks/spi/solver.py:17-52).
 * vars [B,3,H,W] real; x0 [B,1,H,W]; Kmap [B,1,H,W] (K/10 broadcast, only [b,0,0,0] is read). */
int pnpx_spi_admm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* x0,
                  const float* Kmap, const float* sigma_d, const float* mu, int param_stride, int B, int H,
                  int W, int T, void* stream);
/* Training path of ADMMSolver_SPI.forward (same contract): `saved` = 2*T*B*H*W floats (x + u and the denoiser inputs); grads
 * wrt (cat(x, z, u), sigma_d, mu), hyper-parameter gradients [T][B]; work = 3*B*H*W floats.  As in the reference the bisection
 * branch of spi_inverse carries no gradient (transforms.py:404-439), only its K1 == 0 branch does. */
int pnpx_spi_admm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* x0, const float* Kmap,
                        const float* sigma_d, const float* mu, int param_stride, int B, int H, int W, int T,
                        float* saved, unsigned long long* ticket, void* stream);
int pnpx_spi_admm_backward(pnpx_ctx* ctx, const float* x0, const float* Kmap, const float* sigma_d, const float* mu,
                           int param_stride, const float* saved, const float* grad_vars_out, float* grad_vars_in,
                           float* grad_sigma_d, float* grad_mu, float* work, int B, int H, int W, int T,
                           unsigned long long ticket, void* stream);

/* ---- CT: Radon pair standing in for torch_radon (tfpnp/utils/transforms.py:465-508) -------------- */
/* Parallel beam, angles = linspace(0, 179/180*pi, n_view), det = ceil(sqrt(2)*R), unit spacing
 * (transforms.py:487-491).  img [B,1,R,R] <-> sino [B,1,n_view,det].  Discretisation: DESIGN.md. */
int pnpx_radon_det_count(int R);
int pnpx_radon_forward(pnpx_ctx* ctx, const float* img, float* sino, int B, int R, int n_view, void* stream);
int pnpx_radon_backprojection(pnpx_ctx* ctx, const float* sino, float* img, int B, int R, int n_view,
                              void* stream);
/* IADMMSolver_CT.forward (tasks/ct/solver.py:17-53); opnorm = Radon_norm.opnorm (transforms.py:470-474).
 * vars [B,3,R,R] real; y0 [B,1,n_view,det]. */
int pnpx_ct_iadmm(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view,
                  float opnorm, const float* sigma_d, const float* mu, const float* tau, int param_stride,
                  int B, int R, int T, void* stream);
/* Training path of IADMMSolver_CT.forward: `saved` = 3*T*B*R*R floats; grads wrt (cat(x, z, u), sigma_d, mu, tau);
 * work = 6*B*R*R + 2*n_view floats. */
int pnpx_ct_iadmm_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view,
                        float opnorm, const float* sigma_d, const float* mu, const float* tau, int param_stride, int B,
                        int R, int T, float* saved, unsigned long long* ticket, void* stream);
int pnpx_ct_iadmm_backward(pnpx_ctx* ctx, int n_view, float opnorm, const float* sigma_d, const float* mu,
                           const float* tau, int param_stride, const float* saved, const float* grad_vars_out,
                           float* grad_vars_in, float* grad_sigma_d, float* grad_mu, float* grad_tau, float* work,
                           int B, int R, int T, unsigned long long ticket, void* stream);
/* PGSolver_CT.forward (tasks/ct/solver.py:61-87).  vars [B,1,R,R]. */
int pnpx_ct_pg(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view,
               float opnorm, const float* sigma_d, const float* tau, int param_stride, int B, int R, int T,
               void* stream);
/* Training path of PGSolver_CT.forward: `saved` = 2*T*B*R*R floats; grads wrt (x, sigma_d, tau); work = 3*B*R*R + 1 + 2*n_view
 * floats. */
int pnpx_ct_pg_train(pnpx_ctx* ctx, const float* vars_in, float* vars_out, const float* y0, int n_view, float opnorm,
                     const float* sigma_d, const float* tau, int param_stride, int B, int R, int T, float* saved,
                     unsigned long long* ticket, void* stream);
int pnpx_ct_pg_backward(pnpx_ctx* ctx, int n_view, float opnorm, const float* sigma_d, const float* tau,
                        int param_stride, const float* saved, const float* grad_vars_out, float* grad_vars_in,
                        float* grad_sigma_d, float* grad_tau, float* work, int B, int R, int T,
                        unsigned long long ticket, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNPX_H */
