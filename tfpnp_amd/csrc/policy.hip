// Policy actor forward (eval mode) -- SURVEY section 8(f) rank 2.
//
// Replaces ResNetActorBase.forward up to the head activations (tfpnp/policy/network.py:129-147):
//     x = ResNetEncoder(18)(state)            network.py:87-125  (3x3 stride-2 stem, 4 stages of 2 BasicBlocks, each
//                                             stage entered with stride 2; BasicBlock network.py:33-58)
//     x = adaptive_avg_pool2d(x, 1).view(B, 512)
//     probs = softmax(fc_softmax(x));  det = sigmoid(fc_deterministic(x))      (SPI head: 512-64-ReLU-n, :262-268)
// with SynchronizedBatchNorm2d in eval mode (= F.batch_norm on running statistics, sync_batchnorm/batchnorm.py:63-68;
// both the rollout, trainer.py:216-221, and the evaluator, evaluator.py:23, run the actor that way).  Sampling /
// arg-max of idx_stop, log-probabilities and the action range mapping stay in the host mirror (O(B) scalars).
//
// MI355X design: every convolution is one launch of the fp32 MFMA kernel in policy_conv.hip:
//   * BatchNorm folded into weights and bias on the host; ReLU / residual add in the epilogue.
//   * stride-2 3x3 convolutions read a SPACE-TO-DEPTH copy of their input ([4*C][H/2][W/2], written directly by the
//     producer's epilogue): on that grid the convolution is stride 1, and of the 36 (phase, tap) pairs only 9 are
//     non-zero -- a per-(cout tile, K-chunk) tap mask skips the other MFMAs, so no arithmetic is wasted on stride.
//   * the 1x1 stride-2 shortcut is the centre tap of phase (0,0): it rides in the same launch as extra cout tiles
//     (all other chunks masked out) and is written, without ReLU, to a second output.
#include <cmath>
#include <cstring>

#include "common.h"
#include "conv_hs.h"
#include "hs_rec.h"
#include "hs_relayout.h"
#include "policy_conv.h"

namespace pnpx {
namespace {

constexpr float BN_EPS = 1e-5f;
inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// observation [B][C][H][W] -> space-to-depth padded planar [B][4*Cp][H/2+2][W/2+8] (channels >= C stay zero)
__global__ __launch_bounds__(256) void pack_ob_s2d_kernel(const float* __restrict__ ob, float* __restrict__ out, int C,
                                                          int Cp, int H, int W, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  size_t t = i / W;
  const int y = (int)(t % H);
  t /= H;
  const int c = (int)(t % C);
  const size_t b = t / C;
  const int Hp2 = padded_h(H >> 1), Wp2 = pol_wp(W >> 1);
  const int ph = (y & 1) * 2 + (x & 1);
  out[((b * 4 * Cp + (size_t)ph * Cp + c) * Hp2 + (y >> 1) + 1) * Wp2 + (x >> 1) + POL_PADL] = ob[i];
}

// observation [B][C][H][W] fp32 -> half-split HS8 space-to-depth tensor [B][4*Cp/8][H/2+2][W/2+2] (phase-major channel
// groups; channels >= C are zero): the input of the stem on the sparse-tap half-split instance
__global__ __launch_bounds__(256) void pack_ob_s2d_hs_kernel(const float* __restrict__ ob, HsRec* __restrict__ out, int C,
                                                             int Cp, int H, int W, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int W2 = W >> 1, H2 = H >> 1, Gp = Cp >> 3;
  const int x2 = (int)(i % W2);
  size_t t = i / W2;
  const int y2 = (int)(t % H2);
  t /= H2;
  const int g = (int)(t % Gp);
  t /= Gp;
  const int ph = (int)(t % 4);
  const size_t b = t / 4;
  const int y = 2 * y2 + (ph >> 1), x = 2 * x2 + (ph & 1);
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = g * 8 + k;
    v[k] = c < C ? ob[((b * C + c) * H + y) * (size_t)W + x] * HS_ASCALE : 0.f;
  }
  out[((b * 4 * Gp + (size_t)ph * Gp + g) * (H2 + 2) + (y2 + 1)) * (size_t)(W2 + 2) + (x2 + 1)] = hs_pack(v);
}

// global average pool over [B][512][h][w] (padded planar) + the two heads.  One workgroup per observation.
// half-split HS8 [B][C/8][h+2][w+2] -> space-to-depth fp32 planar [B][4*C][h/2+2][w/2+8] (input of a stride-2 conv)
__global__ __launch_bounds__(256) void hs8_to_s2d_kernel(const HsRec* __restrict__ src, float* __restrict__ dst, int C,
                                                         int h, int w, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % w);
  size_t t = i / w;
  const int y = (int)(t % h);
  t /= h;
  const int g = (int)(t % (C >> 3));
  const size_t b = t / (C >> 3);
  float v[8];
  hs_unpack(src[((b * (C >> 3) + g) * (h + 2) + (y + 1)) * (size_t)(w + 2) + (x + 1)], v);
  const int Hp2 = padded_h(h >> 1), Wp2 = pol_wp(w >> 1);
  const int ph = (y & 1) * 2 + (x & 1);
  float* o = dst + ((b * 4 * C + (size_t)ph * C + g * 8) * Hp2 + (y >> 1) + 1) * Wp2 + (x >> 1) + POL_PADL;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[(size_t)k * Hp2 * Wp2] = v[k] * (1.f / HS_ASCALE);
}

// fp32 planar [B][C][h+2][pol_wp(w)] -> half-split HS8 [B][C/8][h+2][w+2] (the stem's space-to-depth output for the stage-0
// entry on the half-split instances)
__global__ __launch_bounds__(256) void planar_to_hs8_kernel(const float* __restrict__ src, HsRec* __restrict__ dst, int C,
                                                            int h, int w, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % w);
  size_t t = i / w;
  const int y = (int)(t % h);
  t /= h;
  const int g = (int)(t % (C >> 3));
  const size_t b = t / (C >> 3);
  const int Hp = padded_h(h), Wp = pol_wp(w);
  const float* p = src + ((b * C + (size_t)g * 8) * Hp + (y + 1)) * Wp + x + POL_PADL;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = p[(size_t)k * Hp * Wp] * HS_ASCALE;
  dst[((b * (C >> 3) + g) * (h + 2) + (y + 1)) * (size_t)(w + 2) + (x + 1)] = hs_pack(v);
}

__global__ __launch_bounds__(256) void pool_heads_kernel(const HsRec* __restrict__ feat, int h, int w,
                                                         const float* __restrict__ sm_w, const float* __restrict__ sm_b,
                                                         const float* __restrict__ d_w, const float* __restrict__ d_b,
                                                         const float* __restrict__ d2_w, const float* __restrict__ d2_b,
                                                         int n_det, int spi, float* __restrict__ probs,
                                                         float* __restrict__ det) {
  __shared__ float f[512];
  __shared__ float hid[64];
  __shared__ float logit[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float inv = 1.f / ((float)(h * w) * HS_ASCALE);
  for (int c = tid; c < 512; c += 256) {
    const HsRec* p = feat + ((size_t)b * 64 + (c >> 3)) * (h + 2) * (w + 2);
    float s = 0.f;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const HsRec& r = p[(y + 1) * (w + 2) + x + 1];
        s += (float)r.hi[c & 7] + (float)r.lo[c & 7];
      }
    f[c] = s * inv;
  }
  __syncthreads();
  auto dot512 = [&](const float* wrow) {
    float s = 0.f;
    for (int k = 0; k < 512; ++k) s = fmaf(wrow[k], f[k], s);
    return s;
  };
  if (tid < 2) logit[tid] = dot512(sm_w + tid * 512) + sm_b[tid];
  if (spi) {
    if (tid >= 64 && tid < 128) hid[tid - 64] = fmaxf(dot512(d_w + (tid - 64) * 512) + d_b[tid - 64], 0.f);
  } else if (tid >= 64 && tid < 64 + n_det) {
    const int j = tid - 64;
    det[(size_t)b * n_det + j] = 1.f / (1.f + expf(-(dot512(d_w + j * 512) + d_b[j])));
  }
  __syncthreads();
  if (tid == 0) {
    const float m = fmaxf(logit[0], logit[1]);
    const float e0 = expf(logit[0] - m), e1 = expf(logit[1] - m);
    probs[b * 2 + 0] = e0 / (e0 + e1);
    probs[b * 2 + 1] = e1 / (e0 + e1);
  }
  if (spi && tid < n_det) {
    float s = d2_b[tid];
    for (int k = 0; k < 64; ++k) s = fmaf(d2_w[tid * 64 + k], hid[k], s);
    det[(size_t)b * n_det + tid] = 1.f / (1.f + expf(-s));
  }
}

// ------------------------------------------------------------------------------------------- parameter layout
struct BnView {
  const float *g, *b, *m, *v;
};
struct Reader {
  const float* p;
  const float* take(size_t n) {
    const float* r = p;
    p += n;
    return r;
  }
  BnView bn(int c) {
    BnView r;
    r.g = take(c);
    r.b = take(c);
    r.m = take(c);
    r.v = take(c);
    return r;
  }
};
inline int stage_planes(int n) { return 64 << n; }   // n = 0..3

// Dense "effective" weights of one launch: E[cout][K][9] (+ bias[cout]), then packed + masked.
struct Eff {
  int cout, K;
  std::vector<float> w, bias;
  Eff(int cout_, int K_) : cout(cout_), K(K_), w((size_t)cout_ * K_ * 9, 0.f), bias(cout_, 0.f) {}
  float& at(int co, int k, int tap) { return w[((size_t)co * K + k) * 9 + tap]; }
};
inline void bn_fold(const BnView& bn, int c, float* scale, float* shift) {
  for (int i = 0; i < c; ++i) {
    scale[i] = bn.g[i] / std::sqrt(bn.v[i] + BN_EPS);
    shift[i] = bn.b[i] - bn.m[i] * scale[i];
  }
}
// 3x3 stride-1 conv + BN -> rows [row0, row0 + cout) of E
void put_conv_s1(Eff& E, int row0, const float* w, const BnView& bn, int cout, int cin) {
  std::vector<float> sc(cout), sh(cout);
  bn_fold(bn, cout, sc.data(), sh.data());
  for (int co = 0; co < cout; ++co) {
    E.bias[row0 + co] = sh[co];
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 9; ++t) E.at(row0 + co, ci, t) = w[((size_t)co * cin + ci) * 9 + t] * sc[co];
  }
}
// 3x3 stride-2 conv + BN over a space-to-depth input with Cp channels per phase
void put_conv_s2(Eff& E, int row0, const float* w, const BnView& bn, int cout, int cin, int Cp) {
  std::vector<float> sc(cout), sh(cout);
  bn_fold(bn, cout, sc.data(), sh.data());
  for (int co = 0; co < cout; ++co) {
    E.bias[row0 + co] = sh[co];
    for (int ci = 0; ci < cin; ++ci)
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) {
          // input row 2*yo + (dy - 1): offset 0 -> phase 0 / same half-res row (tap row 1); offset -1 -> phase 1 /
          // previous row (tap row 0); offset +1 -> phase 1 / same row (tap row 1).  Likewise in x.
          const int py = (dy == 1) ? 0 : 1, ty = (dy == 0) ? 0 : 1;
          const int px = (dx == 1) ? 0 : 1, tx = (dx == 0) ? 0 : 1;
          E.at(row0 + co, (py * 2 + px) * Cp + ci, ty * 3 + tx) = w[((size_t)co * cin + ci) * 9 + dy * 3 + dx] * sc[co];
        }
  }
}
// 1x1 stride-2 conv + BN = centre tap of phase (0,0)
void put_shortcut(Eff& E, int row0, const float* w, const BnView& bn, int cout, int cin) {
  std::vector<float> sc(cout), sh(cout);
  bn_fold(bn, cout, sc.data(), sh.data());
  for (int co = 0; co < cout; ++co) {
    E.bias[row0 + co] = sh[co];
    for (int ci = 0; ci < cin; ++ci) E.at(row0 + co, ci, 4) = w[(size_t)co * cin + ci] * sc[co];
  }
}

struct HostBlob {
  std::vector<float> f;
  void align() { f.resize((f.size() + 255) & ~(size_t)255, 0.f); }
  size_t add(const float* p, size_t n) {
    align();
    const size_t off = f.size();
    f.insert(f.end(), p, p + n);
    return off;
  }
};
struct ConvOff {
  size_t w, bias, steps, nsteps;
};
// Pack one launch: per cout tile the list of K-chunks that carry any weight (PolStep), and only their present tap
// slices [8 channels][64 couts], in list order.  Presence is decided on the values (an all-zero slice contributes
// exactly nothing).
ConvOff pack_eff(HostBlob& H, Eff& E) {
  const int nct = E.cout / 64, nch = E.K / 8;
  ConvOff o;
  H.align();
  o.w = H.f.size();
  std::vector<PolStep> steps((size_t)nct * nch, PolStep{0, 0, 0});
  std::vector<int> nsteps(nct, 0);
  std::vector<float> slice(512);
  unsigned int nslices = 0;
  for (int ct = 0; ct < nct; ++ct)
    for (int ch = 0; ch < nch; ++ch) {
      unsigned short mask = 0;
      const unsigned int first = nslices;
      for (int tap = 0; tap < 9; ++tap) {
        bool any = false;
        for (int c = 0; c < 8; ++c)
          for (int m = 0; m < 64; ++m) {
            const float v = E.at(ct * 64 + m, ch * 8 + c, tap);
            any |= (v != 0.f);
            slice[c * 64 + m] = v;
          }
        if (!any) continue;
        mask |= (unsigned short)(1u << tap);
        H.f.insert(H.f.end(), slice.begin(), slice.end());
        ++nslices;
      }
      if (mask) steps[(size_t)ct * nch + nsteps[ct]++] = PolStep{mask, (unsigned short)ch, first};
    }
  H.f.resize(H.f.size() + 1024, 0.f);   // the 16-byte DMA of the last slice may not over-read, but keep a guard
  o.bias = H.add(E.bias.data(), E.bias.size());
  static_assert(sizeof(PolStep) == 8, "PolStep layout");
  H.align();
  o.steps = H.f.size();
  H.f.resize(H.f.size() + steps.size() * 2, 0.f);
  std::memcpy(H.f.data() + o.steps, steps.data(), steps.size() * sizeof(PolStep));
  H.align();
  o.nsteps = H.f.size();
  H.f.resize(H.f.size() + nsteps.size(), 0.f);
  std::memcpy(H.f.data() + o.nsteps, nsteps.data(), nsteps.size() * sizeof(int));
  return o;
}

// ------------------------------------------------------------------------------------------- activation plan
struct PolAct {
  size_t off = 0;   // floats
  int C = 0, H = 0, W = 0;
};
struct PolicyPlan {
  PolAct ob, stem;               // fp32, space-to-depth
  PolAct stem_hs;                // HS8 space-to-depth of the stem output (stage-0 entry on the half-split instances)
  PolAct ob_hs, stem_o;          // HS8: space-to-depth observation, stem output (64 channels, H/2 x W/2)
  PolAct t1[4], sc[4], o0[4], t2[4], o1[4];   // half-split HS8 (same 4 bytes per value)
  PolAct o1s[3];                 // fp32 space-to-depth copy of o1 for the next stage's stride-2 convolution
  size_t total = 0;              // floats for capB observations
};
PolicyPlan make_policy_plan(int capB, int cin_pad, int H, int W) {
  PolicyPlan P;
  size_t off = 0;
  auto add = [&](PolAct& d, int C, int h, int w) {
    d.off = off;
    d.C = C;
    d.H = h;
    d.W = w;
    off += (size_t)C * padded_h(h) * pol_wp(w) * capB;
    off = (off + 63) & ~(size_t)63;
  };
  auto add_hs = [&](PolAct& d, int C, int h, int w) {   // [C/8][h+2][w+2] records of 8 floats' worth
    d.off = off;
    d.C = C;
    d.H = h;
    d.W = w;
    off += (size_t)C * (h + 2) * (w + 2) * capB;
    off = (off + 63) & ~(size_t)63;
  };
  add(P.ob, 4 * cin_pad, H / 2, W / 2);
  add(P.stem, 4 * 64, H / 4, W / 4);
  add_hs(P.stem_hs, 4 * 64, H / 4, W / 4);
  add_hs(P.ob_hs, 4 * cin_pad, H / 2, W / 2);
  add_hs(P.stem_o, 64, H / 2, W / 2);
  for (int n = 0; n < 4; ++n) {
    const int p = stage_planes(n), h = H >> (n + 2), w = W >> (n + 2);
    add_hs(P.t1[n], p, h, w);
    add_hs(P.sc[n], p, h, w);
    add_hs(P.o0[n], p, h, w);
    add_hs(P.t2[n], p, h, w);
    add_hs(P.o1[n], p, h, w);
    if (n < 3) add(P.o1s[n], 4 * p, h / 2, w / 2);
  }
  P.total = off + (1u << 18);   // slack: overhanging tiles read past their tensor
  return P;
}

}  // namespace

size_t policy_num_params(int num_inputs, int n_det, int spi_head) {
  size_t n = (size_t)64 * num_inputs * 9 + 4 * 64;
  int in_planes = 64;
  for (int s = 0; s < 4; ++s) {
    const int p = stage_planes(s);
    n += (size_t)p * in_planes * 9 + 4 * p + (size_t)p * p * 9 + 4 * p + (size_t)p * in_planes + 4 * p;  // block 0
    n += 2 * ((size_t)p * p * 9 + 4 * p);                                                                 // block 1
    in_planes = p;
  }
  n += 2 * 512 + 2;
  n += spi_head ? (size_t)64 * 512 + 64 + (size_t)n_det * 64 + n_det : (size_t)n_det * 512 + n_det;
  return n;
}

void policy_free(pnpx_ctx* ctx) {
  PolicyNet& N = ctx->policy;
  if (N.weights.p) (void)hipFree(N.weights.p);
  if (N.arena.p) (void)hipFree(N.arena.p);
  N = PolicyNet();
}

int policy_load(pnpx_ctx* ctx, const float* params, size_t n, int num_inputs, int n_det, int spi_head) {
  if (!params || num_inputs < 1 || num_inputs > 64 || n_det < 1 || n_det > 64 ||
      n != policy_num_params(num_inputs, n_det, spi_head)) {
    set_error("pnpx_policy_load: expected %zu parameters for (%d inputs, %d outputs, spi %d), got %zu",
              policy_num_params(num_inputs, n_det, spi_head), num_inputs, n_det, spi_head, n);
    return PNPX_ERR_ARG;
  }
  PNPX_HIP(hipDeviceSynchronize());
  policy_free(ctx);
  PolicyNet& N = ctx->policy;
  N.num_inputs = num_inputs;
  N.cin_pad = (num_inputs + 7) / 8 * 8;
  N.n_det = n_det;
  N.spi_head = spi_head;
  Reader R{params};
  HostBlob H;
  ConvOff off[17];
  int cins[17], couts[17], splits[17];
  int li = 0;
  auto skip_fp32 = [&]() {                  // slot kept for indexing; the stride-1 convolutions only exist half-split
    off[li] = ConvOff{0, 0, 0, 0};
    cins[li] = couts[li] = splits[li] = 0;
    ++li;
  };
  // half-split packing of the stride-1 convolutions (folded weights E.w are [cout][K][9] dense, E.bias the shift)
  size_t hs_w[12], hs_b[12];
  float hs_scale[12];
  int hs_c[12];
  int hi_ = 0;
  size_t stem_w = 0, stem_b = 0;
  float stem_scale = 1.f;
  int stem_K = 0;
  size_t s2_w[4][2] = {}, s2_b[4][2] = {};
  float s2_scale[4][2] = {};
  int s2_K[4][2] = {}, s2_c[4][2] = {};
  auto finish_hs = [&](Eff& E) {
    H.align();
    hs_w[hi_] = H.f.size();
    const size_t n16 = (size_t)E.cout * E.K * 9 * 2;
    H.f.resize(H.f.size() + (n16 + 1) / 2, 0.f);
    hs_scale[hi_] = pack_conv_weights_hs(E.w.data(), E.cout, E.K, 64, reinterpret_cast<uint16_t*>(H.f.data() + hs_w[hi_]));
    hs_b[hi_] = H.add(E.bias.data(), E.bias.size());
    hs_c[hi_] = E.cout;
    ++hi_;
  };
  auto finish = [&](Eff& E, int split) {
    off[li] = pack_eff(H, E);
    cins[li] = E.K;
    couts[li] = E.cout;
    splits[li] = split;
    ++li;
  };
  {  // stem: conv3x3(num_inputs, 64, stride 2) + bn1
    const float* w = R.take((size_t)64 * num_inputs * 9);
    const BnView bn = R.bn(64);
    Eff E(64, 4 * N.cin_pad);
    put_conv_s2(E, 0, w, bn, 64, num_inputs, N.cin_pad);
    finish(E, 64);
    // ... and as a 2x2-window sparse-tap half-split launch over the HS8 space-to-depth observation
    H.align();
    stem_w = H.f.size();
    const size_t n16 = (size_t)E.cout * E.K * 4 * 2;
    H.f.resize(H.f.size() + (n16 + 1) / 2, 0.f);
    stem_scale = pack_conv_weights_hs_taps(E.w.data(), E.cout, E.K, 64, 0x01B, reinterpret_cast<uint16_t*>(H.f.data() + stem_w));
    stem_b = H.add(E.bias.data(), E.bias.size());
    stem_K = E.K;
  }
  int in_planes = 64;
  for (int s = 0; s < 4; ++s) {
    const int p = stage_planes(s);
    // block 0 (stride 2): conv1, bn1, conv2, bn2, shortcut.0 (1x1), shortcut.1 (bn)   -- state_dict order
    const float* w1 = R.take((size_t)p * in_planes * 9);
    const BnView b1 = R.bn(p);
    const float* w2 = R.take((size_t)p * p * 9);
    const BnView b2 = R.bn(p);
    const float* ws = R.take((size_t)p * in_planes);
    const BnView bs = R.bn(p);
    {
      Eff E(2 * p, 4 * in_planes);
      put_conv_s2(E, 0, w1, b1, p, in_planes, in_planes);
      put_shortcut(E, p, ws, bs, p, in_planes);
      finish(E, p);
    }
    {   // the same two convolutions packed for the sparse-tap half-split instances
      Eff E1(p, 4 * in_planes), Es(p, in_planes);
      put_conv_s2(E1, 0, w1, b1, p, in_planes, in_planes);
      put_shortcut(Es, 0, ws, bs, p, in_planes);
      Eff* both[2] = {&E1, &Es};
      const int masks[2] = {0x01B, 0x010};
      for (int k = 0; k < 2; ++k) {
        Eff& E = *both[k];
        int nt = 0;
        for (int t = 0; t < 9; ++t) nt += (masks[k] >> t) & 1;
        H.align();
        s2_w[s][k] = H.f.size();
        const size_t n16 = (size_t)E.cout * E.K * nt * 2;
        H.f.resize(H.f.size() + (n16 + 1) / 2, 0.f);
        s2_scale[s][k] = pack_conv_weights_hs_taps(E.w.data(), E.cout, E.K, 64, masks[k],
                                                   reinterpret_cast<uint16_t*>(H.f.data() + s2_w[s][k]));
        s2_b[s][k] = H.add(E.bias.data(), E.bias.size());
        s2_K[s][k] = E.K;
        s2_c[s][k] = E.cout;
      }
    }
    {
      Eff E(p, p);
      put_conv_s1(E, 0, w2, b2, p, p);
      skip_fp32();
      finish_hs(E);
    }
    // block 1 (stride 1, identity shortcut)
    for (int j = 0; j < 2; ++j) {
      const float* w = R.take((size_t)p * p * 9);
      const BnView b = R.bn(p);
      Eff E(p, p);
      put_conv_s1(E, 0, w, b, p, p);
      skip_fp32();
      finish_hs(E);
    }
    in_planes = p;
  }
  const size_t o_smw = H.add(R.take(2 * 512), 2 * 512);
  const size_t o_smb = H.add(R.take(2), 2);
  size_t o_dw, o_db, o_d2w = 0, o_d2b = 0;
  if (spi_head) {
    o_dw = H.add(R.take((size_t)64 * 512), (size_t)64 * 512);
    o_db = H.add(R.take(64), 64);
    o_d2w = H.add(R.take((size_t)n_det * 64), (size_t)n_det * 64);
    o_d2b = H.add(R.take(n_det), n_det);
  } else {
    o_dw = H.add(R.take((size_t)n_det * 512), (size_t)n_det * 512);
    o_db = H.add(R.take(n_det), n_det);
  }
  H.f.resize(H.f.size() + 8192, 0.f);   // DMA over-read slack
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, H.f.size() * sizeof(float));
  if (e != hipSuccess) {
    set_error("policy weight allocation of %zu bytes failed: %s", H.f.size() * sizeof(float), hipGetErrorString(e));
    return PNPX_ERR_ALLOC;
  }
  N.weights.p = d;
  N.weights.bytes = H.f.size() * sizeof(float);
  PNPX_HIP(hipMemcpy(d, H.f.data(), N.weights.bytes, hipMemcpyHostToDevice));
  const float* base = static_cast<const float*>(d);
  for (int i = 0; i < 17; ++i) {
    N.conv[i].w = base + off[i].w;
    N.conv[i].bias = base + off[i].bias;
    N.conv[i].steps = reinterpret_cast<const PolStep*>(base + off[i].steps);
    N.conv[i].nsteps = reinterpret_cast<const int*>(base + off[i].nsteps);
    N.conv[i].cin = cins[i];
    N.conv[i].cout = couts[i];
    N.conv[i].split_c = splits[i];
  }
  for (int i = 0; i < 12; ++i) {
    N.conv_hs[i].cin = N.conv_hs[i].cout = N.conv_hs[i].cin_pad = hs_c[i];
    N.conv_hs[i].mt = 64;
    N.conv_hs[i].w = const_cast<char*>(reinterpret_cast<const char*>(base + hs_w[i]));
    N.conv_hs[i].inv_scale = 1.0f / (hs_scale[i] * HS_ASCALE);
    N.bias_hs[i] = base + hs_b[i];
  }
  for (int st = 0; st < 4; ++st)
    for (int k = 0; k < 2; ++k) {
      N.s2_hs[st][k].cin = N.s2_hs[st][k].cin_pad = s2_K[st][k];
      N.s2_hs[st][k].cout = s2_c[st][k];
      N.s2_hs[st][k].mt = 64;
      N.s2_hs[st][k].w = const_cast<char*>(reinterpret_cast<const char*>(base + s2_w[st][k]));
      N.s2_hs[st][k].inv_scale = 1.0f / (s2_scale[st][k] * HS_ASCALE);
      N.s2_bias[st][k] = base + s2_b[st][k];
    }
  N.stem_hs.cin = N.stem_hs.cin_pad = stem_K;
  N.stem_hs.cout = 64;
  N.stem_hs.mt = 64;
  N.stem_hs.w = const_cast<char*>(reinterpret_cast<const char*>(base + stem_w));
  N.stem_hs.inv_scale = 1.0f / (stem_scale * HS_ASCALE);
  N.stem_hs_bias = base + stem_b;
  N.fc_sm_w = base + o_smw;
  N.fc_sm_b = base + o_smb;
  N.fc_det_w = base + o_dw;
  N.fc_det_b = base + o_db;
  N.fc_det2_w = spi_head ? base + o_d2w : nullptr;
  N.fc_det2_b = spi_head ? base + o_d2b : nullptr;
  N.loaded = true;
  return PNPX_OK;
}

int policy_forward(pnpx_ctx* ctx, const float* ob, float* probs, float* det, int B, int H, int W, hipStream_t s) {
  PolicyNet& N = ctx->policy;
  if (!N.loaded) {
    set_error("policy forward called before pnpx_policy_load");
    return PNPX_ERR_NO_WEIGHTS;
  }
  if (B <= 0 || H < 32 || W < 32 || (H % 32) || (W % 32)) {
    set_error("policy forward: need B > 0 and H, W positive multiples of 32 (got %d x %d x %d)", B, H, W);
    return PNPX_ERR_SHAPE;
  }
  if (!(B <= N.capB && H == N.capH && W == N.capW)) {
    const bool same = (H == N.capH && W == N.capW);
    const int nb = same ? (B > N.capB ? B : N.capB) : B;
    const PolicyPlan P = make_policy_plan(nb, N.cin_pad, H, W);
    PNPX_HIP(hipDeviceSynchronize());
    if (N.arena.bytes < P.total * sizeof(float)) {
      if (N.arena.p) PNPX_HIP(hipFree(N.arena.p));
      N.arena = DeviceBuf();
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, P.total * sizeof(float));
      if (e != hipSuccess) {
        set_error("policy arena allocation of %zu bytes failed: %s", P.total * sizeof(float), hipGetErrorString(e));
        return PNPX_ERR_ALLOC;
      }
      N.arena.p = p;
      N.arena.bytes = P.total * sizeof(float);
    }
    PNPX_HIP(hipMemset(N.arena.p, 0, P.total * sizeof(float)));
    PNPX_HIP(hipDeviceSynchronize());
    N.capB = nb;
    N.capH = H;
    N.capW = W;
  }
  const PolicyPlan P0 = make_policy_plan(N.capB, N.cin_pad, H, W);
  float* A = static_cast<float*>(N.arena.p);
  const bool all_hs = ctx->opt_policy_s2_hs && (N.stem_hs.cin_pad % 16 == 0);   // every activation is an HS8 tensor (the default)

  // option "chains": n = exactly n chains (when B >= n); 0 = automatic, from the table of every batch size 1..48 at 256 x 256 with
  // one and two chains: two chains pay between the round boundaries of the 8 x 8 / 16 x 16 stages -- B = 9..15
  // (-3..-6 %), 17..24 (-2..-7 %), 33..48 (-8..-15 %) -- and cost up to 12 % elsewhere (B = 32).  q = batch in 256 x 256 images.
  int chains = 1;
  if (all_hs) {
    if (ctx->opt_chains != 0) {
      chains = launch_chains(ctx, B, H, W);
    } else {
      const long long q = (long long)B * H * W / (256 * 256);
      chains = ((q >= 9 && q <= 15) || (q >= 17 && q <= 24) || q >= 33) ? 2 : 1;
    }
    if (chains > B) chains = B;
  }
  // The forward over observations b0 .. b0 + B - 1 (ob / probs / det already point at the first of them) on stream s.  With every
  // activation an HS8 tensor [image][group][h + 2][w + 2], a slice of the batch is a contiguous piece of each: slices run as
  // independent launch chains on side streams like the denoisers' (unet.hip: launch_chains; bit-identical per image) -- the deep
  // 8 x 8 / 16 x 16 stages otherwise step up at every round boundary (B = 33: 1.34 ms against 1.02 at B = 32).
  auto run = [&](int b0, int B, const float* ob, float* probs, float* det, hipStream_t s) -> int {
  PolicyPlan P = P0;
  if (b0) {
    auto shift = [&](PolAct& d) { d.off += (size_t)b0 * d.C * (d.H + 2) * (d.W + 2); };   // HS8 tensors only (all_hs)
    shift(P.stem_hs);
    shift(P.ob_hs);
    shift(P.stem_o);
    for (int n = 0; n < 4; ++n) {
      shift(P.t1[n]);
      shift(P.sc[n]);
      shift(P.o0[n]);
      shift(P.t2[n]);
      shift(P.o1[n]);
      if (n < 3) shift(P.o1s[n]);
    }
  }
  auto ptr = [&](const PolAct& d) { return A + d.off; };

  auto hsc0 = [&](const PolAct& d) { return reinterpret_cast<char*>(A + d.off); };
  const bool stem_on_hs = ctx->opt_policy_s2_hs && (N.stem_hs.cin_pad % 16 == 0);
  if (stem_on_hs) {
    // stem on the sparse-tap half-split instance: HS8 space-to-depth observation -> 2x2-window convolution (+ folded BN,
    // ReLU) -> HS8 [64][H/2][W/2] -> space-to-depth for the stage-0 entry
    const size_t n = (size_t)B * 4 * (N.cin_pad / 8) * (H / 2) * (W / 2);
    hipLaunchKernelGGL(pack_ob_s2d_hs_kernel, g1(n), dim3(256), 0, s, ob, reinterpret_cast<HsRec*>(hsc0(P.ob_hs)),
                       N.num_inputs, N.cin_pad, H, W, n);
    PNPX_LAUNCH_CHECK();
    ConvLayerHs Lh;
    Lh.cin = N.stem_hs.cin;
    Lh.cout = 64;
    Lh.cin_pad = N.stem_hs.cin_pad;
    Lh.mt = 64;
    Lh.w = N.stem_hs.w;
    Lh.b = N.stem_hs_bias;
    Lh.inv_scale = N.stem_hs.inv_scale;
    ConvHsFuse f;
    f.slope = 0.f;
    f.taps = 0x01B;
    f.wreg = 0;
    f.share = chains;
    f.range_flag = ctx->opt_range_guard ? ctx->range_flag_dev : nullptr;
    PNPX_TRY(launch_conv_hs(Lh, hsc0(P.ob_hs), N.stem_hs.cin_pad / 8, nullptr, 0, hsc0(P.stem_o), B, H / 2, W / 2, f, s));
    const size_t n2 = (size_t)B * 4 * 8 * (H / 4) * (W / 4) * 2;
    hipLaunchKernelGGL(hs_s2d_kernel, g1(n2), dim3(256), 0, s, reinterpret_cast<const uint4*>(hsc0(P.stem_o)),
                       reinterpret_cast<uint4*>(hsc0(P.stem_hs)), 8, H / 2, W / 2, n2);
    PNPX_LAUNCH_CHECK();
  } else {
    const size_t n = (size_t)B * N.num_inputs * H * W;
    hipLaunchKernelGGL(pack_ob_s2d_kernel, g1(n), dim3(256), 0, s, ob, ptr(P.ob), N.num_inputs, N.cin_pad, H, W, n);
    PNPX_LAUNCH_CHECK();
    // stem (on the H/2 grid) -> space-to-depth for stage 1
    PNPX_TRY(launch_policy_conv(N.conv[0], ptr(P.ob), ptr(P.stem), nullptr, nullptr, true, B, H / 2, W / 2, s));
  }
  // residual stages.  The stride-2 entry (conv1 + 1x1 shortcut, one fp32 tap-sparse launch) writes its two outputs as
  // half-split HS8 tensors; the three stride-1 convolutions of the stage run on the f16x3 MFMA kernel (conv_hs.hip:
  // folded-BN bias, residual add and ReLU in its epilogue); the stage output is re-laid out space-to-depth in fp32 for
  // the next stage's stride-2 launch.
  auto hsc = [&](const PolAct& d) { return reinterpret_cast<char*>(A + d.off); };
  auto conv_hs = [&](int i, const PolAct& in, const PolAct& out, const PolAct* res, int h, int w) -> int {
    const ConvLayerHsDev& D = N.conv_hs[i];
    ConvLayerHs Lh;
    Lh.cin = D.cin;
    Lh.cout = D.cout;
    Lh.cin_pad = D.cin_pad;
    Lh.mt = D.mt;
    Lh.w = D.w;
    Lh.b = N.bias_hs[i];
    Lh.inv_scale = D.inv_scale;
    ConvHsFuse f;
    f.slope = 0.f;                          // ReLU
    f.share = chains;
    f.res = res ? hsc(*res) : nullptr;
    f.range_flag = ctx->opt_range_guard ? ctx->range_flag_dev : nullptr;
    return launch_conv_hs(Lh, hsc(in), in.C / 8, nullptr, 0, hsc(out), B, h, w, f, s);
  };
  const float* xin = ptr(P.stem);
  for (int st = 0; st < 4; ++st) {
    const int h = H >> (st + 2), w = W >> (st + 2);
    if (!ctx->opt_policy_s2_hs) {
      PNPX_TRY(launch_policy_conv(N.conv[1 + 4 * st], xin, ptr(P.t1[st]), ptr(P.sc[st]), nullptr, false, B, h, w, s, true));
    } else {
      const PolAct& s2in = st == 0 ? P.stem_hs : P.o1s[st - 1];
      if (st == 0 && !stem_on_hs) {
        const size_t n0 = (size_t)B * (P.stem.C / 8) * h * w;
        hipLaunchKernelGGL(planar_to_hs8_kernel, g1(n0), dim3(256), 0, s, ptr(P.stem), reinterpret_cast<HsRec*>(hsc(P.stem_hs)),
                           P.stem.C, h, w, n0);
        PNPX_LAUNCH_CHECK();
      }
      // stride-2 entry on the sparse-tap half-split instances: conv1 = 2x2-window convolution over the HS8 space-to-depth
      // input (taps 0x01B, ReLU), shortcut = 1x1 over its first Cin channels (tap 0x010, linear)
      for (int k = 0; k < 2; ++k) {
        const ConvLayerHsDev& D = N.s2_hs[st][k];
        ConvLayerHs Lh;
        Lh.cin = D.cin;
        Lh.cout = D.cout;
        Lh.cin_pad = D.cin_pad;
        Lh.mt = D.mt;
        Lh.w = D.w;
        Lh.b = N.s2_bias[st][k];
        Lh.inv_scale = D.inv_scale;
        ConvHsFuse f;
        f.slope = k == 0 ? 0.f : 1.f;
        f.taps = k == 0 ? 0x01B : 0x010;
        f.share = chains;
        f.in0_groups = s2in.C / 8;
        f.wreg = 0;
        f.range_flag = ctx->opt_range_guard ? ctx->range_flag_dev : nullptr;
        PNPX_TRY(launch_conv_hs(Lh, hsc(s2in), D.cin_pad / 8, nullptr, 0, hsc(k == 0 ? P.t1[st] : P.sc[st]), B, h, w, f, s));
      }
    }
    PNPX_TRY(conv_hs(3 * st + 0, P.t1[st], P.o0[st], &P.sc[st], h, w));
    PNPX_TRY(conv_hs(3 * st + 1, P.o0[st], P.t2[st], nullptr, h, w));
    PNPX_TRY(conv_hs(3 * st + 2, P.t2[st], P.o1[st], &P.o0[st], h, w));
    if (st < 3 && ctx->opt_policy_s2_hs) {   // HS8 -> HS8 space-to-depth (phase-major groups) for the next stage's entry
      const int G = P.o1[st].C / 8;
      const size_t n2 = (size_t)B * 4 * G * (h / 2) * (w / 2) * 2;
      hipLaunchKernelGGL(hs_s2d_kernel, g1(n2), dim3(256), 0, s, reinterpret_cast<const uint4*>(hsc(P.o1[st])),
                         reinterpret_cast<uint4*>(hsc(P.o1s[st])), G, h, w, n2);
      PNPX_LAUNCH_CHECK();
    } else if (st < 3) {
      const size_t n8 = (size_t)B * (P.o1[st].C / 8) * h * w;
      hipLaunchKernelGGL(hs8_to_s2d_kernel, g1(n8), dim3(256), 0, s, reinterpret_cast<const HsRec*>(hsc(P.o1[st])),
                         ptr(P.o1s[st]), P.o1[st].C, h, w, n8);
      PNPX_LAUNCH_CHECK();
      xin = ptr(P.o1s[st]);
    }
  }
  hipLaunchKernelGGL(pool_heads_kernel, dim3(B), dim3(256), 0, s, reinterpret_cast<const HsRec*>(hsc(P.o1[3])), H / 32, W / 32, N.fc_sm_w, N.fc_sm_b,
                     N.fc_det_w, N.fc_det_b, N.fc_det2_w, N.fc_det2_b, N.n_det, N.spi_head, probs, det);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
  };

  if (chains <= 1) return run(0, B, ob, probs, det, s);
  return fan_out_chains(ctx, chains, B, s, [&](int lo, int hi, hipStream_t st) -> int {
    return run(lo, hi - lo, ob + (size_t)lo * N.num_inputs * H * W, probs + (size_t)lo * 2, det + (size_t)lo * N.n_det, st);
  });
}

}  // namespace pnpx
