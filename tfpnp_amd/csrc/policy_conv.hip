// 3x3 convolution of the policy actor (ResNet-18 encoder, tfpnp/policy/network.py:33-58,87-125) as an fp32-MFMA
// implicit GEMM for gfx950.  Same GEMM view and LDS pipeline as conv3x3.hip (D[cout][pixel], weights = A operand,
// dword LDS-DMA gather of the input halo, 16-byte DMA of the weight slices, two LDS stages, DMA issue interleaved
// with the taps), specialised for what this network needs:
//   * TAP-SPARSE K-chunks.  A stride-2 convolution runs as a stride-1 convolution over the space-to-depth copy of
//     its input ([4*C][H/2][W/2]); there only 9 of the 36 (phase, tap) pairs carry weights.  The 1x1 stride-2
//     shortcut is the centre tap of phase (0,0) and rides in the same launch as extra cout tiles.  Every
//     (cout tile, chunk) has a tap mask; absent taps are neither stored, nor copied, nor multiplied, and chunks
//     without any tap are not visited (PolStep lists).
//   * BATCH-STACKED small images.  Deep stages are 4..16 pixels high: one image cannot fill a workgroup's pixel
//     tile, and one workgroup per (image, cout tile) re-reads the whole weight slice per image (measured: 400 us
//     per stage-4 launch, all of it weight traffic).  Because every plane carries its own zero border, the planes
//     of all B images stacked vertically form one tall zero-separated image on which the same convolution is exact;
//     the gather DMA maps virtual rows to (image, row) and the epilogue drops the border rows.
//   * epilogue: folded-BatchNorm bias (+ residual) + ReLU (linear for the shortcut tiles, second output tensor),
//     optional space-to-depth store for a following stride-2 convolution.
#include "policy_conv.h"

#include "conv_hs.h"

namespace pnpx {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds4(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 4, 0, 0);
}
__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

struct PolArgs {
  const float* in;
  const float* w;
  const float* bias;
  const PolStep* steps;
  const int* nsteps;
  const float* res;
  float* out;
  float* out2;
  int K;                 // input channels
  int H, W, Hp, Wp;      // per-image geometry
  int B;
  int vstack;            // all B images stacked into one virtual image of VH interior rows
  int VH;
  int tilesX, tilesY, nct;
  int split, C_out1, C_out2, s2d;
  int out_hs;
};

constexpr int MT = 64, CC = 8;

template <int NBW, int MBW>
struct Geom {
  static constexpr int MBH = 32 / MBW;
  static constexpr int TW = MBW;
  static constexpr int TH = 4 * NBW * MBH;
  static constexpr int LW4 = (TW + 2 + 3) / 4;      // 16-byte units per LDS halo row
  static constexpr int LW = LW4 * 4;                // LDS row pitch in floats (>= TW + 2)
  static constexpr int LH = TH + 2;
  static constexpr int PLANE = LW * LH;
  static constexpr int IN_UNITS = CC * LH * LW4;    // 16-byte lane loads per chunk
  static constexpr int IN_INSTR = (IN_UNITS + 63) / 64;
  static constexpr int IN_PAD = IN_INSTR * 256;     // floats
  static constexpr int NI = (IN_INSTR + 3) / 4;
  static constexpr int W_ELEMS = 9 * CC * MT;      // LDS slot for all 9 taps (absent ones stay unused)
  static constexpr int W_INSTR = 18;               // two 256-float copies per tap
  static constexpr int NWJ = (W_INSTR + 3) / 4;
  static constexpr int STAGE = IN_PAD + W_ELEMS;
};

// MTB = 32-cout blocks per workgroup: 2 (a whole 64-cout weight tile) or 1 (half of it -- twice as many, half as
// heavy workgroups for the deep stages whose grids would otherwise leave CUs idle and latencies exposed).
template <int NBW, int MBW, int MTB>
__global__ __launch_bounds__(256) void policy_conv_kernel(PolArgs a) {
  using G = Geom<NBW, MBW>;
  __shared__ __attribute__((aligned(16))) float lds[2 * G::STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int t = blockIdx.x;
  constexpr int SPLIT = 2 / MTB;                  // workgroups per 64-cout tile
  const int cth = t % (a.nct * SPLIT);
  const int ct = cth / SPLIT, mhalf = cth % SPLIT; // 64-cout tile, which 32-cout half (MTB == 1)
  t /= a.nct * SPLIT;
  const int tx = t % a.tilesX;
  t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int b = t / a.tilesY;                     // 0 when stacked
  const int x0 = tx * G::TW, y0 = ty * G::TH;
  const int HpWp = a.Hp * a.Wp;
  const size_t img_stride = (size_t)a.K * HpWp;
  const int nchunk = a.K / CC;

  // ---- per-thread source offsets (floats) of the halo's 16-byte units (relative to channel 0 of the chunk, image b);
  //      the LDS image [c][hy][LW] is lane-linear in that order, so unit idx lands at float 4 * idx
  int ioff[G::NI];
#pragma unroll
  for (int k = 0; k < G::NI; ++k) {
    const int idx = (wave + 4 * k) * 64 + lane;
    const int c = idx / (G::LH * G::LW4);
    const int r = idx - c * (G::LH * G::LW4);
    const int hy = r / G::LW4;
    const int hx = (r - hy * G::LW4) * 4;
    int off = 0;
    if (idx < G::IN_UNITS) {
      if (a.vstack) {
        const int vr = y0 + hy;                   // padded virtual row
        const int img = vr / a.Hp;
        const int rr = vr - img * a.Hp;
        off = (img < a.B) ? (int)(img * img_stride) + c * HpWp + rr * a.Wp + hx : 0;
      } else {
        off = c * HpWp + (y0 + hy) * a.Wp + hx;
      }
    }
    ioff[k] = off;
  }
  const float* in_base = a.in + (size_t)b * img_stride + x0 + (POL_PADL - 1);
  const PolStep* steps = a.steps + (size_t)ct * nchunk;
  const int nsteps = a.nsteps[ct];

  constexpr int NS = G::NI + G::NWJ;
  auto issue_slot = [&](int slot, const float* src, const float* wsrc, unsigned mask, float* lstage) {
    if (slot < G::NI) {
      const int instr = wave + 4 * slot;
      if (instr < G::IN_INSTR) glds16(src + ioff[slot], lstage + instr * 256);
    } else {
      const int j = wave + 4 * (slot - G::NI);    // copy j: tap j/2, half j%2
      if (j < G::W_INSTR) {
        const int tap = j >> 1;
        if ((mask >> tap) & 1u) {
          const int rank = __builtin_popcount(mask & ((1u << tap) - 1u));
          glds16(wsrc + rank * 512 + (j & 1) * 256 + lane * 4, lstage + G::IN_PAD + j * 256);
        }
      }
    }
  };
  auto load_step = [&](int i, unsigned& mask, const float*& src, const float*& wsrc) {
    const PolStep st = steps[i];
    mask = __builtin_amdgcn_readfirstlane((unsigned)st.mask);
    const int ch = __builtin_amdgcn_readfirstlane((int)st.chunk);
    const unsigned wo = __builtin_amdgcn_readfirstlane(st.wofs);
    src = in_base + (size_t)ch * CC * HpWp;
    wsrc = a.w + (size_t)wo * 512;
  };

  f32x16 acc[MTB][NBW];
#pragma unroll
  for (int m = 0; m < MTB; ++m)
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int l31 = lane & 31, khalf = lane >> 5;
  const int py = l31 / MBW, px = l31 - py * MBW;
  const int b_lane = khalf * G::PLANE + (wave * NBW * G::MBH + py) * G::LW + px;
  const int a_lane = khalf * MT + l31 + 32 * mhalf;

  unsigned mask = 0, nmask = 0;
  const float *src = nullptr, *wsrc = nullptr;
  if (nsteps > 0) {
    load_step(0, mask, src, wsrc);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) issue_slot(sl, src, wsrc, mask, lds);
  }
  for (int i = 0; i < nsteps; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = i + 1 < nsteps;
    if (more) load_step(i + 1, nmask, src, wsrc);
    float* nstage = lds + ((i + 1) & 1) * G::STAGE;
    const float* lin = lds + (i & 1) * G::STAGE + b_lane;
    const float* lw = lds + (i & 1) * G::STAGE + G::IN_PAD + a_lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      if (more) {
#pragma unroll
        for (int sl = tap; sl < NS; sl += 9) issue_slot(sl, src, wsrc, nmask, nstage);
      }
      if (!((mask >> tap) & 1u)) continue;
#pragma unroll
      for (int cp = 0; cp < CC / 2; ++cp) {
        float av[MTB], bv[NBW];
#pragma unroll
        for (int m = 0; m < MTB; ++m) av[m] = lw[(tap * CC + cp * 2) * MT + m * 32];
#pragma unroll
        for (int n = 0; n < NBW; ++n) bv[n] = lin[(cp * 2) * G::PLANE + (n * G::MBH + dy) * G::LW + dx];
#pragma unroll
        for (int m = 0; m < MTB; ++m)
#pragma unroll
          for (int n = 0; n < NBW; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
      }
    }
    mask = nmask;
  }

  // ---- epilogue
  const bool second = ct >= a.split;
  float* obase = second ? a.out2 : a.out;
  const int Cthis = second ? a.C_out2 : a.C_out1;
  const int c0 = (second ? ct - a.split : ct) * MT;
  const int Hp2 = padded_h(a.H >> 1), Wp2 = pol_wp(a.W >> 1);
#pragma unroll
  for (int n = 0; n < NBW; ++n) {
    const int vy = y0 + (wave * NBW + n) * G::MBH + py;     // interior row of the (virtual) image
    const int x = x0 + px;
    int img = b, y = vy;
    bool ok = vy < a.VH && x < a.W;
    if (a.vstack) {
      const int vr = vy + 1;
      img = vr / a.Hp;
      const int rr = vr - img * a.Hp;
      y = rr - 1;
      ok = ok && rr >= 1 && rr <= a.H && img < a.B;
    }
    if (!ok) continue;
    if (a.out_hs) {
      // half-split store: this lane holds 4 of the 8 channels of each record (channels 4*khalf .. +3 of group q)
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int m = 0; m < MTB; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cl = (m + mhalf) * 32 + j + 8 * q + 4 * khalf;
            float v = acc[m][n][q * 4 + j] + a.bias[ct * MT + cl];
            if (!second) v = fmaxf(v, 0.f);
            v *= HS_ASCALE;
            hi[j] = (_Float16)v;
            lo[j] = (_Float16)(v - (float)hi[j]);
          }
          const int g = ((c0 + (m + mhalf) * 32) >> 3) + q;
          char* rp = reinterpret_cast<char*>(obase) +
                     ((((size_t)img * (Cthis >> 3) + g) * (a.H + 2) + (y + 1)) * (size_t)(a.W + 2) + (x + 1)) * 32 + 8 * khalf;
          *reinterpret_cast<h4*>(rp) = hi;
          *reinterpret_cast<h4*>(rp + 16) = lo;
        }
      }
      continue;
    }
#pragma unroll
    for (int m = 0; m < MTB; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = (m + mhalf) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const int c = c0 + cl;
        float v = acc[m][n][r] + a.bias[ct * MT + cl];
        const size_t off = (((size_t)img * Cthis + c) * a.Hp + (y + 1)) * a.Wp + x + POL_PADL;
        if (a.res) v += a.res[off];
        if (!second) v = fmaxf(v, 0.f);
        if (a.s2d) {
          const int ph = (y & 1) * 2 + (x & 1);
          obase[(((size_t)img * 4 * Cthis + ph * Cthis + c) * Hp2 + (y >> 1) + 1) * Wp2 + (x >> 1) + POL_PADL] = v;
        } else {
          obase[off] = v;
        }
      }
    }
  }
}

template <int NBW, int MBW>
int launch_cfg(PolArgs a, hipStream_t s) {
  using G = Geom<NBW, MBW>;
  a.tilesX = (a.W + G::TW - 1) / G::TW;
  a.tilesY = (a.VH + G::TH - 1) / G::TH;
  const long long grid = (long long)a.nct * a.tilesX * a.tilesY * (a.vstack ? 1 : a.B);
  if (NBW == 1 && grid < 512) {     // too few workgroups for 256 CUs: split every cout tile in two
    hipLaunchKernelGGL((policy_conv_kernel<NBW, MBW, 1>), dim3((unsigned)(2 * grid)), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((policy_conv_kernel<NBW, MBW, 2>), dim3((unsigned)grid), dim3(256), 0, s, a);
  }
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

}  // namespace

int launch_policy_conv(const PolicyConv& L, const float* in, float* out, float* out2, const float* res, bool s2d, int B,
                       int H, int W, hipStream_t s, bool out_hs) {
  if (L.cin % 8 != 0 || L.cout % 64 != 0 || L.split_c % 64 != 0 || (s2d && ((H | W) & 1))) {
    set_error("policy conv: unsupported geometry (cin %d cout %d split %d H %d W %d)", L.cin, L.cout, L.split_c, H, W);
    return PNPX_ERR_SHAPE;
  }
  PolArgs a;
  a.in = in;
  a.w = L.w;
  a.bias = L.bias;
  a.steps = L.steps;
  a.nsteps = L.nsteps;
  a.res = res;
  a.out = out;
  a.out2 = out2;
  a.K = L.cin;
  a.H = H;
  a.W = W;
  a.Hp = padded_h(H);
  a.Wp = pol_wp(W);
  a.B = B;
  a.vstack = (H <= 16 && B > 1 && (size_t)B * L.cin * a.Hp * a.Wp < (1u << 30)) ? 1 : 0;   // int gather offsets
  a.VH = a.vstack ? B * a.Hp - 2 : H;
  a.nct = L.cout / 64;
  a.split = L.split_c / 64;
  a.C_out1 = L.split_c;
  a.C_out2 = L.cout - L.split_c;
  a.s2d = s2d ? 1 : 0;
  a.out_hs = out_hs ? 1 : 0;
  a.tilesX = a.tilesY = 0;
  const int mbw = W >= 32 ? 32 : (W >= 16 ? 16 : 8);
  auto blocks = [&](int nbw) {
    const int th = 4 * nbw * (32 / mbw);
    return (long long)a.nct * ((W + mbw - 1) / mbw) * ((a.VH + th - 1) / th) * (a.vstack ? 1 : B);
  };
  const int nbw = (blocks(2) >= (a.vstack ? 256 : 1024)) ? 2 : 1;
  if (mbw == 32) return nbw == 2 ? launch_cfg<2, 32>(a, s) : launch_cfg<1, 32>(a, s);
  if (mbw == 16) return nbw == 2 ? launch_cfg<2, 16>(a, s) : launch_cfg<1, 16>(a, s);
  return nbw == 2 ? launch_cfg<2, 8>(a, s) : launch_cfg<1, 8>(a, s);
}

}  // namespace pnpx
