// 3x3 / pad 1 convolution + bias + LeakyReLU on the f16 MFMA pipe with fp32-class accuracy ("half-split", HS).
//
// Replaces the same 27 ConvLayer instances as conv3x3.hip (tfpnp/pnp/denoiser/models/unet.py:8-31).
// Every fp32 value v is carried as an unevaluated sum of two halves  v*S = hi + lo,  hi = f16(v*S),
// lo = f16(v*S - hi)  (22 significant bits, relative error 2^-22; S is a power of two that keeps lo out of the
// f16 subnormal range: 16 for activations, per-layer for weights).  A product is evaluated as
//     w*x  ~=  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi          (the dropped w_lo*x_lo term is 2^-22 relative)
// by three v_mfma_f32_32x32x16_f16 accumulating in fp32: products of f16 pairs are exact in fp32, so the only
// rounding besides the operand split is the same fp32 accumulation the plain-fp32 kernel has.  The f16 pipe runs
// 16x the fp32-MFMA rate, so 3 MFMAs per product is 5.3x the fp32-MFMA peak (838 vs 157 TFLOP/s).  Measured drift
// of the whole 30-iteration ADMM loop against an fp64 run is the same as plain fp32's.
//
// Activation layout "HS8" (internal to the denoiser): [B][C/8][H+2][W+2] records of 32 bytes
//   = hi[8 channels] f16 | lo[8 channels] f16;  zero border written once, producers write interiors only.
// GEMM view as in conv3x3.hip: D[cout][pixel], A = weights, B = shifted input; one MFMA k-step = 16 input channels
// of one tap (lanes 0-31 carry channels 0-7, lanes 32-63 channels 8-15 of the chunk).
// Workgroup = 4 waves (one per SIMD, the whole register file each); tile = MT couts x 4*NBW pixel blocks of 32;
// wave = MT x NBW*32 pixels = (MT/32) x NBW accumulators.  Per K-chunk (16 channels) the halo
// [2 groups][hi,lo][TH+2][TW+2] x 16 B and the weight slice [9][hi,lo][2][MT] x 16 B go global->LDS by 16-byte
// LDS-DMA; two stages; the DMA issue of chunk c+1 is interleaved with the taps of chunk c.  Operand fragments
// are single conflict-free ds_read_b128 (pixel / cout stride 16 B), tap shifts are immediates.
// Epilogue: bias + LeakyReLU in fp32, re-split to hi/lo, v_permlane32_swap pairs the two half-waves' 4-channel
// pieces into whole 8-channel records, one coalesced 32-byte record store per lane and group pair.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_hs.h"

namespace pnpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16b(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

template <int MT, int NBW, int MBW, int NSTAGE = 2>
struct HsGeom {
  static constexpr int MBH = 32 / MBW;
  static constexpr int NBLK = 4 * NBW;
  static constexpr int TW = MBW;
  static constexpr int TH = NBLK * MBH;
  static constexpr int LW = TW + 2;
  static constexpr int LH = TH + 2;
  static constexpr int PLANE = LW * LH;                 // pixels per LDS plane
  static constexpr int IN_LOADS = 4 * PLANE;            // 16-byte lane loads per chunk (2 groups x hi/lo)
  static constexpr int IN_INSTR = (IN_LOADS + 63) / 64; // wave-level DMA instructions (1 KiB each)
  static constexpr int NI = (IN_INSTR + 3) / 4;         // DMA slots per wave
  static constexpr int IN_BYTES = IN_INSTR * 1024;
  static constexpr int W_BYTES = 9 * 2 * 2 * MT * 16;   // [tap][hi,lo][kg][MT] x 16 B (multiple of 1 KiB)
  static constexpr int W_INSTR = W_BYTES / 1024;
  static constexpr int NWJ = (W_INSTR + 3) / 4;
  static constexpr int STAGE = IN_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = NSTAGE * STAGE;
  static constexpr int MTB = MT / 32;
  static constexpr int NS = NI + NWJ;
};

// Persistent kernel: a workgroup walks its tiles (XCD-aware order, below) and runs ONE software pipeline over the
// flattened (tile, K-chunk) steps: the DMA of step s+1 (possibly the next tile's first chunk) is issued while
// step s is multiplied, so the load latency is exposed once per workgroup, not once per tile, and the epilogue
// stores of a tile overlap the next tile's first loads.
template <int MT, int NBW, int MBW, int NSTAGE>
__global__ __launch_bounds__(256, 1) void conv_hs_kernel(ConvHsArgs a) {
  using G = HsGeom<MT, NBW, MBW, NSTAGE>;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HpWp = a.Hp * a.Wp;
  const int nch = (a.G0 + a.G1) / 2;
  // XCD-aware tile walk.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own
  // L2.  The nct cout-tiles of one pixel region read the same input halo, so they are given to workgroups of the
  // SAME XCD that run at the same time (consecutive "slots"): step k of workgroup (xcd, slot) handles
  //   j = slot + nslot*k,  pixel region p = 8*(j / nct) + xcd,  cout tile ct = j % nct.
  // nct divides nslot, so a workgroup keeps one cout tile (its weight slice stays hot) for its whole life.
  // With few pixel regions (small batches / deep levels) that grouping would leave whole XCDs idle and funnel every
  // weight byte through one XCD (measured at B=1, 8x8 level: 386 us with the 8 cout tiles on one XCD, 53 us spread
  // over eight), so below 32 regions the work items are simply dealt out to consecutive workgroups (= XCDs).
  const int nregions = a.tilesX * a.tilesY * a.B;
  const int nx = (gridDim.x % 8 == 0 && nregions >= 32) ? 8 : 1;
  const int xcd = blockIdx.x % nx, nslot = gridDim.x / nx;

  // per-thread byte offsets of the halo gather (identical for every chunk and tile)
  int ioff[G::NI];
#pragma unroll
  for (int k = 0; k < G::NI; ++k) {
    const int idx = (wave + 4 * k) * 64 + lane;
    const int q = idx / G::PLANE;               // plane: group = q >> 1, half = q & 1
    const int r = idx - q * G::PLANE;
    const int hy = r / G::LW;
    const int hx = r - hy * G::LW;
    ioff[k] = (idx < G::IN_LOADS) ? (((q >> 1) * HpWp + hy * a.Wp + hx) * 32 + (q & 1) * 16) : 0;
  }

  struct Tile {
    int ct, b, x0, y0;
  };
  auto valid = [&](int j) { return nx * (j / a.nct) + xcd < nregions; };
  auto decode = [&](int j) {
    Tile T;
    T.ct = j % a.nct;
    int t = nx * (j / a.nct) + xcd;
    const int tx = t % a.tilesX;
    t /= a.tilesX;
    const int ty = t % a.tilesY;
    T.b = t / a.tilesY;
    T.x0 = tx * G::TW;
    T.y0 = ty * G::TH;
    return T;
  };
  auto chunk_src = [&](const Tile& T, int chunk) -> const char* {
    const int g0 = chunk * 2;
    const char* src = (g0 < a.G0) ? a.in0 + ((size_t)T.b * a.G0 + g0) * HpWp * 32
                                  : a.in1 + ((size_t)T.b * a.G1 + (g0 - a.G0)) * HpWp * 32;
    return src + ((size_t)T.y0 * a.Wp + T.x0) * 32;
  };
  auto chunk_w = [&](const Tile& T, int chunk) -> const char* {
    return a.wpk + ((size_t)T.ct * nch + chunk) * G::W_BYTES;
  };
  auto issue_slot = [&](int slot, const char* src, const char* wsrc, char* lstage) {
    // the (wave-uniform) guards are compile-time true except on the last slot of each kind
    if (slot < G::NI) {
      const int instr = wave + 4 * slot;
      if (4 * slot + 3 < G::IN_INSTR || instr < G::IN_INSTR) glds16b(src + ioff[slot], lstage + instr * 1024);
    } else {
      const int j = wave + 4 * (slot - G::NI);
      if (4 * (slot - G::NI) + 3 < G::W_INSTR || j < G::W_INSTR)
        glds16b(wsrc + j * 1024 + lane * 16, lstage + G::IN_BYTES + j * 1024);
    }
  };

  f32x16 acc[G::MTB][NBW];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < G::MTB; ++m)
#pragma unroll
      for (int n = 0; n < NBW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };
  zero_acc();

  const int l31 = lane & 31, kg = lane >> 5;
  const int py = l31 / MBW, px = l31 - py * MBW;
  // LDS byte offsets of this lane's operand fragments (tap / tile / hi-lo shifts are compile-time immediates)
  const int b_lane = (kg * 2 * G::PLANE + (wave * NBW * G::MBH + py) * G::LW + px) * 16;
  const int a_lane = G::IN_BYTES + (kg * MT + l31) * 16;

  // one K-chunk of multiply; MORE: also issue the next step's DMA slots.  Explicit software pipeline over the
  // 9 taps: the fragments of tap t+1 are read from LDS before the MFMAs of tap t are issued, and the DMA slots of
  // this tap sit BEHIND those reads (the compiler keeps ds_reads in order with LDS-DMA, so a DMA at the top of a
  // tap would pin the next reads right in front of their first use).
  struct Frags {
    h8 ah[G::MTB], al[G::MTB], bh[NBW], bl[NBW];
  };
  auto load_frags = [&](Frags& f, const char* la, const char* lb, int tap) {
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int m = 0; m < G::MTB; ++m) {
      f.ah[m] = *reinterpret_cast<const h8*>(la + ((tap * 2 + 0) * 2 * MT + m * 32) * 16);
      f.al[m] = *reinterpret_cast<const h8*>(la + ((tap * 2 + 1) * 2 * MT + m * 32) * 16);
    }
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      f.bh[n] = *reinterpret_cast<const h8*>(lb + ((n * G::MBH + dy) * G::LW + dx) * 16);
      f.bl[n] = *reinterpret_cast<const h8*>(lb + (G::PLANE + (n * G::MBH + dy) * G::LW + dx) * 16);
    }
  };
  // KIND: 0 = nothing follows, 1 = the next step's halo + weights come by DMA
  auto body = [&](auto kind_tag, int stage, const char* nsrc, const char* nw) {
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool MORE = (KIND == 1);
    char* nstage = lds + (NSTAGE == 2 ? (stage ^ 1) * G::STAGE : 0);
    const char* lb = lds + stage * G::STAGE + b_lane;
    const char* la = lds + stage * G::STAGE + a_lane;
    Frags fr[2];
    load_frags(fr[0], la, lb, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const Frags& f = fr[tap & 1];
      if (tap + 1 < 9) load_frags(fr[(tap + 1) & 1], la, lb, tap + 1);
      if constexpr (NBW == 1) __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads up front
#pragma unroll
      for (int m = 0; m < G::MTB; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bh[n], acc[m][n], 0, 0, 0);
      if constexpr (MORE) {
#pragma unroll
        for (int sl = tap; sl < G::NS; sl += 9) issue_slot(sl, nsrc, nw, nstage);
      }
#pragma unroll
      for (int m = 0; m < G::MTB; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < G::MTB; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[m], f.bh[n], acc[m][n], 0, 0, 0);
      // schedule of this tap: the next tap's fragment reads are drip-fed between this tap's MFMAs (one ds_read per
      // MFMA) instead of being issued as one burst that lets the matrix pipe run dry
      if constexpr (NBW >= 2) {
        constexpr int NRD = 2 * G::MTB + 2 * NBW;   // ds_read_b128 per tap
        constexpr int NMF = 3 * G::MTB * NBW;
        if (tap + 1 < 9) {
#pragma unroll
          for (int i = 0; i < NRD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
          }
          if (MORE) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // this tap's DMA slots (VMEM read)
          __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
        }
      }
    }
  };

  const int Gout = a.nct * (MT / 8);
  // pack 16 activated values (rows of one 32x32 accumulator, already scaled by HS_ASCALE) into 32-byte HS8 records:
  // after the permlane swaps lanes 0-31 hold all 8 channels of the even group of each pair, lanes 32-63 of the odd.
  auto store_records = [&](const float (&v)[16], char* base, size_t pix_rec, int g_first, size_t group_stride_rec,
                           bool ok) {
    unsigned hp[4][2], lp[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float v0 = v[q * 4 + e * 2], v1 = v[q * 4 + e * 2 + 1];
        const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
        const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
        h2 hh = {h0, h1}, ll = {l0, l1};
        hp[q][e] = __builtin_bit_cast(unsigned, hh);
        lp[q][e] = __builtin_bit_cast(unsigned, ll);
      }
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      unsigned rec[8];  // hi[0..3] dwords, lo[0..3] dwords of one record
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        auto sh = __builtin_amdgcn_permlane32_swap(hp[2 * qp][e], hp[2 * qp + 1][e], false, false);
        auto sl = __builtin_amdgcn_permlane32_swap(lp[2 * qp][e], lp[2 * qp + 1][e], false, false);
        rec[e] = sh[0];
        rec[2 + e] = sh[1];
        rec[4 + e] = sl[0];
        rec[6 + e] = sl[1];
      }
      if (ok) {
        uint4* o = reinterpret_cast<uint4*>(base + ((size_t)(g_first + 2 * qp + kg) * group_stride_rec + pix_rec) * 32);
        o[0] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
        o[1] = make_uint4(rec[4], rec[5], rec[6], rec[7]);
      }
    }
  };

  auto epilogue = [&](const Tile& T) {
    const size_t img_rec = (size_t)T.b * Gout * HpWp;
    // fused MaxPool2d(2) output (models/unet.py:82-85): [B][Gout][H/2+2][W/2+2] records
    const int Hpo = a.H / 2 + 2, Wpo = a.W / 2 + 2;
    const bool do_pool = (MBW == 32) && (NBW >= 2) && (a.pool_out != nullptr);
    [[maybe_unused]] float odot[NBW];   // fused 1x1 out-conv partial sums (MT == 32 only)
#pragma unroll
    for (int n = 0; n < NBW; ++n) odot[n] = 0.f;
#pragma unroll
    for (int m = 0; m < G::MTB; ++m) {
      float bias[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = a.bias[T.ct * MT + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];
      float v[NBW][16];
      if (a.dmask) {
        // input-gradient convolution: the LeakyReLU derivative comes from the saved forward activation (same record
        // position as the output record; this lane's 4 channels of group q are hi[4*kg .. 4*kg+3])
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          const int y = min(T.y0 + (wave * NBW + n) * G::MBH + py, a.H - 1), x = min(T.x0 + px, a.W - 1);
          const size_t rec = img_rec + (size_t)(y + 1) * a.Wp + (x + 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint2 w = *reinterpret_cast<const uint2*>(a.dmask + (rec + (size_t)(T.ct * (MT / 8) + m * 4 + q) * HpWp) * 32 +
                                                            8 * kg);
            const unsigned hh[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool pos = (hh[j] & 0x8000u) == 0 && (hh[j] & 0x7fffu) != 0;
              const float t = acc[m][n][q * 4 + j] * a.inv_scale;
              v[n][q * 4 + j] = (pos ? t : t * a.slope) * HS_ASCALE;
            }
          }
        }
      } else if (a.res) {
        // residual block tail: act(conv + bias + res); res is an HS8 tensor laid out like the output
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          const int y = min(T.y0 + (wave * NBW + n) * G::MBH + py, a.H - 1), x = min(T.x0 + px, a.W - 1);
          const size_t rec = img_rec + (size_t)(y + 1) * a.Wp + (x + 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const char* rp = a.res + (rec + (size_t)(T.ct * (MT / 8) + m * 4 + q) * HpWp) * 32 + 8 * kg;
            const h4 rh = *reinterpret_cast<const h4*>(rp), rl = *reinterpret_cast<const h4*>(rp + 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float t = acc[m][n][q * 4 + j] * a.inv_scale + bias[q * 4 + j] +
                              ((float)rh[j] + (float)rl[j]) * (1.f / HS_ASCALE);
              v[n][q * 4 + j] = (t > 0.f ? t : t * a.slope) * HS_ASCALE;
            }
          }
        }
      } else {
#pragma unroll
        for (int n = 0; n < NBW; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float t = acc[m][n][r] * a.inv_scale + bias[r];
            v[n][r] = (t > 0.f ? t : t * a.slope) * HS_ASCALE;
          }
      }
      if constexpr (MT == 32) {
        if (a.outc_w) {   // out = clamp(x + outc(v) ...): accumulate this lane's 16 channels
          float w16[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) w16[r] = a.outc_w[(r & 3) + 8 * (r >> 2) + 4 * kg];
#pragma unroll
          for (int n = 0; n < NBW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) odot[n] = fmaf(w16[r], v[n][r], odot[n]);
        }
      }
      if (!(MT == 32 && a.outc_w)) {
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          const int y = T.y0 + (wave * NBW + n) * G::MBH + py;
          const int x = T.x0 + px;
          store_records(v[n], a.out, img_rec + (size_t)(y + 1) * a.Wp + (x + 1), T.ct * (MT / 8) + m * 4, HpWp,
                        (y < a.H) && (x < a.W));
        }
      }
      if constexpr (MBW == 32 && NBW >= 2) {
        if (do_pool) {
#pragma unroll
          for (int n = 0; n < NBW; n += 2) {
            float pm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float t = fmaxf(v[n][r], v[n + 1][r]);              // vertical pair: rows y, y+1
              pm[r] = fmaxf(t, __shfl_xor(t, 1, 64));                   // horizontal pair: lanes x, x^1
            }
            const int y = T.y0 + (wave * NBW + n), x = T.x0 + px;       // y even, both rows inside or both outside
            const bool ok = (y + 1 < a.H) && (x + 1 < a.W) && ((px & 1) == 0);
            store_records(pm, a.pool_out, (size_t)T.b * Gout * Hpo * Wpo + (size_t)(y / 2 + 1) * Wpo + (x / 2 + 1),
                          T.ct * (MT / 8) + m * 4, (size_t)Hpo * Wpo, ok);
          }
        }
      }
    }
    if constexpr (MT == 32) {
      if (a.outc_w) {   // finish the fused 1x1 conv + residual + clamp (models/unet.py:63-66, denoiser/base.py:32)
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          const float tot = odot[n] + __shfl_xor(odot[n], 32, 64);     // the other 16 channels live in lane ^ 32
          const int y = T.y0 + (wave * NBW + n) * G::MBH + py;
          const int x = T.x0 + px;
          if (kg == 0 && y < a.H && x < a.W) {
            const size_t o = ((size_t)T.b * a.H + y) * a.W + x;
            const float r = a.x_in[o] + (tot * (1.f / HS_ASCALE) + a.outc_b[0]);
            if (a.out_pre) a.out_pre[o] = r;
            a.out_img[o] = fminf(fmaxf(r, 0.f), 1.f);
          }
        }
      }
    }
  };

  int tile = blockIdx.x / nx;   // j of this workgroup's first step
  if (!valid(tile)) return;
  Tile cur = decode(tile);
  int ch = 0, stage = 0;
  {
    const char* src = chunk_src(cur, 0);
    const char* w = chunk_w(cur, 0);
#pragma unroll
    for (int sl = 0; sl < G::NS; ++sl) issue_slot(sl, src, w, lds);
  }
  while (true) {
    int ntile = tile, nchk = ch + 1;
    if (nchk == nch) {
      nchk = 0;
      ntile = tile + nslot;
    }
    const bool has_next = valid(ntile);
    Tile nxt = cur;
    if (nchk == 0 && has_next) nxt = decode(ntile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NSTAGE == 2) {
      if (!has_next) {
        body(std::integral_constant<int, 0>{}, stage, nullptr, nullptr);
      } else {
        body(std::integral_constant<int, 1>{}, stage, chunk_src(nxt, nchk), chunk_w(nxt, nchk));
      }
    } else {
      // single LDS stage: several workgroups share a CU and cover each other's load latency
      body(std::integral_constant<int, 0>{}, 0, nullptr, nullptr);
      __syncthreads();
      if (has_next) {
        const char* src = chunk_src(nxt, nchk);
        const char* w = chunk_w(nxt, nchk);
#pragma unroll
        for (int sl = 0; sl < G::NS; ++sl) issue_slot(sl, src, w, lds);
      }
    }
    if (ch == nch - 1) {
      epilogue(cur);
      zero_acc();
    }
    if (!has_next) break;
    tile = ntile;
    ch = nchk;
    cur = nxt;
    if (NSTAGE == 2) stage ^= 1;
  }
}

template <int MT, int NBW, int MBW, int NSTAGE>
static int launch_hs_cfg(const ConvHsArgs& a0, int B, int per_cu, hipStream_t s) {
  using G = HsGeom<MT, NBW, MBW, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    PNPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_hs_kernel<MT, NBW, MBW, NSTAGE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set = true;
  }
  ConvHsArgs a = a0;
  a.tilesX = (a.W + G::TW - 1) / G::TW;
  a.tilesY = (a.H + G::TH - 1) / G::TH;
  a.B = B;
  const long long ntiles = (long long)a.nct * a.tilesX * a.tilesY * B;
  // persistent: per_cu workgroups per CU (bounded by the LDS footprint), each walks ntiles/grid tiles
  int fit = (160 * 1024) / G::LDS_BYTES;
  fit = 1;                 // register budget: the kernels are compiled for one wave per SIMD (__launch_bounds__(256, 1))
  if (per_cu > fit) per_cu = fit;
  if (per_cu < 1) per_cu = 1;
  long long grid = 256LL * per_cu;
  if (grid > ntiles) grid = ntiles;
  if (grid >= 8) grid -= grid % 8;          // whole XCD groups (the kernel falls back to a plain walk otherwise)
  if (a.nct > 1 && (grid / 8) % a.nct != 0 && grid >= 8 * a.nct) grid -= grid % (8 * a.nct);
  hipLaunchKernelGGL((conv_hs_kernel<MT, NBW, MBW, NSTAGE>), dim3((unsigned)grid), dim3(256), G::LDS_BYTES, s, a);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

struct HsChoice {
  int nbw, nstage, per_cu;
};

template <int MT, int MBW>
static int launch_hs_mbw(const ConvHsArgs& a, int B, HsChoice c, hipStream_t s) {
  if (c.nstage == 1) {
    if (c.nbw == 4) return launch_hs_cfg<MT, 4, MBW, 1>(a, B, c.per_cu, s);
    if (c.nbw == 2) return launch_hs_cfg<MT, 2, MBW, 1>(a, B, c.per_cu, s);
    return launch_hs_cfg<MT, 1, MBW, 1>(a, B, c.per_cu, s);
  }
  if (c.nbw == 4) return launch_hs_cfg<MT, 4, MBW, 2>(a, B, c.per_cu, s);
  if (c.nbw == 2) return launch_hs_cfg<MT, 2, MBW, 2>(a, B, c.per_cu, s);
  return launch_hs_cfg<MT, 1, MBW, 2>(a, B, c.per_cu, s);
}

template <int MT>
static int launch_hs_mt(const ConvHsArgs& a, int B, hipStream_t s) {
  const int mbw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : 8);
  auto blocks = [&](int nbw) {
    const int th = 4 * nbw * (32 / mbw);
    return (long long)a.nct * ((a.W + mbw - 1) / mbw) * ((a.H + th - 1) / th) * B;
  };
  // Tile height by a small cost model: 256 persistent workgroups process ceil(tiles/256) rounds of tiles whose cost
  // is ~ (NBW + 0.3) (0.3 = per-tile prologue/epilogue/barrier overhead in units of one 128-pixel block row,
  // calibrated with tools/tune_hs.py).  Matters for batch sizes that do not fill the rounds (idx_left compaction).
  HsChoice c{4, 2, 1};
  {
    double best = 1e30;
    for (int nbw : {4, 2, 1}) {
      const long long nt = blocks(nbw);
      const double cost = (double)((nt + 255) / 256) * (nbw + 0.3);
      if (cost < best * 0.999) {
        best = cost;
        c.nbw = nbw;
      }
    }
  }
  // Production configurations are double-buffered with >= 100 KB of LDS per workgroup, i.e. exactly one resident
  // workgroup per CU by construction.  Two co-resident workgroups per CU (single-stage or 80-KiB variants,
  // reachable through the PNPX_HS_* hook below) were 3-6 % faster on some layers, but the <MT=32, NBW=2> instance
  // then produced rare stale 1-KiB operand pieces (a handful of tiles per launch; root cause not identified --
  // see DESIGN.md "open issues"), so co-residency is not used.
  if (MT == 32) c = HsChoice{4, 2, 1};
  // experiment hook: PNPX_HS_<MT>_<W>="nbw,nstage,per_cu"
  char key[64];
  snprintf(key, sizeof(key), "PNPX_HS_%d_%d", MT, a.W);
  if (const char* e = getenv(key)) {
    int x, y, z;
    if (sscanf(e, "%d,%d,%d", &x, &y, &z) == 3) c = HsChoice{x, y, z};
  }
  snprintf(key, sizeof(key), "PNPX_HS_%d_%d_G%d", MT, a.W, a.G0 + a.G1);
  if (const char* e = getenv(key)) {
    int x, y, z;
    if (sscanf(e, "%d,%d,%d", &x, &y, &z) == 3) c = HsChoice{x, y, z};
  }
  if (a.pool_out && c.nbw < 2) c.nbw = 2;   // the fused 2x2 pool pairs two rows held by one wave
  if (mbw == 32) return launch_hs_mbw<MT, 32>(a, B, c, s);
  if (mbw == 16) return launch_hs_mbw<MT, 16>(a, B, c, s);
  return launch_hs_mbw<MT, 8>(a, B, c, s);
}

bool conv_hs_can_pool(int H, int W) { return W >= 32 && (W % 2) == 0 && (H % 2) == 0; }

int launch_conv_hs(const ConvLayerHs& L, const char* in0, int G0, const char* in1, int G1, char* out, int B, int H,
                   int W, const ConvHsFuse& fuse, hipStream_t s) {
  if ((G0 + G1) * 8 != L.cin_pad || (G0 & 1) || (G1 & 1)) {
    set_error("conv_hs: channel groups %d+%d incompatible with packed layer (cin_pad %d)", G0, G1, L.cin_pad);
    return PNPX_ERR_SHAPE;
  }
  ConvHsArgs a;
  a.in0 = in0;
  a.G0 = G0;
  a.in1 = in1 ? in1 : in0;
  a.G1 = G1;
  a.wpk = L.w;
  a.bias = L.b;
  a.out = out;
  a.pool_out = conv_hs_can_pool(H, W) ? fuse.pool_out : nullptr;
  a.outc_w = (L.mt == 32 && L.cout == 32) ? fuse.outc_w : nullptr;
  a.outc_b = fuse.outc_b;
  a.x_in = fuse.x_in;
  a.out_img = fuse.out_img;
  a.out_pre = fuse.out_pre;
  if (fuse.outc_w && !a.outc_w) {
    set_error("conv_hs: fused out-conv needs a 32-channel layer");
    return PNPX_ERR_SHAPE;
  }
  a.H = H;
  a.W = W;
  a.Hp = H + 2;
  a.Wp = W + 2;
  a.nct = L.cout / L.mt;
  a.inv_scale = L.inv_scale;
  a.slope = fuse.slope;
  a.dmask = fuse.dmask;
  a.res = fuse.res;
  a.tilesX = a.tilesY = 0;
  a.B = B;
  if (L.mt == 64) return launch_hs_mt<64>(a, B, s);
  if (L.mt == 32) return launch_hs_mt<32>(a, B, s);
  set_error("conv_hs: no kernel for mt=%d", L.mt);
  return PNPX_ERR_SHAPE;
}

// ---- host-side weight packing -------------------------------------------------------------------------
static inline uint16_t f16_bits(_Float16 h) {
  uint16_t u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}

int conv_hs_mt(int cout) { return cout >= 64 ? 64 : 32; }

// w[cout][cin][3][3] fp32 -> [cout/mt][cin_pad/16][tap][hi,lo][kg][mt][8] f16, scaled by a power of two.
// Returns the scale s (weights are stored as split(s*w)).
float pack_conv_weights_hs(const float* w, int cout, int cin, int mt, uint16_t* dst) {
  const int cin_pad = (cin + 15) / 16 * 16;
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)cout * cin * 9; ++i) mx = std::fmax(mx, std::fabs(w[i]));
  int e = 0;
  if (mx > 0.f) {
    std::frexp(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
    e = 14 - e;                  // s*mx in [2^13, 2^14)
  }
  const float s = std::ldexp(1.0f, e);
  const int nct = cout / mt, nch = cin_pad / 16;
  for (int ct = 0; ct < nct; ++ct)
    for (int ch = 0; ch < nch; ++ch)
      for (int tap = 0; tap < 9; ++tap)
        for (int kgi = 0; kgi < 2; ++kgi)
          for (int m = 0; m < mt; ++m)
            for (int el = 0; el < 8; ++el) {
              const int co = ct * mt + m, ci = ch * 16 + kgi * 8 + el;
              const float v = (ci < cin) ? w[((size_t)co * cin + ci) * 9 + tap] * s : 0.f;
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (float)hi);
              const size_t base = ((((size_t)ct * nch + ch) * 9 + tap) * 2) * 2 * mt * 8;   // start of [hi,lo] pair
              dst[base + ((size_t)(0 * 2 + kgi) * mt + m) * 8 + el] = f16_bits(hi);
              dst[base + ((size_t)(1 * 2 + kgi) * mt + m) * 8 + el] = f16_bits(lo);
            }
  return s;
}

}  // namespace pnpx
