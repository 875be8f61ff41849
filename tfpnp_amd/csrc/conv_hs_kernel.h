// Device code of the half-split ("HS") 3x3 convolution: kernel template + per-instance launcher.
// Included by the translation units that instantiate it (conv_hs.hip: forward epilogues; conv_hs_bwd.hip: the
// input-gradient epilogue; conv_hs_res.hip: the residual-block epilogue of the policy network).  See conv_hs.hip
// for the numerical scheme and the data layouts.
#pragma once
#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "conv_hs.h"
#include "hs_rec.h"

namespace pnpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16b(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

// Epilogue flavours (compile-time: each instance carries only the registers / scalars its own epilogue needs)
enum HsEpi {
  EPI_ACT = 0,    // out = act(conv + bias) [+ fused 2x2 max-pool output]
  EPI_OUTC = 1,   // network tail: clamp(x + outc(act(conv + bias)) + b) written as an fp32 image (MT == 32)
  EPI_DMASK = 2,  // input-gradient convolution: out = conv * (saved activation > 0 ? 1 : slope)
  EPI_RES = 3     // out = act(conv + bias + res)
};

#ifndef HS_DMA_TAPS
#define HS_DMA_TAPS 9
#endif
constexpr int HS_BIAS_BYTES = 4096;   // LDS copy of the layer's (pre-scaled) bias vector, cout <= 1024

// WREG = 1 ("weights in registers", the cin = cout = 32 layers): the layer's whole weight slice (2 K-chunks x 9 taps x
// hi/lo = 36 fragments of 16 B per lane) is loaded into registers once per workgroup, a pipeline step is a whole TILE
// (CPS = 2 chunks: both halo planes sets in one stage, one barrier per tile instead of two, no weight DMA, no A-fragment
// ds_reads), and the LDS stage holds halos only.
// TAPS = bit mask of the 3x3 taps a layer has (bit dy*3+dx; 0x1FF = all nine).  Sparse-tap layers store, copy and
// multiply only their taps: 0x010 = 1x1 convolution, 0x01B = taps (0,0) (0,1) (1,0) (1,1) = a stride-2 3x3 convolution
// evaluated as a 2x2-window convolution over the space-to-depth input (policy ResNet stage entries).
constexpr int hs_ntaps(int mask) {
  int n = 0;
  for (int t = 0; t < 9; ++t) n += (mask >> t) & 1;
  return n;
}
constexpr int hs_nth_tap(int mask, int i) {   // the i-th set bit
  for (int t = 0; t < 9; ++t)
    if ((mask >> t) & 1) {
      if (i == 0) return t;
      --i;
    }
  return 0;
}
template <int MT, int NBW, int MBW, int NW, int WREG = 0, int TAPS = 0x1FF>
struct HsGeom {
  static constexpr int NTAPS = hs_ntaps(TAPS);
  static constexpr int CPS = WREG ? 2 : 1;              // K-chunks per pipeline step
  static constexpr int MBH = 32 / MBW;
  static constexpr int NBLK = NW * NBW;
  static constexpr int TW = MBW;
  static constexpr int TH = NBLK * MBH;
  static constexpr int LW = TW + 2;
  static constexpr int LH = TH + 2;
  static constexpr int PLANE = LW * LH;                 // pixels per LDS plane
  static constexpr int IN_LOADS = 4 * PLANE;            // 16-byte lane loads per chunk (2 groups x hi/lo)
  static constexpr int IN_INSTR_C = (IN_LOADS + 63) / 64; // wave-level DMA instructions (1 KiB each) per chunk
  static constexpr int IN_BYTES_C = IN_INSTR_C * 1024;
  static constexpr int IN_INSTR = CPS * IN_INSTR_C;
  static constexpr int NI = (IN_INSTR + NW - 1) / NW;   // DMA slots per wave
  static constexpr int IN_BYTES = IN_INSTR * 1024;
  static constexpr int W_BYTES_FULL = NTAPS * 2 * 2 * MT * 16;   // [tap][hi,lo][kg][MT] x 16 B (multiple of 1 KiB)
  static constexpr int W_BYTES = WREG ? 0 : W_BYTES_FULL;    // ... staged through LDS
  static constexpr int W_INSTR = W_BYTES / 1024;
  static constexpr int NWJ = (W_INSTR + NW - 1) / NW;
  static constexpr int STAGE = IN_BYTES + W_BYTES;
  static constexpr int BIAS_OFF = 2 * STAGE;
  static constexpr int BIAS_BYTES = WREG ? 256 : HS_BIAS_BYTES;   // WREG: one 32-cout tile
  static constexpr int XWIN_W = TW + 4, XWIN_H = TH + 4;          // WREG == 2: fp32 network-input window of a tile
  static constexpr int XWIN_OFF = 2 * STAGE + BIAS_BYTES;
  static constexpr int XWIN_BYTES = WREG == 2 ? ((XWIN_W * XWIN_H * 4 + 255) & ~255) : 0;
  static constexpr int LDS_USED = 2 * STAGE + BIAS_BYTES + XWIN_BYTES;
  // One workgroup per CU BY CONSTRUCTION, and NO other workgroup that needs LDS beside it: every instance requests the CU's whole
  // 160 KiB.  (r1-r3 padded the request past 80 KiB, which kept a second conv_hs workgroup out but let small-LDS kernels of other
  // streams in.  On this pool's MI355X boxes a wave executing packed-fp32 VALU instructions on a CU that hosts another kernel's
  // dense f16 MFMA wave computes wrong values in lanes 48-63 -- tools/attic/stress_aggressor.py, DESIGN.md appendix r4 -- so conv_hs
  // keeps LDS-using neighbours off its CUs; kernels without LDS can still share the CU's free registers.)
  static constexpr int LDS_BYTES = 160 * 1024;
  static constexpr int MTB = MT / 32;
  static constexpr int NS = NI + NWJ;
  static constexpr int NST = MTB * NBW * 4;             // 16-byte record stores per wave and tile
  static constexpr int NST_POOL = MTB * (NBW / 2) * 4;  // ... of the fused pool output
  static_assert(LDS_USED <= 160 * 1024, "tile does not fit the LDS");
  static_assert(NST + NST_POOL <= 63, "vmcnt is a 6-bit counter");
};

// lo halves of a hi/lo pair: f16(v0 - hi.lo16) | f16(v1 - hi.hi16) << 16, straight from the packed hi register
// (v - hi is exact in fp32, so this is the single rounding of the reference split).
__device__ __forceinline__ unsigned hs_lo_pair(unsigned hi_pk, float neg_one, float v0, float v1) {
  unsigned lo_pk;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo_pk)
      : "v"(hi_pk), "s"(neg_one), "v"(v0), "v"(v1));
  return lo_pk;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Persistent kernel: a workgroup walks its tiles (XCD-aware order, below) and runs ONE software pipeline over the
// flattened (tile, K-chunk) steps: the DMA of step s+1 (possibly the next tile's first chunk) is issued while
// step s is multiplied, so the load latency is exposed once per workgroup, not once per tile.  The record stores of
// a finished tile are never waited for: the wait in front of a step's barrier is a COUNTED vmcnt that covers only
// the LDS-DMA of that step (issued before the stores; the counter retires in order), so the stores drain while the
// next tile is multiplied.
//
// UPS = 1 ("fused bilinear x2"): the second source `in1` is the LOW-resolution tensor (a.ups_h x a.ups_w) and the
// workgroup carries four extra PRODUCER waves (one per SIMD, VALU only) that interpolate the halo of every K-chunk taken
// from it straight into the LDS stage (align_corners=True, ATen's association -- what upsample2x_hs_kernel writes to
// memory otherwise): the up-sampled tensor is never written to or read from HBM.  Three low-resolution windows LR[3] are
// in flight; in step s the producers (A) start the LDS-DMA of the window of step s+3 into LR[s % 3], (B) convert the
// window of step s+2 (landed: every wave waits for its own DMA in front of the step's barrier) from hi/lo records to
// fp32 IN PLACE (32 bytes either way: each source record is unpacked once instead of once per output pixel that
// touches it, ~4x), and (C) interpolate the window of step s+1 into the halo planes of stage (s+1) & 1 -- while the
// MFMA waves multiply stage s & 1 and DMA only the weights of step s+1.  The first two chunks of a tile come from `in0`
// (G0 >= 4), so the only producer work that looks into the next tile is (A) in a tile's last step.  Every wave executes
// the same barrier sequence.
constexpr int HS_UPS_WAVES = 4;
template <int MBW, int NBLK>
struct HsUpsGeom {   // low-resolution window feeding one (TH+2) x (TW+2) halo
  static constexpr int TW = MBW, TH = NBLK * (32 / MBW);
  static constexpr int LRW = TW / 2 + 3, LRH = TH / 2 + 3;
  static constexpr int RECS = LRW * LRH;                       // records per channel group
  static constexpr int PIECES = 2 * RECS * 2;                  // 2 groups x 16-byte halves
  static constexpr int INSTR = (PIECES + 63) / 64;
  static constexpr int BYTES = INSTR * 1024;
};

template <int MT, int NBW, int MBW, int NW, int EPI, int UPS = 0, int WREG = 0, int TAPS = 0x1FF>
// (register budget: four-wave instances with at most four 32 x 32 accumulators per wave are compiled for 256 registers like the
// eight-wave ones -- with the 512-register budget of one wave per SIMD the compiler keeps the accumulators in VGPRs across the
// loop and copies them to AGPRs and back around every step's MFMAs: 64 v_accvgpr moves per step.  Occupancy is set by LDS.)
__global__ __launch_bounds__((NW + HS_UPS_WAVES * UPS) * 64,
                             (NW == 4 && !UPS && !WREG && (MT / 32) * NBW <= (EPI == EPI_DMASK ? 2 : 4)) ? 2 : (NW + HS_UPS_WAVES * UPS) / 4) void conv_hs_kernel(ConvHsArgs a) {
  using G = HsGeom<MT, NBW, MBW, NW, WREG, TAPS>;
  static_assert(TAPS == 0x1FF || (!WREG && !UPS), "sparse-tap layers: generic instances only");
  static_assert(!(WREG && UPS) && (!WREG || MT == 32), "WREG: 32-cout single-source layers only");
  constexpr bool FIRST = (WREG == 2);   // the layer's input halo is computed from the fp32 network input (no halo DMA)
  using U = HsUpsGeom<MBW, NW * NBW>;
  constexpr int NT = (NW + HS_UPS_WAVES * UPS) * 64;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PNPX_TUNING
  if (a.wgt && tid == 0) a.wgt[2 * blockIdx.x] = wall_clock64();
#endif
  const int HpWp = a.Hp * a.Wp;
  const int nch = (a.G0 + a.G1) / 2 / G::CPS;   // pipeline steps per tile
  // XCD-aware tile walk.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own
  // L2.  The nct cout-tiles of one pixel region read the same input halo, so they are given to workgroups of the
  // SAME XCD that run at the same time (consecutive "slots"): step k of workgroup (xcd, slot) handles
  //   j = slot + nslot*k,  pixel region p = 8*(j / nct) + xcd,  cout tile ct = j % nct.
  // nct divides nslot, so a workgroup keeps one cout tile (its weight slice stays hot) for its whole life.
  // With few pixel regions (small batches / deep levels) that grouping would leave whole XCDs idle and funnel every
  // weight byte through one XCD (measured at B=1, 8x8 level: 386 us with the 8 cout tiles on one XCD, 53 us spread
  // over eight), so below 32 regions the work items are simply dealt out to consecutive workgroups (= XCDs).
  const int nregions = a.tilesX * a.tilesY * a.B;
  const int nx = (!a.plain_walk && gridDim.x % 8 == 0 && nregions >= 32) ? 8 : 1;
  const int xcd = blockIdx.x % nx, nslot = gridDim.x / nx;

  // the layer's bias vector, pre-scaled by HS_ASCALE, stays in LDS for the life of the workgroup (a global load in
  // the epilogue would make the compiler drain the in-flight LDS-DMA queue in front of it)
  if constexpr (EPI != EPI_DMASK) {
    float* lbias = reinterpret_cast<float*>(lds + G::BIAS_OFF);
    for (int i = tid; i < a.nct * MT; i += NT) lbias[i] = a.bias[i] * HS_ASCALE;
    __syncthreads();
  }

  // per-thread byte offsets of the halo gather (identical for every chunk and tile)
  int ioff[G::NI];
#pragma unroll
  for (int k = 0; k < G::NI; ++k) {
    const int instr = wave + NW * k;
    const int cc = instr / G::IN_INSTR_C;       // chunk within the step (always 0 unless WREG)
    const int idx = (instr - cc * G::IN_INSTR_C) * 64 + lane;
    const int q = idx / G::PLANE;               // plane: group = q >> 1, half = q & 1
    const int r = idx - q * G::PLANE;
    const int hy = r / G::LW;
    const int hx = r - hy * G::LW;
    ioff[k] = (idx < G::IN_LOADS) ? (((2 * cc + (q >> 1)) * HpWp + hy * a.Wp + hx) * 32 + (q & 1) * 16) : 0;
#ifdef PNPX_TUNING   // ablation 4 (invalid results): address the halo as if hi / lo were separate dense planes
    if ((a.abl & 4) && idx < G::IN_LOADS) ioff[k] = (q * HpWp + hy * a.Wp + hx) * 16;
#endif
  }

  // Weight slices are packed [cout / w_mt][chunk][tap][hi,lo][kg][w_mt] x 16 B.  A 32-cout instance may run over a
  // 64-cout packing ("half tiles": twice the tiles for batches that do not fill the chip, same K order -> same bits):
  // its slice is then every other 512-byte run of the 64-cout slice, gathered by the DMA's per-lane source address.
  const bool half_tiles = (MT == 32) && (a.w_mt == 64);
  // A tile carries its three base addresses (first / second source at the tile's halo origin, weight slice of its cout tile),
  // computed once where the walk is decoded: a pipeline step then costs one multiply-add per operand instead of the whole
  // 64-bit index arithmetic (one wave per SIMD has nothing to hide scalar code behind)
  struct Tile {
    int ct, b, x0, y0;
    const char *s0, *s1, *w;
  };
  const size_t plane_bytes = (size_t)HpWp * 32;                         // one channel group of one image
  const size_t w_step = half_tiles ? 2 * (size_t)G::W_BYTES : (size_t)G::W_BYTES;   // weight bytes per K-chunk of a cout tile
  auto fdiv = [](unsigned n, const HsFastDiv& f) { return f.d > 1 ? __umulhi(n, f.m) : n; };
  auto valid = [&](int j) { return nx * (int)fdiv(j, a.div_nct) + xcd < nregions; };
  auto decode = [&](int j) {
    Tile T;
    const unsigned q = fdiv(j, a.div_nct);
    T.ct = j - (int)q * a.nct;
    const unsigned t = nx * q + xcd;
    const unsigned t1 = fdiv(t, a.div_tx);
    const int tx = t - t1 * a.tilesX;
    const unsigned t2 = fdiv(t1, a.div_ty);
    const int ty = t1 - t2 * a.tilesY;
    T.b = t2;
    T.x0 = tx * G::TW;
    T.y0 = ty * G::TH;
    const size_t pix = ((size_t)T.y0 * a.Wp + T.x0) * 32;
    T.s0 = a.in0 + (size_t)T.b * a.G0t * plane_bytes + pix;
    T.s1 = a.in1 + ((size_t)T.b * a.G1) * plane_bytes + pix - (size_t)a.G0 * plane_bytes;   // + g0 * plane_bytes for g0 >= G0
    T.w = half_tiles ? a.wpk + (size_t)(T.ct >> 1) * nch * w_step + (T.ct & 1) * 512 : a.wpk + (size_t)T.ct * nch * w_step;
    return T;
  };
  auto chunk_src = [&](const Tile& T, int chunk) -> const char* {   // chunk = pipeline step (CPS K-chunks)
    const int g0 = chunk * 2 * G::CPS;
#ifdef PNPX_TUNING
    if (a.abl & (4 | 16)) {
      const char* src = (g0 < a.G0) ? a.in0 + ((size_t)T.b * a.G0t + g0) * HpWp * 32
                                    : a.in1 + ((size_t)T.b * a.G1 + (g0 - a.G0)) * HpWp * 32;
      if (a.abl & 4) return src + ((size_t)T.y0 * a.Wp + T.x0) * 16;
      return a.in0 + (size_t)g0 * HpWp * 32 + (size_t)(blockIdx.x & 7) * a.Wp * 32;   // ablation 16 (invalid results): every tile reads the same cache-resident halo
    }
#endif
    return ((g0 < a.G0) ? T.s0 : T.s1) + (size_t)g0 * plane_bytes;
  };
  auto chunk_w = [&](const Tile& T, int chunk) -> const char* {
    if constexpr (WREG) return a.wpk;   // unused: no weight DMA
    return T.w + (size_t)chunk * w_step;
  };
  bool next_halo_by_dma = true;   // UPS: false while the next step's chunk comes from the low-resolution source
  auto issue_slot = [&](int slot, const char* src, const char* wsrc, char* lstage) {
    if constexpr (FIRST) return;
    // the (wave-uniform) guards are compile-time true except on the last slot of each kind
    if (slot < G::NI) {
      if (UPS && !next_halo_by_dma) return;
      const int instr = wave + NW * slot;
      if (NW * slot + NW - 1 < G::IN_INSTR || instr < G::IN_INSTR) glds16b(src + ioff[slot], lstage + instr * 1024);
    } else {
      const int j = wave + NW * (slot - G::NI);
      if (NW * (slot - G::NI) + NW - 1 < G::W_INSTR || j < G::W_INSTR)
        // 16-byte piece i = j*64 + lane of the slice: row r = i / 32 (tap, half, kg), cout m = i % 32
        glds16b(wsrc + (half_tiles ? (j * 2 + (lane >> 5)) * 1024 + (lane & 31) * 16 : j * 1024 + lane * 16),
                lstage + G::IN_BYTES + j * 1024);
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
  // WREG: this lane's A fragments of the whole layer (cout tile 0; the launcher guarantees nct == 1, w_mt == 32)
  [[maybe_unused]] h8 areg[WREG ? 2 : 1][WREG ? 9 : 1][2];
  if constexpr (WREG) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          areg[cc][tap][h] = *reinterpret_cast<const h8*>(a.wpk + (size_t)cc * G::W_BYTES_FULL +
                                                           ((tap * 2 + h) * 2 * MT + kg * MT + l31) * 16);
  }

  // one K-chunk of multiply; MORE: also issue the next step's DMA slots.  Explicit software pipeline over the
  // 9 taps: the fragments of tap t+1 are read from LDS before the MFMAs of tap t are issued, and the DMA slots of
  // this tap sit BEHIND those reads (the compiler keeps ds_reads in order with LDS-DMA, so a DMA at the top of a
  // tap would pin the next reads right in front of their first use).
  struct Frags {
    h8 ah[G::MTB], al[G::MTB], bh[NBW], bl[NBW];
  };
  // `tap` runs over the CPS * 9 (chunk, tap) pairs of a step
  auto load_frags = [&](Frags& f, const char* la, const char* lb, int tapx) {
    const int cc = tapx / G::NTAPS, tap = tapx % G::NTAPS;     // tap = index among the layer's taps (weight packing order)
    const int rt = hs_nth_tap(TAPS, tap);                      // ... and which of the 3x3 positions it is
    const int dy = rt / 3, dx = rt % 3;
    if constexpr (WREG) {
      f.ah[0] = areg[cc][tap][0];
      f.al[0] = areg[cc][tap][1];
    } else {
#pragma unroll
      for (int m = 0; m < G::MTB; ++m) {
        f.ah[m] = *reinterpret_cast<const h8*>(la + ((tap * 2 + 0) * 2 * MT + m * 32) * 16);
        f.al[m] = *reinterpret_cast<const h8*>(la + ((tap * 2 + 1) * 2 * MT + m * 32) * 16);
      }
    }
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      f.bh[n] = *reinterpret_cast<const h8*>(lb + cc * G::IN_BYTES_C + ((n * G::MBH + dy) * G::LW + dx) * 16);
      f.bl[n] = *reinterpret_cast<const h8*>(lb + cc * G::IN_BYTES_C + (G::PLANE + (n * G::MBH + dy) * G::LW + dx) * 16);
    }
  };
  // KIND: 0 = nothing follows, 1 = the next step's halo + weights come by DMA
  auto body = [&](auto kind_tag, int stage, const char* nsrc, const char* nw) {
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool MORE = (KIND == 1);
    char* nstage = lds + (stage ^ 1) * G::STAGE;
    const char* lb = lds + stage * G::STAGE + b_lane;
    const char* la = lds + stage * G::STAGE + a_lane;
    constexpr int NTAP = G::NTAPS * G::CPS;                       // (chunk, tap) pairs of one step
    constexpr int DMA_TAPS = WREG ? NTAP : (HS_DMA_TAPS < NTAP ? HS_DMA_TAPS : NTAP);   // taps the next step's DMA slots are spread over
    Frags fr[2];
    load_frags(fr[0], la, lb, 0);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const Frags& f = fr[tap & 1];
      if (tap + 1 < NTAP) load_frags(fr[(tap + 1) & 1], la, lb, tap + 1);
      if constexpr (NBW == 1) __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads up front
#pragma unroll
      for (int m = 0; m < G::MTB; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bh[n], acc[m][n], 0, 0, 0);
      if constexpr (MORE) {
        // the next step's DMA slots are spread over the first DMA_TAPS taps of this step
        if (tap < DMA_TAPS) {
#pragma unroll
          for (int sl = tap; sl < G::NS; sl += DMA_TAPS) issue_slot(sl, nsrc, nw, nstage);
        }
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
        constexpr int NRD = (WREG ? 0 : 2 * G::MTB) + 2 * NBW;   // ds_read_b128 per tap
        constexpr int NMF = 3 * G::MTB * NBW;
        if (tap + 1 < NTAP) {
#pragma unroll
          for (int i = 0; i < NRD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
          }
          if (MORE && tap < DMA_TAPS)
            __builtin_amdgcn_sched_group_barrier(0x020, (G::NS + DMA_TAPS - 1) / DMA_TAPS, 0);   // this tap's DMA slots
          __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
        }
      }
    }
  };

  const int Gout = a.nct * (MT / 8);
  const float c16 = a.inv_scale * HS_ASCALE;   // accumulator -> HS_ASCALE * value (a power of two)
  const float neg_one = a.neg_one;             // -1.0f in a scalar register
  h2 range_chk = {(_Float16)0.f, (_Float16)0.f};
  // pack 16 activated values (rows of one 32x32 accumulator, already scaled by HS_ASCALE) into 32-byte HS8 records.
  // The weight rows of every 32-cout block are packed in the order that makes the MFMA's C layout put channels
  // 16*kg .. 16*kg+15 of the block into lane half kg (hs_row_channel): register r IS channel 16*kg + r, so a lane holds
  // two complete 8-channel records (groups 2*kg and 2*kg + 1) and nothing has to move between lanes.
  // `rsrc` is a buffer descriptor of one image's tensor slice; masked lanes pass an out-of-range offset (the store is
  // dropped by the bounds check), so every wave issues exactly the same number of stores (see the counted vmcnt).
  auto store_records = [&](const float (&v)[16], __amdgpu_buffer_rsrc_t rsrc, int pix_rec, int g_first,
                           int group_stride_rec, bool ok) {
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      unsigned rec[8];  // hi[0..3] dwords, lo[0..3] dwords of one record (channels 8*qp .. 8*qp+7 of this lane)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = v[qp * 8 + e * 2], v1 = v[qp * 8 + e * 2 + 1];
        const h2 hh = {(_Float16)v0, (_Float16)v1};
        range_chk = hh * (h2){(_Float16)0.f, (_Float16)0.f} + range_chk;   // inf * 0 = NaN: sticky per lane
        rec[e] = __builtin_bit_cast(unsigned, hh);
        rec[4 + e] = hs_lo_pair(rec[e], neg_one, v0, v1);
      }
      const int off = ok ? ((g_first + 2 * kg + qp) * group_stride_rec + pix_rec) * 32 : (int)0x80000000;
#ifdef PNPX_TUNING   // ablation (invalid results): conversion work kept alive, stores dropped
      if (a.abl & 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(rec[i]));
        asm volatile("" ::"v"(off));
        continue;
      }
      if (a.abl & 8) {   // ablation 8 (invalid results): store hi / lo as two dense planes
        const int poff = ok ? ((g_first + 2 * kg + qp) * 2 * group_stride_rec + pix_rec) * 16 : (int)0x80000000;
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){rec[0], rec[1], rec[2], rec[3]}, rsrc, poff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){rec[4], rec[5], rec[6], rec[7]}, rsrc, poff, group_stride_rec * 16, 0);
        continue;
      }
#endif
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){rec[0], rec[1], rec[2], rec[3]}, rsrc, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){rec[4], rec[5], rec[6], rec[7]}, rsrc, off, 16, 0);
    }
  };

  // EPI_DMASK: the hi halves of the saved forward activation at this lane's output positions (their signs are LeakyReLU').
  // Fetched at the FIRST step of a tile, ahead of that step's DMA issue, so that they land under the tile's main loop: issued in
  // the epilogue they cost a full exposed load latency per tile (r3: +190 us on a 110 us launch of the 32 -> 32 layers).
  [[maybe_unused]] uint2 mk[EPI == EPI_DMASK ? G::MTB : 1][EPI == EPI_DMASK ? NBW : 1][4];
  [[maybe_unused]] auto fetch_masks = [&](const Tile& T) {
    const size_t img_rec = (size_t)T.b * Gout * HpWp;
#pragma unroll
    for (int m = 0; m < G::MTB; ++m)
#pragma unroll
      for (int n = 0; n < NBW; ++n) {
        const int y = min(T.y0 + (wave * NBW + n) * G::MBH + py, a.H - 1), x = min(T.x0 + px, a.W - 1);
        const size_t rec = img_rec + (size_t)(y + 1) * a.Wp + (x + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          mk[m][n][q] = a.dmask ? *reinterpret_cast<const uint2*>(a.dmask + (rec + (size_t)(T.ct * (MT / 8) + m * 4 + 2 * kg + (q >> 1)) * HpWp) * 32 + 8 * (q & 1))
                                : make_uint2(0x3c003c00u, 0x3c003c00u);
      }
  };

  auto epilogue = [&](const Tile& T) {
    const size_t img_rec = (size_t)T.b * Gout * HpWp;
    const int img_bytes = Gout * HpWp * 32;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out + img_rec * 32, 0, img_bytes, 0x00020000);
    // fused MaxPool2d(2) output (models/unet.py:82-85): [B][Gout][H/2+2][W/2+2] records
    const int Hpo = a.H / 2 + 2, Wpo = a.W / 2 + 2;
    const bool do_pool = (EPI == EPI_ACT) && (MBW == 32) && (NBW >= 2) && (a.pool_out != nullptr);
    [[maybe_unused]] float odot[NBW];   // fused 1x1 out-conv partial sums (EPI_OUTC)
#pragma unroll
    for (int n = 0; n < NBW; ++n) odot[n] = 0.f;
#pragma unroll
    for (int m = 0; m < G::MTB; ++m) {
      float bias[16];
      if constexpr (EPI != EPI_DMASK) {
        const char* lb = lds + G::BIAS_OFF + (T.ct * MT + m * 32 + 16 * kg) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(lb + 16 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) bias[q * 4 + j] = bq[j];
        }
      }
      float v[NBW][16];
      if constexpr (EPI == EPI_DMASK) {
        // input-gradient convolution: the LeakyReLU derivative comes from the saved forward activation (same record
        // position as the output record; this lane's channels 4*q .. 4*q+3 are hi[4*(q&1) ..] of group 2*kg + (q>>1))
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // mk: fetched by fetch_masks() at the tile's first step (0x3c00 = "positive" without a mask: linear epilogue)
            const uint2 w = mk[m][n][q];
            const unsigned hh[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool pos = (hh[j] & 0x8000u) == 0 && (hh[j] & 0x7fffu) != 0;
              const float t = acc[m][n][q * 4 + j] * c16;
              v[n][q * 4 + j] = pos ? t : t * a.slope;
            }
          }
        }
      } else if constexpr (EPI == EPI_RES) {
        // residual block tail: act(conv + bias + res); res is an HS8 tensor laid out like the output
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          const int y = min(T.y0 + (wave * NBW + n) * G::MBH + py, a.H - 1), x = min(T.x0 + px, a.W - 1);
          const size_t rec = img_rec + (size_t)(y + 1) * a.Wp + (x + 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            h4 rh = {0, 0, 0, 0}, rl = {0, 0, 0, 0};
            if (a.res) {
              const char* rp = a.res + (rec + (size_t)(T.ct * (MT / 8) + m * 4 + 2 * kg + (q >> 1)) * HpWp) * 32 + 8 * (q & 1);
              rh = *reinterpret_cast<const h4*>(rp);
              rl = *reinterpret_cast<const h4*>(rp + 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float t = __builtin_fmaf(acc[m][n][q * 4 + j], c16, bias[q * 4 + j]) + ((float)rh[j] + (float)rl[j]);
              v[n][q * 4 + j] = __builtin_fmaxf(t, t * a.slope);
            }
          }
        }
      } else {
#pragma unroll
        for (int n = 0; n < NBW; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float t = __builtin_fmaf(acc[m][n][r], c16, bias[r]);
            v[n][r] = __builtin_fmaxf(t, t * a.slope);   // = t > 0 ? t : slope * t  for 0 <= slope <= 1
          }
      }
      if constexpr (EPI == EPI_OUTC) {
        // out = clamp(x + outc(v) ...): accumulate this lane's 16 channels
        float w16[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) w16[r] = a.outc_w[16 * kg + r];
#pragma unroll
        for (int n = 0; n < NBW; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) odot[n] = fmaf(w16[r], v[n][r], odot[n]);
      } else {
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          const int y = T.y0 + (wave * NBW + n) * G::MBH + py;
          const int x = T.x0 + px;
          store_records(v[n], orsrc, (y + 1) * a.Wp + (x + 1), T.ct * (MT / 8) + m * 4, HpWp, (y < a.H) && (x < a.W));
        }
      }
      if constexpr (EPI == EPI_ACT && MBW == 32 && NBW >= 2) {
        if (do_pool) {
          const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
              a.pool_out + (size_t)T.b * Gout * Hpo * Wpo * 32, 0, Gout * Hpo * Wpo * 32, 0x00020000);
#pragma unroll
          for (int n = 0; n < NBW; n += 2) {
            float pm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float t = __builtin_fmaxf(v[n][r], v[n + 1][r]);    // vertical pair: rows y, y+1
              const float o = __builtin_bit_cast(                      // horizontal pair: lanes x, x^1 (DPP quad_perm 1,0,3,2)
                  float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));
              pm[r] = __builtin_fmaxf(t, o);
            }
            const int y = T.y0 + (wave * NBW + n), x = T.x0 + px;       // y even, both rows inside or both outside
            const bool ok = (y + 1 < a.H) && (x + 1 < a.W) && ((px & 1) == 0);
            store_records(pm, prsrc, (y / 2 + 1) * Wpo + (x / 2 + 1), T.ct * (MT / 8) + m * 4, Hpo * Wpo, ok);
          }
        }
      }
    }
    if constexpr (EPI == EPI_OUTC) {
      // finish the fused 1x1 conv + residual + clamp (models/unet.py:63-66, denoiser/base.py:32)
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
  };

  // FIRST: the fp32 input window of the NEXT tile goes global -> LDS by 4-byte LDS-DMA right after this tile's halo has
  // been generated, i.e. BEFORE this tile's record stores (vmcnt retires in order: a load behind the stores would wait for
  // them); window elements outside the image are fetched from a zero word.  Element i lands at byte 4 i.
  [[maybe_unused]] auto xwin_fetch = [&](const Tile& T) {
    const float* xb = a.first_x + (size_t)T.b * a.H * a.W;
    char* xw = lds + G::XWIN_OFF;
#pragma unroll
    for (int k = 0; k < (G::XWIN_BYTES / 256 + NW - 1) / NW; ++k) {
      const int instr = wave + NW * k;
      if (instr < G::XWIN_BYTES / 256) {
        const int i = instr * 64 + lane;
        const int wy = i / G::XWIN_W, wx = i - wy * G::XWIN_W;
        const int yy = T.y0 - 2 + wy, xc = T.x0 - 2 + wx;
        const bool ok = (i < G::XWIN_W * G::XWIN_H) && (yy >= 0) && (yy < a.H) && (xc >= 0) && (xc < a.W);
        const float* src = ok ? xb + (size_t)yy * a.W + xc : a.first_zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xw + instr * 256), 4, 0, 0);
      }
    }
  };
  // FIRST: this tile's (TH+2) x (TW+2) x 32-channel input halo = the network's first convolution evaluated on the fly
  // (exact fp32 FMA chains in conv_first_hs_kernel's order, same hi/lo split: bit-identical to the tensor it replaces;
  // halo pixels outside the image are the tensor's zero border).  One (pixel, 8-channel group) item per lane and round;
  // the 100 left-over pixels of every group go to a different pair of waves so that all waves do five rounds.
  // The first layer's weights + bias live in LDS for the life of the workgroup: group g's 8 x 18 weights and 8 biases
  // (608 B) sit in the 768-byte slack behind chunk region g of the two halo stages (a chunk region is IN_BYTES_C = a whole
  // number of 1-KiB DMA pieces; the planes end 768 B earlier and nothing else writes there in FIRST mode).
  // (Measured alternatives: wave-uniform scalar weight loads per item 0.285 ms per folded launch, scalar weights hoisted
  // per channel over two pixels 0.45 ms (143 spilled registers); this form 0.211 ms = the two separate launches + 0.034.)
  static_assert(!FIRST || G::IN_BYTES_C - 4 * G::PLANE * 16 >= 608, "no slack for the first layer's weights");
  [[maybe_unused]] auto first_wtab = [&](int g) -> float* {
    return reinterpret_cast<float*>(lds + (g >> 1) * G::STAGE + (g & 1) * G::IN_BYTES_C + 4 * G::PLANE * 16);
  };
  if constexpr (FIRST) {
    for (int i = tid; i < 4 * 152; i += NT) {
      const int g = i / 152, k = i - g * 152;
      first_wtab(g)[k] = k < 144 ? a.first_w[g * 144 + k] : a.first_b[g * 8 + (k - 144)];
    }
    __syncthreads();
  }
  [[maybe_unused]] auto gen_halo = [&](const Tile& T, char* lstage) {
    const float sg = a.first_sigma[(size_t)T.b * a.first_sigma_stride];
    const float* xw = reinterpret_cast<const float*>(lds + G::XWIN_OFF);   // window origin = image (y0 - 2, x0 - 2), 0 outside
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
      const float* wt = first_wtab(g);
      const int rot = (tid + NT - g * 128) % NT;     // lane's slot in this group's item list
#pragma unroll 1
      for (int r = rot; r < G::PLANE; r += NT) {
        const int hy = r / G::LW, hx = r - hy * G::LW;
        const int y = T.y0 - 1 + hy, x = T.x0 - 1 + hx;
        const bool inside = (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        float xi[9], si[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yy = y + t / 3 - 1, xc = x + t % 3 - 1;
          const bool in = inside && (yy >= 0) && (yy < a.H) && (xc >= 0) && (xc < a.W);
          xi[t] = xw[(hy + t / 3) * G::XWIN_W + hx + t % 3];       // (yy - (y0 - 2), xc - (x0 - 2)); zero outside the image
          si[t] = in ? sg : 0.f;
        }
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float acc = wt[144 + c];
#pragma unroll
          for (int t = 0; t < 9; ++t) acc = fmaf(wt[c * 18 + t], xi[t], acc);
#pragma unroll
          for (int t = 0; t < 9; ++t) acc = fmaf(wt[c * 18 + 9 + t], si[t], acc);
          v[c] = inside ? fmaxf(acc, acc * a.first_slope) * HS_ASCALE : 0.f;
        }
        const HsRec rec = hs_pack(v);
        char* dst = lstage + (g >> 1) * G::IN_BYTES_C + (((g & 1) * 2) * G::PLANE + r) * 16;
        *reinterpret_cast<h8v*>(dst) = rec.hi;
        *reinterpret_cast<h8v*>(dst + G::PLANE * 16) = rec.lo;
      }
    }
  };

#ifdef HS_TRACE
  int tk = 0;
  const bool tr_on = a.trace && blockIdx.x == 8 && wave == 0;
  auto mark = [&](int tag) {
    if (tr_on && tk < 4000) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (lane == 0) a.trace[tk] = (t << 4) | (unsigned)tag;
      ++tk;
    }
  };
#else
  auto mark = [](int) {};
#endif
  // ---- UPS: producer waves (see the kernel comment)
  const bool producer = UPS && wave >= NW;
  if (UPS && producer) __builtin_amdgcn_s_setprio(3);   // little work, but on the critical path of every barrier
  const int pw = wave - NW;                                   // producer wave 0..3
  char* const lr_base = lds + G::BIAS_OFF + HS_BIAS_BYTES;    // LR[3], U::BYTES each
  auto chunk_is_up = [&](int c) { return UPS && 2 * c >= a.G0; };
  auto lr_origin = [&](const Tile& T, int& r_lo, int& c_lo) {
    r_lo = (int)(a.ups_sy * (float)max(T.y0 - 1, 0));
    c_lo = (int)(a.ups_sx * (float)max(T.x0 - 1, 0));
  };
  // LDS-DMA of the low-resolution window of chunk c (2 groups x U::RECS records x 32 B)
  auto lr_issue = [&](const Tile& T, int c, char* dst) {
    int r_lo, c_lo;
    lr_origin(T, r_lo, c_lo);
    const int lw2 = a.ups_w + 2;
    const char* base = a.in1 + ((size_t)T.b * a.G1 + (2 * c - a.G0)) * (size_t)(a.ups_h + 2) * lw2 * 32;
#pragma unroll
    for (int k = 0; k < (U::INSTR + HS_UPS_WAVES - 1) / HS_UPS_WAVES; ++k) {
      const int instr = pw + HS_UPS_WAVES * k;
      if (instr < U::INSTR) {
        int piece = instr * 64 + lane;
        if (piece >= U::PIECES) piece = 0;                    // tail lanes re-read piece 0 into the padding of the buffer
        // LDS layout [group][half][record] x 16 B (planar: neighbouring records 16 B apart -> conflict-free reads)
        const int pl = piece / U::RECS;                       // plane = group * 2 + half
        const int rec = piece - pl * U::RECS;
        const int g = pl >> 1;
        const int rr = rec / U::LRW, cc = rec - rr * U::LRW;
        const int yy = min(r_lo + rr, a.ups_h - 1), xx = min(c_lo + cc, a.ups_w - 1);
        glds16b(base + (((size_t)g * (a.ups_h + 2) + (yy + 1)) * lw2 + (xx + 1)) * 32 + (pl & 1) * 16, dst + instr * 1024);
      }
    }
  };
  // (B) hi/lo records -> 8 fp32 values, in place (same 32 bytes)
  auto lr_convert = [&](char* lr) {
    for (int r = pw * 64 + lane; r < 2 * U::RECS; r += HS_UPS_WAVES * 64) {
      const int g = r / U::RECS, rec = r - g * U::RECS;
      char* p0 = lr + ((g * 2 + 0) * U::RECS + rec) * 16;     // hi plane -> channels 0..3
      char* p1 = lr + ((g * 2 + 1) * U::RECS + rec) * 16;     // lo plane -> channels 4..7
      HsRec rc;
      rc.hi = *reinterpret_cast<const h8v*>(p0);
      rc.lo = *reinterpret_cast<const h8v*>(p1);
      float v[8];
      hs_unpack(rc, v);
      *reinterpret_cast<f32x4*>(p0) = (f32x4){v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(p1) = (f32x4){v[4], v[5], v[6], v[7]};
    }
  };
  // (C) interpolate the (TH+2) x (TW+2) halo of one up-sampled chunk from the fp32 window into the hi / lo planes of a
  // stage: per halo pixel four source offsets and two weights, then loads + packed-fp32 FMAs + the hi/lo split.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto lr_interpolate = [&](const Tile& T, const char* lr, char* lstage) {
    int r_lo, c_lo;
    lr_origin(T, r_lo, c_lo);
    for (int p = pw * 64 + lane; p < G::PLANE; p += HS_UPS_WAVES * 64) {
      const int hy_ = p / G::LW, hx_ = p - hy_ * G::LW;
      const int y = T.y0 - 1 + hy_, x = T.x0 - 1 + hx_;
      const bool inside = (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
      int o00 = 0, o01 = 0, o10 = 0, o11 = 0;
      float ly = 0.f, lx = 0.f, hy = 0.f, hx = 0.f;          // outside the image: all weights zero (the zero border)
      if (inside) {
        const float fy = a.ups_sy * y, fx = a.ups_sx * x;
        const int yq0 = (int)fy, xq0 = (int)fx;
        const int yq1 = yq0 + (yq0 < a.ups_h - 1 ? 1 : 0), xq1 = xq0 + (xq0 < a.ups_w - 1 ? 1 : 0);
        ly = fy - yq0;
        lx = fx - xq0;
        hy = 1.f - ly;
        hx = 1.f - lx;
        o00 = ((yq0 - r_lo) * U::LRW + (xq0 - c_lo)) * 16;
        o01 = ((yq0 - r_lo) * U::LRW + (xq1 - c_lo)) * 16;
        o10 = ((yq1 - r_lo) * U::LRW + (xq0 - c_lo)) * 16;
        o11 = ((yq1 - r_lo) * U::LRW + (xq1 - c_lo)) * 16;
      }
      const f32x2 wy0 = {hy, hy}, wy1 = {ly, ly}, wx0 = {hx, hx}, wx1 = {lx, lx};
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // 4 channels at a time, two packed-fp32 lanes each
          const char* tg = lr + (g * 2 + q) * U::RECS * 16;
          const f32x4 v00 = *reinterpret_cast<const f32x4*>(tg + o00);
          const f32x4 v01 = *reinterpret_cast<const f32x4*>(tg + o01);
          const f32x4 v10 = *reinterpret_cast<const f32x4*>(tg + o10);
          const f32x4 v11 = *reinterpret_cast<const f32x4*>(tg + o11);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x2 a00 = {v00[2 * e], v00[2 * e + 1]}, a01 = {v01[2 * e], v01[2 * e + 1]};
            const f32x2 a10 = {v10[2 * e], v10[2 * e + 1]}, a11 = {v11[2 * e], v11[2 * e + 1]};
            // ATen's association: hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11)
            const f32x2 top = wx0 * a00 + wx1 * a01;
            const f32x2 bot = wx0 * a10 + wx1 * a11;
            const f32x2 o = wy0 * top + wy1 * bot;
            const h2 hh = {(_Float16)o[0], (_Float16)o[1]};
            hi[q * 2 + e] = __builtin_bit_cast(unsigned, hh);
            lo[q * 2 + e] = hs_lo_pair(hi[q * 2 + e], a.neg_one, o[0], o[1]);
          }
        }
        *reinterpret_cast<u32x4*>(lstage + ((g * 2 + 0) * G::PLANE + p) * 16) = (u32x4){hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(lstage + ((g * 2 + 1) * G::PLANE + p) * 16) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
      }
    }
  };
  // which (tile, chunk) runs k steps ahead of (cur, ch); false if none / not an up-sampled chunk.  Only k = 3 in a tile's
  // last step reaches the next tile's first up-sampled chunk (chunks 0 and 1 of a tile come from in0).
  int lr_i = 0;   // step index mod 3

  int tile = blockIdx.x / nx;   // j of this workgroup's first step
  if (!valid(tile)) return;
  Tile cur = decode(tile);
  int ch = 0, stage = 0;
  if constexpr (FIRST) xwin_fetch(cur);
  if (!producer) {
    const char* src = chunk_src(cur, 0);
    const char* w = chunk_w(cur, 0);
#pragma unroll
    for (int sl = 0; sl < G::NS; ++sl) issue_slot(sl, src, w, lds);
  } else if (2 < nch && chunk_is_up(2)) {
    lr_issue(cur, 2, lr_base + 2 * U::BYTES);   // the window of step 2 (its (A) would have been in step -1)
  }
  bool stores_behind = false;   // the last VMEM operations of this wave are the record stores of a finished tile
  const bool pool_on = (EPI == EPI_ACT) && (MBW == 32) && (NBW >= 2) && (a.pool_out != nullptr);
  while (true) {
    int ntile = tile, nchk = ch + 1;
    bool has_next = true;
    Tile nxt = cur;
    if (nchk == nch) {   // tile boundary: the only place the walk is decoded
      nchk = 0;
      ntile = tile + nslot;
      has_next = valid(ntile);
      if (has_next) nxt = decode(ntile);
    }
    // this step's operands have landed (every wave waits for its own DMA, then the barrier publishes them); stores
    // of the tile finished in the previous step stay in flight
    mark(1);
    if (EPI != EPI_OUTC && stores_behind) {
      if (pool_on)
        wait_vmcnt<G::NST + G::NST_POOL>();
      else
        wait_vmcnt<G::NST>();
    } else {
      wait_vmcnt<0>();
    }
    mark(2);
    __builtin_amdgcn_s_barrier();
    mark(3);
    if constexpr (FIRST) {
      // the counted wait + barrier above published this tile's fp32 input window (its 4-byte LDS-DMA was issued one tile ago,
      // before that tile's stores).  Compute the 32-channel halo from it (nobody reads this stage: its last readers passed
      // the previous barrier), then start the next tile's window.
      gen_halo(cur, lds + stage * G::STAGE);
      __syncthreads();
      if (has_next) xwin_fetch(nxt);
    }
    if (producer) {
      const int i0 = lr_i, i1 = lr_i == 2 ? 0 : lr_i + 1, i2 = lr_i == 0 ? 2 : lr_i - 1;   // s % 3, (s+1) % 3, (s+2) % 3
      // (A) window of step s+3
      if (ch + 3 < nch) {
        if (chunk_is_up(ch + 3)) lr_issue(cur, ch + 3, lr_base + i0 * U::BYTES);
      } else if (ch == nch - 1 && has_next && 2 < nch && chunk_is_up(2)) {
        lr_issue(nxt, 2, lr_base + i0 * U::BYTES);
      }
      // (B) window of step s+2: chunk ch+2 of this tile (chunks 0 / 1 of the next tile are not up-sampled)
      if (ch + 2 < nch && chunk_is_up(ch + 2)) lr_convert(lr_base + i2 * U::BYTES);
      // (C) halo of step s+1
      if (ch + 1 < nch && chunk_is_up(ch + 1)) lr_interpolate(cur, lr_base + i1 * U::BYTES, lds + (stage ^ 1) * G::STAGE);
      if (!has_next) break;
      tile = ntile;
      ch = nchk;
      cur = nxt;
      stage ^= 1;
      lr_i = i1;
      continue;
    }
    next_halo_by_dma = !(has_next && chunk_is_up(nchk));
    if constexpr (EPI == EPI_DMASK) {
      if (ch == 0) fetch_masks(cur);
    }
    if (!has_next) {
      body(std::integral_constant<int, 0>{}, stage, nullptr, nullptr);
    } else {
      body(std::integral_constant<int, 1>{}, stage, chunk_src(nxt, nchk), chunk_w(nxt, nchk));
    }
    mark(4);
    stores_behind = false;
    if (ch == nch - 1) {
#ifdef PNPX_TUNING      // ablation (invalid results): accumulators kept alive, no epilogue at all
      if (a.abl & 2) {
#pragma unroll
        for (int m = 0; m < G::MTB; ++m)
#pragma unroll
          for (int n = 0; n < NBW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[m][n][r]));
      } else
#endif
      {
        epilogue(cur);
        stores_behind = true;
      }
      zero_acc();
      mark(5);
    }
    if (!has_next) break;
    tile = ntile;
    ch = nchk;
    cur = nxt;
    stage ^= 1;
  }
#ifdef PNPX_TUNING
  if (a.wgt && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.wgt[2 * blockIdx.x + 1] = wall_clock64();
  }
#endif
  // half-split range guard: a stored hi half overflowed f16 (|value| >= 4095) or was NaN
  if constexpr (EPI != EPI_OUTC) {
    if (a.range_flag) {
      const bool bad = (range_chk[0] != range_chk[0]) || (range_chk[1] != range_chk[1]);
      if (bad) __hip_atomic_fetch_or(a.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// hipFuncSetAttribute once per (instance, device); thread-safe
inline int ensure_dyn_lds(const void* func, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, bool> done;
  int dev = 0;
  PNPX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto key = std::make_pair(func, dev);
  if (done.count(key)) return PNPX_OK;
  PNPX_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done[key] = true;
  return PNPX_OK;
}

template <int MT, int NBW, int MBW, int NW, int EPI, int UPS = 0, int WREG = 0, int TAPS = 0x1FF>
static int launch_hs_cfg(const ConvHsArgs& a0, int B, hipStream_t s) {
  using G = HsGeom<MT, NBW, MBW, NW, WREG, TAPS>;
  constexpr int LDS_REQ = UPS ? G::LDS_USED + 3 * HsUpsGeom<MBW, NW * NBW>::BYTES : G::LDS_BYTES;   // UPS: window pool behind the used part
  static_assert(G::LDS_USED + UPS * 3 * HsUpsGeom<MBW, NW * NBW>::BYTES <= 160 * 1024, "no LDS room for the low-resolution windows");
  PNPX_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&conv_hs_kernel<MT, NBW, MBW, NW, EPI, UPS, WREG, TAPS>), LDS_REQ));
  ConvHsArgs a = a0;
  a.tilesX = (a.W + G::TW - 1) / G::TW;
  a.tilesY = (a.H + G::TH - 1) / G::TH;
  a.B = B;
  const long long ntiles = (long long)a.nct * a.tilesX * a.tilesY * B;
  a.div_nct = hs_fastdiv(a.nct);
  a.div_tx = hs_fastdiv(a.tilesX);
  a.div_ty = hs_fastdiv(a.tilesY);
  {
    const long long dmax = std::max<long long>(a.nct, std::max(a.tilesX, a.tilesY));
    if ((ntiles + 8 * 256) * dmax >= (1LL << 32)) {
      set_error("conv_hs: %lld tiles exceed the range of the tile-walk decoder", ntiles);
      return PNPX_ERR_SHAPE;
    }
  }
  // persistent: one workgroup per CU (the LDS request makes a second one impossible), each walks ntiles/grid tiles
  long long grid = 256;
  int lds_req = LDS_REQ;
#ifdef PNPX_TUNING   // co-residency experiments (tools/micro/coresident_check.py): PNPX_HS_PERCU=2 drops the LDS padding
  if (const char* e = getenv("PNPX_HS_PERCU")) {
    const int per_cu = atoi(e);
    if (per_cu > 1 && per_cu * G::LDS_USED <= 160 * 1024) {
      grid = 256LL * per_cu;
      lds_req = G::LDS_USED + UPS * 3 * HsUpsGeom<MBW, NW * NBW>::BYTES;
    }
  }
#endif
  // A launch whose tiles fit one round gets exactly one workgroup per tile and the plain walk (tile j = blockIdx.x: consecutive
  // workgroups = consecutive XCDs take consecutive cout tiles of a region, so an XCD's L2 still sees only nct / 8-ths of the weights).
  // The XCD-grouped walk wants 8 | grid and nct | grid / 8; rounding a small grid down to that made two rounds out of one
  // (192 tiles on 128 workgroups at B = 6, 144 on 128 at B = 9: 85 instead of 44 us per launch at the 32 x 32 level), and an
  // unrounded grouped walk leaves a ragged second round.  Larger launches run on all 256 workgroups, where both conditions hold.
  if (grid >= ntiles) {
    grid = ntiles;
    a.plain_walk = 1;
  } else {
    if (grid >= 8) grid -= grid % 8;          // whole XCD groups (the kernel falls back to a plain walk otherwise)
    const bool grouped = (long long)a.tilesX * a.tilesY * a.B >= 32;
    if (grouped && a.nct > 1 && (grid / 8) % a.nct != 0 && grid >= 8 * a.nct) grid -= grid % (8 * a.nct);
  }
#ifdef PNPX_TUNING   // PNPX_HS_WGT=<file>: per-workgroup start / end stamps of every launch (tools/wg_spread.py)
  static unsigned long long* wbuf = nullptr;
  const char* wfile = getenv("PNPX_HS_WGT");
  hipEvent_t wev0 = nullptr, wev1 = nullptr;
  if (wfile) {
    if (!wbuf) PNPX_HIP(hipMalloc(&wbuf, 2 * 256 * 8));
    PNPX_HIP(hipMemsetAsync(wbuf, 0, 2 * 256 * 8, s));
    a.wgt = wbuf;
    PNPX_HIP(hipEventCreate(&wev0));
    PNPX_HIP(hipEventCreate(&wev1));
    PNPX_HIP(hipEventRecord(wev0, s));
  }
#endif
#ifdef HS_TRACE
  static unsigned long long* tbuf = nullptr;
  const char* tfile = getenv("PNPX_HS_TRACE");
  if (tfile) {
    if (!tbuf) PNPX_HIP(hipMalloc(&tbuf, 4096 * 8));
    PNPX_HIP(hipMemsetAsync(tbuf, 0, 4096 * 8, s));
    a.trace = tbuf;
  }
#endif
  hipLaunchKernelGGL((conv_hs_kernel<MT, NBW, MBW, NW, EPI, UPS, WREG, TAPS>), dim3((unsigned)grid),
                     dim3((NW + HS_UPS_WAVES * UPS) * 64), lds_req, s, a);
  PNPX_LAUNCH_CHECK();
#ifdef PNPX_TUNING
  if (wfile) {
    PNPX_HIP(hipEventRecord(wev1, s));
    std::vector<unsigned long long> host(512);
    PNPX_HIP(hipStreamSynchronize(s));
    PNPX_HIP(hipMemcpy(host.data(), wbuf, 512 * 8, hipMemcpyDeviceToHost));
    float evms = 0.f;
    PNPX_HIP(hipEventElapsedTime(&evms, wev0, wev1));
    (void)hipEventDestroy(wev0);
    (void)hipEventDestroy(wev1);
    if (FILE* f = fopen(wfile, "a")) {
      fprintf(f, "conv_hs<%d,%d,%d,%d,e%d> W=%d H=%d B=%d G=%d nct=%d grid=%lld tiles=%lld event_us=%.1f |", MT, NBW, MBW, NW, EPI,
              a.W, a.H, B, a.G0 + a.G1, a.nct, grid, ntiles, evms * 1e3);
      for (long long i = 0; i < grid; ++i) fprintf(f, " %llu:%llu", host[2 * i], host[2 * i + 1]);
      fprintf(f, "\n");
      fclose(f);
    }
  }
#endif
#ifdef HS_TRACE
  if (tfile) {
    static std::vector<unsigned long long> host(4096);
    PNPX_HIP(hipStreamSynchronize(s));
    PNPX_HIP(hipMemcpy(host.data(), tbuf, 4096 * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(tfile, "a")) {
      fprintf(f, "# conv_hs<%d,%d,%d,%d,e%d> W=%d H=%d B=%d G=%d nct=%d grid=%lld\n", MT, NBW, MBW, NW, EPI, a.W, a.H, B,
              a.G0 + a.G1, a.nct, grid);
      unsigned long long prev = 0;
      for (int i = 0; i < 4096 && host[i]; ++i) {
        const unsigned long long t = host[i] >> 4;
        fprintf(f, "%d:%llu ", (int)(host[i] & 15), prev ? t - prev : 0ull);
        prev = t;
      }
      fprintf(f, "\n");
      fclose(f);
    }
  }
#endif
  return PNPX_OK;
}

struct HsChoice {
  int nbw, nw;
};

template <int MT, int MBW, int EPI, int TAPS = 0x1FF>
static int launch_hs_mbw(const ConvHsArgs& a, int B, HsChoice c, hipStream_t s) {
  if (c.nw == 8) {
    if (c.nbw >= 2) return launch_hs_cfg<MT, 2, MBW, 8, EPI, 0, 0, TAPS>(a, B, s);
    return launch_hs_cfg<MT, 1, MBW, 8, EPI, 0, 0, TAPS>(a, B, s);
  }
  if (c.nbw == 4) return launch_hs_cfg<MT, 4, MBW, 4, EPI, 0, 0, TAPS>(a, B, s);
  if (c.nbw == 2) return launch_hs_cfg<MT, 2, MBW, 4, EPI, 0, 0, TAPS>(a, B, s);
  return launch_hs_cfg<MT, 1, MBW, 4, EPI, 0, 0, TAPS>(a, B, s);
}

HsChoice hs_choose(int mt, const ConvHsArgs& a, int B);   // conv_hs.hip

template <int MT, int EPI, int TAPS = 0x1FF>
static int launch_hs_mt(const ConvHsArgs& a, int B, hipStream_t s) {
  const int mbw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : 8);
  const HsChoice c = hs_choose(MT, a, B);
  if (mbw == 32) return launch_hs_mbw<MT, 32, EPI, TAPS>(a, B, c, s);
  if (mbw == 16) return launch_hs_mbw<MT, 16, EPI, TAPS>(a, B, c, s);
  return launch_hs_mbw<MT, 8, EPI, TAPS>(a, B, c, s);
}

// entry points of the other translation units
int launch_conv_hs_dmask(const ConvHsArgs& a, int mt, int B, hipStream_t s);   // conv_hs_bwd.hip
int launch_conv_hs_res(const ConvHsArgs& a, int mt, int B, hipStream_t s);     // conv_hs_res.hip
int launch_conv_hs_taps(const ConvHsArgs& a, int mt, int taps, int B, hipStream_t s);   // conv_hs_taps.hip (EPI_ACT)

}  // namespace pnpx
