// Winograd F(2x2, 3x3) on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for the 3x3 layers of the fp32 kernel family with cout % 32 == 0
// (option fp32_winograd; conv_mode 0 layouts: padded planar fp32 activations, common.h).
//
// Replaces the same Conv2d 3x3 + bias + LeakyReLU layers as conv3x3.hip (tfpnp/pnp/denoiser/models/unet.py:8-31) with 16 instead
// of 36 multiplications per 2x2 output tile.  The round-4 probe of the same algebra on the half-split f16 kernel lost
// (profiles/r4_winograd_probe.md): there a product costs 6 matrix-pipe clocks per k-value and the input transform (fp32 VALU plus
// the hi/lo re-split) and the LDS traffic of 16 positions (~135 KB per stage) bound the stage, not the MFMAs.  Here a product
// costs 32 clocks per k-value: a pipeline stage is 2048 clocks of matrix work against the SAME ~1000 clocks of LDS traffic and
// ~300 of VALU, so the MFMA count is what counts -- PROVIDED the side work is issued between the MFMAs (stage()).
//
// Dataflow, 64-cout tile (Cfg<64>; Cfg<32> differs in the constants only -- see Cfg).  One workgroup = 4 waves, one per SIMD;
// 64 couts x 64 tiles = a 16 x 16-pixel region; persistent, XCD-grouped walk:
//   wave (wm, wn) owns cout block wm x tile block wn and all 16 Winograd positions: 16 accumulators of 32 x 32 = all 256 AGPRs,
//   addressed by name from inline assembly (see below).
//   K is walked in chunks of 16 input channels, a chunk in four stages (position rows a = 0..3; 4 positions x 8 MFMAs per wave).
//   Per stage the transformed weights U[a][0..3] of the chunk (16 KiB, packed on the host from an fp64 transform) arrive by 16-byte
//   LDS-DMA three stages ahead (ring of 4); the raw 18 x 18 x 16-channel halo of the NEXT chunk arrives by dword LDS-DMA gather
//   during stage 0 (two buffers); V[a][0..3] = row a of B^T d B is computed by all 256 threads one stage ahead (constants +-1).
//   MFMA k-slot (lane half kg, step m) = input channel 8 kg + m of the chunk, for A and B alike, so a lane's 8 operands per position
//   are two ds_read_b128 each.  Output transform A^T M A in registers, bias + LeakyReLU, 8-byte stores.
//
// Measured (B = 48, 256 x 256, profiles/r4_fp32_winograd.md): 1.5 - 2.0x over conv3x3.hip on the 64-cout-tile layers (e.g. 256 -> 256
// at 32 x 32: 0.469 -> 0.249 ms), 1.3 - 1.6x on the 32-cout-tile (full-resolution) layers; all 27 convolutions 15.4 -> 9.2 ms
// (200 TF/s algorithmic against the 157 TF/s fp32 MFMA peak; MFMA pipe 70 - 75 % busy on the deep layers at the ~2.05 GHz the
// chip holds under this kernel).  Relative error vs the fp64 oracle 7e-7 (direct kernel: 1.1e-6; fewer, shorter sums).
#include <cmath>
#include <utility>
#include <vector>

#include "common.h"
#include "conv3x3.h"
#include "wino_common.h"

namespace pnpx {

namespace {

using namespace wino;

// Two tile shapes.  CT = couts per workgroup tile:
//   CT 64: region 16 x 16 px = 64 tiles, K chunks of 16 channels, waves 2 (cout blocks) x 2 (tile blocks), 32 MFMAs per stage
//   CT 32: region 16 x 32 px = 128 tiles, K chunks of  8 channels, waves 1 x 4 (tile blocks),             16 MFMAs per stage
// (a 32-cout tile with 16-channel chunks would need 174 KB of LDS for its 128-tile halo and V buffers).  V is 16 KiB per stage in both.
// Diagnostic builds (tools/attic/wino_ablate.sh): -DWINO_ABL=bits removes parts of a stage (results are then wrong; timing only):
// 1 the closing wait + barrier, 2 the transform, 4 the operand reads, 8 the LDS-DMA issue, 16 the MFMAs, 32 the epilogue's stores,
// 64 its bias loads, 128 its residual loads.
#ifndef WINO_ABL
#define WINO_ABL 0
#endif

template <int CT_>
struct Cfg {
  static constexpr int CT = CT_;
  static constexpr int CK = (CT == 64) ? 16 : 8;          // input channels per chunk
  static constexpr int HALVES = CK / 8;                   // 16-byte operand reads per position and operand
  static constexpr int KS = CK / 2;                       // MFMA k-steps per position
  static constexpr int NM = 4 * KS;                       // MFMAs per stage and wave
  static constexpr int RPXW = (CT == 64) ? 16 : 32;       // region width in pixels (height 16)
  static constexpr int TX = RPXW / 2, NT = 8 * TX;        // tiles per row, per region
  static constexpr int RW = RPXW + 2, RPX = 18 * RW;      // raw halo
  static constexpr int RAW_ELEMS = CK * RPX;
  static constexpr int RAW_INSTR = (RAW_ELEMS + 63) / 64; // dword gathers of 64 lanes: 81 / 77
  static constexpr int RAW_BYTES = RAW_INSTR * 256;
  static constexpr int RAW_PER_WAVE = (RAW_INSTR + 3) / 4;
  static_assert(RAW_INSTR % 4 == 1, "wave 0 issues one gather more than the others");
  static constexpr int UQ = 16 * CK * CT;                 // bytes of one stage's weights: 4 positions x CK x CT floats
  static constexpr int NUW = UQ / 4096;                   // 16-byte LDS-DMA instructions per wave and stage: 4 / 1
  static constexpr int VQ = 16384, NU = 4;
  static constexpr int OFF_U = 0, OFF_V = NU * UQ, OFF_RAW = OFF_V + 2 * VQ;
  static constexpr int LDS_USED = OFF_RAW + 2 * RAW_BYTES;   // 139776 / 88576
};
constexpr int LDS_REQ = 160 * 1024;                 // the whole CU (see conv_hs_kernel.h: no LDS-using neighbours)

// accumulator IDX (0..15) = AGPRs 16 * IDX .. 16 * IDX + 15, by name
#define WINO_MFMA(IDX, x, y) \
  asm volatile("v_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(x), "v"(y), "n"(16 * (IDX)), "n"(16 * (IDX) + 15))
#define WINO_MFMA_FROM_ZERO(IDX, x, y) \
  asm volatile("v_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, 0" ::"v"(x), "v"(y), "n"(16 * (IDX)), "n"(16 * (IDX) + 15))

template <int CT, bool FUSE_OUTC, bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_kernel(WinoArgs a) {
  static_assert(!FUSE_OUTC || CT == 32, "the fused out-conv needs all 32 couts of a pixel in one wave");
  using C = Cfg<CT>;
  constexpr int CK = C::CK, HALVES = C::HALVES, KS = C::KS, TX = C::TX, NT = C::NT, RW = C::RW, RPX = C::RPX;
  constexpr int RAW_ELEMS = C::RAW_ELEMS, RAW_BYTES = C::RAW_BYTES, RAW_PER_WAVE = C::RAW_PER_WAVE;
  constexpr int UQ = C::UQ, NUW = C::NUW, VQ = C::VQ, OFF_U = C::OFF_U, OFF_V = C::OFF_V, OFF_RAW = C::OFF_RAW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HpWp = a.Hp * a.Wp;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
  const int nregions = a.rx * a.ry * a.B;
  const int nx = (gridDim.x % 8 == 0 && nregions >= 32) ? 8 : 1;       // XCD-grouped walk as conv_hs (siblings share the halo in L2)
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx, nslot = gridDim.x / nx;

  struct Tile {
    int ct, b, x0, y0;
    const float* s0;   // halo origin in channel 0 of the first source
    const float* s1;   // ... of the second source, pre-offset by -C0 channels
    const float* w;
    int ok, pad_;      // (ints, no tail padding: a struct copy with padding bytes goes through scratch)
  };
  auto decode = [&](int k) {
    Tile T;
    const int j = slot + nslot * k;
    const int q = j / a.nct;
    T.ct = j - q * a.nct;
    const int reg = nx * q + xcd;
    T.ok = reg < nregions;
    T.pad_ = 0;
    const int t1 = reg / a.rx;
    const int tx = reg - t1 * a.rx;
    const int t2 = t1 / a.ry;
    const int ty = t1 - t2 * a.ry;
    T.b = t2;
    T.x0 = tx * C::RPXW;
    T.y0 = ty * 16;
    const size_t pix = (size_t)T.y0 * a.Wp + T.x0 + (PADL - 1);
    T.s0 = a.in0 + (size_t)T.b * a.C0 * HpWp + pix;
    T.s1 = a.in1 + ((long long)T.b * a.C1 - a.C0) * (long long)HpWp + (long long)pix;
    T.w = a.u + (size_t)T.ct * a.nch * 4 * (UQ / 4);
    return T;
  };

  // byte offsets of this lane's halo elements from the chunk's origin: unsigned 32-bit, so the gather is one SGPR base + one VGPR each
  unsigned roff[RAW_PER_WAVE];
#pragma unroll
  for (int k = 0; k < RAW_PER_WAVE; ++k) {
    const int idx = (wave + 4 * k) * 64 + lane;
    const int c = idx / RPX, r = idx - c * RPX;
    const int hy = r / RW, hx = r - hy * RW;
    roff[k] = (idx < RAW_ELEMS) ? 4u * (unsigned)(c * HpWp + hy * a.Wp + hx) : 0u;
  }
  auto issue_raw = [&](const float* s0, const float* s1, int c, int rbuf) {
    const float* src = ((CK * c < a.C0) ? s0 : s1) + (size_t)CK * c * HpWp;
    const unsigned dst = lds0 + OFF_RAW + rbuf * RAW_BYTES + wave * 256;
#pragma unroll
    for (int k = 0; k < RAW_PER_WAVE; ++k)
      if (k < RAW_PER_WAVE - 1 || wave == 0) glds4(src, roff[k], dst + k * 1024);
  };
  const unsigned uoff = lane * 16u;
  auto issue_u = [&](const float* w, int c, int aa) {
    const float* s = w + ((size_t)c * 4 + aa) * (UQ / 4) + wave * 256;
#pragma unroll
    for (int k = 0; k < NUW; ++k) glds16(s + k * 1024, uoff, lds0 + OFF_U + aa * UQ + (wave + 4 * k) * 1024);
  };

  // transform role: one tile and 4 channels of the chunk per thread.  CT 64: tile = lane, channels 8 * tkg + 4 * thf .. + 3 by wave;
  // CT 32: tile = tid & 127, channels 4 * tkg .. + 3 with tkg = tid >> 7
  // LDS banks: the halo reads are 8 bytes per lane at dword (2 tty) * RW + 2 ttx (+ row, + channel plane); the 16 lanes the LDS
  // serves together must cover 16 distinct bank pairs.  CT 32: one tile row (ttx 0..15) = 16 lanes.  CT 64 (8 tiles per row, row
  // step 36 dwords = 2 pairs mod 16): tile rows t and t + 4 are 8 pairs apart, so lanes 0-7 / 8-15 take rows (0, 4), 16-31 (1, 5), ...
  // (with tile = lane the rows of a group overlapped in 6 of 8 pairs: the counters showed half of all LDS cycles as conflicts).
  const int tt = (CT == 64) ? ((lane & 7) + 8 * (4 * ((lane >> 3) & 1) + (lane >> 4))) : (tid & 127);
  const int tkg = (CT == 64) ? (wave & 1) : (wave >> 1), thf = (CT == 64) ? (wave >> 1) : 0;
  const int tty = tt / TX, ttx = tt % TX;
  const int t_rd = (((tkg * (CK / 2) + thf * 4) * RPX) + (2 * tty) * RW + 2 * ttx) * 4;
  const int t_wr = (((tkg * HALVES + thf) * NT) + tt) * 16;          // + b * 4096

  const int wm = (CT == 64) ? (wave & 1) : 0, wn = (CT == 64) ? (wave >> 1) : wave;
  const int l31 = lane & 31, kg = lane >> 5;
  const int a_lane = ((kg * HALVES) * CT + wm * 32 + l31) * 16;      // + half * CT * 16 + b * UQ / 4
  const int b_lane = ((kg * HALVES) * NT + wn * 32 + l31) * 16;      // + half * NT * 16 + b * 4096
  // The 16 accumulators (position (A, b) = AGPRs 16 * (4 A + b) .. + 15, all 256) are addressed by NAME from inline assembly.
  // As C++ values the register allocator split their live ranges across VGPRs and scratch (161 spills, reloads inside every
  // stage); by name they never move.  The compiler must not use AGPRs itself: it has no reason to while the kernel's VGPR
  // demand stays far below 256 (167 here; check after a change: no v_accvgpr_* outside the ASM blocks of the ISA, no spills),
  // and the clobber below makes the kernel descriptor allocate all 256 (.agpr_count in the code object's metadata).
  asm volatile("" ::: "a0", "a255");

  // the work of one transform (row TA of B^T d B for this thread's tile and 4 channels -> V[vbuf][0..3]), cut into 12 slices that
  // stage() drops between its MFMAs: 4 x (8 halo reads), 4 x (4 adds), 4 x (4 adds + one 16-byte write)
  struct Tr {
    float d[2][4][4];   // [row RA / RB][channel][x]
    float r[4][4];      // [x][channel]
  };
  auto transform_slice = [&](auto ta_tag, int sl, Tr& t, int rbuf, int vbuf) {
    constexpr int A = decltype(ta_tag)::value;
    constexpr int RA = (A == 0) ? 0 : (A == 2 ? 2 : 1);
    constexpr int RB = (A == 0) ? 2 : (A == 1 ? 2 : (A == 2 ? 1 : 3));
    if (sl < 4) {
      const int e = sl;
      const char* rb = lds + OFF_RAW + rbuf * RAW_BYTES + t_rd;
#pragma unroll
      for (int x = 0; x < 4; x += 2) {      // 8-byte reads (the dword index is even: RPX, RW and 2 ttx are)
        const f32x2 da = *reinterpret_cast<const f32x2*>(rb + (e * RPX + RA * RW + x) * 4);
        const f32x2 db = *reinterpret_cast<const f32x2*>(rb + (e * RPX + RB * RW + x) * 4);
        t.d[0][e][x] = da[0];
        t.d[0][e][x + 1] = da[1];
        t.d[1][e][x] = db[0];
        t.d[1][e][x + 1] = db[1];
      }
    } else if (sl < 8) {
      const int e = sl - 4;
#pragma unroll
      for (int x = 0; x < 4; ++x)      // rows of B^T: d0 - d2, d1 + d2, d2 - d1, d1 - d3
        t.r[x][e] = (A == 1) ? t.d[0][e][x] + t.d[1][e][x] : t.d[0][e][x] - t.d[1][e][x];
    } else {
      const int bb = sl - 8;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = (bb == 0) ? t.r[0][e] - t.r[2][e] : (bb == 1) ? t.r[1][e] + t.r[2][e] : (bb == 2) ? t.r[2][e] - t.r[1][e] : t.r[1][e] - t.r[3][e];
      *reinterpret_cast<f32x4*>(lds + OFF_V + vbuf * VQ + t_wr + bb * 4096) = v;
    }
  };
  // the whole transform at once (prologue only)
  auto transform = [&](auto ta_tag, int rbuf, int vbuf) {
    Tr t;
#pragma unroll
    for (int sl = 0; sl < 12; ++sl) transform_slice(ta_tag, sl, t, rbuf, vbuf);
  };

  // One pipeline stage = position row A: 32 MFMAs per wave (4 positions x 8 k-steps), issued one at a time with a slice of the
  // stage's other work behind each so that the matrix pipe (64 clocks per MFMA) never waits for the wave's VALU / LDS / DMA issue:
  //   before MFMA 0     the operands MFMA 0 / 1 need; the rest of the stage's 16 operand reads and the 16 halo reads of the NEXT
  //                     stage's transform (row TA) follow in the first 10 slots, the transform's arithmetic and V writes in the
  //                     next 8, then this stage's weight slice and (stage 0) the next chunk's halo by LDS-DMA  -- see side()
  // The fences pin that order; left alone the scheduler puts all the side work first and the MFMAs in one block behind it.
  // (No run-time conditions in here: a stage has to stay ONE basic block, or the compiler peels and unswitches the chunk loop into
  //  versions with a branch behind every MFMA, and the first chunk of every tile ran ~6x slower than the others.)
  auto sync_all = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // end of a stage: this wave's LDS-DMA for the NEXT stage (and, at a = 2, the next chunk's halo) has landed -- a counted wait:
  // N = VMEM operations issued after it (younger weight slices, halo pieces, the previous tile's stores) that may stay in flight
  auto sync_counted = [&](auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    if (WINO_ABL & 1) return;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __syncthreads();
  };
  // MFMA operands of the stage in flight: [position][8-channel half].  Kernel scope: the first reads of stage s + 1 (positions 0, 1)
  // are issued by stage s right behind its barrier, which stands a few MFMAs BEFORE the stage's end -- those MFMAs cover the LDS
  // latency that would otherwise sit between the barrier and the next stage's first MFMA (measured time-neutral against the barrier
  // at the very end: the head latency was not what the stage waits for).
  f32x4 af[4][HALVES], bf[4][HALVES];
  auto read_operand = [&](int A, int b, int h) {
    if (WINO_ABL & 4) {
      af[b][h] = bf[b][h] = (f32x4){1.f, 1.f, 1.f, 1.f};
      return;
    }
    af[b][h] = *reinterpret_cast<const f32x4*>(lds + OFF_U + A * UQ + a_lane + b * (UQ / 4) + h * (CT * 16));
    bf[b][h] = *reinterpret_cast<const f32x4*>(lds + OFF_V + (A & 1) * VQ + b_lane + b * 4096 + h * (NT * 16));
  };
  struct Side {
    int t_rbuf, t_vbuf;    // transform row TA of halo buffer t_rbuf into V[t_vbuf]
    const float* u_src;    // weight slice u_src -> ring slot u_slot
    int u_slot;
    const float* raw_src;  // RAW stages: halo of the chunk at raw_src -> halo buffer raw_buf
    int raw_buf;
  };
  auto stage = [&](auto a_tag, auto ta_tag, auto raw_tag, auto zero_tag, auto wait_tag, const Side& sd) {
    constexpr int A = decltype(a_tag)::value;
    constexpr bool RAW = decltype(raw_tag)::value;
    constexpr bool ZERO = decltype(zero_tag)::value;   // first chunk of a tile: the first MFMA of each position starts from C = 0
    Tr t;
    auto operand = [&](int b, int h) { read_operand(A, b, h); };
    auto tslice = [&](int q) {
      if (!(WINO_ABL & 2)) transform_slice(ta_tag, q, t, sd.t_rbuf, sd.t_vbuf);
    };
    auto dma_u = [&]() {
      if (WINO_ABL & 8) return;
      const float* s = sd.u_src + wave * 256;
#pragma unroll
      for (int k = 0; k < NUW; ++k) glds16(s + k * 1024, uoff, lds0 + OFF_U + sd.u_slot * UQ + (wave + 4 * k) * 1024);
    };
    auto dma_raw = [&](int k0, int k1) {
      if (RAW && !(WINO_ABL & 8)) {
        const unsigned dst = lds0 + OFF_RAW + sd.raw_buf * RAW_BYTES + wave * 256;
#pragma unroll
        for (int k = k0; k < k1; ++k)
          if (k < RAW_PER_WAVE - 1 || (k == RAW_PER_WAVE - 1 && wave == 0)) glds4(sd.raw_src, roff[k], dst + k * 1024);
      }
    };
    // The side work behind MFMA number sl.  LDS reads (the rest of the stage's 16 operand reads, 16 halo reads per lane) are spread
    // over the first third of the stage, then the transform's arithmetic and V writes, then the DMA issue; the stage's barrier
    // stands behind slot BAR (everything the NEXT stage reads is complete by then: this stage's V writes and the LDS-DMA the
    // counted wait covers; nothing reads this stage's U / V any more: the remaining MFMAs have their operands in registers).
    constexpr int BAR = (CT == 64) ? 27 : 12;
    auto side = [&](int sl) {
      if (CT == 64) {
        if (sl == 0) operand(0, 1);
        else if (sl == 1) operand(1, 1);
        else if (sl < 10) {
          const int q = sl - 2;                      // even: halo reads of channel q / 2; odd: one operand pair of positions 2, 3
          if ((q & 1) == 0) tslice(q >> 1);
          else operand(2 + (q >> 2), (q >> 1) & 1);
        } else if (sl < 18) tslice(4 + (sl - 10));
        else if (sl == 18) dma_u();
        else if (sl <= BAR) dma_raw(3 * (sl - 19), 3 * (sl - 19) + 3);
      } else {
        if (sl == 0) tslice(0);
        else if (sl == 1) operand(2, 0);
        else if (sl == 2) tslice(1);
        else if (sl == 3) operand(3, 0);
        else if (sl == 4) tslice(2);
        else if (sl == 5) tslice(3);
        else if (sl < 10) {
          tslice(4 + 2 * (sl - 6));
          tslice(5 + 2 * (sl - 6));
          dma_raw(4 * (sl - 6), 4 * (sl - 6) + 4);
        } else if (sl == 10) dma_u();
        else if (sl == 11) dma_raw(16, RAW_PER_WAVE);
      }
      if (sl == BAR) {
        sync_counted(wait_tag);
        read_operand((A + 1) & 3, 0, 0);
        read_operand((A + 1) & 3, 1, 0);
      }
    };
    // (the operands of the first k-steps -- positions 0, 1, first 8-channel half -- were read by the previous stage behind its barrier)
#pragma unroll
    for (int pair = 0; pair < 2; ++pair)
#pragma unroll
      for (int m = 0; m < KS; ++m)
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
          const int b = 2 * pair + bl;
          const int sl = pair * (2 * KS) + m * 2 + bl;
          {
            const float x = af[b][m >> 2][m & 3], y = bf[b][m >> 2][m & 3];
            const bool from_zero = ZERO && m == 0;
            if (!(WINO_ABL & 16)) switch (b) {
              case 0: if (from_zero) WINO_MFMA_FROM_ZERO(4 * A + 0, x, y); else WINO_MFMA(4 * A + 0, x, y); break;
              case 1: if (from_zero) WINO_MFMA_FROM_ZERO(4 * A + 1, x, y); else WINO_MFMA(4 * A + 1, x, y); break;
              case 2: if (from_zero) WINO_MFMA_FROM_ZERO(4 * A + 2, x, y); else WINO_MFMA(4 * A + 2, x, y); break;
              default: if (from_zero) WINO_MFMA_FROM_ZERO(4 * A + 3, x, y); else WINO_MFMA(4 * A + 3, x, y); break;
            }
          }
          side(sl);
          __builtin_amdgcn_sched_barrier(0);
        }
  };

  auto epilogue = [&](const Tile& T) {
    const int tile = wn * 32 + l31, ty = tile / TX, tx = tile % TX;
    const int cbase = T.ct * CT + wm * 32 + 4 * kg;              // C layout: row = (r & 3) + 8 * (r >> 2) + 4 * kg
    float* ob = a.out + ((size_t)T.b * a.Cout) * HpWp + (size_t)(T.y0 + 2 * ty + 1) * a.Wp + T.x0 + 2 * tx + PADL;
    const int Hp_pool = padded_h(a.H / 2), Wp_pool = padded_w(a.W / 2);
    const size_t HpWp_pool = (size_t)Hp_pool * Wp_pool;
    float* pb = a.pool ? a.pool + (size_t)T.b * a.Cout * HpWp_pool + (size_t)(T.y0 / 2 + ty + 1) * Wp_pool + T.x0 / 2 + tx + PADL : nullptr;
    // the last MFMAs were issued before the stage's closing barrier; their results must have left the matrix pipe before an
    // AGPR read (a software-visible hazard on gfx950 that the compiler cannot see through the named registers)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = (WINO_ABL & 64) ? 0.f : a.bias[cbase + (r & 3) + 8 * (r >> 2)];
    // RES instances (the DRUNet's ResBlock skip): all 32 residual pairs up front -- loaded per row behind a run-time condition they
    // cost sixteen serialised load latencies per tile
    f32x2 rs0[RES ? 16 : 1], rs1[RES ? 16 : 1];
    if (RES && !(WINO_ABL & 128)) {
      const float* rb = a.res + (ob - a.out);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* rp = rb + (size_t)(cbase + (r & 3) + 8 * (r >> 2)) * HpWp;
        rs0[r] = *reinterpret_cast<const f32x2*>(rp);
        rs1[r] = *reinterpret_cast<const f32x2*>(rp + a.Wp);
      }
    }
    float ow[16];
    float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;       // FUSE_OUTC: this lane's 16 channels of the 1x1 out-conv, per pixel
    if (FUSE_OUTC) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ow[r] = a.outc_w[cbase + (r & 3) + 8 * (r >> 2)];
    }
    static_for<16>([&](auto r_tag) {
      constexpr int R = decltype(r_tag)::value;
      float t0[4], t1[4];
      static_for<4>([&](auto i_tag) {
        constexpr int I = decltype(i_tag)::value;
        float m0, m1, m2, m3;
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m0) : "n"(16 * (4 * I + 0) + R));
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m1) : "n"(16 * (4 * I + 1) + R));
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m2) : "n"(16 * (4 * I + 2) + R));
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m3) : "n"(16 * (4 * I + 3) + R));
        t0[I] = m0 + m1 + m2;
        t1[I] = m1 - m2 - m3;
      });
      const int co = cbase + (R & 3) + 8 * (R >> 2);
      const float bias = bias_r[R];
      float y00 = t0[0] + t0[1] + t0[2] + bias, y10 = t0[1] - t0[2] - t0[3] + bias;
      float y01 = t1[0] + t1[1] + t1[2] + bias, y11 = t1[1] - t1[2] - t1[3] + bias;
      y00 = y00 > 0.f ? y00 : y00 * a.slope;
      y01 = y01 > 0.f ? y01 : y01 * a.slope;
      y10 = y10 > 0.f ? y10 : y10 * a.slope;
      y11 = y11 > 0.f ? y11 : y11 * a.slope;
      float* o = ob + (size_t)co * HpWp;
      if (RES && !(WINO_ABL & 128)) {
        y00 += rs0[RES ? R : 0][0];
        y01 += rs0[RES ? R : 0][1];
        y10 += rs1[RES ? R : 0][0];
        y11 += rs1[RES ? R : 0][1];
      }
      if (FUSE_OUTC) {
        s00 = fmaf(ow[R], y00, s00);
        s01 = fmaf(ow[R], y01, s01);
        s10 = fmaf(ow[R], y10, s10);
        s11 = fmaf(ow[R], y11, s11);
      } else if (!(WINO_ABL & 32) || y00 == 1.2345e-33f) {
        *reinterpret_cast<f32x2*>(o) = (f32x2){y00, y01};
        *reinterpret_cast<f32x2*>(o + a.Wp) = (f32x2){y10, y11};
        if (a.pool) pb[(size_t)co * HpWp_pool] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));   // same association as maxpool2_kernel
      }
    });
    if (FUSE_OUTC) {
      // the other 16 channels sit in lane ^ 32; channels 0-3, 8-11, ... here (kg 0), 4-7, 12-15, ... there
      s00 += __shfl_xor(s00, 32);
      s01 += __shfl_xor(s01, 32);
      s10 += __shfl_xor(s10, 32);
      s11 += __shfl_xor(s11, 32);
      const size_t px0 = ((size_t)T.b * a.H + T.y0 + 2 * ty) * a.W + T.x0 + 2 * tx;
      const float ob0 = a.outc_b[0];
      // both stores by every lane (kg 0 and 1 write identical values): the stage waits count exactly NST store instructions
      const f32x2 xi0 = *reinterpret_cast<const f32x2*>(a.x_img + px0), xi1 = *reinterpret_cast<const f32x2*>(a.x_img + px0 + a.W);
      const float v00 = xi0[0] + (s00 + ob0), v01 = xi0[1] + (s01 + ob0), v10 = xi1[0] + (s10 + ob0), v11 = xi1[1] + (s11 + ob0);
      *reinterpret_cast<f32x2*>(a.img_pre + px0) = (f32x2){v00, v01};
      *reinterpret_cast<f32x2*>(a.img_pre + px0 + a.W) = (f32x2){v10, v11};
      *reinterpret_cast<f32x2*>(a.img + px0) = (f32x2){fminf(fmaxf(v00, 0.f), 1.f), fminf(fmaxf(v01, 0.f), 1.f)};
      *reinterpret_cast<f32x2*>(a.img + px0 + a.W) = (f32x2){fminf(fmaxf(v10, 0.f), 1.f), fminf(fmaxf(v11, 0.f), 1.f)};
    }
  };

  constexpr int NST = FUSE_OUTC ? 4 : 32;  // store instructions per wave and tile (a fused max-pool adds 16: the waits then err on the safe side)
  constexpr int NRAW = RAW_PER_WAVE - 1;   // halo gathers per wave (wave 0 issues one more: counted conservatively)

  int k = 0;
  Tile T = decode(0);
  if (!T.ok) return;
  issue_raw(T.s0, T.s1, 0, 0);
  issue_u(T.w, 0, 0);
  issue_u(T.w, 0, 1);
  issue_u(T.w, 0, 2);
  sync_all();
  transform(std::integral_constant<int, 0>{}, 0, 0);
  sync_all();
  read_operand(0, 0, 0);
  read_operand(0, 1, 0);

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using Yes = std::true_type;
  using No = std::false_type;
  // One chunk of 16 input channels = four stages.  FIRST: the chunk follows an epilogue, whose NST stores sit in the VMEM queue
  // between the weight slices of the previous stage and this one's.  "n" names the chunk after this one (in the next tile after a
  // tile's last chunk; after the very last chunk of the walk it is this tile's chunk 0 again: the loads land in buffers nobody reads,
  // which costs one chunk of DMA per workgroup and keeps every stage free of conditions).
  auto chunk = [&](auto first_tag, const float* w_c, const float* nu, const float* nraw, int rcur) {
    constexpr int EPI = decltype(first_tag)::value ? NST : 0;
    const int rnext = rcur ^ 1;
    stage(I0{}, I1{}, Yes{}, first_tag, std::integral_constant<int, NUW + EPI + NUW + NRAW>{}, Side{rcur, 1, w_c + 3 * (UQ / 4), 3, nraw, rnext});
    stage(I1{}, I2{}, No{}, first_tag, std::integral_constant<int, EPI + NUW + NRAW + NUW>{}, Side{rcur, 0, nu, 0, nullptr, 0});
    stage(I2{}, I3{}, No{}, first_tag, std::integral_constant<int, 2 * NUW>{}, Side{rcur, 1, nu + (UQ / 4), 1, nullptr, 0});
    stage(I3{}, I0{}, No{}, first_tag, std::integral_constant<int, 2 * NUW>{}, Side{rnext, 0, nu + 2 * (UQ / 4), 2, nullptr, 0});
  };
  auto raw_of = [&](const Tile& X, int c) { return ((CK * c < a.C0) ? X.s0 : X.s1) + (size_t)CK * c * HpWp; };

  int cc = 0;
  for (;;) {
    Tile Tn = decode(k + 1);
    const bool more = Tn.ok;
    if (!more) Tn = T;
    for (int c = 0; c < a.nch; ++c, ++cc) {
      const bool last_c = (c == a.nch - 1);
      const float* w_c = T.w + (size_t)c * 4 * (UQ / 4);
      const float* nu = last_c ? Tn.w : w_c + 4 * (UQ / 4);
      const float* nraw = last_c ? raw_of(Tn, 0) : raw_of(T, c + 1);
      if (c == 0) chunk(Yes{}, w_c, nu, nraw, cc & 1);
      else chunk(No{}, w_c, nu, nraw, cc & 1);
    }
    epilogue(T);
    if (!more) break;
    T = Tn;
    ++k;
  }
  sync_all();   // the surplus loads of the last chunk
}

}  // namespace

// cout tile of a layer: 64 where it divides, else 32; 0 = the direct kernel (conv3x3.hip) runs the layer
static int wino_ct(int C0, int C1, int cout, int H, int W) {
  if (C0 <= 0 || H < 16 || H % 16 != 0) return 0;
  if (cout % 64 == 0) return (C0 % 16 == 0 && C1 % 16 == 0 && W % 16 == 0) ? 64 : 0;   // (the packing follows cout alone)
  if (cout % 32 == 0) return (C0 % 8 == 0 && C1 % 8 == 0 && W % 32 == 0) ? 32 : 0;
  return 0;
}
bool conv3x3_wino_ok(int C0, int C1, int cout, int H, int W) { return wino_ct(C0, C1, cout, H, W) != 0; }

bool conv3x3_wino_packs(int cout, int cin) { return (cout % 64 == 0 && cin % 16 == 0) || (cout % 32 == 0 && cin % 8 == 0); }

size_t conv3x3_wino_floats(int cout, int cin) { return (size_t)cout * cin * 16; }

// w[cout][cin][3][3] -> U = G g G^T in fp64 -> [cout/CT][cin/CK][a][b][kg 2][half][m CT][4] fp32, CT = 64 (CK 16) where 64 divides
// cout, else 32 (CK 8): the tile shape is a property of the layer, so one packing per layer
void pack_conv_weights_wino(const float* w, int cout, int cin, float* dst) {
  static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  const int CT = (cout % 64 == 0) ? 64 : 32, CK = (CT == 64) ? 16 : 8, HALVES = CK / 8;
  const int nct = cout / CT, nch = cin / CK;
  for (int ct = 0; ct < nct; ++ct)
    for (int ch = 0; ch < nch; ++ch)
      for (int m = 0; m < CT; ++m)
        for (int kgi = 0; kgi < 2; ++kgi)
          for (int hf = 0; hf < HALVES; ++hf)
            for (int e = 0; e < 4; ++e) {
              const int co = ct * CT + m, ci = ch * CK + kgi * (CK / 2) + hf * 4 + e;
              const float* g = w + ((size_t)co * cin + ci) * 9;
              for (int aa = 0; aa < 4; ++aa)
                for (int bb = 0; bb < 4; ++bb) {
                  double s = 0;
                  for (int k2 = 0; k2 < 3; ++k2)
                    for (int l = 0; l < 3; ++l) s += G[aa][k2] * (double)g[k2 * 3 + l] * G[bb][l];
                  const size_t o = ((((((size_t)ct * nch + ch) * 4 + aa) * 4 + bb) * 2 + kgi) * HALVES + hf) * CT * 4 + (size_t)m * 4 + e;
                  dst[o] = (float)s;
                }
            }
}

template <int CT, bool FUSE_OUTC, bool RES>
static int launch_wino(WinoArgs a, hipStream_t s) {
  using C = Cfg<CT>;
  a.nct = a.Cout / CT;
  a.nch = (a.C0 + a.C1) / C::CK;
  a.rx = a.W / C::RPXW;
  a.ry = a.H / 16;
  static std::once_flag attr_once[64];      // (host threads of different contexts launch concurrently)
  int dev = 0;
  PNPX_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64) {
    hipError_t e = hipSuccess;
    std::call_once(attr_once[dev], [&] {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32_kernel<CT, FUSE_OUTC, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REQ);
    });
    PNPX_HIP(e);
  }
  const long long ntiles = (long long)a.rx * a.ry * a.B * a.nct;
  long long grid = 256;
  if (grid >= ntiles) grid = ntiles;
  else if ((grid / 8) % a.nct != 0 && grid >= 8LL * a.nct) grid -= grid % (8 * a.nct);
  hipLaunchKernelGGL((conv3x3_wino_f32_kernel<CT, FUSE_OUTC, RES>), dim3((unsigned)grid), dim3(256), LDS_REQ, s, a);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

// the layer can run with the UNet's out-conv fused into it (conv3x3_wino_outc)
bool conv3x3_wino_outc_ok(int cin, int cout, int H, int W) { return cout == 32 && wino_ct(cin, 0, cout, H, W) == 32; }

int launch_conv3x3_wino_outc(const float* u, const float* bias, const float* in0, int cin, const float* outc_w, const float* outc_b,
                             const float* x_img, float* img, float* img_pre, int B, int H, int W, hipStream_t s) {
  if (!conv3x3_wino_outc_ok(cin, 32, H, W)) {
    set_error("conv3x3_wino_outc: unsupported geometry (%d -> 32 channels, %d x %d)", cin, H, W);
    return PNPX_ERR_SHAPE;
  }
  WinoArgs a{};
  a.in0 = in0;
  a.in1 = in0;
  a.u = u;
  a.bias = bias;
  a.out = nullptr;
  a.outc_w = outc_w;
  a.outc_b = outc_b;
  a.x_img = x_img;
  a.img = img;
  a.img_pre = img_pre ? img_pre : img;
  a.B = B;
  a.H = H;
  a.W = W;
  a.Hp = padded_h(H);
  a.Wp = padded_w(W);
  a.C0 = cin;
  a.C1 = 0;
  a.Cout = 32;
  a.slope = 0.2f;
  return launch_wino<32, true, false>(a, s);
}

int launch_conv3x3_wino(const float* u, const float* bias, int cout, const float* in0, int C0, const float* in1, int C1,
                        float* out, int B, int H, int W, hipStream_t s, float slope, const float* res, float* pool_out) {
  const int ct = wino_ct(C0, C1, cout, H, W);
  if (!ct) {
    set_error("conv3x3_wino: unsupported geometry (%d + %d -> %d channels, %d x %d)", C0, C1, cout, H, W);
    return PNPX_ERR_SHAPE;
  }
  WinoArgs a{};
  a.in0 = in0;
  a.in1 = in1 ? in1 : in0;
  a.u = u;
  a.bias = bias;
  a.res = res;
  a.pool = pool_out;
  a.out = out;
  a.B = B;
  a.H = H;
  a.W = W;
  a.Hp = padded_h(H);
  a.Wp = padded_w(W);
  a.C0 = C0;
  a.C1 = C1;
  a.Cout = cout;
  a.slope = slope;
  if (res) return ct == 64 ? launch_wino<64, false, true>(a, s) : launch_wino<32, false, true>(a, s);
  return ct == 64 ? launch_wino<64, false, false>(a, s) : launch_wino<32, false, false>(a, s);
}

}  // namespace pnpx
