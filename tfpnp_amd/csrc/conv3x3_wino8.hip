// Winograd F(2x2, 3x3) on the exact-fp32 MFMA, EIGHT waves per workgroup: the 16 Winograd positions of an output block are split
// over the two waves of a SIMD (round 5).  Same layers, same packed weights (pack_conv_weights_wino), same activation layout and the
// same algebra as conv3x3_wino.hip; what changes is who holds the accumulators.
//
// Why.  conv3x3_wino.hip gives one wave all 16 positions of its 32-cout x 32-tile block = 256 accumulator registers, which leaves ONE
// wave per SIMD: whenever that wave waits (operand latency at the head of a stage, an LDS write behind its reads, the stage's
// barrier) the matrix pipe idles, and the round-4 counters show it 25-30 % idle on the deep layers and > 50 % on the full-resolution
// ones (profiles/r4_denoiser_pmc_fp32.md, r4_fp32_winograd.md: transform -18 %, barrier -10 %, operand reads -8 % exposed).  The 16
// positions are independent GEMMs, so they can be split without reading any operand twice: wave p of a SIMD's pair takes position
// COLUMNS b = 2p, 2p + 1 (all four rows a) = 8 accumulators = 128 AGPRs, two waves per SIMD, and one wave's side work and waits hide
// behind the other's MFMAs.  The price is the output transform A^T M A, which needs all 16 positions of a tile: each wave reduces
// its columns over the rows (S = A^T M, two values per column), the pair swaps half of S through LDS once per tile (32 floats per
// lane each way) and each wave finishes 8 of the block's 16 accumulator rows (= couts).
//
// Roles.  p = 0 waves run the input transform (CT 64; with the 32-cout tile both do), p = 1 waves issue every LDS-DMA (weights,
// halo).  A role is a compile-time parameter of the whole tile loop (one branch at the top), so that every pipeline stage stays one
// basic block with a fixed instruction order.
//
// Tile shapes (Cfg8): CT 64 = 64 couts x 16 x 16 px, 16-channel chunks, one position row per stage (4 stages per chunk, 16 MFMAs per
// wave and stage = 32 per SIMD between barriers, as before); CT 32 = 32 couts x 16 x 32 px, 8-channel chunks, TWO position rows per
// stage (2 stages per chunk; the 4-wave kernel had 16 MFMAs per SIMD between barriers on these full-resolution layers, now 32).
#include <cmath>
#include <utility>
#include <vector>

#include "common.h"
#include "conv3x3.h"
#include "wino_common.h"

namespace pnpx {

namespace {

using namespace wino;

template <int CT_>
struct Cfg8 {
  static constexpr int CT = CT_;
  static constexpr int CK = (CT == 64) ? 16 : 8;          // input channels per chunk
  static constexpr int HALVES = CK / 8;                   // 16-byte operand reads per position and operand
  static constexpr int KS = CK / 2;                       // MFMA k-steps per position
  static constexpr int RS = (CT == 64) ? 1 : 2;           // position rows per pipeline stage
  static constexpr int NSTG = 4 / RS;                     // stages per chunk
  static constexpr int NPOS = 2 * RS;                     // positions per wave and stage (its two columns of the stage's rows)
  static constexpr int NM = NPOS * KS;                    // MFMAs per wave and stage
  static_assert(NM == 16, "the side-work schedule below is written for 16 MFMA slots per stage");
  static constexpr int RPXW = (CT == 64) ? 16 : 32;       // region width in pixels (height 16)
  static constexpr int TX = RPXW / 2, NT = 8 * TX;        // tiles per row, per region
  static constexpr int RW = RPXW + 2, RPX = 18 * RW;      // raw halo
  // channel-plane stride of the halo in LDS.  CT 64: 328 instead of 324 -- the transform's lanes alternate between two channel pairs
  // (2 planes apart), and 2 * 328 dwords = 8 bank pairs (mod 16) puts the second pair's reads on the banks the first leaves free
  static constexpr int RPXL = (CT == 64) ? 328 : RPX;
  static constexpr int RAW_ELEMS = CK * RPXL;
  static constexpr int RAW_INSTR = (RAW_ELEMS + 63) / 64; // dword gathers of 64 lanes: 82 / 77
  static constexpr int RAW_BYTES = RAW_INSTR * 256;
  static constexpr int RAW_PER_WAVE = (RAW_INSTR + 7) / 8;   // every wave issues its eighth (instruction wave + 8 k): 11 / 10 ...
  static constexpr int NRAW = RAW_INSTR / 8;                 // ... of which the last exists for the first RAW_INSTR % 8 waves only: >= 10 / 9
  static constexpr int UQ = 16 * CK * CT;                 // bytes of one position row of weights: 4 positions x CK x CT floats
  static constexpr int USTG = RS * UQ;                    // ... of one stage: 16 KiB / 8 KiB
  static constexpr int NUW = USTG / 8192;                 // 16-byte LDS-DMA instructions per wave and stage: 2 / 1
  static constexpr int VROW = 16384;                      // bytes of one position row of V: 4 positions x CK x NT floats (both shapes)
  static constexpr int VSTG = RS * VROW;
  static constexpr int OFF_U = 0, OFF_V = 4 * USTG, OFF_RAW = OFF_V + 2 * VSTG;   // U: ring of four stage slots
  static constexpr int LDS_USED = OFF_RAW + 2 * RAW_BYTES;   // 140288 / 137728
  static constexpr int BAR = 12;                          // MFMA slot the stage's barrier stands behind
  // UPS instances: the low-resolution window of a chunk's halo (NR x NQP source pixels per channel, 16-byte granules of 4 channels:
  // [quad][row][column][4]) is staged by LDS-DMA two (CT 32: three) chunks ahead and interpolated into the halo buffer one chunk ahead
  static constexpr int NR = 11, NQP = (CT == 64) ? 12 : 20;     // 18 (34) output pixels span <= 10 (18) source pixels + 1
  static constexpr int QPLANE = NR * NQP * 16;               // bytes per channel quad
  static constexpr int ST_ELEMS = CK * NR * NQP;
  static constexpr int ST_INSTR = (ST_ELEMS + 63) / 64;      // 33 / 28
  static constexpr int ST_STRIDE = ST_INSTR * 256;           // bytes per staging buffer: 8448 / 7168
  static constexpr int ST_PER_WAVE = (ST_INSTR + 7) / 8, NSW = ST_INSTR / 8;   // 5, >= 4 / 4, >= 3
  static constexpr int OFF_ST = LDS_USED, OFF_TAB = OFF_ST + 2 * ST_STRIDE;     // tables: 18 row + RW column entries of 16 bytes
  static constexpr int LDS_USED_UPS = OFF_TAB + (18 + RW) * 16;
  static constexpr int NPOSI = 18 * RW;                      // halo pixels per channel: 324 / 612
};
constexpr int LDS_REQ8 = 160 * 1024;                // the whole CU (see conv_hs_kernel.h: no LDS-using neighbours)

// all 128 accumulator registers: named as clobbers of every MFMA statement they keep the register allocator out of a0 .. a127 for
// the whole tile loop (left alone it parks ordinary values there -- the unified register file of gfx950 lets it -- and the MFMAs
// then accumulate on top of them)
#define WINO8_ACC \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
  "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", \
  "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", \
  "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", \
  "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", \
  "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", \
  "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", \
  "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
// accumulator IDX (0..7) = AGPRs 16 * IDX .. 16 * IDX + 15, by name (see conv3x3_wino.hip)
// -DWINO8_PRIO=n (experiment): the wave raises its issue priority around every MFMA, so that the SIMD's arbiter prefers a ready MFMA over
// the other wave's VALU / LDS instruction
#ifdef WINO8_PRIO
#define WINO8_PRIO_UP "s_setprio 3\n\t"
#define WINO8_PRIO_DOWN "\n\ts_setprio 0"
#else
#define WINO8_PRIO_UP
#define WINO8_PRIO_DOWN
#endif
#define WINO8_MFMA(IDX, x, y) \
  asm volatile(WINO8_PRIO_UP "v_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, a[%2:%3]" WINO8_PRIO_DOWN ::"v"(x), "v"(y), "n"(16 * (IDX)), "n"(16 * (IDX) + 15) : WINO8_ACC)
#define WINO8_MFMA_FROM_ZERO(IDX, x, y) \
  asm volatile(WINO8_PRIO_UP "v_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, 0" WINO8_PRIO_DOWN ::"v"(x), "v"(y), "n"(16 * (IDX)), "n"(16 * (IDX) + 15) : WINO8_ACC)

// Diagnostic builds (-DWINO8_TRACE, tools/trace_wino8.py): waves 0 and 4 of workgroup 0 stamp s_memtime behind every stage's barrier and
// at the epilogue's phases into a device array the tool reads back through pnpx_debug_wino8_trace.
// -DWINO8_ABL=bits (timing only, results wrong): 1 no transform, 2 operand reads replaced by constants, 4 no LDS-DMA issue, 8 no stage wait +
// barrier, 16 no MFMAs
#ifndef WINO8_ABL
#define WINO8_ABL 0
#endif
#ifdef WINO8_TRACE
__device__ unsigned long long wino8_trace_buf[2][2048];
__device__ __forceinline__ void wino8_stamp(int wave, int lane, int& n, int tag) {
  if (blockIdx.x == 0 && (wave & 3) == 0 && n < 2047) {
    const unsigned long long t = __builtin_readcyclecounter();
    if (lane == 0) wino8_trace_buf[wave >> 2][n] = (t << 8) | (unsigned)tag;
    n = __builtin_amdgcn_readfirstlane(n + 1);
    if (lane == 0) wino8_trace_buf[wave >> 2][2047] = n;
  }
}
#define WINO8_STAMP(tag) wino8_stamp(wave, lane, trace_n, tag)
#else
#define WINO8_STAMP(tag)
#endif

template <bool RAWDMA_, bool STDMA_, int IQ_>
struct StageKind {
  static constexpr bool RAWDMA = RAWDMA_, STDMA = STDMA_;
  static constexpr int IQ = IQ_;
};

using I0 = std::integral_constant<int, 0>;
using I1 = std::integral_constant<int, 1>;
using I2 = std::integral_constant<int, 2>;
using I3 = std::integral_constant<int, 3>;
using Yes = std::true_type;
using No = std::false_type;

template <int CT, bool FUSE_OUTC, bool RES, bool UPS, bool KSPLIT>
__global__ __launch_bounds__(512, 1) void conv3x3_wino8_f32_kernel(WinoArgs a) {
  static_assert(!FUSE_OUTC || CT == 32, "the fused out-conv needs all 32 couts of a pixel in one SIMD's wave pair");
  static_assert(!UPS || (!FUSE_OUTC && !RES), "the up-sampling instances are the UNet's plain decoder entries");
  static_assert(!KSPLIT || (CT == 64 && !FUSE_OUTC && !RES && !UPS), "K-split: the plain 64-cout instance (the deep levels' layers)");
  using C = Cfg8<CT>;
  constexpr int CK = C::CK, HALVES = C::HALVES, KS = C::KS, RS = C::RS, NSTG = C::NSTG, NPOS = C::NPOS;
  constexpr int TX = C::TX, NT = C::NT, RW = C::RW, RPX = C::RPX, RPXL = C::RPXL;
  constexpr int RAW_BYTES = C::RAW_BYTES, RAW_PER_WAVE = C::RAW_PER_WAVE;
  constexpr int UQ = C::UQ, USTG = C::USTG, NUW = C::NUW, VROW = C::VROW, VSTG = C::VSTG;
  constexpr int OFF_U = C::OFF_U, OFF_V = C::OFF_V, OFF_RAW = C::OFF_RAW, BAR = C::BAR;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WINO8_TRACE
  int trace_n = 0;
#endif
  const int sw = wave & 3;                       // the pair's index (waves sw and sw + 4 share a cout block x tile block)
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
    const float* s1u;  // UPS: origin of the low-resolution window in the second source, pre-offset by -C0 channels
    int ok, rlo;       // (ints, no tail padding: a struct copy with padding bytes goes through scratch)
    int clo, c0;       // c0 (KSPLIT): first channel chunk of this work item
    int item, g;       // KSPLIT: tile number (scratch slab) and piece
  };
  const int hpwp_lo = UPS ? (a.ups_h + 2) * (a.ups_w + 2 * PADL) : 0;
  auto decode = [&](int k) {
    Tile T;
    const int j = slot + nslot * k;
    // KSPLIT: the pieces of a tile are `ksplit` consecutive groups of nct items (v = piece * nct + cout tile): they run side by side
    // on one XCD, beside the sibling cout tiles that share their halo
    const int vnct = KSPLIT ? a.nct * a.ksplit : a.nct;
    const int q = j / vnct;
    const int v = j - q * vnct;
    T.g = KSPLIT ? v / a.nct : 0;
    T.ct = KSPLIT ? v - T.g * a.nct : v;
    T.c0 = KSPLIT ? T.g * a.nch : 0;
    const int reg = nx * q + xcd;
    T.ok = reg < nregions;
    T.item = reg * a.nct + T.ct;
    T.rlo = T.clo = 0;
    T.s1u = nullptr;
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
    T.w = a.u + ((size_t)T.ct * (KSPLIT ? a.nch_all : a.nch) + T.c0) * 4 * (UQ / 4);
    if (UPS) {
      // window of source pixels: anchored at the source pixel of the halo's first row / column, pulled back so that its NR rows and
      // NQP columns stay inside the padded low-resolution tensor (whose padding is zero: the only out-of-image taps have weight 0)
      const int ya = T.y0 > 0 ? T.y0 - 1 : 0, xa = T.x0 > 0 ? T.x0 - 1 : 0;
      const int rl = (int)(a.ups_sy * (float)ya), cl = (int)(a.ups_sx * (float)xa);
      const int rmax = a.ups_h + 1 - C::NR, cmax = a.ups_w + PADL - C::NQP;
      T.rlo = rl < rmax ? rl : rmax;
      T.clo = cl < cmax ? cl : cmax;
      T.s1u = a.in1 + ((long long)T.b * a.C1 - a.C0) * (long long)hpwp_lo + (long long)(T.rlo + 1) * (a.ups_w + 2 * PADL) + T.clo + PADL;
    }
    return T;
  };
  auto raw_of = [&](const Tile& X, int c) {
    const int ca = KSPLIT ? c + X.c0 : c;     // (a piece's prefetch beyond its last chunk names the next item's first chunks, as at a tile's end)
    return ((CK * ca < a.C0) ? X.s0 : X.s1) + (size_t)CK * ca * HpWp;
  };

  const int wm = (CT == 64) ? (sw & 1) : 0, wn = (CT == 64) ? (sw >> 1) : sw;
  const int l31 = lane & 31, kg = lane >> 5;
  const int a_lane = ((kg * HALVES) * CT + wm * 32 + l31) * 16;      // + half * CT * 16 + b * UQ / 4 (+ row, + slot)
  const int b_lane = ((kg * HALVES) * NT + wn * 32 + l31) * 16;      // + half * NT * 16 + b * 4096   (+ row, + buffer)
  // The 8 accumulators of a wave (position (row A, column 2 p + bl) = AGPRs 16 * (2 A + bl) .. + 15, 128 in all) are addressed by
  // name; the clobber makes the kernel descriptor allocate them and keeps the compiler's own values out (two waves per SIMD:
  // 128 AGPRs + at most 128 VGPRs each -- tests/test_build_invariants.py checks the code object for spills and stray v_accvgpr).
  asm volatile("" ::: WINO8_ACC);

  auto wait_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto barrier = [&]() { __builtin_amdgcn_s_barrier(); };
  auto sync_all = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ------------------------------------------------------------------------------------------------------------------------
  auto body = [&](auto role_tag) {
    constexpr int P = decltype(role_tag)::value;          // this wave's position columns: 2 P, 2 P + 1

    // -- LDS-DMA, an eighth per wave: per-lane byte offsets of the halo elements from the chunk's origin (one SGPR base + one VGPR each)
    unsigned roff[RAW_PER_WAVE];
#pragma unroll
    for (int k = 0; k < RAW_PER_WAVE; ++k) {
      const int idx = (wave + 8 * k) * 64 + lane;
      const int c = idx / RPXL, r = idx - c * RPXL;
      const int hy = r / RW, hx = r - hy * RW;
      roff[k] = (c < CK && r < RPX) ? 4u * (unsigned)(c * HpWp + hy * a.Wp + hx) : 0u;
    }
    const unsigned uoff = lane * 16u;
    auto dma_u = [&](const float* src, int uslot) {     // one stage's weights (USTG bytes) -> ring slot uslot
      if (WINO8_ABL & 4) return;
      const float* s = src + wave * 256;
#pragma unroll
      for (int k = 0; k < NUW; ++k) glds16(s + k * 2048, uoff, lds0 + OFF_U + uslot * USTG + (wave + 8 * k) * 1024);
    };
    auto dma_raw = [&](const float* src, int rbuf, int k0, int k1) {
      if (WINO8_ABL & 4) return;
      const unsigned dst = lds0 + OFF_RAW + rbuf * RAW_BYTES + wave * 256;
#pragma unroll
      for (int k = k0; k < k1; ++k)
        if (k < RAW_PER_WAVE) {
          if (k < C::NRAW || wave + 8 * k < C::RAW_INSTR) glds4(src, roff[k], dst + k * 2048);
        }
    };

    // -- UPS: staging DMA of a chunk's low-resolution window, an eighth per wave (offsets are tile-independent: no clamping, see decode)
    constexpr int NR = C::NR, NQP = C::NQP, QPLANE = C::QPLANE, ST_STRIDE = C::ST_STRIDE, SPW = C::ST_PER_WAVE;
    constexpr int OFF_ST = C::OFF_ST, OFF_TAB = C::OFF_TAB, NPOSI = C::NPOSI;
    unsigned soff[UPS ? SPW : 1];
    if (UPS) {
      const int wp_lo = a.ups_w + 2 * PADL;
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
        const int idx = (wave + 8 * k) * 64 + lane;
        const int e = idx & 3, g = idx >> 2;
        const int q = g % NQP, g2 = g / NQP;
        const int rr = g2 % NR, Q = g2 / NR;
        soff[k] = (Q < CK / 4) ? 4u * (unsigned)((4 * Q + e) * hpwp_lo + rr * wp_lo + q) : 0u;
      }
    }
    auto dma_st = [&](const float* src, int sbuf) {
      if (!UPS || (WINO8_ABL & 4)) return;
      const unsigned dst = lds0 + OFF_ST + sbuf * ST_STRIDE + wave * 256;
#pragma unroll
      for (int k = 0; k < SPW; ++k)
        if (k < C::NSW || wave + 8 * k < C::ST_INSTR) glds4(src, soff[UPS ? k : 0], dst + k * 2048);
    };
    // per-tile tables of the interpolation: entry = (byte offset into a staging plane, weight of the nearer tap h, of the farther tap l, 0);
    // rows 0..17 of the halo, then its RW columns.  Same arithmetic as upsample2x_v4_kernel (unet.hip): f = s * index, i0 = (int) f, l = f - i0,
    // h = 1 - l.  Pixels outside the image (the convolution's zero padding) get weights 0.
    auto write_tables = [&](const Tile& X) {
      if (!UPS) return;
      if (tid < 18 + RW) {
        const bool row = tid < 18;
        const int i = row ? tid : tid - 18;
        const int g = (row ? X.y0 : X.x0) - 1 + i;                      // image coordinate of the halo row / column
        const bool valid = g >= 0 && g < (row ? a.H : a.W);
        const float f = (row ? a.ups_sy : a.ups_sx) * (float)g;
        const int i0 = (int)f;
        const float l = f - (float)i0, h = 1.f - l;
        const int off = row ? (i0 - X.rlo) * NQP * 16 : (i0 - X.clo) * 16;
        f32x4 ent;
        ent[0] = __int_as_float(valid ? off : 0);
        ent[1] = valid ? h : 0.f;
        ent[2] = valid ? l : 0.f;
        ent[3] = 0.f;
        *reinterpret_cast<f32x4*>(lds + OFF_TAB + tid * 16) = ent;
      }
    };
    // interpolation of one channel quad of a staged chunk into a halo buffer: one halo pixel per thread and pass (CT 64: 324 pixels,
    // threads beyond repeat the last one; CT 32: 612 pixels = a full pass + a second one of the p = 0 waves), 6 slices per pass:
    // 0 table entries, 1 the four taps (16 bytes = 4 channels each), 2..5 one channel each: v = hy (hx v00 + lx v01) + ly (hx v10 + lx v11)
    struct Ip {
      float hy, ly, hx, lx;
      int base;
      f32x4 v00, v01, v10, v11;
    };
    const int ipos = tid < NPOSI ? tid : NPOSI - 1;                      // pass A: halo pixel `tid`, the quad's 4 channels
    // pass B (CT 32: 612 pixels > 512 threads): the remaining 100 pixels x 4 channels, one output per thread
    const int ipos_b = (512 + (tid >> 2) < NPOSI) ? 512 + (tid >> 2) : NPOSI - 1;
    auto interp_tables = [&](int pp, Ip& ip) {
      const int hyi = pp / RW, hxi = pp - hyi * RW;
      const f32x4 rt = *reinterpret_cast<const f32x4*>(lds + OFF_TAB + hyi * 16);
      const f32x4 ct = *reinterpret_cast<const f32x4*>(lds + OFF_TAB + (18 + hxi) * 16);
      ip.hy = rt[1];
      ip.ly = rt[2];
      ip.hx = ct[1];
      ip.lx = ct[2];
      ip.base = __float_as_int(rt[0]) + __float_as_int(ct[0]);
    };
    auto interp_slice = [&](auto q_tag, int sl, Ip& ip, int sbuf, int rbuf) {
      constexpr int Q = decltype(q_tag)::value;
      if (sl == 0) {
        interp_tables(ipos, ip);
      } else if (sl == 1) {
        const char* b = lds + OFF_ST + sbuf * ST_STRIDE + Q * QPLANE + ip.base;
        ip.v00 = *reinterpret_cast<const f32x4*>(b);
        ip.v01 = *reinterpret_cast<const f32x4*>(b + 16);
        ip.v10 = *reinterpret_cast<const f32x4*>(b + NQP * 16);
        ip.v11 = *reinterpret_cast<const f32x4*>(b + NQP * 16 + 16);
      } else {
        const int e = sl - 2;
        const float o = ip.hy * (ip.hx * ip.v00[e] + ip.lx * ip.v01[e]) + ip.ly * (ip.hx * ip.v10[e] + ip.lx * ip.v11[e]);
        *reinterpret_cast<float*>(lds + OFF_RAW + rbuf * RAW_BYTES + ((4 * Q + e) * RPXL + ipos) * 4) = o;
      }
    };
    auto interp_mini = [&](auto q_tag, int sl, Ip& ip, int sbuf, int rbuf) {      // slices 0 tables, 1 taps, 2 output
      constexpr int Q = decltype(q_tag)::value;
      const int e4 = (tid & 3) * 4;
      if (sl == 0) {
        interp_tables(ipos_b, ip);
      } else if (sl == 1) {
        const char* b = lds + OFF_ST + sbuf * ST_STRIDE + Q * QPLANE + ip.base + e4;
        ip.v00[0] = *reinterpret_cast<const float*>(b);
        ip.v01[0] = *reinterpret_cast<const float*>(b + 16);
        ip.v10[0] = *reinterpret_cast<const float*>(b + NQP * 16);
        ip.v11[0] = *reinterpret_cast<const float*>(b + NQP * 16 + 16);
      } else {
        const float o = ip.hy * (ip.hx * ip.v00[0] + ip.lx * ip.v01[0]) + ip.ly * (ip.hx * ip.v10[0] + ip.lx * ip.v11[0]);
        *reinterpret_cast<float*>(lds + OFF_RAW + rbuf * RAW_BYTES + (4 * Q * RPXL + ipos_b) * 4 + e4 * RPXL) = o;
      }
    };

    // -- input transform, shared by all eight waves.  CT 32: one tile and 4 channels (one 16-byte V write per position) per thread:
    // (tile half, channel quad, row of the stage) by wave.  CT 64: one tile and TWO channels per thread, lanes alternating between the
    // two channel pairs of a quad (so a wave writes contiguous 8-byte pieces and, with RPXL, reads conflict-free: a 16-lane group =
    // one tile row x 2 pairs = 16 distinct bank pairs); (tile half, channel quad) by wave.
    constexpr int TCH = (CT == 64) ? 2 : 4;
    const int cb = (CT == 64) ? (lane & 1) : 0;
    const int tt = (CT == 64) ? ((wave >> 2) * 32 + (lane >> 1)) : ((sw & 1) * 64 + lane);
    const int tkg = (CT == 64) ? (sw & 1) : (sw >> 1), thf = (CT == 64) ? (sw >> 1) : 0;
    const int tty = tt / TX, ttx = tt % TX;
    const int t_rd = (((tkg * (CK / 2) + thf * 4 + 2 * cb) * RPXL) + (2 * tty) * RW + 2 * ttx) * 4;
    const int t_wr = (((tkg * HALVES + thf) * NT) + tt) * 16 + cb * 8;          // + b * 4096 (+ row, + buffer)
    struct Tr {
      float d[2][TCH][4];   // [row RA / RB][channel][x]
      float r[4][TCH];      // [x][channel]
    };
    constexpr int NSL = 3 * TCH;      // slices: TCH x (halo reads of one channel), TCH x (row combination), 4 x (column combination + write) -- 8 / 12
    // row TA of B^T d B for this thread's tile and channels -> V at byte offset vdst, in slices the stage drops between its MFMAs
    auto transform_slice = [&](auto ta_tag, int sl, Tr& t, int rbuf, int vdst) {
      constexpr int A = decltype(ta_tag)::value;
      constexpr int RA = (A == 0) ? 0 : (A == 2 ? 2 : 1);
      constexpr int RB = (A == 0) ? 2 : (A == 1 ? 2 : (A == 2 ? 1 : 3));
      if (sl < TCH) {
        const int e = sl;
        const char* rb = lds + OFF_RAW + rbuf * RAW_BYTES + t_rd;
#pragma unroll
        for (int x = 0; x < 4; x += 2) {
          const f32x2 da = *reinterpret_cast<const f32x2*>(rb + (e * RPXL + RA * RW + x) * 4);
          const f32x2 db = *reinterpret_cast<const f32x2*>(rb + (e * RPXL + RB * RW + x) * 4);
          t.d[0][e][x] = da[0];
          t.d[0][e][x + 1] = da[1];
          t.d[1][e][x] = db[0];
          t.d[1][e][x + 1] = db[1];
        }
      } else if (sl < 2 * TCH) {
        const int e = sl - TCH;
#pragma unroll
        for (int x = 0; x < 4; ++x)      // rows of B^T: d0 - d2, d1 + d2, d2 - d1, d1 - d3
          t.r[x][e] = (A == 1) ? t.d[0][e][x] + t.d[1][e][x] : t.d[0][e][x] - t.d[1][e][x];
      } else if (CT == 64) {
        const int bb = (sl - 2 * TCH) >> 1, e = (sl - 2 * TCH) & 1;     // one position column per two slices: channel 0, then channel 1 + the write
        (void)e;
        if (((sl - 2 * TCH) & 1) == 0) return;
        f32x2 v;
#pragma unroll
        for (int q = 0; q < 2; ++q)
          v[q] = (bb == 0) ? t.r[0][q] - t.r[2][q] : (bb == 1) ? t.r[1][q] + t.r[2][q] : (bb == 2) ? t.r[2][q] - t.r[1][q] : t.r[1][q] - t.r[3][q];
        *reinterpret_cast<f32x2*>(lds + vdst + t_wr + bb * 4096) = v;
      } else {
        const int bb = sl - 2 * TCH;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = (bb == 0) ? t.r[0][e] - t.r[2][e] : (bb == 1) ? t.r[1][e] + t.r[2][e] : (bb == 2) ? t.r[2][e] - t.r[1][e] : t.r[1][e] - t.r[3][e];
        *reinterpret_cast<f32x4*>(lds + vdst + t_wr + bb * 4096) = v;
      }
    };
    constexpr int NSLT = (CT == 64) ? 2 * TCH + 8 : NSL;      // slice indices in use: CT 64 0..11 (the write slices are the odd ones of 4..11), CT 32 0..11
    // V of stage SN (the rows this thread transforms for it): byte offset
    auto vdst_of = [&](auto sn_tag) {
      constexpr int SN = decltype(sn_tag)::value;
      return OFF_V + ((CT == 64) ? (SN & 1) * VSTG : SN * VSTG + P * VROW);
    };

    // -- MFMA operands of the stage in flight: [position = 2 * (row in stage) + bl][8-channel half]
    f32x4 af[NPOS][HALVES], bf[NPOS][HALVES];
    auto read_operand = [&](auto s_tag, int uslot, int pos, int h) {
      constexpr int S = decltype(s_tag)::value;
      constexpr int VB = (CT == 64) ? (S & 1) : S;
      const int r = pos >> 1, b = 2 * P + (pos & 1);
      if (WINO8_ABL & 2) {
        af[pos][h] = bf[pos][h] = (f32x4){1.f, 1.f, 1.f, 1.f};
        return;
      }
      af[pos][h] = *reinterpret_cast<const f32x4*>(lds + OFF_U + uslot * USTG + r * UQ + b * (UQ / 4) + a_lane + h * (CT * 16));
      bf[pos][h] = *reinterpret_cast<const f32x4*>(lds + OFF_V + VB * VSTG + r * VROW + b * 4096 + b_lane + h * (NT * 16));
    };

    struct Side {
      int t_rbuf;            // halo buffer the transform for the next stage reads
      int u_cur, u_next;     // weight-ring slots of this stage and of the next one (operand reads)
      const float* u_src;    // DMA: weights u_src -> ring slot u_dst
      int u_dst;
      const float* raw_src;  // DMA (halo stages): halo at raw_src -> halo buffer raw_buf
      int raw_buf;
      const float* st_src;   // UPS: staging DMA of the window at st_src -> staging buffer st_buf
      int st_buf;
      int i_sbuf, i_rbuf;    // UPS: interpolation from staging buffer i_sbuf into halo buffer i_rbuf
    };
    // what a stage does besides its MFMAs / transform / weight DMA: RAWDMA the halo gathers, STDMA the staging gathers, IQ >= 0 the
    // interpolation of channel quad IQ
    // (StageKind<RAWDMA, STDMA, IQ>, namespace scope)
    // One pipeline stage S: 16 MFMAs per wave, one at a time with a slice of the stage's side work behind each (conv3x3_wino.hip
    // explains why the order is pinned by fences and why a stage must stay one basic block).  Behind slot BAR: the counted wait
    // (DMA waves: N = VMEM operations issued after the ones the NEXT stage needs) + barrier, then the next stage's first operands.
    auto stage = [&](auto s_tag, auto zero_tag, auto wait_tag, auto prefetch_tag, auto kind_tag, const Side& sd) {
      using K = decltype(kind_tag);
      constexpr int S = decltype(s_tag)::value;
      constexpr bool PREFETCH = decltype(prefetch_tag)::value;   // false: the tile's last stage (the epilogue follows and wants the registers)
      constexpr int SN = (S + 1) % NSTG;
      constexpr bool ZERO = decltype(zero_tag)::value;
      constexpr int WAITN = decltype(wait_tag)::value;
      constexpr bool HALOST = (CT == 64) ? (S == 0) : (S == 1);     // the stage of a chunk that fetches halos / staging windows
      constexpr bool RAWST = HALOST && K::RAWDMA;
      constexpr bool STST = UPS && HALOST && K::STDMA;
      constexpr int IQ = UPS ? K::IQ : -1;
      using IQt = std::integral_constant<int, IQ < 0 ? 0 : IQ>;
      Ip ipa, ipb;
      auto islice = [&](int q) {
        if (IQ >= 0) interp_slice(IQt{}, q, ipa, sd.i_sbuf, sd.i_rbuf);
      };
      auto imini = [&](int q) {
        if (IQ >= 0 && CT == 32) interp_mini(IQt{}, q, ipb, sd.i_sbuf, sd.i_rbuf);
      };
      using TA = std::integral_constant<int, (CT == 64) ? SN : 2 * SN + P>;
      using SNt = std::integral_constant<int, SN>;
      Tr t;
      auto operand = [&](int pos, int h) { read_operand(s_tag, sd.u_cur, pos, h); };
      auto tslice = [&](int q) {
        if (!(WINO8_ABL & 1)) transform_slice(TA{}, q, t, sd.t_rbuf, vdst_of(SNt{}));
      };
      auto side = [&](int sl) {
        if (CT == 64) {
          // transform slices 0, 1: halo reads of the two channels; 2, 3: row combinations; 4..11: four position columns (even = nothing,
          // odd = combine + write) -- done by slot 6; the interpolation of an up-sampled chunk (tables, taps, four channels) follows in
          // slots 5..10, so that the two never hold their registers together
          if (sl == 0) operand(0, 1);
          else if (sl == 1) operand(1, 1);
          if (sl < 2) tslice(sl);
          if (sl == 2) dma_u(sd.u_src, sd.u_dst);
          if (RAWST && sl >= 3 && sl < 9) dma_raw(sd.raw_src, sd.raw_buf, 2 * (sl - 3), 2 * (sl - 3) + 2);
          if (STST && sl == 9) dma_st(sd.st_src, sd.st_buf);
          if (IQ < 0) {
            if (sl == 4) tslice(2);
            if (sl == 5) tslice(3);
            if (sl >= 6 && sl < 10) tslice(5 + 2 * (sl - 6));
          } else {
            if (sl == 3) tslice(2), tslice(3);
            if (sl == 4) tslice(5), tslice(7);
            if (sl == 5) tslice(9), tslice(11), islice(0);
            if (sl == 6) islice(1);
            if (sl >= 8 && sl < 12) islice(2 + (sl - 8));
          }
        } else {
          // transform slices 0-3: halo reads of the four channels; 4-7: row combinations (a channel's right behind the next one's reads,
          // so that two channels' raw values are live at a time); 8-11: position columns + writes.  UPS: pass A of the interpolation in
          // slots 7..10 (where the stage's first operands and the raw values are dead), the single-output pass B in 9..11
          if (sl == 0) tslice(0);
          else if (sl == 1) operand(2, 0);
          else if (sl == 2) tslice(1);
          else if (sl == 3) operand(3, 0);
          else if (sl == 4) tslice(4), tslice(2);
          else if (sl == 5) tslice(5), tslice(3);
          else if (sl == 6) tslice(6), tslice(7);
          else if (sl == 7) tslice(8), tslice(9), islice(0);
          else if (sl == 8) tslice(10), tslice(11), islice(1);
          else if (sl == 9) islice(2), islice(3), imini(0);
          else if (sl == 10) islice(4), islice(5), imini(1), dma_u(sd.u_src, sd.u_dst);
          else if (sl == 11) imini(2);
          if (RAWST && sl >= 6 && sl < 11) dma_raw(sd.raw_src, sd.raw_buf, 2 * (sl - 6), 2 * (sl - 6) + 2);
          if (STST && sl == 9) dma_st(sd.st_src, sd.st_buf);
        }
        if (sl == BAR) {
          if (!(WINO8_ABL & 8)) {
            WINO8_STAMP(32 + S);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");
            WINO8_STAMP(48 + S);
            barrier();
          }
          WINO8_STAMP(S);
          if (PREFETCH) {
            read_operand(SNt{}, sd.u_next, 0, 0);
            read_operand(SNt{}, sd.u_next, 1, 0);
          }
        }
      };
      // (the operands of the first k-steps -- positions 0, 1, first 8-channel half -- were read by the previous stage behind its barrier)
#pragma unroll
      for (int pair = 0; pair < NPOS / 2; ++pair)
#pragma unroll
        for (int m = 0; m < KS; ++m)
#pragma unroll
          for (int bl = 0; bl < 2; ++bl) {
            const int pos = 2 * pair + bl;
            const int sl = pair * (2 * KS) + m * 2 + bl;
            const float x = af[pos][m >> 2][m & 3], y = bf[pos][m >> 2][m & 3];
            const bool from_zero = ZERO && m == 0;
            if (!(WINO8_ABL & 16)) switch (pos) {     // accumulator of position (row RS * S + pos / 2, column bl) = 2 * row + bl
              case 0: if (from_zero) WINO8_MFMA_FROM_ZERO(2 * (RS * S) + 0, x, y); else WINO8_MFMA(2 * (RS * S) + 0, x, y); break;
              case 1: if (from_zero) WINO8_MFMA_FROM_ZERO(2 * (RS * S) + 1, x, y); else WINO8_MFMA(2 * (RS * S) + 1, x, y); break;
              case 2: if (from_zero) WINO8_MFMA_FROM_ZERO((2 * (RS * S) + 2) & 7, x, y); else WINO8_MFMA((2 * (RS * S) + 2) & 7, x, y); break;
              default: if (from_zero) WINO8_MFMA_FROM_ZERO((2 * (RS * S) + 3) & 7, x, y); else WINO8_MFMA((2 * (RS * S) + 3) & 7, x, y); break;
            }
            side(sl);
            __builtin_amdgcn_sched_barrier(0);
          }
    };

    // -- epilogue: S = A^T M over this wave's two columns, half of it swapped with the pair's other wave, A-transform of the rows,
    // bias + activation, stores.  The p = 0 wave finishes accumulator rows R 0..7, the p = 1 wave R 8..15 of the 32 x 32 block.
    // xoff: this pair's 8 KiB of LDS nobody reads or fills during the epilogue; zoff (FUSE_OUTC): 1 KiB more.
    auto epilogue = [&](const Tile& T, int xsend, int xrecv, int zoff) {
      const int tile = wn * 32 + l31, ty = tile / TX, tx = tile % TX;
      const int cbase = T.ct * CT + wm * 32 + 4 * kg;              // C layout: row = (r & 3) + 8 * (r >> 2) + 4 * kg
      float* ob = a.out + ((size_t)T.b * a.Cout) * HpWp + (size_t)(T.y0 + 2 * ty + 1) * a.Wp + T.x0 + 2 * tx + PADL;
      const int Hp_pool = padded_h(a.H / 2), Wp_pool = padded_w(a.W / 2);
      const size_t HpWp_pool = (size_t)Hp_pool * Wp_pool;
      float* pb = a.pool ? a.pool + (size_t)T.b * a.Cout * HpWp_pool + (size_t)(T.y0 / 2 + ty + 1) * Wp_pool + T.x0 / 2 + tx + PADL : nullptr;
      constexpr int R0 = 8 * P;          // the accumulator rows this wave finishes; it sends the other eight
      constexpr int RX = 8 - R0;
      // the last MFMAs were issued behind the stage's closing barrier; their results must have left the matrix pipe before an AGPR read
      WINO8_STAMP(16);
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory", WINO8_ACC);
      float bias_r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bias_r[j] = a.bias[cbase + ((R0 + j) & 3) + 8 * ((R0 + j) >> 2)];
      // column sums of accumulator row R over the position rows A: (S0[bl 0], S0[bl 1], S1[bl 0], S1[bl 1]), S0 = (M0 + M1) + M2, S1 = (M1 - M2) - M3
      auto column_sums = [&](auto r_tag) {
        constexpr int R = decltype(r_tag)::value;
        f32x4 sv;
        static_for<2>([&](auto bl_tag) {
          constexpr int BL = decltype(bl_tag)::value;
          float m0, m1, m2, m3;
          asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m0) : "n"(16 * (0 + BL) + R) : WINO8_ACC);
          asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m1) : "n"(16 * (2 + BL) + R) : WINO8_ACC);
          asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m2) : "n"(16 * (4 + BL) + R) : WINO8_ACC);
          asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m3) : "n"(16 * (6 + BL) + R) : WINO8_ACC);
          sv[BL] = (m0 + m1) + m2;
          sv[2 + BL] = (m1 - m2) - m3;
        });
        return sv;
      };
      // the swap: rows RX .. RX + 7 of this wave's sums go to the pair's other wave, both directions at once through two 8 KiB buffers
      // per pair (xsend / xrecv: LDS nobody reads or fills during the epilogue -- see the call).  Sums are made on the fly; `theirs` is
      // the only array held.
      f32x4 theirs[8];
      {
        char* xs = lds + xsend + lane * 16;
        static_for<8>([&](auto j_tag) {
          *reinterpret_cast<f32x4*>(xs + decltype(j_tag)::value * 1024) = column_sums(std::integral_constant<int, RX + decltype(j_tag)::value>{});
        });
        wait_lds();
        barrier();      // #1
        WINO8_STAMP(17);
        const char* xr = lds + xrecv + lane * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) theirs[j] = *reinterpret_cast<const f32x4*>(xr + j * 1024);
      }
      // RES instances (the DRUNet's ResBlock skip): residual pairs in two batches of four rows (all eight up front do not fit beside
      // `theirs` in the 128 registers a wave has here)
      f32x2 rs0[RES ? 4 : 1], rs1[RES ? 4 : 1];
      auto load_res = [&](int j0) {
        const float* rb = a.res + (ob - a.out);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* rp = rb + (size_t)(cbase + ((R0 + j0 + q) & 3) + 8 * ((R0 + j0 + q) >> 2)) * HpWp;
          rs0[RES ? q : 0] = *reinterpret_cast<const f32x2*>(rp);
          rs1[RES ? q : 0] = *reinterpret_cast<const f32x2*>(rp + a.Wp);
        }
      };
      float ow[8];
      float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;       // FUSE_OUTC: this lane's 8 channels of the 1x1 out-conv, per pixel
      if (FUSE_OUTC) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ow[j] = a.outc_w[cbase + ((R0 + j) & 3) + 8 * ((R0 + j) >> 2)];
      }
      wait_lds();
      // KSPLIT: [tile][piece][wave][row][lane] f32x4
      float* const part_tile = KSPLIT ? a.part + ((size_t)T.item * a.ksplit * 8 + wave) * 2048 + lane * 4 : nullptr;
      float* const part_mine = KSPLIT ? part_tile + (size_t)T.g * (8 * 2048) : nullptr;
      static_for<8>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int R = R0 + j;
        if (RES && (j & 3) == 0) load_res(j);
        const int co = cbase + (R & 3) + 8 * (R >> 2);
        // columns 0, 1 from the p = 0 wave, 2, 3 from the p = 1 wave: Y[i][0] = (S[i][0] + S[i][1]) + S[i][2], Y[i][1] = (S[i][1] - S[i][2]) - S[i][3]
        const f32x4 mine = column_sums(std::integral_constant<int, R>{});
        const f32x4 lo = (P == 0) ? mine : theirs[j], hi = (P == 0) ? theirs[j] : mine;
        const float bias = bias_r[j];
        float y00 = (lo[0] + lo[1]) + hi[0], y01 = (lo[1] - hi[0]) - hi[1];
        float y10 = (lo[2] + lo[3]) + hi[2], y11 = (lo[3] - hi[2]) - hi[3];
        if (KSPLIT) {      // this piece's share of the sums over the input channels: wino8_ksplit_finish_kernel adds the pieces
          *reinterpret_cast<f32x4*>(part_mine + j * 256) = (f32x4){y00, y01, y10, y11};
          return;
        }
        y00 += bias;
        y01 += bias;
        y10 += bias;
        y11 += bias;
        // LeakyReLU with 0 <= slope <= 1 (0.2; 0 = ReLU; 1 = linear) as max(y, slope y): two instructions instead of three per value
        y00 = fmaxf(y00, y00 * a.slope);
        y01 = fmaxf(y01, y01 * a.slope);
        y10 = fmaxf(y10, y10 * a.slope);
        y11 = fmaxf(y11, y11 * a.slope);
        if (RES) {
          const f32x2 r0 = rs0[RES ? (j & 3) : 0], r1 = rs1[RES ? (j & 3) : 0];
          if (a.res_mask) {      // adjoint convolution: x LeakyReLU'(saved forward activation)
            y00 *= r0[0] > 0.f ? 1.f : a.mask_slope;
            y01 *= r0[1] > 0.f ? 1.f : a.mask_slope;
            y10 *= r1[0] > 0.f ? 1.f : a.mask_slope;
            y11 *= r1[1] > 0.f ? 1.f : a.mask_slope;
          } else {
            y00 += r0[0];
            y01 += r0[1];
            y10 += r1[0];
            y11 += r1[1];
          }
        }
        if (FUSE_OUTC) {
          s00 = fmaf(ow[j], y00, s00);
          s01 = fmaf(ow[j], y01, s01);
          s10 = fmaf(ow[j], y10, s10);
          s11 = fmaf(ow[j], y11, s11);
        } else {
          float* o = ob + (size_t)co * HpWp;
          *reinterpret_cast<f32x2*>(o) = (f32x2){y00, y01};
          *reinterpret_cast<f32x2*>(o + a.Wp) = (f32x2){y10, y11};
          if (a.pool) pb[(size_t)co * HpWp_pool] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));   // same association as maxpool2_kernel
        }
      });
      WINO8_STAMP(19);
      if (FUSE_OUTC) {
        // this wave holds couts 16 P .. 16 P + 15 (4 kg + (R & 3) + 8 (R >> 2)): the other lane half, then the other wave (through LDS)
        s00 += __shfl_xor(s00, 32);
        s01 += __shfl_xor(s01, 32);
        s10 += __shfl_xor(s10, 32);
        s11 += __shfl_xor(s11, 32);
        char* zb = lds + zoff + lane * 16;
        if (P == 1) {
          *reinterpret_cast<f32x4*>(zb) = (f32x4){s00, s01, s10, s11};
          wait_lds();
        }
        barrier();                                                   // #3
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (P == 0) {
          o = *reinterpret_cast<const f32x4*>(zb);
          wait_lds();
        }
        barrier();                                                   // #4: the 1 KiB is read (it lies in LDS the next stages refill)
        if (P == 0) {
          s00 += o[0];
          s01 += o[1];
          s10 += o[2];
          s11 += o[3];
          const size_t px0 = ((size_t)T.b * a.H + T.y0 + 2 * ty) * a.W + T.x0 + 2 * tx;
          const float ob0 = a.outc_b[0];
          const f32x2 xi0 = *reinterpret_cast<const f32x2*>(a.x_img + px0), xi1 = *reinterpret_cast<const f32x2*>(a.x_img + px0 + a.W);
          const float v00 = xi0[0] + (s00 + ob0), v01 = xi0[1] + (s01 + ob0), v10 = xi1[0] + (s10 + ob0), v11 = xi1[1] + (s11 + ob0);
          *reinterpret_cast<f32x2*>(a.img_pre + px0) = (f32x2){v00, v01};
          *reinterpret_cast<f32x2*>(a.img_pre + px0 + a.W) = (f32x2){v10, v11};
          *reinterpret_cast<f32x2*>(a.img + px0) = (f32x2){fminf(fmaxf(v00, 0.f), 1.f), fminf(fmaxf(v01, 0.f), 1.f)};
          *reinterpret_cast<f32x2*>(a.img + px0 + a.W) = (f32x2){fminf(fmaxf(v10, 0.f), 1.f), fminf(fmaxf(v11, 0.f), 1.f)};
        }
      } else {
        barrier();                                                   // #3: the exchange buffer is free again (the next stages overwrite it)
      }
      WINO8_STAMP(20);
    };

    // VMEM operations a DMA wave leaves in its queue across an epilogue (its stores); halo gathers per DMA wave (wave 0: one more)
    constexpr int NST = KSPLIT ? 8 : (FUSE_OUTC ? (P == 0 ? 4 : 0) : 16);      // (KSPLIT: the eight partial rows)
    constexpr int NRAW = C::NRAW;

    int k = 0;
    Tile T = decode(0);
    if (!T.ok) return;
    WINO8_STAMP(14);
    // prologue: first halo(s) and weight stages, first transform, first operands
    if (CT == 64) {
      dma_raw(raw_of(T, 0), 0, 0, RAW_PER_WAVE);
      dma_u(T.w, 0);
      dma_u(T.w + (UQ / 4), 1);
      dma_u(T.w + 2 * (UQ / 4), 2);
    } else {
      dma_raw(T.s0, 0, 0, RAW_PER_WAVE);
      dma_raw(raw_of(T, 1), 1, 0, RAW_PER_WAVE);
      dma_u(T.w, 0);
      dma_u(T.w + (USTG / 4), 1);
      dma_u(T.w + 2 * (USTG / 4), 2);      // (nch >= 2: chunk 1 exists)
    }
    sync_all();
    {
      Tr t;
#pragma unroll
      for (int sl = 0; sl < NSLT; ++sl) transform_slice(std::integral_constant<int, (CT == 64) ? 0 : P>{}, sl, t, 0, vdst_of(I0{}));
    }
    sync_all();
    WINO8_STAMP(15);
    read_operand(I0{}, 0, 0, 0);
    read_operand(I0{}, 0, 1, 0);

    // One chunk = NSTG stages.  FIRST: the chunk follows an epilogue, whose stores sit in the DMA wave's queue.  nu = weights of the
    // next chunk, nraw = halo of the chunk the halo stage fetches (CT 64: the next one, CT 32: the one after) -- in the next tile(s)
    // at a tile's end; after the walk's last chunk they name this tile's first chunks again (surplus loads into buffers nobody reads,
    // so that no stage carries a condition).
    // K1 / K2 / K3: the chunk 1 / 2 / 3 after this one comes from the up-sampled source (UPS instances; same tile).  CT 64: stage 0
    // fetches the next chunk's halo (or, K1, interpolates its quads 1-3 in stages 0-2) and stages the window of the chunk after (K2),
    // whose quad 0 stage 3 interpolates.  CT 32: stage 0 interpolates quad 1 of the next chunk (K1); stage 1 fetches the halo of the
    // chunk after (or, K2, interpolates its quad 0) and stages the window of the third (K3).
    auto chunk = [&](auto first_tag, auto k1_tag, auto k2_tag, auto k3_tag, const float* w_c, const float* nu, const float* nu2,
                     const float* nraw, const float* nst, int cc) {
      constexpr int EPI = decltype(first_tag)::value ? NST : 0;
      constexpr bool K1 = UPS && decltype(k1_tag)::value, K2 = UPS && decltype(k2_tag)::value, K3 = UPS && decltype(k3_tag)::value;
      constexpr int NSW = C::NSW;
      using Pre = Yes;   // (a tile's last stage could leave its prefetch to the epilogue's registers: not needed, the epilogue fits)
      const int rcur = cc & 1, rnext = rcur ^ 1;
      if constexpr (CT == 64) {
        using S0k = StageKind<!K1, K2, K1 ? 1 : -1>;
        using S1k = StageKind<false, false, K1 ? 2 : -1>;
        using S2k = StageKind<false, false, K1 ? 3 : -1>;
        using S3k = StageKind<false, false, K2 ? 0 : -1>;
        constexpr int HALO = (K1 ? 0 : NRAW) + (K2 ? NSW : 0);      // VMEM operations of stage 0 beyond its weight slice
        stage(I0{}, first_tag, std::integral_constant<int, NUW + EPI + NUW + HALO>{}, Yes{}, S0k{},
              Side{rcur, 0, 1, w_c + 3 * (UQ / 4), 3, nraw, rnext, nst, rcur, rnext, rnext});
        stage(I1{}, first_tag, std::integral_constant<int, EPI + NUW + HALO + NUW>{}, Yes{}, S1k{},
              Side{rcur, 1, 2, nu, 0, nullptr, 0, nullptr, 0, rnext, rnext});
        stage(I2{}, first_tag, std::integral_constant<int, 2 * NUW>{}, Yes{}, S2k{}, Side{rcur, 2, 3, nu + (UQ / 4), 1, nullptr, 0, nullptr, 0, rnext, rnext});
        stage(I3{}, first_tag, std::integral_constant<int, 2 * NUW>{}, Pre{}, S3k{}, Side{rnext, 3, 0, nu + 2 * (UQ / 4), 2, nullptr, 0, nullptr, 0, rcur, rcur});
      } else {
        using S0k = StageKind<false, false, K1 ? 1 : -1>;
        using S1k = StageKind<!K2, K3, K2 ? 0 : -1>;
        // weight ring: stage g = 2 cc + s reads slot g & 3 and fetches the weights of stage g + 3 (the slot the stage before just left):
        // stage 0 those of the next chunk's stage 1, stage 1 those of the chunk after's stage 0 (nu2) -- 2.4 stages of lead (1.4 with a
        // lead of two stages left the full-resolution layers waiting ~600 clocks per chunk for their weight slice behind the halo traffic)
        const int u0 = 2 * rcur, n0 = 2 * rnext;      // ring slots of this chunk's and the next chunk's stages
        stage(I0{}, first_tag, std::integral_constant<int, NUW + EPI>{}, Yes{}, S0k{},
              Side{rcur, u0, u0 + 1, nu + (USTG / 4), n0 + 1, nullptr, 0, nullptr, 0, rnext, rnext});
        stage(I1{}, first_tag, std::integral_constant<int, 2 * NUW + EPI + (K2 ? 0 : NRAW) + (K3 ? NSW : 0)>{}, Pre{}, S1k{},
              Side{rnext, u0 + 1, n0, nu2, u0, nraw, rcur, nst, rnext, rcur, rcur});
      }
    };

    int cc = 0;
    for (;;) {
      Tile Tn = decode(k + 1);
      const bool more = Tn.ok;
      if (!more) Tn = T;
      for (int c = 0; c < a.nch; ++c, ++cc) {
        const float* w_c = T.w + (size_t)c * 4 * (UQ / 4);
        const float* nu = (c + 1 < a.nch) ? w_c + 4 * (UQ / 4) : Tn.w;
        const float* nu2 = (c + 2 < a.nch) ? w_c + 8 * (UQ / 4) : Tn.w + (size_t)(c + 2 - a.nch) * 4 * (UQ / 4);      // (CT 32)
        constexpr int AH = (CT == 64) ? 1 : 2;        // chunks ahead the halo stage fetches; the staging stage: one more
        const float* nraw = (c + AH < a.nch) ? raw_of(T, c + AH) : raw_of(Tn, c + AH - a.nch);
        if constexpr (!UPS) {
          if (c == 0) chunk(Yes{}, No{}, No{}, No{}, w_c, nu, nu2, nraw, nullptr, cc);
          else chunk(No{}, No{}, No{}, No{}, w_c, nu, nu2, nraw, nullptr, cc);
        } else {
          const int cu = a.C0 / CK;                   // first chunk of the up-sampled source (host: >= 3 / 4, so chunk 0 is plain)
          auto ups = [&](int j) { return c + j < a.nch && c + j >= cu; };
          const bool k1 = ups(1), k2 = ups(2), k3 = (CT == 32) && ups(3);
          const float* nst = T.s1u + (size_t)CK * (c + AH + 1) * hpwp_lo;      // window of chunk c + 2 (CT 32: c + 3)
          if (c == 0) {
            write_tables(T);                          // (read from the first interpolation on, several barriers later)
            chunk(Yes{}, No{}, No{}, No{}, w_c, nu, nu2, nraw, nst, cc);
          } else if constexpr (CT == 64) {
            if (!k1 && !k2) chunk(No{}, No{}, No{}, No{}, w_c, nu, nu2, nraw, nst, cc);
            else if (!k1 && k2) chunk(No{}, No{}, Yes{}, No{}, w_c, nu, nu2, nraw, nst, cc);
            else if (k1 && k2) chunk(No{}, Yes{}, Yes{}, No{}, w_c, nu, nu2, nraw, nst, cc);
            else chunk(No{}, Yes{}, No{}, No{}, w_c, nu, nu2, nraw, nst, cc);
          } else {
            if (!k1 && !k2 && !k3) chunk(No{}, No{}, No{}, No{}, w_c, nu, nu2, nraw, nst, cc);
            else if (!k1 && !k2) chunk(No{}, No{}, No{}, Yes{}, w_c, nu, nu2, nraw, nst, cc);
            else if (!k1 && k3) chunk(No{}, No{}, Yes{}, Yes{}, w_c, nu, nu2, nraw, nst, cc);
            else if (k1 && k2 && k3) chunk(No{}, Yes{}, Yes{}, Yes{}, w_c, nu, nu2, nraw, nst, cc);
            else if (k1 && k2) chunk(No{}, Yes{}, Yes{}, No{}, w_c, nu, nu2, nraw, nst, cc);
            else if (k1) chunk(No{}, Yes{}, No{}, No{}, w_c, nu, nu2, nraw, nst, cc);
            else chunk(No{}, No{}, Yes{}, No{}, w_c, nu, nu2, nraw, nst, cc);      // (k2 alone: two-chunk up-sampled sources)
          }
        }
      }
      // LDS free during the epilogue (nothing reads it, no DMA in flight targets it, the next stages refill it only behind the
      // epilogue's closing barrier): CT 64 -- weight slot 3 and V[1] (the last stage's), the halo buffer of the tile's last chunk, and
      // the bytes of the CU's 160 KiB beyond LDS_USED; CT 32 -- V[1], the weight slot of the tile's LAST stage (the slot of its last
      // chunk's first stage is being refilled: the ring runs three stages ahead), the bytes beyond LDS_USED (UPS instances: their
      // staging buffers and tables, dead at a tile's end).  Buffer A carries p = 1 -> p = 0, buffer B p = 0 -> p = 1, 8 KiB per pair
      // each; the fused out-conv's 1 KiB per pair re-uses the pair's buffer B once its p = 1 wave has read it.
      const int rl = (cc - 1) & 1;
      const int bufA = (CT == 64) ? ((sw < 2) ? OFF_U + 3 * USTG + sw * 8192 : OFF_V + VSTG + (sw - 2) * 8192) : OFF_V + VSTG + sw * 8192;
      const int bufB = (CT == 64) ? ((sw < 2) ? OFF_RAW + rl * RAW_BYTES + sw * 8192 : C::LDS_USED + (sw - 2) * 8192)
                                  : ((sw < 1) ? OFF_U + (2 * rl + 1) * USTG : C::LDS_USED + (sw - 1) * 8192);
      static_assert(C::LDS_USED + (CT == 64 ? 2 : 3) * 8192 <= LDS_REQ8 && 2 * 8192 <= RAW_BYTES && USTG >= 8192, "exchange buffers");
      epilogue(T, P == 1 ? bufA : bufB, P == 1 ? bufB : bufA, bufB);
      if (!more) break;
      T = Tn;
      ++k;
    }
    sync_all();   // the surplus loads of the last chunk
  };

  if (wave < 4) body(I0{});
  else body(I1{});
}

// K-split, second launch: out = act(bias + piece 0 + piece 1 [+ piece 2 + piece 3]) per output, pieces added left to right (a fixed
// order: the bits do not depend on which workgroup ran which piece, or when).  One workgroup per tile, the thread <-> output map of the main
// kernel's epilogue (wave = 4 P + sw: cout half wm = sw & 1, tile half wn = sw >> 1; lane: tile l31, cout group kg; the wave's rows
// R = 8 P + j), so every load is the 16-byte record its writer stored: [tile][piece][wave][row][lane] f32x4.
// A kernel boundary, not an in-kernel arrival protocol: the protocol (write-through sc1 slab stores, device-scope counter per (tile,
// wave), last arriver sums) was built first and measured -- the returned atomic and the drained stores cost ~6 us per work item with
// nothing to hide them behind (+19 % on the split layers at B = 48), and an agent-scope release fence instead (buffer_wbl2) ~60 us per
// item under this kernel's output traffic.  The boundary costs ~2 us per split layer.
__global__ __launch_bounds__(512) void wino8_ksplit_finish_kernel(WinoArgs a) {
  using C = Cfg8<64>;
  const int item = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int reg = item / a.nct, ct = item - reg * a.nct;
  const int t1 = reg / a.rx, txr = reg - t1 * a.rx, b = t1 / a.ry, tyr = t1 - b * a.ry;
  const int x0 = txr * C::RPXW, y0 = tyr * 16;
  const int P = wave >> 2, sw = wave & 3, wm = sw & 1, wn = sw >> 1, l31 = lane & 31, kg = lane >> 5;
  const int tile = wn * 32 + l31, ty = tile / C::TX, tx = tile % C::TX;
  const int cbase = ct * 64 + wm * 32 + 4 * kg;
  const size_t HpWp = (size_t)a.Hp * a.Wp;
  float* ob = a.out + ((size_t)b * a.Cout) * HpWp + (size_t)(y0 + 2 * ty + 1) * a.Wp + x0 + 2 * tx + PADL;
  const int Hp_pool = padded_h(a.H / 2), Wp_pool = padded_w(a.W / 2);
  const size_t HpWp_pool = (size_t)Hp_pool * Wp_pool;
  float* pb = a.pool ? a.pool + (size_t)b * a.Cout * HpWp_pool + (size_t)(y0 / 2 + ty + 1) * Wp_pool + x0 / 2 + tx + PADL : nullptr;
  const float* part = a.part + ((size_t)item * a.ksplit * 8 + wave) * 2048 + lane * 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int R = 8 * P + j;
    const int co = cbase + (R & 3) + 8 * (R >> 2);
    f32x4 y = *reinterpret_cast<const f32x4*>(part + j * 256);
    for (int g = 1; g < a.ksplit; ++g) {
      const f32x4 pg = *reinterpret_cast<const f32x4*>(part + (size_t)g * (8 * 2048) + j * 256);
      y[0] += pg[0];
      y[1] += pg[1];
      y[2] += pg[2];
      y[3] += pg[3];
    }
    const float bias = a.bias[co];
    float y00 = y[0] + bias, y01 = y[1] + bias, y10 = y[2] + bias, y11 = y[3] + bias;
    y00 = fmaxf(y00, y00 * a.slope);
    y01 = fmaxf(y01, y01 * a.slope);
    y10 = fmaxf(y10, y10 * a.slope);
    y11 = fmaxf(y11, y11 * a.slope);
    float* o = ob + (size_t)co * HpWp;
    *reinterpret_cast<f32x2*>(o) = (f32x2){y00, y01};
    *reinterpret_cast<f32x2*>(o + a.Wp) = (f32x2){y10, y11};
    if (a.pool) pb[(size_t)co * HpWp_pool] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));   // same association as maxpool2_kernel
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// which layers the 8-wave kernel can run: the 4-wave kernel's geometries (conv3x3_wino_ok) with >= 2 chunks on the 32-cout tile
// (its halo stage fetches two chunks ahead)
static int wino8_ct(int C0, int C1, int cout, int H, int W) {
  if (C0 <= 0 || H < 16 || H % 16 != 0) return 0;
  if (cout % 64 == 0) return (C0 % 16 == 0 && C1 % 16 == 0 && W % 16 == 0) ? 64 : 0;
  if (cout % 32 == 0) return (C0 % 8 == 0 && C1 % 8 == 0 && W % 32 == 0 && C0 + C1 >= 16) ? 32 : 0;
  return 0;
}
bool conv3x3_wino8_ok(int C0, int C1, int cout, int H, int W) { return wino8_ct(C0, C1, cout, H, W) != 0; }

// r6: which layers CAN split their channel chunks over several work items, and into how many pieces.  A rule of the LAYER'S geometry alone
// (never of the batch): whenever a layer splits, the summation tree of an output is the same, and per-image results are bit-identical across
// batch sizes, slices and launch chains among the calls that split (unet.hip decides per call WHETHER the deep levels split: option
// fp32_ksplit).  t = tiles per image: <= 8 (the 16 x 16 level of a 256 x 256 input: one region x 8 cout tiles) -> 4 pieces, <= 16
// (32 x 32: four regions x 4) -> 2; every piece keeps >= 4 chunks.  At B = 6 this puts 192 instead of 48 / 96 workgroups on the chip
// (measured per batch size in profiles/r6_ksplit.txt).
// rule: 1 = the default (4 pieces up to 8 tiles per image, 2 up to 16); any other value (tuning) = pieces for t <= 8 in bits 0-3, for
// t <= 16 in bits 4-7.
int conv3x3_wino8_ksplit(int C0, int C1, int cout, int H, int W, int rule) {
  if (wino8_ct(C0, C1, cout, H, W) != 64 || rule <= 0) return 1;
  const int t = (H / 16) * (W / 16) * (cout / 64), nch = (C0 + C1) / 16;
  const int g8 = rule == 1 ? 4 : (rule & 15), g16 = rule == 1 ? 2 : ((rule >> 4) & 15);
  int g = t <= 8 ? g8 : (t <= 16 ? g16 : 1);
  if (g != 1 && g != 2 && g != 4) g = 1;
  while (g > 1 && g * t > 32) g /= 2;      // the scratch slab holds 32 (tile, piece) slots per image
  while (g > 1 && (nch % g != 0 || nch / g < 4)) g /= 2;
  return g;
}
size_t conv3x3_wino8_ksplit_bytes_per_image() { return (size_t)32 * 8 * 2048 * sizeof(float); }     // <= 32 (tile, piece) slots of 64 KiB

template <int CT, bool FUSE_OUTC, bool RES, bool UPS = false, bool KSPLIT = false>
static int launch_wino8(WinoArgs a, hipStream_t s) {
  using C = Cfg8<CT>;
  static_assert(C::LDS_USED <= LDS_REQ8 && C::LDS_USED_UPS <= LDS_REQ8, "LDS budget");
  a.nct = a.Cout / CT;
  a.nch = (a.C0 + a.C1) / C::CK;
  a.nch_all = a.nch;
  if (KSPLIT) a.nch /= a.ksplit;
  else a.ksplit = 1;
  a.rx = a.W / C::RPXW;
  a.ry = a.H / 16;
  static std::once_flag attr_once[64];
  int dev = 0;
  PNPX_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64) {
    hipError_t e = hipSuccess;
    std::call_once(attr_once[dev], [&] {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino8_f32_kernel<CT, FUSE_OUTC, RES, UPS, KSPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REQ8);
    });
    PNPX_HIP(e);
  }
  const int vnct = a.nct * a.ksplit;                 // work items per region
  const long long nreg = (long long)a.rx * a.ry * a.B, ntiles = nreg * vnct;
  long long grid = 256;
  if (grid >= ntiles) grid = ntiles;
  else if ((grid / 8) % vnct != 0 && grid >= 8LL * vnct) grid -= grid % (8 * vnct);
  hipLaunchKernelGGL((conv3x3_wino8_f32_kernel<CT, FUSE_OUTC, RES, UPS, KSPLIT>), dim3((unsigned)grid), dim3(512), LDS_REQ8, s, a);
  PNPX_LAUNCH_CHECK();
  if (KSPLIT) {
    hipLaunchKernelGGL(wino8_ksplit_finish_kernel, dim3((unsigned)(nreg * a.nct)), dim3(512), 0, s, a);
    PNPX_LAUNCH_CHECK();
  }
  return PNPX_OK;
}

int launch_conv3x3_wino8_outc(const float* u, const float* bias, const float* in0, int cin, const float* outc_w, const float* outc_b,
                              const float* x_img, float* img, float* img_pre, int B, int H, int W, hipStream_t s) {
  if (wino8_ct(cin, 0, 32, H, W) != 32) {
    set_error("conv3x3_wino8_outc: unsupported geometry (%d -> 32 channels, %d x %d)", cin, H, W);
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
  return launch_wino8<32, true, false>(a, s);
}

// The decoder entries of the UNet with the bilinear x2 (align_corners) up-sampling of their second source inside the kernel: in1 is the
// LOW-resolution tensor (C1 channels, H/2 x W/2, padded planar).  The first source must hold >= 3 (cout % 64 == 0) / 4 chunks.
bool conv3x3_wino8_ups_ok(int C0, int C1, int cout, int H, int W) {
  const int ct = wino8_ct(C0, C1, cout, H, W);
  if (!ct || C1 <= 0 || H % 2 || W % 2) return false;
  const int h = H / 2, w = W / 2;
  if (ct == 64) return C0 / 16 >= 3 && h + 1 >= Cfg8<64>::NR && w + PADL >= Cfg8<64>::NQP;
  return C0 / 8 >= 4 && h + 1 >= Cfg8<32>::NR && w + PADL >= Cfg8<32>::NQP;
}

int launch_conv3x3_wino8_ups(const float* u, const float* bias, int cout, const float* in0, int C0, const float* in1_lowres, int C1,
                             float* out, int B, int H, int W, hipStream_t s, float slope) {
  if (!conv3x3_wino8_ups_ok(C0, C1, cout, H, W)) {
    set_error("conv3x3_wino8_ups: unsupported geometry (%d + up(%d) -> %d channels, %d x %d)", C0, C1, cout, H, W);
    return PNPX_ERR_SHAPE;
  }
  WinoArgs a{};
  a.in0 = in0;
  a.in1 = in1_lowres;
  a.u = u;
  a.bias = bias;
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
  a.ups_h = H / 2;
  a.ups_w = W / 2;
  // models/unet.py:99 (align_corners = True): source = destination * (in - 1) / (out - 1); the same two floats as unet.hip hands upsample2x_v4_kernel
  a.ups_sy = (H > 1) ? (float)(H / 2 - 1) / (float)(H - 1) : 0.f;
  a.ups_sx = (W > 1) ? (float)(W / 2 - 1) / (float)(W - 1) : 0.f;
  return wino8_ct(C0, C1, cout, H, W) == 64 ? launch_wino8<64, false, false, true>(a, s) : launch_wino8<32, false, false, true>(a, s);
}

// Input-gradient (adjoint) convolution of the backward pass on the same kernel: u = pack_conv_weights_wino of the transposed, tap-flipped
// weights; no bias (zero_bias: >= cout zeros), linear, optionally x LeakyReLU'(dmask) with dmask = the saved forward activation the gradient
// flows into (models/unet.py:8-18 backwards).
int launch_conv3x3_wino8_grad(const float* u, const float* zero_bias, int cout, const float* gin, int cin, float* gout, const float* dmask,
                              float mask_slope, int B, int H, int W, hipStream_t s) {
  const int ct = wino8_ct(cin, 0, cout, H, W);
  if (!ct) {
    set_error("conv3x3_wino8_grad: unsupported geometry (%d -> %d channels, %d x %d)", cin, cout, H, W);
    return PNPX_ERR_SHAPE;
  }
  WinoArgs a{};
  a.in0 = gin;
  a.in1 = gin;
  a.u = u;
  a.bias = zero_bias;
  a.res = dmask;
  a.res_mask = 1;
  a.mask_slope = mask_slope;
  a.out = gout;
  a.B = B;
  a.H = H;
  a.W = W;
  a.Hp = padded_h(H);
  a.Wp = padded_w(W);
  a.C0 = cin;
  a.C1 = 0;
  a.Cout = cout;
  a.slope = 1.0f;
  if (dmask) return ct == 64 ? launch_wino8<64, false, true>(a, s) : launch_wino8<32, false, true>(a, s);
  return ct == 64 ? launch_wino8<64, false, false>(a, s) : launch_wino8<32, false, false>(a, s);
}

int launch_conv3x3_wino8(const float* u, const float* bias, int cout, const float* in0, int C0, const float* in1, int C1,
                         float* out, int B, int H, int W, hipStream_t s, float slope, const float* res, float* pool_out, float* ks_part,
                         int ks_rule) {
  const int ct = wino8_ct(C0, C1, cout, H, W);
  if (!ct) {
    set_error("conv3x3_wino8: unsupported geometry (%d + %d -> %d channels, %d x %d)", C0, C1, cout, H, W);
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
  if (res) return ct == 64 ? launch_wino8<64, false, true>(a, s) : launch_wino8<32, false, true>(a, s);
  // K-split instances: when the caller hands over the scratch slab (>= conv3x3_wino8_ksplit_bytes_per_image() per image of the batch)
  // and the layer's geometry calls for it
  const int g = ks_part ? conv3x3_wino8_ksplit(C0, C1, cout, H, W, ks_rule) : 1;
  if (g > 1) {
    a.ksplit = g;
    a.part = ks_part;
    return launch_wino8<64, false, false, false, true>(a, s);
  }
  return ct == 64 ? launch_wino8<64, false, false>(a, s) : launch_wino8<32, false, false>(a, s);
}

}  // namespace pnpx

#ifdef WINO8_TRACE
extern "C" int pnpx_debug_wino8_trace(unsigned long long* host_out) {      // [2][2048]; diagnostic builds only
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pnpx::wino8_trace_buf), sizeof(unsigned long long) * 2 * 2048);
}
#endif
