// Probe: Winograd F(2x2, 3x3) on the half-split (hi/lo f16, 3 MFMAs per product) scheme of conv_hs,
// as a stand-alone kernel for layers with cin % 16 == 0 and cout % 64 == 0 over the HS8 record layout (csrc/hs_rec.h).
// Question it answers: with 16 products instead of 36 per 2x2 output tile (2.25x fewer v_mfma_f32_32x32x16_f16), does a
// whole layer run >= 1.25x faster than the direct conv_hs launch of the same layer on the same box?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/wino_probe.hip -o /tmp/wino_probe
//   /tmp/wino_probe <B> <H> <cin> <cout> [iters]        (H = W, multiple of 16)
//
// Dataflow (one workgroup = 4 waves, one per SIMD, 512 registers each; 64 couts x 64 tiles = a 16 x 16-pixel region):
//   wave (wm, wn) owns cout block wm (32 couts) x tile block wn (32 tiles) and ALL 16 Winograd positions of it:
//   16 accumulators of 32 x 32 = 256 registers.  K is walked in chunks of 16 input channels; a chunk is four STAGES, one
//   per position row a (4 positions, 12 MFMAs per wave).  Per stage: the transformed weights U[a][0..3] of the chunk
//   (16 KiB) arrive by LDS-DMA; the transformed inputs V[a][0..3] (16 KiB: [b][hi,lo][kg][tile] x 16 B, exactly the B
//   fragments) are computed by all 256 threads from the raw 18 x 18 halo (LDS-DMA, double-buffered per chunk) one stage
//   ahead: reconstruct hi + lo in fp32, B^T d B row a (constants +-1: exact up to fp32 rounding of 4-term sums), re-split.
//   Output transform A^T M A in registers in the epilogue (lane-local), then bias + LeakyReLU + hi/lo split + stores.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr float ASCALE = 16.f;
constexpr int RW = 18, RPX = RW * RW;              // raw halo of a 16 x 16 region
constexpr int RAW_PIECES = 4 * RPX;                // [g][half][px] x 16 B
constexpr int RAW_INSTR = (RAW_PIECES + 63) / 64;  // 21
constexpr int RAW_BYTES = RAW_INSTR * 1024;
constexpr int UQ = 16384, VQ = 16384;
constexpr int NU = 4;                               // U ring: the DMA runs three stages ahead (LDS-DMA issue -> landed is ~1 us)
constexpr int OFF_U = 0, OFF_V = NU * UQ, OFF_RAW = NU * UQ + 2 * VQ;
constexpr int LDS_BYTES = OFF_RAW + 2 * RAW_BYTES;   // 108544 > 80 KiB: one workgroup per CU

struct WinoArgs {
  const char* in;     // HS8 [B][cin/8][H+2][W+2]
  const char* u;      // [cout/64][cin/16][a 4][b 4][hi,lo][kg 2][m 64][8] f16
  const float* bias;  // [cout]
  char* out;          // HS8 [B][cout/8][H+2][W+2]
  int B, H, W, G, nct, nch;
  int rx, ry;         // regions per row / column
  float c16, slope, neg_one;
  int abl;            // ablation bits (invalid results): 1 = no input transform, 2 = no MFMAs, 4 = no DMA, 8 = no epilogue, 16 = no barriers
};

__device__ __forceinline__ void glds16b(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ unsigned lo_pair(unsigned hi_pk, float neg_one, float v0, float v1) {
  unsigned lo_pk;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo_pk)
      : "v"(hi_pk), "s"(neg_one), "v"(v0), "v"(v1));
  return lo_pk;
}

__global__ __launch_bounds__(256, 1) void wino_hs_kernel(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Wp = a.W + 2, HpWp = (a.H + 2) * Wp;
  const int nregions = a.rx * a.ry * a.B;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;

  struct Tile {
    int ct, b, x0, y0;
    const char* src;   // halo origin of channel group 0
    const char* w;     // weight slice of cout tile ct
    bool ok;
  };
  auto decode = [&](int k) {
    Tile T;
    const int j = slot + nslot * k;
    const int q = j / a.nct;
    T.ct = j - q * a.nct;
    const int reg = 8 * q + xcd;
    T.ok = reg < nregions;
    const int t1 = reg / a.rx;
    const int tx = reg - t1 * a.rx;
    const int t2 = t1 / a.ry;
    const int ty = t1 - t2 * a.ry;
    T.b = t2;
    T.x0 = tx * 16;
    T.y0 = ty * 16;
    T.src = a.in + ((size_t)T.b * a.G * HpWp + (size_t)T.y0 * Wp + T.x0) * 32;
    T.w = a.u + (size_t)T.ct * a.nch * 4 * UQ;
    return T;
  };

  // raw-halo DMA: per-lane source offsets of this wave's pieces (instruction = wave + 4k)
  int roff[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int idx = (wave + 4 * k) * 64 + lane;
    const int q = idx / RPX, r = idx - q * RPX;
    const int hy = r / RW, hx = r - hy * RW;
    roff[k] = (idx < RAW_PIECES) ? (((q >> 1) * HpWp + hy * Wp + hx) * 32 + (q & 1) * 16) : 0;
  }
  auto issue_raw = [&](const Tile& T, int c, int k, int rbuf) {
    const int instr = wave + 4 * k;
    if (instr < RAW_INSTR && !(a.abl & 4)) glds16b(T.src + (size_t)c * 2 * HpWp * 32 + roff[k], lds + OFF_RAW + rbuf * RAW_BYTES + instr * 1024);
  };
  auto issue_u = [&](const Tile& T, int c, int aa, int ubuf) {
    if (a.abl & 4) return;
    const char* s = T.w + ((size_t)c * 4 + aa) * UQ + lane * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) glds16b(s + (wave + 4 * k) * 1024, lds + OFF_U + ubuf * UQ + (wave + 4 * k) * 1024);
  };

  // transform role: tile tt (8 x 8 tiles of the region), channel group tkg, channel half tch (4 channels)
  const int tt = lane, tkg = wave & 1, tch = wave >> 1;
  const int tty = tt >> 3, ttx = tt & 7;
  const int t_rd = ((tkg * 2) * RPX + (2 * tty) * RW + 2 * ttx) * 16 + tch * 8;   // hi plane; lo plane + RPX * 16
  const int t_wr = (tkg * 64 + tt) * 16 + tch * 8;                                 // + (b * 2 + half) * 2048
  const float neg_one = a.neg_one;

  auto transform = [&](auto a_tag, int rbuf, int vbuf) {
    if (a.abl & 1) return;
    constexpr int A = decltype(a_tag)::value;
    constexpr int RA = (A == 0) ? 0 : (A == 2 ? 2 : 1);
    constexpr int RB = (A == 0) ? 2 : (A == 1 ? 2 : (A == 2 ? 1 : 3));
    constexpr float SG = (A == 1) ? 1.f : -1.f;
    const char* rb = lds + OFF_RAW + rbuf * RAW_BYTES + t_rd;
    float r[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const h4 ha = *reinterpret_cast<const h4*>(rb + (RA * RW + x) * 16);
      const h4 la = *reinterpret_cast<const h4*>(rb + (RA * RW + x) * 16 + RPX * 16);
      const h4 hb = *reinterpret_cast<const h4*>(rb + (RB * RW + x) * 16);
      const h4 lb = *reinterpret_cast<const h4*>(rb + (RB * RW + x) * 16 + RPX * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[x][e] = ((float)ha[e] + (float)la[e]) + SG * ((float)hb[e] + (float)lb[e]);
    }
    char* vb = lds + OFF_V + vbuf * VQ + t_wr;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = (b == 0) ? r[0][e] - r[2][e] : (b == 1) ? r[1][e] + r[2][e] : (b == 2) ? r[2][e] - r[1][e] : r[1][e] - r[3][e];
      const h2 p0 = {(_Float16)v[0], (_Float16)v[1]}, p1 = {(_Float16)v[2], (_Float16)v[3]};
      const unsigned h0 = __builtin_bit_cast(unsigned, p0), h1 = __builtin_bit_cast(unsigned, p1);
      const unsigned l0 = lo_pair(h0, neg_one, v[0], v[1]), l1 = lo_pair(h1, neg_one, v[2], v[3]);
      *reinterpret_cast<u32x2*>(vb + (b * 2 + 0) * 2048) = (u32x2){h0, h1};
      *reinterpret_cast<u32x2*>(vb + (b * 2 + 1) * 2048) = (u32x2){l0, l1};
    }
  };

  // MFMA role: cout block wm, tile block wn
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, kg = lane >> 5;
  const int a_lane = (kg * 64 + wm * 32 + l31) * 16;
  const int b_lane = (kg * 64 + wn * 32 + l31) * 16;
  f32x16 acc[4][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  auto multiply = [&](auto a_tag) {
    if (a.abl & 2) return;
    constexpr int A = decltype(a_tag)::value;
    const char* ua = lds + OFF_U + A * UQ + a_lane;
    const char* vb = lds + OFF_V + (A & 1) * VQ + b_lane;
    h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      ah[b] = *reinterpret_cast<const h8*>(ua + (b * 2 + 0) * 2048);
      al[b] = *reinterpret_cast<const h8*>(ua + (b * 2 + 1) * 2048);
      bh[b] = *reinterpret_cast<const h8*>(vb + (b * 2 + 0) * 2048);
      bl[b] = *reinterpret_cast<const h8*>(vb + (b * 2 + 1) * 2048);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[A][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bh[b], acc[A][b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[A][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bl[b], acc[A][b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[A][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b], bh[b], acc[A][b], 0, 0, 0);
  };

  auto epilogue = [&](const Tile& T) {
    if (a.abl & 8) return;
    const int Gout = a.nct * 8;
    const int ty = 4 * wn + (l31 >> 3), tx = l31 & 7;
    const int cbase = T.ct * 64 + wm * 32 + 16 * kg;
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = a.bias[cbase + r] * ASCALE;
    char* ob = a.out + ((size_t)T.b * Gout + (cbase >> 3)) * HpWp * 32;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      float y[2][2][8];
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int r = qp * 8 + r8;
        float t0[4], t1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          t0[i] = acc[i][0][r] + acc[i][1][r] + acc[i][2][r];
          t1[i] = acc[i][1][r] - acc[i][2][r] - acc[i][3][r];
        }
        y[0][0][r8] = t0[0] + t0[1] + t0[2];
        y[1][0][r8] = t0[1] - t0[2] - t0[3];
        y[0][1][r8] = t1[0] + t1[1] + t1[2];
        y[1][1][r8] = t1[1] - t1[2] - t1[3];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int py = T.y0 + 2 * ty + i + 1, px = T.x0 + 2 * tx + j + 1;
          char* o = ob + ((size_t)py * Wp + px) * 32 + (size_t)qp * HpWp * 32;
          unsigned rec[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u0 = __builtin_fmaf(y[i][j][2 * e], a.c16, bias[qp * 8 + 2 * e]);
            const float u1 = __builtin_fmaf(y[i][j][2 * e + 1], a.c16, bias[qp * 8 + 2 * e + 1]);
            const float v0 = __builtin_fmaxf(u0, u0 * a.slope), v1 = __builtin_fmaxf(u1, u1 * a.slope);
            const h2 hh = {(_Float16)v0, (_Float16)v1};
            rec[e] = __builtin_bit_cast(unsigned, hh);
            rec[4 + e] = lo_pair(rec[e], neg_one, v0, v1);
          }
          *reinterpret_cast<u32x4*>(o) = (u32x4){rec[0], rec[1], rec[2], rec[3]};
          *reinterpret_cast<u32x4*>(o + 16) = (u32x4){rec[4], rec[5], rec[6], rec[7]};
        }
    }
    zero_acc();
  };

  auto sync_all = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // end of a stage: the LDS-DMA this wave issued for the NEXT stage (and, at a = 2, the next chunk's halo) has landed --
  // a counted wait: N = VMEM operations issued after it (younger weight slices, the halo pieces, the previous tile's
  // stores), which may stay in flight; then the V fragments written this stage are visible and everyone is done reading
  auto sync_counted = [&](auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    if (!(a.abl & 16)) __syncthreads();
  };
  constexpr int NST = 16;   // record stores per wave and tile
  constexpr int NRAW = 5;   // halo pieces per wave (wave 0 issues 6: counted conservatively)

  int k = 0;
  Tile T = decode(0);
  if (!T.ok) return;
  // prologue: halo of (T, 0), weight slices of stages 0..2, V of stage 0
#pragma unroll
  for (int q = 0; q < 6; ++q) issue_raw(T, 0, q, 0);
  issue_u(T, 0, 0, 0);
  issue_u(T, 0, 1, 1);
  issue_u(T, 0, 2, 2);
  sync_all();
  transform(std::integral_constant<int, 0>{}, 0, 0);
  sync_all();

  int cc = 0;           // running chunk counter (raw buffer parity)
  bool after_epi = false;
  for (;;) {
    Tile Tn = decode(k + 1);
    for (int c = 0; c < a.nch; ++c, ++cc) {
      const bool last_c = (c == a.nch - 1);
      const bool has_next = !last_c || Tn.ok;
      const Tile& Tc = last_c ? Tn : T;       // tile of the next chunk
      const int cn = last_c ? 0 : c + 1;
      const int rcur = cc & 1, rnext = rcur ^ 1;
      // stage a = 0: issue U(c, 3) and the next chunk's halo
      issue_u(T, c, 3, 3);
      if (has_next) {
#pragma unroll
        for (int q = 0; q < 6; ++q) issue_raw(Tc, cn, q, rnext);
      }
      transform(std::integral_constant<int, 1>{}, rcur, 1);
      multiply(std::integral_constant<int, 0>{});
      if (!has_next) sync_all();
      else if (after_epi) sync_counted(std::integral_constant<int, 4 + NST + 4 + NRAW>{});
      else sync_counted(std::integral_constant<int, 4 + 4 + NRAW>{});
      // stage a = 1
      if (has_next) issue_u(Tc, cn, 0, 0);
      transform(std::integral_constant<int, 2>{}, rcur, 0);
      multiply(std::integral_constant<int, 1>{});
      if (!has_next) sync_all();
      else if (after_epi) sync_counted(std::integral_constant<int, NST + 4 + NRAW + 4>{});
      else sync_counted(std::integral_constant<int, 4 + NRAW + 4>{});
      after_epi = false;
      // stage a = 2
      if (has_next) issue_u(Tc, cn, 1, 1);
      transform(std::integral_constant<int, 3>{}, rcur, 1);
      multiply(std::integral_constant<int, 2>{});
      if (!has_next) sync_all();
      else sync_counted(std::integral_constant<int, 8>{});
      // stage a = 3
      if (has_next) {
        issue_u(Tc, cn, 2, 2);
        transform(std::integral_constant<int, 0>{}, rnext, 0);
      }
      multiply(std::integral_constant<int, 3>{});
      if (!has_next) sync_all();
      else sync_counted(std::integral_constant<int, 8>{});
    }
    epilogue(T);
    after_epi = true;
    if (!Tn.ok) break;
    T = Tn;
    ++k;
  }
}

// ------------------------------------------------------------------ host side
__global__ void fill_input(char* t, int B, int G, int H, int W, unsigned seed) {
  // interior records: pseudo-random activations (ReLU-like: half of them small), x16, split; border stays zero
  const size_t n = (size_t)B * G * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % W, y = (i / W) % H;
    const size_t bg = i / ((size_t)W * H);
    _Float16* rec = reinterpret_cast<_Float16*>(t + ((bg * (H + 2) + y + 1) * (W + 2) + x + 1) * 32);
    for (int e = 0; e < 8; ++e) {
      unsigned h = (unsigned)(i * 8 + e) * 2654435761u + seed;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
      float u = (h & 0xffffff) / 16777216.f, s = ((h >> 24) & 0xff) / 256.f;
      float v = (u - 0.35f) * 1.7f;
      v = v > 0 ? v : 0.2f * v;
      v *= (0.25f + s);
      const float vs = v * ASCALE;
      const _Float16 hi = (_Float16)vs;
      rec[e] = hi;
      rec[8 + e] = (_Float16)(vs - (float)hi);
    }
  }
}

// fp64 direct convolution of image b from the same records and the ORIGINAL fp32 weights; error statistics vs `out`
__global__ void check_kernel(const char* in, const float* w, const float* bias, const char* out, int b, int cin, int cout,
                             int H, int W, float slope, double* stats) {
  const int n = cout * H * W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = i % W, y = (i / W) % H, co = i / (W * H);
  const int Wp = W + 2, HpWp = (H + 2) * Wp;
  double s = 0;
  for (int ci = 0; ci < cin; ++ci) {
    const char* base = in + ((size_t)(b * (cin / 8) + ci / 8) * HpWp) * 32;
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) {
        const _Float16* rec = reinterpret_cast<const _Float16*>(base + ((size_t)(y + k) * Wp + x + l) * 32);
        const double v = ((double)(float)rec[ci & 7] + (double)(float)rec[8 + (ci & 7)]) / ASCALE;
        s += v * (double)w[((size_t)co * cin + ci) * 9 + k * 3 + l];
      }
  }
  s += bias[co];
  s = s > 0 ? s : slope * s;
  const _Float16* orec = reinterpret_cast<const _Float16*>(out + (((size_t)(b * (cout / 8) + co / 8) * HpWp) + (size_t)(y + 1) * Wp + x + 1) * 32);
  const double got = ((double)(float)orec[co & 7] + (double)(float)orec[8 + (co & 7)]) / ASCALE;
  const double err = fabs(got - s);
  atomicMax(reinterpret_cast<unsigned long long*>(&stats[0]), __double_as_longlong(err));
  atomicMax(reinterpret_cast<unsigned long long*>(&stats[1]), __double_as_longlong(fabs(s)));
  atomicAdd(&stats[2], err * err);
  atomicAdd(&stats[3], s * s);
}

static inline int row_channel(int row) {   // conv_hs.hip::hs_row_channel
  const int kg = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
  return 16 * kg + r;
}
static inline uint16_t f16_bits(_Float16 h) {
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 48, H = argc > 2 ? atoi(argv[2]) : 128;
  const int cin = argc > 3 ? atoi(argv[3]) : 64, cout = argc > 4 ? atoi(argv[4]) : 64;
  const int iters = argc > 5 ? atoi(argv[5]) : 20;
  const int W = H;
  if (H % 16 || cin % 16 || cout % 64) { fprintf(stderr, "unsupported shape\n"); return 2; }
  const int G = cin / 8, Gout = cout / 8, nct = cout / 64, nch = cin / 16;
  const size_t in_bytes = (size_t)B * G * (H + 2) * (W + 2) * 32, out_bytes = (size_t)B * Gout * (H + 2) * (W + 2) * 32;
  char *d_in, *d_out, *d_u;
  float *d_w, *d_bias;
  double* d_stats;
  CK(hipMalloc(&d_in, in_bytes));
  CK(hipMalloc(&d_out, out_bytes));
  CK(hipMemset(d_in, 0, in_bytes));
  CK(hipMemset(d_out, 0, out_bytes));
  fill_input<<<4096, 256>>>(d_in, B, G, H, W, 12345u);

  // weights: He-normal-ish, fp32; U = G g G^T in fp64, scaled by a power of two, split
  std::vector<float> w((size_t)cout * cin * 9), bias(cout);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  const double sd = std::sqrt(2.0 / (1.04 * 9 * cin));
  for (auto& v : w) { const double u1 = rnd() + 1e-12, u2 = rnd(); v = (float)(sd * std::sqrt(-2 * std::log(u1)) * std::cos(6.283185307179586 * u2)); }
  for (auto& v : bias) v = (float)(0.05 * (rnd() - 0.5));
  static const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  std::vector<double> U((size_t)cout * cin * 16);
  double mx = 0;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      const float* g = &w[((size_t)co * cin + ci) * 9];
      for (int aa = 0; aa < 4; ++aa)
        for (int bb = 0; bb < 4; ++bb) {
          double s = 0;
          for (int k2 = 0; k2 < 3; ++k2)
            for (int l = 0; l < 3; ++l) s += Gm[aa][k2] * (double)g[k2 * 3 + l] * Gm[bb][l];
          U[((size_t)co * cin + ci) * 16 + aa * 4 + bb] = s;
          mx = std::fmax(mx, std::fabs(s));
        }
    }
  int e;
  std::frexp(mx, &e);
  const float wscale = std::ldexp(1.0f, 14 - e);
  std::vector<uint16_t> upk((size_t)nct * nch * 16 * 2 * 2 * 64 * 8);
  for (int ct = 0; ct < nct; ++ct)
    for (int ch = 0; ch < nch; ++ch)
      for (int p = 0; p < 16; ++p)
        for (int kgi = 0; kgi < 2; ++kgi)
          for (int m = 0; m < 64; ++m)
            for (int el = 0; el < 8; ++el) {
              const int co = ct * 64 + (m & ~31) + row_channel(m & 31), ci = ch * 16 + kgi * 8 + el;
              const float v = (float)(U[((size_t)co * cin + ci) * 16 + p] * wscale);
              const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
              const size_t base = (((size_t)ct * nch + ch) * 16 + p) * 2 * 2 * 64 * 8;
              upk[base + ((size_t)(0 * 2 + kgi) * 64 + m) * 8 + el] = f16_bits(hi);
              upk[base + ((size_t)(1 * 2 + kgi) * 64 + m) * 8 + el] = f16_bits(lo);
            }
  CK(hipMalloc(&d_u, upk.size() * 2));
  CK(hipMemcpy(d_u, upk.data(), upk.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_w, w.size() * 4));
  CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_bias, cout * 4));
  CK(hipMemcpy(d_bias, bias.data(), cout * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_stats, 4 * sizeof(double)));

  WinoArgs a;
  a.in = d_in; a.u = d_u; a.bias = d_bias; a.out = d_out;
  a.B = B; a.H = H; a.W = W; a.G = G; a.nct = nct; a.nch = nch;
  a.rx = W / 16; a.ry = H / 16;
  a.c16 = ASCALE / (wscale * ASCALE);
  a.slope = 0.2f; a.neg_one = -1.f; a.abl = 0;
  CK(hipFuncSetAttribute((const void*)wino_hs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const long long ntiles = (long long)a.rx * a.ry * B * nct;
  const int grid = (int)std::min<long long>(256, (ntiles + 7) / 8 * 8);
  wino_hs_kernel<<<grid, 256, LDS_BYTES>>>(a);
  CK(hipDeviceSynchronize());
  for (int b : {0, B - 1}) {
    CK(hipMemset(d_stats, 0, 4 * sizeof(double)));
    const int n = cout * H * W;
    check_kernel<<<(n + 255) / 256, 256>>>(d_in, d_w, d_bias, d_out, b, cin, cout, H, W, 0.2f, d_stats);
    double s[4];
    CK(hipMemcpy(s, d_stats, sizeof(s), hipMemcpyDeviceToHost));
    printf("check image %d: max|err| %.3e  max|ref| %.3e  rel-L2 %.3e\n", b, s[0], s[1], std::sqrt(s[2] / s[3]));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  const double flop = 2.0 * 9 * cin * cout * (double)H * W * B;
  if (getenv("WINO_ABL"))
    for (int abl : {1, 2, 4, 8, 13, 11, 14, 7, 9, 5, 15, 31}) {
      a.abl = abl;
      for (int i = 0; i < 3; ++i) wino_hs_kernel<<<grid, 256, LDS_BYTES>>>(a);
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) wino_hs_kernel<<<grid, 256, LDS_BYTES>>>(a);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  abl %2d: %.1f us\n", abl, ms / iters * 1e3);
    }
  a.abl = 0;
  for (int i = 0; i < 5; ++i) wino_hs_kernel<<<grid, 256, LDS_BYTES>>>(a);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) wino_hs_kernel<<<grid, 256, LDS_BYTES>>>(a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  printf("wino_hs B=%d %dx%d %d->%d: %.1f us per launch = %.1f TF/s algorithmic (direct-conv FLOPs), grid %d, tiles %lld\n", B, H, W,
         cin, cout, ms * 1e3, flop / ms / 1e9, grid, ntiles);
  return 0;
}
