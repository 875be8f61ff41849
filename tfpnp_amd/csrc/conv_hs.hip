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
#include <type_traits>

#include "common.h"
#include "conv_hs.h"

namespace pnpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16b(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

template <int MT, int NBW, int MBW>
struct HsGeom {
  static constexpr int MBH = 32 / MBW;
  static constexpr int NBLK = 4 * NBW;
  static constexpr int TW = MBW;
  static constexpr int TH = NBLK * MBH;
  static constexpr int LW = TW + 2;
  static constexpr int LH = TH + 2;
  static constexpr int PLANE = LW * LH;                 // pixels per LDS plane
  static constexpr int IN_LOADS = 4 * PLANE;            // 16-byte lane loads per chunk (2 groups x hi/lo)
  static constexpr int NI = (IN_LOADS + 255) / 256;     // DMA slots per wave (4 waves x 64 lanes x 16 B each)
  static constexpr int IN_BYTES = NI * 4096;            // padded: every wave issues every slot, no branches
  static constexpr int W_BYTES = 9 * 2 * 2 * MT * 16;   // [tap][hi,lo][kg][MT] x 16 B
  static constexpr int NWJ = (W_BYTES + 4095) / 4096;
  static constexpr int W_PAD = NWJ * 4096;
  static constexpr int STAGE = IN_BYTES + W_PAD;
  static constexpr int LDS_BYTES = 2 * STAGE;
  static constexpr int MTB = MT / 32;
  static constexpr int NS = NI + NWJ;
};

// Persistent kernel: a workgroup walks tiles blockIdx.x, +gridDim.x, ... and runs ONE software pipeline over the
// flattened (tile, K-chunk) steps: the DMA of step s+1 (possibly the next tile's first chunk) is issued while
// step s is multiplied, so the load latency is exposed once per workgroup, not once per tile, and the epilogue
// stores of a tile overlap the next tile's first loads.
template <int MT, int NBW, int MBW>
__global__ __launch_bounds__(256, 1) void conv_hs_kernel(ConvHsArgs a) {
  using G = HsGeom<MT, NBW, MBW>;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HpWp = a.Hp * a.Wp;
  const int nch = (a.G0 + a.G1) / 2;
  const int ntiles = a.nct * a.tilesX * a.tilesY * a.B;

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
  auto decode = [&](int t) {
    Tile T;
    T.ct = t % a.nct;
    t /= a.nct;
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
    if (slot < G::NI) {
      glds16b(src + ioff[slot], lstage + (wave + 4 * slot) * 1024);
    } else {
      const int j = wave + 4 * (slot - G::NI);
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
  auto body = [&](auto more_tag, int stage, const char* nsrc, const char* nw) {
    constexpr bool MORE = decltype(more_tag)::value;
    char* nstage = lds + (stage ^ 1) * G::STAGE;
    const char* lb = lds + stage * G::STAGE + b_lane;
    const char* la = lds + stage * G::STAGE + a_lane;
    Frags fr[2];
    load_frags(fr[0], la, lb, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const Frags& f = fr[tap & 1];
      if (tap + 1 < 9) load_frags(fr[(tap + 1) & 1], la, lb, tap + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads here (the scheduler would sink them to their use)
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
    }
  };

  const int Gout = a.nct * (MT / 8);
  auto epilogue = [&](const Tile& T) {
#pragma unroll
    for (int m = 0; m < G::MTB; ++m) {
      float bias[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = a.bias[T.ct * MT + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];
#pragma unroll
      for (int n = 0; n < NBW; ++n) {
        const int y = T.y0 + (wave * NBW + n) * G::MBH + py;
        const int x = T.x0 + px;
        const bool ok = (y < a.H) && (x < a.W);
        // hi/lo pairs packed two channels per dword: hp[q][0..1] = channels (r&3)=0..3 of 8-channel group q
        unsigned hp[4][2], lp[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float v0 = acc[m][n][q * 4 + e * 2] * a.inv_scale + bias[q * 4 + e * 2];
            float v1 = acc[m][n][q * 4 + e * 2 + 1] * a.inv_scale + bias[q * 4 + e * 2 + 1];
            v0 = (v0 > 0.f ? v0 : v0 * a.slope) * HS_ASCALE;
            v1 = (v1 > 0.f ? v1 : v1 * a.slope) * HS_ASCALE;
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
            h2 hh = {h0, h1}, ll = {l0, l1};
            hp[q][e] = __builtin_bit_cast(unsigned, hh);
            lp[q][e] = __builtin_bit_cast(unsigned, ll);
          }
        // pair groups (0,1) and (2,3): after the swaps lanes 0-31 hold all 8 channels of the even group,
        // lanes 32-63 all 8 channels of the odd group (vdst = even-group register, src = odd-group register).
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          unsigned rec[8];  // hi[0..3] dwords, lo[0..3] dwords of one 32-byte record
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
            const int g = T.ct * (MT / 8) + m * 4 + 2 * qp + kg;
            uint4* o =
                reinterpret_cast<uint4*>(a.out + ((((size_t)T.b * Gout + g) * a.Hp + (y + 1)) * a.Wp + (x + 1)) * 32);
            o[0] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
            o[1] = make_uint4(rec[4], rec[5], rec[6], rec[7]);
          }
        }
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
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
      ntile = tile + gridDim.x;
    }
    const bool has_next = ntile < ntiles;
    Tile nxt = cur;
    if (nchk == 0 && has_next) nxt = decode(ntile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (has_next) {
      body(std::true_type{}, stage, chunk_src(nxt, nchk), chunk_w(nxt, nchk));
    } else {
      body(std::false_type{}, stage, nullptr, nullptr);
    }
    if (ch == nch - 1) {
      epilogue(cur);
      zero_acc();
    }
    if (!has_next) break;
    tile = ntile;
    ch = nchk;
    cur = nxt;
    stage ^= 1;
  }
}

template <int MT, int NBW, int MBW>
static int launch_hs_cfg(const ConvHsArgs& a0, int B, hipStream_t s) {
  using G = HsGeom<MT, NBW, MBW>;
  static bool attr_set = false;
  if (!attr_set) {
    PNPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_hs_kernel<MT, NBW, MBW>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set = true;
  }
  ConvHsArgs a = a0;
  a.tilesX = (a.W + G::TW - 1) / G::TW;
  a.tilesY = (a.H + G::TH - 1) / G::TH;
  a.B = B;
  const long long ntiles = (long long)a.nct * a.tilesX * a.tilesY * B;
  // persistent: one workgroup per CU slot (LDS footprint decides how many fit), each walks ntiles/grid tiles
  const int per_cu = (G::LDS_BYTES <= 80 * 1024) ? 2 : 1;
  long long grid = 256LL * per_cu;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL((conv_hs_kernel<MT, NBW, MBW>), dim3((unsigned)grid), dim3(256), G::LDS_BYTES, s, a);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

template <int MT>
static int launch_hs_mt(const ConvHsArgs& a, int B, hipStream_t s) {
  const int mbw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : 8);
  auto blocks = [&](int nbw) {
    const int th = 4 * nbw * (32 / mbw);
    return (long long)a.nct * ((a.W + mbw - 1) / mbw) * ((a.H + th - 1) / th) * B;
  };
  // one workgroup per CU is resident: want >= ~3 waves of 256 workgroups, else shrink the tile
  int nbw = 4;
  if (blocks(4) < 700) nbw = 2;
  if (nbw == 2 && blocks(2) < 700) nbw = 1;
  if (mbw == 32) {
    if (nbw == 4) return launch_hs_cfg<MT, 4, 32>(a, B, s);
    if (nbw == 2) return launch_hs_cfg<MT, 2, 32>(a, B, s);
    return launch_hs_cfg<MT, 1, 32>(a, B, s);
  }
  if (mbw == 16) {
    if (nbw == 4) return launch_hs_cfg<MT, 4, 16>(a, B, s);
    if (nbw == 2) return launch_hs_cfg<MT, 2, 16>(a, B, s);
    return launch_hs_cfg<MT, 1, 16>(a, B, s);
  }
  if (nbw == 4) return launch_hs_cfg<MT, 4, 8>(a, B, s);
  if (nbw == 2) return launch_hs_cfg<MT, 2, 8>(a, B, s);
  return launch_hs_cfg<MT, 1, 8>(a, B, s);
}

int launch_conv_hs(const ConvLayerHs& L, const char* in0, int G0, const char* in1, int G1, char* out, int B, int H,
                   int W, hipStream_t s) {
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
  a.H = H;
  a.W = W;
  a.Hp = H + 2;
  a.Wp = W + 2;
  a.nct = L.cout / L.mt;
  a.inv_scale = L.inv_scale;
  a.slope = 0.2f;
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
