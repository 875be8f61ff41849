// 3x3 / pad 1 / stride 1 convolution + bias + LeakyReLU as an fp32-MFMA implicit GEMM for gfx950.
//
// Replaces the 27 ConvLayer instances of the reference UNet (tfpnp/pnp/denoiser/models/unet.py:8-31):
// Conv2d(k=3, padding=1, bias) followed by LeakyReLU(0.2).  fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32): the reference's 1e-4 parity bar rules out bf16/fp16 operands.
//
// GEMM view per image:  D[cout][pixel] = sum_{tap, cin} Wt[cout][tap, cin] * X[tap, cin][pixel]
//   M = cout  (A operand = weights),  N = pixels (B operand = shifted input),  K = 9 * Cin.
// D rows = cout means each accumulator register holds 32 consecutive pixels of one output channel
// across lanes 0..31 -> every epilogue store is a fully coalesced 128-byte row segment.
//
// Work decomposition (256 threads = 4 waves, one per SIMD):
//   workgroup tile = MT output channels x (4*NBW) pixel blocks; a pixel block is 32 pixels laid out as
//   MBW wide x (32/MBW) high; blocks are stacked vertically, so the tile is MBW x (4*NBW*32/MBW) pixels.
//   wave w owns NBW pixel blocks and all MT channels: MT/32 x NBW accumulators of 32x32 (16 VGPRs each).
// K loop: Cin is consumed in chunks of CC channels.  Per chunk the (TH+2) x (TW+2) x CC input halo and the
//   [9][CC][MT] weight slice are copied global->LDS by LDS-DMA (global_load_lds, no VGPR round trip):
//   inputs as a dword gather (LDS image is planar [cc][hy][hx], so a lane's 32 pixels are bank-conflict
//   free for every tap), weights as a linear 16-byte copy of a block pre-packed on the host.
//   Two LDS stages: chunk c+1 is in flight while chunk c is multiplied; one barrier per chunk.  The DMA issue
//   instructions of chunk c+1 are interleaved with the taps of chunk c (PMC: with all copies issued up front
//   the MFMA pipe idled 18-30 % because co-resident workgroups run in lockstep and issue at the same time).
// Zero padding comes from the padded activation layout (common.h): no bounds checks on loads.  Tiles that
// overhang the image read whatever follows in the arena and mask their stores (columns of D are
// independent, so garbage inputs only reach masked outputs).
#include "common.h"
#include "conv3x3.h"

namespace pnpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds4(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 4, 0, 0);
}
__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

template <int MT, int NBW, int CC, int MBW, int NTAP = 9>
struct ConvGeom {
  static constexpr int MBH = 32 / MBW;
  static constexpr int NBLK = 4 * NBW;
  static constexpr int TW = MBW;
  static constexpr int TH = NBLK * MBH;
  static constexpr int LW = TW + 2;
  static constexpr int LH = TH + 2;
  static constexpr int PLANE = LW * LH;
  static constexpr int IN_ELEMS = CC * PLANE;
  static constexpr int IN_INSTR = (IN_ELEMS + 63) / 64;
  static constexpr int IN_PAD = IN_INSTR * 64;
  static constexpr int NI = (IN_INSTR + 3) / 4;
  static constexpr int W_ELEMS = NTAP * CC * MT;      // NTAP 1: a 1x1 convolution = the centre tap alone
  static constexpr int W_INSTR = (W_ELEMS + 255) / 256;
  static constexpr int W_PAD = W_INSTR * 256;
  static constexpr int STAGE = IN_PAD + W_PAD;
  static constexpr int MTB = MT / 32;
};

template <int MT, int NBW, int CC, int MBW, int NTAP = 9>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(ConvArgs a) {
  using G = ConvGeom<MT, NBW, CC, MBW, NTAP>;
  __shared__ __attribute__((aligned(16))) float lds[2 * G::STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- block -> (cout tile, x tile, y tile, image); cout tile fastest so that a cout slice of the
  //      weights stays on one XCD's L2 (blocks are dealt round-robin to the 8 XCDs).
  int t = blockIdx.x;
  const int ct = t % a.nct;
  t /= a.nct;
  const int tx = t % a.tilesX;
  t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int b = t / a.tilesY;
  const int x0 = tx * G::TW, y0 = ty * G::TH;
  const int HpWp = a.Hp * a.Wp;
  const int nch = (a.C0 + a.C1) / CC;

  // ---- per-thread gather offsets for the input halo (same for every chunk)
  int ioff[G::NI];
#pragma unroll
  for (int k = 0; k < G::NI; ++k) {
    const int idx = (wave + 4 * k) * 64 + lane;
    const int c = idx / G::PLANE;
    const int r = idx - c * G::PLANE;
    const int hy = r / G::LW;
    const int hx = r - hy * G::LW;
    ioff[k] = (idx < G::IN_ELEMS) ? c * HpWp + hy * a.Wp + hx : 0;
  }
  const size_t tile_org = (size_t)y0 * a.Wp + x0 + (PADL - 1);
  const float* wbase = a.wpk + (size_t)ct * nch * G::W_ELEMS;

  // One K-chunk's copies are NS DMA "slots" per wave: NI input gathers (4 B/lane) + NWJ weight copies (16 B/lane).
  // Issuing them costs the wave tens of cycles each, so they are spread over the 9 taps of the PREVIOUS chunk's
  // multiply (slot s is issued before tap s % 9): the MFMA pipe keeps running underneath.
  constexpr int NWJ = (G::W_INSTR + 3) / 4;
  constexpr int NS = G::NI + NWJ;
  auto chunk_src = [&](int chunk) -> const float* {
    const int c0 = chunk * CC;
    const float* src = (c0 < a.C0) ? a.in0 + ((size_t)b * a.C0 + c0) * HpWp
                                   : a.in1 + ((size_t)b * a.C1 + (c0 - a.C0)) * HpWp;
    return src + tile_org;
  };
  auto issue_slot = [&](int slot, const float* src, const float* wsrc, float* lstage) {
    if (slot < G::NI) {
      const int instr = wave + 4 * slot;
      if (instr < G::IN_INSTR) glds4(src + ioff[slot], lstage + instr * 64);
    } else {
      const int j = wave + 4 * (slot - G::NI);
      if (j < G::W_INSTR) glds16(wsrc + j * 256 + lane * 4, lstage + G::IN_PAD + j * 256);
    }
  };

  f32x16 acc[G::MTB][NBW];
#pragma unroll
  for (int m = 0; m < G::MTB; ++m)
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int l31 = lane & 31, khalf = lane >> 5;
  const int py = l31 / MBW, px = l31 - py * MBW;
  const int b_lane = khalf * G::PLANE + (wave * NBW * G::MBH + py) * G::LW + px;
  const int a_lane = khalf * MT + l31;

  {
    const float* src = chunk_src(0);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) issue_slot(sl, src, wbase, lds);
  }
  for (int ch = 0; ch < nch; ++ch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = ch + 1 < nch;
    const float* nsrc = more ? chunk_src(ch + 1) : nullptr;
    const float* nw = wbase + (size_t)(ch + 1) * G::W_ELEMS;
    float* nstage = lds + ((ch + 1) & 1) * G::STAGE;
    const float* lin = lds + (ch & 1) * G::STAGE + b_lane;
    const float* lw = lds + (ch & 1) * G::STAGE + G::IN_PAD + a_lane;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int dy = NTAP == 1 ? 1 : tap / 3, dx = NTAP == 1 ? 1 : tap % 3;
      if (more) {
#pragma unroll
        for (int sl = tap; sl < NS; sl += NTAP) issue_slot(sl, nsrc, nw, nstage);
      }
#pragma unroll
      for (int cp = 0; cp < CC / 2; ++cp) {
        float av[G::MTB], bv[NBW];
#pragma unroll
        for (int m = 0; m < G::MTB; ++m) av[m] = lw[(tap * CC + cp * 2) * MT + m * 32];
#pragma unroll
        for (int n = 0; n < NBW; ++n) bv[n] = lin[(cp * 2) * G::PLANE + (n * G::MBH + dy) * G::LW + dx];
#pragma unroll
        for (int m = 0; m < G::MTB; ++m)
#pragma unroll
          for (int n = 0; n < NBW; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias + LeakyReLU, coalesced row stores into the padded output
  const int Cout = a.nct * MT;
#pragma unroll
  for (int m = 0; m < G::MTB; ++m) {
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bias[r] = a.mode == 0 ? a.bias[ct * MT + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf] : 0.f;
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      const int y = y0 + (wave * NBW + n) * G::MBH + py;
      const int x = x0 + px;
      if (y < a.H && x < a.W) {
        const size_t off = (((size_t)b * Cout + ct * MT + m * 32 + 4 * khalf) * a.Hp + (y + 1)) * a.Wp + x + PADL;
        float* o = a.out + off;
        unsigned hsbits = 0;   // mode 4: bit r = saved activation of register r's channel is > 0
        if (a.mode == 4) {
          const int g0 = (ct * MT + m * 32) >> 3;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const size_t rec = (((size_t)b * (Cout >> 3) + g0 + q) * (a.H + 2) + (y + 1)) * (size_t)(a.W + 2) + (x + 1);
            const uint2 w = *reinterpret_cast<const uint2*>(a.dmask_hs + rec * 32 + 8 * khalf);   // hi[4*khalf .. +3]
            const unsigned short hh[4] = {(unsigned short)(w.x & 0xffff), (unsigned short)(w.x >> 16),
                                          (unsigned short)(w.y & 0xffff), (unsigned short)(w.y >> 16)};
#pragma unroll
            for (int j = 0; j < 4; ++j)   // f16 > 0: sign bit clear and not (+)zero
              hsbits |= (unsigned)((hh[j] & 0x8000u) == 0 && (hh[j] & 0x7fffu) != 0) << (q * 4 + j);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const size_t ro = (size_t)((r & 3) + 8 * (r >> 2)) * HpWp;
          float v = acc[m][n][r] + bias[r];
          if (a.mode == 0) {
            v = v > 0.f ? v : v * a.slope;
            if (a.res) v += a.res[off + ro];
          } else if (a.mode == 2) v = a.dmask[off + ro] > 0.f ? v : v * a.slope;
          else if (a.mode == 4) v = ((hsbits >> r) & 1u) ? v : v * a.slope;
          o[ro] = v;
        }
      }
    }
  }
}

template <int MT, int NBW, int CC, int MBW, int NTAP = 9>
static int launch_cfg(const ConvArgs& a0, int B, hipStream_t s) {
  using G = ConvGeom<MT, NBW, CC, MBW, NTAP>;
  ConvArgs a = a0;
  a.tilesX = (a.W + G::TW - 1) / G::TW;
  a.tilesY = (a.H + G::TH - 1) / G::TH;
  const long long grid = (long long)a.nct * a.tilesX * a.tilesY * B;
  hipLaunchKernelGGL((conv3x3_mfma_kernel<MT, NBW, CC, MBW, NTAP>), dim3((unsigned)grid), dim3(256), 0, s, a);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

template <int MT, int CC>
static int launch_mt_cc(const ConvArgs& a, int B, hipStream_t s) {
  // pixel-block shape by image width; 2 blocks per wave unless the grid would be too small to fill
  // 256 CUs twice over.
  const int mbw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : 8);
  auto blocks = [&](int nbw) {
    const int th = 4 * nbw * (32 / mbw);
    return (long long)a.nct * ((a.W + mbw - 1) / mbw) * ((a.H + th - 1) / th) * B;
  };
  const int nbw = (blocks(2) >= 1024) ? 2 : 1;
  if (mbw == 32) return nbw == 2 ? launch_cfg<MT, 2, CC, 32>(a, B, s) : launch_cfg<MT, 1, CC, 32>(a, B, s);
  if (mbw == 16) return nbw == 2 ? launch_cfg<MT, 2, CC, 16>(a, B, s) : launch_cfg<MT, 1, CC, 16>(a, B, s);
  return nbw == 2 ? launch_cfg<MT, 2, CC, 8>(a, B, s) : launch_cfg<MT, 1, CC, 8>(a, B, s);
}

// 1x1 convolution (+ bias, LeakyReLU(slope), optional residual) on the same kernel with the centre tap alone: cout % 64 == 0,
// cin % 8 == 0, weights packed by pack_conv_weights_1x1 (the DRUNet's strided / transposed 2x2 layers in fp32 mode)
int launch_conv1x1_act(const ConvLayer& L, const float* in0, float* out, int B, int H, int W, float slope, const float* res, hipStream_t s) {
  if (L.mt != 64 || L.cc != 8 || L.cin % 8 != 0 || L.cout % 64 != 0) {
    set_error("conv1x1: unsupported packing (cin %d cout %d mt %d cc %d)", L.cin, L.cout, L.mt, L.cc);
    return PNPX_ERR_SHAPE;
  }
  ConvArgs a;
  a.in0 = in0;
  a.C0 = L.cin;
  a.in1 = in0;
  a.C1 = 0;
  a.wpk = L.w;
  a.bias = L.b;
  a.out = out;
  a.H = H;
  a.W = W;
  a.Hp = padded_h(H);
  a.Wp = padded_w(W);
  a.nct = L.cout / 64;
  a.slope = slope;
  a.mode = 0;
  a.dmask = nullptr;
  a.res = res;
  const int mbw = W >= 32 ? 32 : (W >= 16 ? 16 : 8);
  if (mbw == 32) return launch_cfg<64, 2, 8, 32, 1>(a, B, s);
  if (mbw == 16) return launch_cfg<64, 2, 8, 16, 1>(a, B, s);
  return launch_cfg<64, 2, 8, 8, 1>(a, B, s);
}

// w[cout][cin] -> [cout/64][cin/8][8][64]
void pack_conv_weights_1x1(const float* w, int cout, int cin, float* dst) {
  const int nct = cout / 64, nch = cin / 8;
  for (int ct = 0; ct < nct; ++ct)
    for (int ch = 0; ch < nch; ++ch)
      for (int c = 0; c < 8; ++c)
        for (int m = 0; m < 64; ++m) dst[(((size_t)ct * nch + ch) * 8 + c) * 64 + m] = w[(size_t)(ct * 64 + m) * cin + ch * 8 + c];
}

int conv_pack_mt(int cout) { return cout >= 64 ? 64 : 32; }
int conv_pack_cc(int cin) { return cin % 8 == 0 ? 8 : 2; }

int launch_conv3x3(const ConvLayer& L, const float* in0, int C0, const float* in1, int C1, float* out, int B,
                   int H, int W, hipStream_t s) {
  return launch_conv3x3_act(L, in0, C0, in1, C1, out, B, H, W, 0.2f, nullptr, s);
}

int launch_conv3x3_act(const ConvLayer& L, const float* in0, int C0, const float* in1, int C1, float* out, int B,
                       int H, int W, float slope, const float* res, hipStream_t s) {
  if (C0 + C1 != L.cin || C0 % L.cc != 0 || C1 % L.cc != 0 || L.cout % L.mt != 0) {
    set_error("conv3x3: channel split %d+%d incompatible with packed layer (cin %d, cc %d)", C0, C1, L.cin, L.cc);
    return PNPX_ERR_SHAPE;
  }
  ConvArgs a;
  a.in0 = in0;
  a.C0 = C0;
  a.in1 = in1 ? in1 : in0;
  a.C1 = C1;
  a.wpk = L.w;
  a.bias = L.b;
  a.out = out;
  a.H = H;
  a.W = W;
  a.Hp = padded_h(H);
  a.Wp = padded_w(W);
  a.nct = L.cout / L.mt;
  a.slope = slope;
  a.mode = 0;
  a.dmask = nullptr;
  a.res = res;
  if (L.mt == 64 && L.cc == 8) return launch_mt_cc<64, 8>(a, B, s);
  if (L.mt == 32 && L.cc == 8) return launch_mt_cc<32, 8>(a, B, s);
  if (L.mt == 32 && L.cc == 2) return launch_mt_cc<32, 2>(a, B, s);
  set_error("conv3x3: no kernel for mt=%d cc=%d", L.mt, L.cc);
  return PNPX_ERR_SHAPE;
}

int launch_conv3x3_grad(const ConvLayer& L, const float* gin, float* gout, const float* dmask, int B, int H, int W,
                        hipStream_t s, const char* dmask_hs, float mask_slope) {
  ConvArgs a;
  a.in0 = gin;
  a.C0 = L.cin;
  a.in1 = gin;
  a.C1 = 0;
  a.wpk = L.w;
  a.bias = nullptr;
  a.out = gout;
  a.H = H;
  a.W = W;
  a.Hp = padded_h(H);
  a.Wp = padded_w(W);
  a.nct = L.cout / L.mt;
  a.slope = mask_slope;
  a.mode = dmask_hs ? 4 : (dmask ? 2 : 1);
  a.dmask = dmask;
  a.dmask_hs = dmask_hs;
  if (L.cc != 8 || L.cin % 8 != 0 || L.cout % L.mt != 0) {
    set_error("conv3x3_grad: unsupported packing (cin %d cout %d mt %d cc %d)", L.cin, L.cout, L.mt, L.cc);
    return PNPX_ERR_SHAPE;
  }
  if (L.mt == 64) return launch_mt_cc<64, 8>(a, B, s);
  return launch_mt_cc<32, 8>(a, B, s);
}

void pack_conv_weights_transposed(const float* w, int cout, int cin, int cout_pad, int mt, int cc, float* dst) {
  // adjoint conv: input channels = cout (forward outputs), output channels = cin (padded to cout_pad)
  const int nct = cout_pad / mt, nch = cout / cc;
  for (int ct = 0; ct < nct; ++ct)
    for (int ch = 0; ch < nch; ++ch)
      for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < cc; ++c)
          for (int m = 0; m < mt; ++m) {
            const int ci = ct * mt + m;      // forward input channel = adjoint output channel
            const int co = ch * cc + c;      // forward output channel = adjoint input channel
            const float v = (ci < cin) ? w[((size_t)co * cin + ci) * 9 + (8 - tap)] : 0.f;
            dst[((((size_t)ct * nch + ch) * 9 + tap) * cc + c) * mt + m] = v;
          }
}

// Host-side repack: w[cout][cin][3][3] -> [cout/mt][cin/cc][tap][cc][mt]
void pack_conv_weights(const float* w, int cout, int cin, int mt, int cc, float* dst) {
  const int nct = cout / mt, nch = cin / cc;
  for (int ct = 0; ct < nct; ++ct)
    for (int ch = 0; ch < nch; ++ch)
      for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < cc; ++c)
          for (int m = 0; m < mt; ++m) {
            const int co = ct * mt + m, ci = ch * cc + c;
            dst[((((size_t)ct * nch + ch) * 9 + tap) * cc + c) * mt + m] = w[((size_t)co * cin + ci) * 9 + tap];
          }
}

}  // namespace pnpx
