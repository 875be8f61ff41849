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
// Epilogue: bias + LeakyReLU in fp32, re-split to hi/lo; the weight rows of each 32-cout block are packed in the order
// that leaves 16 consecutive channels in every lane (hs_row_channel), so a lane owns two whole 8-channel records and
// stores them with two 16-byte buffer stores each -- no cross-lane traffic.
#include <cstdlib>
#include <cstring>

#include "conv_hs_kernel.h"

namespace pnpx {

bool conv_hs_can_pool(int H, int W) { return W >= 32 && (W % 2) == 0 && (H % 2) == 0; }

#ifdef PNPX_TUNING
// Tuning builds only (libpnpx_tune.so, `make tuning`): PNPX_HS_<MT>_<W>="nbw,nw" overrides the launch table.
static bool tuning_override(int mt, int W, HsChoice* c) {
  char key[64];
  snprintf(key, sizeof(key), "PNPX_HS_%d_%d", mt, W);
  if (const char* e = getenv(key)) {
    int x, y;
    if (sscanf(e, "%d,%d", &x, &y) == 2) {
      *c = HsChoice{x, y};
      return true;
    }
  }
  return false;
}
#endif

// Launch table.  Tile = MT couts x (NW waves x NBW blocks of 32 pixels).
//  * NBW (tile height) by a small cost model: 256 persistent workgroups process ceil(tiles/256) rounds of tiles whose
//    cost is ~ (blocks + 0.3) (0.3 = per-tile prologue/epilogue/barrier overhead in units of one 128-pixel block row,
//    calibrated with tools/tune_hs.py).  Matters for batch sizes that do not fill the rounds (idx_left compaction).
//  * NW: 4 waves (one per SIMD, the whole register file each) or 8 waves (two per SIMD, 256 registers each: one
//    wave's ds_read / DMA issue / epilogue overlaps the other's MFMAs).
HsChoice hs_choose(int mt, const ConvHsArgs& a, int B) {
  const int mbw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : 8);
  auto blocks = [&](int rows) {   // rows = NW * NBW block rows per tile
    const int th = rows * (32 / mbw);
    return (long long)a.nct * ((a.W + mbw - 1) / mbw) * ((a.H + th - 1) / th) * B;
  };
  // r5: beside `share - 1` other launch chains (small and medium batches, unet.hip launch_chains) the chains drift out of phase and their
  // workgroups pack: a launch then costs its FRACTION of a round of 256 / share workgroups (at least one tile time), not a whole number
  // of rounds of 256 -- and a tile's two waves per SIMD pay even in a one-tile workgroup (one wave's loads and epilogue hide behind the
  // other's MFMAs while the other chain keeps the remaining CUs busy).  Measured against the lone-launch table (tools/tune_rule.sh, wall
  // clock of whole forwards): B = 6 -1.7 %, 10 -4.0 %, 12 -4.7 %, 18 -6.1 %, 24 -6.3 %, 38 -8.3 %, 44 -6.1 %; whole rounds of 128
  // workgroups instead (rule 1) lose 2.8 % at B = 10 / 12.  A lone launch (share = 1: B <= 4, full rounds, B >= 47) plans as before.
  const int share = a.share > 1 ? a.share : 1;
  int rule = 6;      // bit 1: two waves per SIMD beside another chain; bit 2: fractional rounds of 256 / share; bit 0 (experiment): whole rounds of 256 / share
#ifdef PNPX_TUNING
  if (const char* e = getenv("PNPX_HS_RULE")) rule = atoi(e);
#endif
  const long long budget = (rule & 1) ? 256 / share : 256;
  const bool smooth = (rule & 4) != 0 && share > 1;
  auto rounds = [&](long long nt) {
    if (smooth) {
      const double r = (double)nt * share / 256.0;
      return r < 1.0 ? 1.0 : r;
    }
    return (double)((nt + budget - 1) / budget);
  };
  HsChoice c{4, 4};
  double best = 1e30;
  // (half tiles -- 32 couts over a 64-cout packing, deep levels of small batches -- with one block per wave are bound by the LDS-DMA
  // latency of a 27-MFMA step, not by its MFMAs: beside another chain a 4-row tile costs what an 8-row tile costs.  32-pixel-wide
  // blocks only: at the 16-pixel level the 8-row tile measured slower.  tools/tune_small.sh 6: -2.3 % per forward at B = 6.)
  const bool latency_bound_rows4 = share > 1 && mt == 32 && a.w_mt == 64 && mbw == 32;
  for (int rows : {16, 8, 4}) {
    const double cost = rounds(blocks(rows)) * ((rows == 4 && latency_bound_rows4 ? 2 : rows / 4) + 0.3);
    if (cost < best * 0.999) {
      best = cost;
      c.nbw = rows / 4;
    }
  }
  if (mt == 32 && a.w_mt == 32) {   // 32-cout layers: 16-row tiles unless a half-empty last round makes 8-row tiles cheaper
    const double c16 = rounds(blocks(16)) * 4.3, c8 = rounds(blocks(8)) * 2.3;
    c.nbw = (c8 < 0.9 * c16) ? 2 : 4;
  }
  // two waves per SIMD (same tile, half the blocks per wave) once every workgroup has at least two tiles to walk -- or, beside another
  // chain, always: one wave's loads and epilogue then hide behind the other's MFMAs even in a one-tile workgroup
  if (c.nbw >= 2 && (blocks(4 * c.nbw) >= 512 || (share > 1 && (rule & 2))) && !(a.pool_out && c.nbw < 4)) c = HsChoice{c.nbw / 2, 8};
#ifdef PNPX_TUNING
  tuning_override(mt, a.W, &c);
#endif
  if (a.pool_out && c.nbw < 2) c.nbw = 2;   // the fused 2x2 pool pairs two rows held by one wave
  return c;
}

int launch_conv_hs(const ConvLayerHs& L, const char* in0, int G0, const char* in1, int G1, char* out, int B, int H,
                   int W, const ConvHsFuse& fuse, hipStream_t s) {
  if ((G0 + G1) * 8 != L.cin_pad || (G0 & 1) || (G1 & 1)) {
    set_error("conv_hs: channel groups %d+%d incompatible with packed layer (cin_pad %d)", G0, G1, L.cin_pad);
    return PNPX_ERR_SHAPE;
  }
  if (L.cout > 1024 || !(fuse.slope >= 0.f && fuse.slope <= 1.f)) {   // 4 KiB LDS bias table
    set_error("conv_hs: cout %d > 1024 or activation slope %g outside [0, 1]", L.cout, (double)fuse.slope);
    return PNPX_ERR_SHAPE;
  }
  if ((long long)(L.cout / 8) * (H + 2) * (W + 2) * 32 >= (1LL << 31)) {
    set_error("conv_hs: one image's output tensor exceeds 2 GiB (%d channels, %d x %d)", L.cout, H, W);
    return PNPX_ERR_SHAPE;
  }
  ConvHsArgs a;
  a.in0 = in0;
  a.G0 = G0;
  a.in1 = in1 ? in1 : in0;
  a.G1 = G1;
  a.G0t = fuse.in0_groups > 0 ? fuse.in0_groups : G0;
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
  a.w_mt = L.mt;
  a.inv_scale = L.inv_scale;
  a.slope = fuse.slope;
  a.dmask = fuse.dmask;
  a.res = fuse.res;
  a.neg_one = -1.0f;
  a.range_flag = fuse.range_flag;
  a.trace = nullptr;
  a.wgt = nullptr;
  a.abl = 0;
  a.share = fuse.share;
#ifdef PNPX_TUNING
  if (const char* e = getenv("PNPX_HS_ABL")) a.abl = atoi(e);
#endif
  a.tilesX = a.tilesY = 0;
  a.B = B;
  a.ups_h = a.ups_w = 0;
  a.ups_sy = a.ups_sx = 0.f;
  a.first_x = a.first_sigma = a.first_w = a.first_b = a.first_zero = nullptr;
  a.first_sigma_stride = 0;
  a.first_slope = 0.f;
  if (fuse.ups_h) {   // fused bilinear x2 of the second source: one instance, picked here
    if (!conv_hs_can_fuse_upsample(L, G0, G1, H, W) || fuse.ups_h * 2 != H || fuse.ups_w * 2 != W || !in1 || fuse.pool_out ||
        fuse.outc_w || fuse.dmask || fuse.res) {
      set_error("conv_hs: fused up-sampling is not available for this layer / geometry");
      return PNPX_ERR_SHAPE;
    }
    a.pool_out = nullptr;
    a.ups_h = fuse.ups_h;
    a.ups_w = fuse.ups_w;
    a.ups_sy = (H > 1) ? (float)(fuse.ups_h - 1) / (float)(H - 1) : 0.f;
    a.ups_sx = (W > 1) ? (float)(fuse.ups_w - 1) / (float)(W - 1) : 0.f;
    // (64-cout layers: 8-row tiles -- the 16-row tile's two stages leave no LDS for the three low-resolution windows)
    if (L.mt == 64) return launch_hs_cfg<64, 1, 32, 8, EPI_ACT, 1>(a, B, s);
    return launch_hs_cfg<32, 2, 32, 8, EPI_ACT, 1>(a, B, s);
  }
  if (L.mt != 64 && L.mt != 32) {
    set_error("conv_hs: no kernel for mt=%d", L.mt);
    return PNPX_ERR_SHAPE;
  }
  // Half tiles: when the 64-cout tiling leaves more than half of the 256 CUs without a tile (small batches, deep
  // levels), run the 32-cout instance over the same 64-cout weight packing: twice the tiles, identical K order (bit-
  // identical results), weight slice gathered by the DMA.
  int mt_run = L.mt;
  if (L.mt == 64) {
    const HsChoice c64 = hs_choose(64, a, B);
    const int mbw = W >= 32 ? 32 : (W >= 16 ? 16 : 8);
    const int th = c64.nw * c64.nbw * (32 / mbw);
    const long long tiles64 = (long long)a.nct * ((W + mbw - 1) / mbw) * ((H + th - 1) / th) * B;
    bool half = tiles64 <= 128;
#ifdef PNPX_TUNING   // PNPX_HS_HALF_<W>=1 / 0: force half tiles on / off at that width
    {
      char key[64];
      snprintf(key, sizeof(key), "PNPX_HS_HALF_%d", W);
      if (const char* e = getenv(key)) half = atoi(e) != 0;
    }
#endif
    if (half) {
      mt_run = 32;
      a.nct = L.cout / 32;
    }
  }
  if (fuse.taps != 0x1FF) {
    if (a.dmask || a.res || a.outc_w || a.pool_out || fuse.ups_h || fuse.first_x || G1 != 0) {
      set_error("conv_hs: sparse-tap layers support the plain activation epilogue and one source only");
      return PNPX_ERR_SHAPE;
    }
    return launch_conv_hs_taps(a, mt_run, fuse.taps, B, s);
  }
  if (a.dmask) return launch_conv_hs_dmask(a, mt_run, B, s);
  if (a.res) return launch_conv_hs_res(a, mt_run, B, s);
  // weights-in-registers instances: 32 -> 32 channels from one source, 32-pixel-wide blocks, enough tiles to fill the chip
  if (fuse.first_x) {   // first convolution folded into this layer's loader: one instance (checked by the caller through
                        // conv_hs_can_fold_first)
    if (!conv_hs_can_fold_first(L, G0, B, H, W, fuse) || G1 != 0) {
      set_error("conv_hs: the first convolution cannot be folded into this layer / geometry");
      return PNPX_ERR_SHAPE;
    }
    a.first_x = fuse.first_x;
    a.first_sigma = fuse.first_sigma;
    a.first_sigma_stride = fuse.first_sigma_stride;
    a.first_w = fuse.first_w;
    a.first_b = fuse.first_b;
    a.first_zero = fuse.first_zero;
    a.first_slope = fuse.first_slope;
    return launch_hs_cfg<32, 2, 32, 8, EPI_ACT, 0, 2>(a, B, s);
  }
  if (fuse.wreg && L.mt == 32 && L.cout == 32 && G0 == 4 && G1 == 0 && W >= 32) {
    const bool w8 = fuse.wreg == 2;
    const long long tiles = (long long)((W + 31) / 32) * ((H + 15) / 16) * B;   // both shapes: 16-row tiles
    // (a half-empty last round costs more than the registers buy: 384 tiles on 256 workgroups run 25 % longer than they should,
    // the generic path then picks 8-row tiles)
    const bool full_rounds = tiles * 100 >= ((tiles + 255) / 256) * 256 * 85;
    if (tiles >= 256 && full_rounds) {
      if (a.outc_w) return w8 ? launch_hs_cfg<32, 2, 32, 8, EPI_OUTC, 0, 1>(a, B, s) : launch_hs_cfg<32, 4, 32, 4, EPI_OUTC, 0, 1>(a, B, s);
      return w8 ? launch_hs_cfg<32, 2, 32, 8, EPI_ACT, 0, 1>(a, B, s) : launch_hs_cfg<32, 4, 32, 4, EPI_ACT, 0, 1>(a, B, s);
    }
  }
  if (a.outc_w) return launch_hs_mt<32, EPI_OUTC>(a, B, s);
  if (mt_run == 64) return launch_hs_mt<64, EPI_ACT>(a, B, s);
  return launch_hs_mt<32, EPI_ACT>(a, B, s);
}

bool conv_hs_can_fold_first(const ConvLayerHs& L, int G0, int B, int H, int W, const ConvHsFuse& fuse) {
  const long long tiles = (long long)((W + 31) / 32) * ((H + 15) / 16) * B;
  return fuse.wreg != 0 && L.mt == 32 && L.cout == 32 && L.cin == 32 && G0 == 4 && W >= 32 && tiles >= 256 && !fuse.pool_out &&
         !fuse.outc_w && !fuse.dmask && !fuse.res && !fuse.ups_h;
}

// The fused bilinear x2 instances: 32 output channels (16-row tiles) or 64-cout tiles (8-row tiles), 32-pixel-wide blocks, the first two
// K-chunks from the skip source.
bool conv_hs_can_fuse_upsample(const ConvLayerHs& L, int G0, int G1, int H, int W) {
  return ((L.mt == 32 && L.cout == 32) || (L.mt == 64 && L.cout % 64 == 0)) && G0 >= 4 && !(G0 & 1) && G1 >= 2 && !(G1 & 1) && W >= 32 &&
         !(H & 1) && !(W & 1);
}

// ---- host-side weight packing -------------------------------------------------------------------------
static inline uint16_t f16_bits(_Float16 h) {
  uint16_t u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}

int conv_hs_mt(int cout) { return cout >= 64 ? 64 : 32; }

// MFMA C layout (32x32): lane half kg, register r holds row (r&3) + 8*(r>>2) + 4*kg.  Weight row i of every 32-cout block
// is given channel 16*kg + r of the block, so that a lane's 16 registers are 16 consecutive channels = two whole HS8
// records (conv_hs_kernel.h::store_records) and the epilogue needs no cross-lane traffic.
static inline int hs_row_channel(int row) {
  const int kg = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
  return 16 * kg + r;
}

// w[cout][cin][3][3] fp32 -> [cout/mt][cin_pad/16][tap][hi,lo][kg][mt][8] f16 (rows permuted per 32-block, above),
// scaled by a power of two.
// Returns the scale s (weights are stored as split(s*w)).
float pack_conv_weights_hs(const float* w, int cout, int cin, int mt, uint16_t* dst) {
  return pack_conv_weights_hs_taps(w, cout, cin, mt, 0x1FF, dst);
}

float pack_conv_weights_hs_taps(const float* w, int cout, int cin, int mt, int tapmask, uint16_t* dst) {
  int taps[9], nt = 0;
  for (int t = 0; t < 9; ++t)
    if ((tapmask >> t) & 1) taps[nt++] = t;
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
      for (int ti = 0; ti < nt; ++ti)
        for (int kgi = 0; kgi < 2; ++kgi)
          for (int m = 0; m < mt; ++m)
            for (int el = 0; el < 8; ++el) {
              const int tap = taps[ti];
              const int co = ct * mt + (m & ~31) + hs_row_channel(m & 31), ci = ch * 16 + kgi * 8 + el;
              const float v = (ci < cin) ? w[((size_t)co * cin + ci) * 9 + tap] * s : 0.f;
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (float)hi);
              const size_t base = ((((size_t)ct * nch + ch) * nt + ti) * 2) * 2 * mt * 8;   // start of [hi,lo] pair
              dst[base + ((size_t)(0 * 2 + kgi) * mt + m) * 8 + el] = f16_bits(hi);
              dst[base + ((size_t)(1 * 2 + kgi) * mt + m) * 8 + el] = f16_bits(lo);
            }
  return s;
}

}  // namespace pnpx
