// Generic 2-D FFT entry (pnpx_fft2), twiddle tables and launch planning.  Kernels: fft_lds.h.
#include <cmath>

#include "fft_lds.h"

namespace pnpx {

int ctx_twiddle(pnpx_ctx* ctx, int N, const float2** out) {
  if (N < 1 || N > FFT_MAX_N) {
    set_error("FFT length %d unsupported (1..%d)", N, FFT_MAX_N);
    return PNPX_ERR_SHAPE;
  }
  auto it = ctx->twiddle.find(N);
  if (it == ctx->twiddle.end()) {
    std::vector<float2> h(N);
    for (int m = 0; m < N; ++m) {
      const double a = -2.0 * M_PI * (double)m / (double)N;
      h[m] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    float2* d = nullptr;
    PNPX_HIP(hipMalloc(&d, sizeof(float2) * N));
    PNPX_HIP(hipMemcpy(d, h.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
    it = ctx->twiddle.emplace(N, d).first;
  }
  *out = it->second;
  return PNPX_OK;
}

// Stage radices for a length that is not a power of two: 4s and 2s first, then the odd prime factors.
static int factorise(int N, PassGeom* g) {
  g->nrad = 0;
  int n = N;
  auto push = [&](int r) {
    if (g->nrad >= 12) return false;
    g->rad[g->nrad++] = (unsigned short)r;
    return true;
  };
  while (n % 4 == 0) {
    if (!push(4)) return -1;
    n /= 4;
  }
  for (int p = 2; p <= n; ++p)
    while (n % p == 0) {
      if (!push(p)) return -1;
      n /= p;
    }
  return 0;
}

static int make_pass(pnpx_ctx* ctx, int n_img, int H, int W, int N, bool centered, int lines, PassGeom* g) {
  const float2* tw;
  PNPX_TRY(ctx_twiddle(ctx, N, &tw));
  *g = PassGeom();
  g->H = H;
  g->W = W;
  g->logN = ilog2_exact(N);
  g->odd = (centered && (N & 1)) ? 1 : 0;
  if (g->logN < 0 && factorise(N, g) != 0) {
    set_error("FFT length %d has too many prime factors", N);
    return PNPX_ERR_SHAPE;
  }
  g->lines = lines;
  g->n_img = n_img;
  g->scale = (float)(1.0 / std::sqrt((double)N));
  g->centered = centered ? 1 : 0;
  g->tw = tw;
  return PNPX_OK;
}

int make_fft_plan(pnpx_ctx* ctx, int n_img, int H, int W, bool centered, FftPlan2D* P) {
  if (H < 1 || W < 1 || H > FFT_MAX_N || W > FFT_MAX_N || n_img <= 0) {
    set_error("fft2: H=%d W=%d unsupported (1..%d)", H, W, FFT_MAX_N);
    return PNPX_ERR_SHAPE;
  }
  const int total_rows = n_img * H;
  const int tile_points = ctx->opt_fft_tile > 0 ? ctx->opt_fft_tile : FFT_TILE_POINTS;
  int lr = tile_points / W;
  if (lr < 1) lr = 1;
  if (lr > total_rows) lr = total_rows;
  int lc = tile_points / H;
  if (lc < 1) lc = 1;
  if (lc > W) lc = W;
  if (lc > 64) lc = 64;
  PNPX_TRY(make_pass(ctx, n_img, H, W, W, centered, lr, &P->rows));
  PNPX_TRY(make_pass(ctx, n_img, H, W, H, centered, lc, &P->cols));
  // XCD-affine image mapping (fft_lds.h): every pass of a chain keeps an image on one XCD, so the k-space round trips
  // between passes are served by that XCD's L2
  if (ctx->opt_fft_affine && n_img >= 8) {
    P->rows.affine = (H % lr == 0) ? 1 : 0;
    P->cols.affine = 1;
  }
  P->grid_rows = dim3((total_rows + lr - 1) / lr);
  P->grid_cols = dim3((W + lc - 1) / lc, n_img);
  P->lds_rows = sizeof(float2) * 2 * (size_t)lr * (W + 1);
  P->lds_cols = sizeof(float2) * 2 * (size_t)lc * (H + 1);
  // 256-point lines: the register-radix-16 kernels (16-line tiles; option fft_fast, default on)
  if (ctx->opt_fft_fast) {
    P->fast256_rows = (W == 256 && H % FFT256_LINES == 0);
    P->fast256_cols = (H == 256 && W % FFT256_LINES == 0);
    if (ctx->opt_fft_affine && n_img >= 8) {
      if (P->fast256_rows) P->rows.affine = 1;
      if (P->fast256_cols) P->cols.affine = 1;
    }
  }
  return PNPX_OK;
}

int fft2(pnpx_ctx* ctx, const float* in, float* out, int n_img, int H, int W, bool inverse, bool centered,
         hipStream_t s) {
  FftPlan2D P;
  PNPX_TRY(make_fft_plan(ctx, n_img, H, W, centered, &P));
  LoadC li{reinterpret_cast<const float2*>(in), H, W};
  LoadC lo{reinterpret_cast<const float2*>(out), H, W};
  StoreC so{reinterpret_cast<float2*>(out), H, W};
  if (!inverse) {
    PNPX_TRY((launch_rows<false>(P, li, so, s)));
    PNPX_TRY((launch_cols<false, false>(P, lo, MidNone(), so, s)));
  } else {
    PNPX_TRY((launch_rows<true>(P, li, so, s)));
    PNPX_TRY((launch_cols<true, false>(P, lo, MidNone(), so, s)));
  }
  return PNPX_OK;
}

}  // namespace pnpx
