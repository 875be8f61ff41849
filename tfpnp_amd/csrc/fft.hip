// Generic 2-D FFT entry (pnpx_fft2), twiddle tables and launch planning.  Kernels: fft_lds.h.
#include <cmath>

#include "fft_lds.h"

namespace pnpx {

int ctx_twiddle(pnpx_ctx* ctx, int N, const float2** out) {
  const int lg = ilog2_exact(N);
  if (lg < 0) {
    set_error("FFT length %d unsupported (power of two in [2,1024] required)", N);
    return PNPX_ERR_SHAPE;
  }
  if (!ctx->twiddle[lg]) {
    std::vector<float2> h(N);
    for (int m = 0; m < N; ++m) {
      const double a = -2.0 * M_PI * (double)m / (double)N;
      h[m] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    float2* d = nullptr;
    PNPX_HIP(hipMalloc(&d, sizeof(float2) * N));
    PNPX_HIP(hipMemcpy(d, h.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
    ctx->twiddle[lg] = d;
  }
  *out = ctx->twiddle[lg];
  return PNPX_OK;
}

int make_fft_plan(pnpx_ctx* ctx, int n_img, int H, int W, bool centered, FftPlan2D* P) {
  const int lw = ilog2_exact(W), lh = ilog2_exact(H);
  if (lw < 0 || lh < 0 || n_img <= 0) {
    set_error("fft2: H=%d W=%d unsupported (powers of two in [2,1024] required)", H, W);
    return PNPX_ERR_SHAPE;
  }
  const float2 *tww, *twh;
  PNPX_TRY(ctx_twiddle(ctx, W, &tww));
  PNPX_TRY(ctx_twiddle(ctx, H, &twh));
  const int total_rows = n_img * H;
  int lr = FFT_TILE_POINTS / W;
  if (lr < 1) lr = 1;
  if (lr > total_rows) lr = total_rows;
  int lc = FFT_TILE_POINTS / H;
  if (lc < 1) lc = 1;
  if (lc > W) lc = W;
  if (lc > 64) lc = 64;
  P->rows = PassGeom{H, W, lw, lr, n_img, (float)(1.0 / std::sqrt((double)W)), centered ? 1 : 0, tww};
  P->cols = PassGeom{H, W, lh, lc, n_img, (float)(1.0 / std::sqrt((double)H)), centered ? 1 : 0, twh};
  P->grid_rows = dim3((total_rows + lr - 1) / lr);
  P->grid_cols = dim3(W / lc, n_img);
  P->lds_rows = sizeof(float2) * 2 * (size_t)lr * (W + 1);
  P->lds_cols = sizeof(float2) * 2 * (size_t)lc * (H + 1);
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
