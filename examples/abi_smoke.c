/* A C consumer of libpnpx.so with no Python and no torch in the process: the drop-in boundary on its own.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/abi_smoke.c -Ltfpnp_amd -lpnpx \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/tfpnp_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/abi_smoke && /tmp/abi_smoke
 * (the HIP runtime API is used for device memory only; the macro is what hip_runtime_api.h wants from a non-hipcc
 * compiler)
 *
 * Loads deterministic synthetic UNet weights, runs the denoiser and five CS-MRI ADMM iterations (reference call
 * sites: tfpnp/pnp/denoiser/base.py:23-32, tasks/csmri/solver.py:29-57) on a 2 x 64 x 64 batch and checks the
 * contract a caller relies on: status codes, finite outputs in [0,1], bit-identical repeat, the ADMM fixed-point
 * structure of the returned state, error reporting for a bad shape.  Prints "abi_smoke OK" and exits 0 on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pnpx.h"

#define CHECK(call)                                                                       \
  do {                                                                                    \
    int _s = (call);                                                                      \
    if (_s != 0) {                                                                        \
      fprintf(stderr, "%s failed with %d: %s\n", #call, _s, pnpx_last_error());           \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)
#define HIP(call)                                                            \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) {                                                  \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(_e));            \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static unsigned long long rng = 88172645463325252ull;
static float unif(void) { /* xorshift64 -> (-1, 1) */
  rng ^= rng << 13;
  rng ^= rng >> 7;
  rng ^= rng << 17;
  return (float)((rng >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f;
}

int main(void) {
  const int B = 2, H = 64, W = 64, T = 5;
  const size_t n = pnpx_unet_num_params();
  float* wts = (float*)malloc(n * sizeof(float));
  /* state_dict order: 27 x (weight[cout][cin][3][3], bias[cout]) + outc; a uniform He-like scale is enough here */
  for (size_t i = 0; i < n; ++i) wts[i] = unif() * 0.04f;

  pnpx_ctx* ctx = NULL;
  CHECK(pnpx_ctx_create(0, &ctx));
  CHECK(pnpx_unet_load(ctx, wts, n));

  const size_t npx = (size_t)B * H * W;
  float *h_x = (float*)malloc(npx * sizeof(float)), *h_o = (float*)malloc(npx * sizeof(float)),
        *h_o2 = (float*)malloc(npx * sizeof(float));
  for (size_t i = 0; i < npx; ++i) h_x[i] = 0.5f + 0.5f * unif();
  float h_sigma[2] = {0.1f, 0.2f};
  float *d_x, *d_o, *d_sigma;
  HIP(hipMalloc((void**)&d_x, npx * 4));
  HIP(hipMalloc((void**)&d_o, npx * 4));
  HIP(hipMalloc((void**)&d_sigma, sizeof(h_sigma)));
  HIP(hipMemcpy(d_x, h_x, npx * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_sigma, h_sigma, sizeof(h_sigma), hipMemcpyHostToDevice));

  CHECK(pnpx_unet_denoise(ctx, d_x, d_sigma, d_o, NULL, B, H, W, NULL));
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(h_o, d_o, npx * 4, hipMemcpyDeviceToHost));
  CHECK(pnpx_unet_denoise(ctx, d_x, d_sigma, d_o, NULL, B, H, W, NULL));
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(h_o2, d_o, npx * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < npx; ++i) {
    if (!(h_o[i] >= 0.f && h_o[i] <= 1.f)) {
      fprintf(stderr, "denoiser output %zu = %g outside [0,1]\n", i, h_o[i]);
      return 1;
    }
  }
  if (memcmp(h_o, h_o2, npx * 4) != 0) {
    fprintf(stderr, "denoiser is not bit-reproducible\n");
    return 1;
  }

  /* CS-MRI ADMM: state [B,3,H,W,2] = (x, z, u); fully sampled k-space and u = 0 make z = x a fixed point of the
   * data step, so after any number of iterations z == x and u == 0 (up to fp32 round-off). */
  const size_t nstate = (size_t)B * 3 * H * W * 2, nk = (size_t)B * H * W * 2;
  float *h_v = (float*)calloc(nstate, 4), *h_y0 = (float*)calloc(nk, 4), *h_vo = (float*)malloc(nstate * 4);
  unsigned char* h_mask = (unsigned char*)calloc(npx, 1); /* nothing sampled: the blend leaves k-space untouched */
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < H * W; ++i) {
      const float v = h_x[(size_t)b * H * W + i];
      h_v[(((size_t)b * 3 + 0) * H * W + i) * 2] = v; /* x */
      h_v[(((size_t)b * 3 + 1) * H * W + i) * 2] = v; /* z */
    }
  float h_par[2 * 5];
  for (int i = 0; i < 10; ++i) h_par[i] = 0.1f;
  float *d_v, *d_vo, *d_y0, *d_sig, *d_mu;
  unsigned char* d_mask;
  HIP(hipMalloc((void**)&d_v, nstate * 4));
  HIP(hipMalloc((void**)&d_vo, nstate * 4));
  HIP(hipMalloc((void**)&d_y0, nk * 4));
  HIP(hipMalloc((void**)&d_mask, npx));
  HIP(hipMalloc((void**)&d_sig, sizeof(h_par)));
  HIP(hipMalloc((void**)&d_mu, sizeof(h_par)));
  HIP(hipMemcpy(d_v, h_v, nstate * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_y0, h_y0, nk * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_mask, h_mask, npx, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_sig, h_par, sizeof(h_par), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_mu, h_par, sizeof(h_par), hipMemcpyHostToDevice));
  CHECK(pnpx_csmri_admm(ctx, d_v, d_vo, d_y0, d_mask, d_sig, d_mu, /*stride*/ T, B, H, W, T, NULL));
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(h_vo, d_vo, nstate * 4, hipMemcpyDeviceToHost));
  double dz = 0, du = 0, dim = 0;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < H * W; ++i)
      for (int c = 0; c < 2; ++c) {
        const float x = h_vo[(((size_t)b * 3 + 0) * H * W + i) * 2 + c];
        const float z = h_vo[(((size_t)b * 3 + 1) * H * W + i) * 2 + c];
        const float u = h_vo[(((size_t)b * 3 + 2) * H * W + i) * 2 + c];
        if (!isfinite(x) || !isfinite(z) || !isfinite(u)) {
          fprintf(stderr, "non-finite solver state\n");
          return 1;
        }
        dz = fmax(dz, fabs((double)z - x));
        du = fmax(du, fabs((double)u));
        if (c == 1) dim = fmax(dim, fabs((double)x));
      }
  if (dz > 1e-4 || du > 1e-4 || dim != 0.0) {
    fprintf(stderr, "ADMM fixed-point structure violated: |z-x| %g |u| %g |Im x| %g\n", dz, du, dim);
    return 1;
  }

  /* error contract: unsupported geometry -> non-zero status + message, nothing thrown */
  if (pnpx_unet_denoise(ctx, d_x, d_sigma, d_o, NULL, B, 8, 8, NULL) == 0 || strlen(pnpx_last_error()) == 0) {
    fprintf(stderr, "bad shape was not rejected\n");
    return 1;
  }
  CHECK(pnpx_ctx_destroy(ctx));
  printf("abi_smoke OK (%s; |z-x| %.1e, |u| %.1e)\n", pnpx_version(), dz, du);
  return 0;
}
