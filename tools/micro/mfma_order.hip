// Micro-benchmark: does the ORDER of a fixed set of v_mfma_f32_32x32x16_f16 matter under the power cap?  Same 2 A x 4 B random
// operands, 8 independent accumulators, three issue orders: A-stationary (a0 b0..b3, a1 b0..b3), B-stationary (b0 a0 a1, b1 a0 a1 ..)
// and fully alternating (both operands change on every issue).  Operand bit activity is what the cap binds through
// (profiles/r2_mfma_toggle.md); this asks whether holding one operand across consecutive issues saves any of it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ h8 rnd_frag(unsigned seed) {
  union { h8 v; unsigned short s[8]; } u;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (seed + i) * 2654435761u + 12345u;
    h ^= h >> 13;
    u.s[i] = (unsigned short)(0x3800u | (h & 0x07ffu) | ((h >> 3) & 0x8000u));
  }
  return u.v;
}

template <int ORDER>
__global__ __launch_bounds__(256) void k(float* out, int rounds) {
  const int tid = threadIdx.x;
  h8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = rnd_frag(tid * 64 + i * 8);
  for (int i = 0; i < 4; ++i) b[i] = rnd_frag(tid * 64 + 16 + i * 8);
  f32x16 acc[2][4];
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 4; ++n)
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
#define MF(m, n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0)
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if constexpr (ORDER == 0) {        // A-stationary
        MF(0, 0); MF(0, 1); MF(0, 2); MF(0, 3); MF(1, 0); MF(1, 1); MF(1, 2); MF(1, 3);
      } else if constexpr (ORDER == 1) { // B-stationary
        MF(0, 0); MF(1, 0); MF(0, 1); MF(1, 1); MF(0, 2); MF(1, 2); MF(0, 3); MF(1, 3);
      } else {                           // both change on every issue
        MF(0, 0); MF(1, 1); MF(0, 2); MF(1, 3); MF(0, 1); MF(1, 0); MF(0, 3); MF(1, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 4; ++n)
      for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

template <int ORDER>
static void run(const char* name, float* d_out) {
  const double flops_per_round = 4.0 * 8 * 32768;
  const int rounds = 300000, reps = 10;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double sum = 0;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<ORDER>, dim3(256), dim3(256), 0, 0, d_out, rounds);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r >= reps / 2) sum += flops_per_round * rounds * 4 * 256 / (ms * 1e-3) / 1e12;
  }
  printf("%-44s sustained %7.1f TF/s\n", name, sum / (reps - reps / 2));
}

int main() {
  float* d_out;
  (void)hipMalloc(&d_out, 64);
  run<0>("A-stationary (4 issues per A)", d_out);
  run<1>("B-stationary (2 issues per B)", d_out);
  run<2>("both operands change every issue", d_out);
  run<0>("A-stationary (again)", d_out);
  return 0;
}
