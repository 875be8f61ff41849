// Micro-benchmark: how much does operand bit activity move the power-capped v_mfma_f32_32x32x16_f16 rate?
// Register-only loop, one wave per SIMD; A operands random f16 in [0.5, 2); B operands the same with their `zb` low
// mantissa bits cleared (zb = 0: fully random; 10: powers of two) -- the question behind it: would truncating the `lo`
// halves of the half-split operands (they only need a few bits) buy clock under the power cap?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_toggle.hip -o /tmp/mfma_toggle && /tmp/mfma_toggle
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ h8 rnd_frag(unsigned seed, unsigned mask) {
  union { h8 v; unsigned short s[8]; } u;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (seed + i) * 2654435761u + 12345u;
    h ^= h >> 13;
    u.s[i] = (unsigned short)((0x3800u | (h & 0x07ffu) | ((h >> 3) & 0x8000u)) & mask);
  }
  return u.v;
}

__global__ __launch_bounds__(256) void k(float* out, int rounds, unsigned maskA, unsigned maskB) {
  const int tid = threadIdx.x;
  h8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = rnd_frag(tid * 64 + i * 8, maskA);
  for (int i = 0; i < 4; ++i) b[i] = rnd_frag(tid * 64 + 16 + i * 8, maskB);
  f32x16 acc[2][4];
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 4; ++n)
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
  }
  float s = 0.f;
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 4; ++n)
      for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  if (s == 12345.678f) out[0] = s;
}

static void run(const char* name, unsigned maskA, unsigned maskB, float* d_out) {
  const double flops_per_round = 4.0 * 8 * 32768;
  int rounds = 300000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double sum = 0;
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_out, rounds, maskA, maskB);
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
  run("A random, B random", 0xffff, 0xffff, d_out);
  run("A random, B 4 low mantissa bits cleared", 0xffff, 0xfff0, d_out);
  run("A random, B 6 low mantissa bits cleared", 0xffff, 0xffc0, d_out);
  run("A random, B 8 low mantissa bits cleared", 0xffff, 0xff00, d_out);
  run("A random, B mantissa cleared (powers of 2)", 0xffff, 0xfc00, d_out);
  run("A and B 6 low mantissa bits cleared", 0xffc0, 0xffc0, d_out);
  run("A and B mantissa cleared", 0xfc00, 0xfc00, d_out);
  run("A random, B zero", 0xffff, 0x0000, d_out);
  run("A random, B random (again)", 0xffff, 0xffff, d_out);
  return 0;
}
