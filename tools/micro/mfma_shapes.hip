// Micro-benchmark: sustained (power-capped) rate of the two dense f16 MFMA shapes of gfx950 on toggling operands,
// register-only loops, one or two waves per SIMD:  v_mfma_f32_32x32x16_f16 (what conv_hs issues) against
// v_mfma_f32_16x16x32_f16 (same FLOPs per cycle on paper, half the accumulator traffic per FLOP, twice the A/B operand
// traffic).  Question: does the shape change the energy per FLOP, i.e. the clock the power cap allows?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ h8 rnd_frag(unsigned seed) {
  union { h8 v; unsigned short s[8]; } u;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (seed + i) * 2654435761u + 12345u;
    h ^= h >> 13;
    u.s[i] = (unsigned short)(0x3800u | (h & 0x07ffu) | ((h >> 3) & 0x8000u));   // |x| in [0.5, 2), random sign / mantissa
  }
  return u.v;
}

// SHAPE 0: 32x32x16, 8 accumulators x 16 regs;  SHAPE 1: 16x16x32, 32 accumulators x 4 regs.  Both: 128 accumulator
// registers, 2 A and 4 B fragments, every MFMA of a round independent of the others; FLOPs per round: 8 x 32768 = 32 x 8192.
template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int rounds) {
  const int tid = threadIdx.x;
  h8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = rnd_frag(tid * 64 + i * 8);
  for (int i = 0; i < 4; ++i) b[i] = rnd_frag(tid * 64 + 16 + i * 8);
  float s = 0.f;
  if constexpr (SHAPE == 0) {
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
    for (int m = 0; m < 2; ++m)
      for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  } else {
    f32x4 acc[4][8];
    for (int m = 0; m < 4; ++m)
      for (int n = 0; n < 8; ++n)
        for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 8; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m & 1], b[n & 3], acc[m][n], 0, 0, 0);
    }
    for (int m = 0; m < 4; ++m)
      for (int n = 0; n < 8; ++n)
        for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  }
  if (s == 12345.678f) out[0] = s;
}

template <int SHAPE>
static void run(const char* name, int threads, float* d_out) {
  const double flops_per_round = 4.0 * 8 * 32768;     // per wave
  int rounds = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, d_out, 2000);   // warm-up
  hipDeviceSynchronize();
  double best = 0, sum = 0;
  const int reps = 12;                                   // ~3 s per configuration: long enough for the power cap to settle
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, d_out, rounds);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = flops_per_round * rounds * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
    if (r >= reps / 2) sum += tf;                        // second half: settled clocks
    if (tf > best) best = tf;
    if (r == 0 && ms < 200) rounds = (int)(rounds * 250.0 / ms);
  }
  printf("%-14s %d waves/SIMD: sustained %7.1f TF/s (settled mean), best burst %7.1f TF/s\n", name, threads / 256, sum / (reps - reps / 2),
         best);
}

int main() {
  float* d_out;
  hipMalloc(&d_out, 64);
  run<0>("32x32x16_f16", 256, d_out);
  run<1>("16x16x32_f16", 256, d_out);
  run<0>("32x32x16_f16", 512, d_out);
  run<1>("16x16x32_f16", 512, d_out);
  run<0>("32x32x16_f16", 256, d_out);
  run<1>("16x16x32_f16", 256, d_out);
  return 0;
}
