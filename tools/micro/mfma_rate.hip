// Micro-benchmark: achievable v_mfma_f32_32x32x16_f16 rate with ONE wave per SIMD (256 threads / CU, 512 registers),
// 8 accumulators of 32x32 per wave, 3 MFMAs on each per tap (the MT=64 / NBW=4 shape of conv_hs.hip: hi*hi, hi*lo,
// lo*hi; same-accumulator MFMAs are 8 instructions apart), with and without the 12
// ds_read_b128 fragment reads per 24 MFMAs, and with the reads hoisted one "tap" ahead.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MODE 0: MFMA only, 1: + 12 ds_read_b128 per 24 MFMA (prefetched one tap ahead), 2: reads right before use,
// 3: MODE 1 + 19 LDS-DMA instructions (1 KiB each) per wave per 9 taps, spread over the taps + s_waitcnt vmcnt(0) and
//    s_barrier every 9 taps (conv_hs's chunk step), 4: same DMA issued as one burst at the start of the step,
// 5: MODE 1 + the barrier only
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, int data, const char* src) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  // DATA 0: near-constant operands; DATA 1: pseudo-random f16 bit patterns with |x| in [0.5, 2) (realistic toggling)
  for (int i = tid; i < 16384; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + 12345u;
    h ^= h >> 13;
    const unsigned short lo16 = (unsigned short)(0x3800u | (h & 0x07ffu) | ((h >> 3) & 0x8000u));
    const unsigned short hi16 = (unsigned short)(0x3800u | ((h >> 11) & 0x07ffu) | ((h >> 7) & 0x8000u));
    reinterpret_cast<unsigned*>(lds)[i] = data ? ((unsigned)hi16 << 16 | lo16) : 0x1400u;
  }
  __syncthreads();
  f32x16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  h8 a[2][2], b[4][2], na[2][2], nb[4][2];
  const char* base = lds + (tid & 63) * 16;
  auto load = [&](h8 (&A)[2][2], h8 (&B)[4][2], int off) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h) A[m][h] = *reinterpret_cast<const h8*>(base + ((off + m * 2 + h) & 31) * 1024);
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int h = 0; h < 2; ++h) B[n][h] = *reinterpret_cast<const h8*>(base + ((off + 4 + n * 2 + h) & 31) * 1024);
  };
  load(a, b, 0);
  constexpr bool RD1 = (MODE == 1 || MODE >= 3);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* dma_dst = lds + 32768;                       // 2 x 38 KiB staging area behind the fragment table
  const char* gsrc = src + (size_t)blockIdx.x * 4096 + (tid & 63) * 16;
  auto dma = [&](int j, int stage) {                 // one 1-KiB piece
    __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + j * 1024), (lptr_t)(dma_dst + stage * 38912 + (wave * 19 + j) % 38 * 1024), 16, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
    const int tap = it % 9, stage = (it / 9) & 1;
    if (MODE >= 3 && tap == 0) {
      if (MODE != 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 19; ++j) dma(j, stage);
      }
    }
    if (MODE == 3) {
      dma(2 * tap, stage);
      dma(2 * tap + 1, stage);
      if (tap == 8) dma(18, stage);
    }
    if (RD1) load(na, nb, it + 1);
    if (MODE == 2) load(a, b, it);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], b[n][0], acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], b[n][1], acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][1], b[n][0], acc[m][n], 0, 0, 0);
    if (RD1) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) a[m][h] = na[m][h];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h) b[n][h] = nb[n][h];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, float* d, int data, const char* src) {
  const int iters = 3996;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 110 * 1024, 0, d, iters, data, src);   // 110 KB LDS: one workgroup per CU
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = 24.0 * iters * 4 * 256;   // per wave 24 per iteration, 4 waves, 256 CUs
  const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s (f16 MFMA)  = %5.1f %% of 2500;  cycles/MFMA/SIMD at 2.4 GHz: %.1f\n", name, ms, tf,
         tf / 25.0, ms * 1e-3 * 2.4e9 / (24.0 * iters));
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  char* src;
  hipMalloc(&src, 256 * 4096 + 64 * 1024);
  hipMemset(src, 0x3c, 256 * 4096 + 64 * 1024);
  for (int data = 0; data < 2; ++data) {
    printf("operands: %s\n", data ? "pseudo-random f16" : "constant");
    run<0>("MFMA only", d, data, src);
    run<1>("+12 ds_read_b128 / 24 MFMA, one tap ahead", d, data, src);
    run<2>("+12 ds_read_b128 / 24 MFMA, right before use", d, data, src);
    run<5>("reads + s_barrier every 9 taps", d, data, src);
    run<3>("reads + barrier + 19 LDS-DMA / 9 taps, spread", d, data, src);
    run<4>("reads + barrier + 19 LDS-DMA / 9 taps, burst", d, data, src);
  }
  return 0;
}
