// Does the 32-byte-record access pattern of the HS8 layout (each lane: two 16-byte accesses 16 B apart, lanes 32 B apart,
// i.e. every wave-instruction touches its 2-KiB span at 50 % density) cost bandwidth against a planar hi / lo layout
// (every wave-instruction dense)?   hipcc --offload-arch=gfx950 -O3 rec_density.hip -o rec_density && ./rec_density
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void write_rec(uint4* dst, size_t nrec) {   // record layout: hi | lo per pixel
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nrec; i += stride) {
    dst[2 * i] = make_uint4((unsigned)i, 1, 2, 3);
    dst[2 * i + 1] = make_uint4((unsigned)i, 5, 6, 7);
  }
}
__global__ __launch_bounds__(256) void write_planar(uint4* dst, size_t nrec) {   // planar: hi plane, lo plane
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nrec; i += stride) {
    dst[i] = make_uint4((unsigned)i, 1, 2, 3);
    dst[nrec + i] = make_uint4((unsigned)i, 5, 6, 7);
  }
}
__global__ __launch_bounds__(256) void read_rec(const uint4* src, size_t nrec, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < nrec; i += stride) {
    const uint4 a = src[2 * i], b = src[2 * i + 1];
    acc += a.x ^ b.y ^ a.z ^ b.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void read_planar(const uint4* src, size_t nrec, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < nrec; i += stride) {
    const uint4 a = src[i], b = src[nrec + i];
    acc += a.x ^ b.y ^ a.z ^ b.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// one 16-byte piece per lane per instruction, pieces 32 B apart (what one conv_hs LDS-DMA / store instruction does)
__global__ __launch_bounds__(256) void read_half(const uint4* src, size_t nrec, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < nrec; i += stride) acc += src[2 * i].x;
  if (acc == 0x12345678u) *sink = acc;
}

int main() {
  const size_t nrec = (size_t)48 * 1024 * 1024;   // 1.5 GiB of 32-byte records
  uint4* buf;
  unsigned* sink;
  CK(hipMalloc(&buf, nrec * 32));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, nrec * 32));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grids[] = {2048, 8192, 65536};
  for (int g : grids) {
    auto time = [&](auto launch, const char* name, double bytes) {
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("grid %6d  %-28s %7.3f ms  %6.2f TB/s\n", g, name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
      return 0;
    };
    time([&] { hipLaunchKernelGGL(write_rec, dim3(g), dim3(256), 0, 0, buf, nrec); }, "write 32-B records", nrec * 32.0);
    time([&] { hipLaunchKernelGGL(write_planar, dim3(g), dim3(256), 0, 0, buf, nrec); }, "write planar hi|lo", nrec * 32.0);
    time([&] { hipLaunchKernelGGL(read_rec, dim3(g), dim3(256), 0, 0, buf, nrec, sink); }, "read 32-B records", nrec * 32.0);
    time([&] { hipLaunchKernelGGL(read_planar, dim3(g), dim3(256), 0, 0, buf, nrec, sink); }, "read planar hi|lo", nrec * 32.0);
    time([&] { hipLaunchKernelGGL(read_half, dim3(g), dim3(256), 0, 0, buf, nrec, sink); }, "read hi halves only (stride 32)", nrec * 16.0);
  }
  return 0;
}
