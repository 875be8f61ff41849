// Episode orchestration on the device -- the caller contract of PnPEnv.step (tfpnp/env/base.py:157-191):
//   * live-row gather / write-back of the environment state (`x[self.idx_left, ...]`, `state[...][idx_left] = ...`,
//     base.py:162-172) as ONE launch over all tensors of the state,
//   * the shrinking live set (`self.idx_left = self.idx_left[idx_stop == 0]`, `len(self.idx_left) == 0`,
//     base.py:180-182) as a device-side stream compaction whose count is the ONE host read of a step,
//   * the policy observation (`get_policy_ob`, tasks/csmri/env.py:14-23 and siblings: complex2real / complex2channel
//     views concatenated on the channel axis) packed by one kernel straight from the state tensors.
// All of it is HBM-bound byte shuffling: 16-byte accesses, rows on grid.y, tensors on grid.z.
#include "common.h"

namespace pnpx {

constexpr int ENV_MAX_T = 12;

struct RowsArgs {
  const char* src[ENV_MAX_T];
  char* dst[ENV_MAX_T];
  unsigned long long row_bytes[ENV_MAX_T];
  const long long* idx;   // [n_rows] row numbers in the big (un-compacted) tensors
  int n_rows;
};

// SCATTER = false: dst[r] = src[idx[r]];  true: dst[idx[r]] = src[r]
template <bool SCATTER>
__global__ __launch_bounds__(256) void rows_copy_kernel(RowsArgs a) {
  const int t = blockIdx.z, r = blockIdx.y;
  const unsigned long long nb = a.row_bytes[t];
  const long long big = a.idx[r];
  const char* s = a.src[t] + (SCATTER ? (unsigned long long)r : (unsigned long long)big) * nb;
  char* d = a.dst[t] + (SCATTER ? (unsigned long long)big : (unsigned long long)r) * nb;
  const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  if (((nb | (unsigned long long)(uintptr_t)a.src[t] | (unsigned long long)(uintptr_t)a.dst[t]) & 15ull) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    uint4* d4 = reinterpret_cast<uint4*>(d);
    for (unsigned long long i = i0; i < nb / 16; i += stride) d4[i] = s4[i];
  } else if (((nb | (unsigned long long)(uintptr_t)a.src[t] | (unsigned long long)(uintptr_t)a.dst[t]) & 3ull) == 0) {
    const unsigned* s1 = reinterpret_cast<const unsigned*>(s);
    unsigned* d1 = reinterpret_cast<unsigned*>(d);
    for (unsigned long long i = i0; i < nb / 4; i += stride) d1[i] = s1[i];
  } else {
    for (unsigned long long i = i0; i < nb; i += stride) d[i] = s[i];
  }
}

// idx_out[0..n_live) = idx_left[i] for the i with idx_stop[i] == 0, in order; one workgroup (batches are <= a few
// thousand items).  The count goes to a device word and to a host-mapped word.
__global__ __launch_bounds__(256) void live_compact_kernel(const long long* __restrict__ idx_left,
                                                           const long long* __restrict__ idx_stop, int n,
                                                           long long* __restrict__ idx_out, int* __restrict__ n_out_host) {
  __shared__ int wave_tot[4];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int i = base + tid;
    const bool keep = (i < n) && (idx_stop[i] == 0);
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (keep) idx_out[off + before] = idx_left[i];
    __syncthreads();
    if (tid == 0) carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (tid == 0) __hip_atomic_store(n_out_host, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct PackArgs {
  const void* src[ENV_MAX_T];
  int kind[ENV_MAX_T];      // 0 raw fp32 [B,c,H,W]; 1 real part of [B,c,H,W,2]; 2 re/im as channels of [B,c,H,W,2]; 3 raw u8
  int c[ENV_MAX_T];         // source channels
  int ch0[ENV_MAX_T];       // first output channel
  int n_t, C_out, HW;
  const long long* idx;     // row numbers in the sources, or null: row r
  float* out;               // [n_rows, C_out, H, W]
};

// one thread per 4 consecutive pixels of one output channel
__global__ __launch_bounds__(256) void policy_ob_pack_kernel(PackArgs a) {
  const int r = blockIdx.z;
  const int co = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;   // pixel quad
  if (q * 4 >= a.HW) return;
  int t = 0;
#pragma unroll
  for (int k = 1; k < ENV_MAX_T; ++k)
    if (k < a.n_t && co >= a.ch0[k]) t = k;
  const int kind = a.kind[t];
  const int cl = co - a.ch0[t];
  const size_t row = a.idx ? (size_t)a.idx[r] : (size_t)r;
  const int p0 = q * 4;
  const int np = min(4, a.HW - p0);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (kind == 0) {
    const float* s = static_cast<const float*>(a.src[t]) + (row * a.c[t] + cl) * (size_t)a.HW + p0;
    for (int j = 0; j < np; ++j) v[j] = s[j];
  } else if (kind == 3) {
    const unsigned char* s = static_cast<const unsigned char*>(a.src[t]) + (row * a.c[t] + cl) * (size_t)a.HW + p0;
    for (int j = 0; j < np; ++j) v[j] = s[j] ? 1.f : 0.f;
  } else {
    const int cs = (kind == 1) ? cl : (cl >> 1);
    const int part = (kind == 1) ? 0 : (cl & 1);
    const float* s = static_cast<const float*>(a.src[t]) + ((row * a.c[t] + cs) * (size_t)a.HW + p0) * 2 + part;
    for (int j = 0; j < np; ++j) v[j] = s[2 * j];
  }
  float* o = a.out + ((size_t)r * a.C_out + co) * (size_t)a.HW + p0;
  if (np == 4 && ((a.HW & 3) == 0)) {
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    for (int j = 0; j < np; ++j) o[j] = v[j];
  }
}

}  // namespace pnpx

using namespace pnpx;

#define LOCK_CTX(ctx)                         \
  if (!(ctx)) {                               \
    pnpx::set_error("null context");          \
    return PNPX_ERR_ARG;                      \
  }                                           \
  std::lock_guard<std::mutex> _lk((ctx)->mu); \
  PNPX_HIP(hipSetDevice((ctx)->device))

static int rows_copy(pnpx_ctx* ctx, bool scatter, int n_tensors, const void* const* src_host, void* const* dst_host,
                     const size_t* row_bytes_host, const int64_t* idx, int n_rows, void* stream) {
  LOCK_CTX(ctx);
  if (n_tensors < 0 || n_rows < 0 || (n_tensors > 0 && (!src_host || !dst_host || !row_bytes_host)) ||
      (n_rows > 0 && !idx)) {
    set_error("rows gather/scatter: bad arguments");
    return PNPX_ERR_ARG;
  }
  if (n_rows == 0) return PNPX_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int t0 = 0; t0 < n_tensors; t0 += ENV_MAX_T) {
    const int nt = (n_tensors - t0 < ENV_MAX_T) ? (n_tensors - t0) : ENV_MAX_T;
    RowsArgs a{};
    size_t mx = 0;
    for (int t = 0; t < nt; ++t) {
      a.src[t] = static_cast<const char*>(src_host[t0 + t]);
      a.dst[t] = static_cast<char*>(dst_host[t0 + t]);
      a.row_bytes[t] = row_bytes_host[t0 + t];
      if (!a.src[t] || !a.dst[t]) {
        set_error("rows gather/scatter: null tensor pointer");
        return PNPX_ERR_ARG;
      }
      if (a.row_bytes[t] > mx) mx = a.row_bytes[t];
    }
    a.idx = reinterpret_cast<const long long*>(idx);
    a.n_rows = n_rows;
    size_t gx = (mx / 16 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;   // 64 x 256 lanes x 16 B = 256 KiB per sweep of a row; longer rows loop
    const dim3 grid((unsigned)gx, (unsigned)n_rows, (unsigned)nt);
    if (scatter)
      hipLaunchKernelGGL(rows_copy_kernel<true>, grid, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(rows_copy_kernel<false>, grid, dim3(256), 0, s, a);
    PNPX_LAUNCH_CHECK();
  }
  return PNPX_OK;
}

extern "C" {

int pnpx_rows_gather(pnpx_ctx* ctx, int n_tensors, const void* const* src_host, void* const* dst_host,
                     const size_t* row_bytes_host, const int64_t* idx, int n_rows, void* stream) {
  return rows_copy(ctx, false, n_tensors, src_host, dst_host, row_bytes_host, idx, n_rows, stream);
}

int pnpx_rows_scatter(pnpx_ctx* ctx, int n_tensors, const void* const* src_host, void* const* dst_host,
                      const size_t* row_bytes_host, const int64_t* idx, int n_rows, void* stream) {
  return rows_copy(ctx, true, n_tensors, src_host, dst_host, row_bytes_host, idx, n_rows, stream);
}

int pnpx_live_compact(pnpx_ctx* ctx, const int64_t* idx_left, const int64_t* idx_stop, int n, int64_t* idx_out,
                      int* n_live_host, void* stream) {
  if (!ctx) {
    set_error("null context");
    return PNPX_ERR_ARG;
  }
  if (n < 0 || !n_live_host || (n > 0 && (!idx_left || !idx_stop || !idx_out))) {
    set_error("pnpx_live_compact: bad arguments");
    return PNPX_ERR_ARG;
  }
  if (n == 0) {
    *n_live_host = 0;
    return PNPX_OK;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Result slot: one pinned, host-mapped word per (calling thread, device) -- NOT a field of the (possibly shared,
  // e.g. default) context -- so the context mutex is not needed at all and other threads' stateless ops on the same
  // context never queue behind this call's stream synchronisation.
  struct Slot {
    int* host = nullptr;
    int* dev = nullptr;
  };
  thread_local std::map<int, Slot> slots;
  PNPX_HIP(hipSetDevice(ctx->device));
  Slot& sl = slots[ctx->device];
  if (!sl.host) {
    void* h = nullptr;
    void* d = nullptr;
    PNPX_HIP(hipHostMalloc(&h, 64, hipHostMallocMapped));
    hipError_t e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
      (void)hipHostFree(h);
      return hip_fail(e, "hipHostGetDevicePointer(live count)", __FILE__, __LINE__);
    }
    sl.host = static_cast<int*>(h);
    sl.dev = static_cast<int*>(d);
  }
  hipLaunchKernelGGL(live_compact_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<const long long*>(idx_left),
                     reinterpret_cast<const long long*>(idx_stop), n, reinterpret_cast<long long*>(idx_out), sl.dev);
  PNPX_LAUNCH_CHECK();
  PNPX_HIP(hipStreamSynchronize(s));   // the one host read of an env step: `all_done` is a Python bool in the contract
  *n_live_host = *static_cast<volatile int*>(sl.host);
  return PNPX_OK;
}

int pnpx_policy_ob_pack(pnpx_ctx* ctx, int n_entries, const void* const* src_host, const int* kind_host,
                        const int* channels_host, const int64_t* idx, int n_rows, int H, int W, float* out,
                        void* stream) {
  LOCK_CTX(ctx);
  if (n_entries <= 0 || n_entries > ENV_MAX_T || !src_host || !kind_host || !channels_host || n_rows < 0 || H <= 0 ||
      W <= 0 || (n_rows > 0 && !out)) {
    set_error("pnpx_policy_ob_pack: bad arguments (at most %d entries)", ENV_MAX_T);
    return PNPX_ERR_ARG;
  }
  if (n_rows == 0) return PNPX_OK;
  PackArgs a{};
  int co = 0;
  for (int t = 0; t < n_entries; ++t) {
    if (!src_host[t] || kind_host[t] < 0 || kind_host[t] > 3 || channels_host[t] <= 0) {
      set_error("pnpx_policy_ob_pack: bad entry %d", t);
      return PNPX_ERR_ARG;
    }
    a.src[t] = src_host[t];
    a.kind[t] = kind_host[t];
    a.c[t] = channels_host[t];
    a.ch0[t] = co;
    co += channels_host[t] * (kind_host[t] == 2 ? 2 : 1);
  }
  a.n_t = n_entries;
  a.C_out = co;
  a.HW = H * W;
  a.idx = reinterpret_cast<const long long*>(idx);
  a.out = out;
  const int quads = (a.HW + 3) / 4;
  hipLaunchKernelGGL(policy_ob_pack_kernel, dim3((quads + 255) / 256, co, n_rows), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

}  // extern "C"
