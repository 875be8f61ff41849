// LDS-staged batched 1-D FFT passes for gfx950 (power-of-two sizes 2..1024), used as the row and column
// passes of every 2-D transform on the hot path.
//
// Replaces tfpnp/utils/transforms.py:68-103 (fft2/ifft2 = ifftshift -> torch.fft(x, 2, normalized=True) ->
// fftshift) and the plain torch.fft / torch.ifft calls of cdp_forward/backward (transforms.py:300,318).
// The reference materialises four rolls per centered transform (transforms.py:215-257); for even sizes
//     fftshift(F(ifftshift(x)))[k] = (-1)^(k + N/2) * F((-1)^n x[n])[k]
// so the shifts are exact sign flips applied at load and store (no data movement, bit-identical to rolls).
//
// A workgroup (256 threads) owns a tile of L lines x N points in LDS (line stride N+1 float2 to break the
// power-of-two bank stride of column tiles) and runs an autosort Stockham FFT: radix-4 stages plus one
// radix-2 stage when log2(N) is odd, ping-ponging between two LDS buffers, one barrier per stage.
// Twiddles e^{-2 pi i m / N} come from a per-size table computed on the host in double precision.
// Load / Mid / Store functors fuse the neighbouring pointwise work (x+u, k-space blend, dual update ...)
// into the passes, so a fused prox step is: row pass -> column pass (fwd, pointwise, inverse) -> row pass.
#pragma once
#include "common.h"

namespace pnpx {

constexpr int FFT_THREADS = 256;
constexpr int FFT_TILE_POINTS = 1024;   // complex points per workgroup tile (r3 sweep at 48 x 256^2: 512 / 1024 / 2048 / 4096 -> 124 / 109 / 140 / 210 us per ADMM iteration for the three passes)
constexpr int FFT_MAX_N = 2048;   // one line (+ ping-pong copy) must fit LDS

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-LDS Stockham FFT of `lines` lines of N points (line stride N+1).  Data must already be in d0 (a
// barrier is issued first).  Returns the buffer that holds the result; a barrier has been issued after
// the last stage.
template <bool INV>
__device__ inline float2* lds_fft(float2* d0, float2* d1, int N, int logN, int lines,
                                  const float2* __restrict__ tw) {
  const int ls = N + 1;
  int Ns = 1, rem = logN;
  __syncthreads();
  while (Ns < N) {
    if (rem >= 2) {
      const int q = N >> 2, logq = logN - 2;
      const int tstep = N / (4 * Ns);
      for (int item = threadIdx.x; item < lines * q; item += FFT_THREADS) {
        const int l = item >> logq, j = item & (q - 1);
        const int k = j & (Ns - 1);
        const float2* s = d0 + l * ls + j;
        float2 v0 = s[0], v1 = s[q], v2 = s[2 * q], v3 = s[3 * q];
        if (k) {
          float2 t1 = tw[k * tstep], t2 = tw[2 * k * tstep], t3 = tw[3 * k * tstep];
          if (INV) { t1.y = -t1.y; t2.y = -t2.y; t3.y = -t3.y; }
          v1 = cmul(v1, t1);
          v2 = cmul(v2, t2);
          v3 = cmul(v3, t3);
        }
        const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y), a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
        const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
        const float2 dd = make_float2(v1.x - v3.x, v1.y - v3.y);
        const float2 a3 = INV ? make_float2(-dd.y, dd.x) : make_float2(dd.y, -dd.x);  // * (+-i)
        float2* o = d1 + l * ls + ((j - k) << 2) + k;
        o[0] = make_float2(a0.x + a2.x, a0.y + a2.y);
        o[Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
        o[2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
        o[3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
      }
      Ns <<= 2;
      rem -= 2;
    } else {
      const int q = N >> 1, logq = logN - 1;
      const int tstep = N / (2 * Ns);
      for (int item = threadIdx.x; item < lines * q; item += FFT_THREADS) {
        const int l = item >> logq, j = item & (q - 1);
        const int k = j & (Ns - 1);
        const float2* s = d0 + l * ls + j;
        float2 v0 = s[0], v1 = s[q];
        if (k) {
          float2 t1 = tw[k * tstep];
          if (INV) t1.y = -t1.y;
          v1 = cmul(v1, t1);
        }
        float2* o = d1 + l * ls + ((j - k) << 1) + k;
        o[0] = make_float2(v0.x + v1.x, v0.y + v1.y);
        o[Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
      }
      Ns <<= 1;
      rem -= 1;
    }
    __syncthreads();
    float2* t = d0;
    d0 = d1;
    d1 = t;
  }
  return d0;
}

// General-size stage: N = R * q, sub-transform length Ns so far.  O(R^2) butterflies straight from the definition
// (R is a small prime factor in practice); twiddle and DFT_R phase are combined into one table index.
template <bool INV>
__device__ inline void lds_fft_stage_any(const float2* d0, float2* d1, int N, int R, int Ns, int lines,
                                         const float2* __restrict__ tw) {
  const int ls = N + 1, q = N / R, tstep = N / (R * Ns);
  for (int item = threadIdx.x; item < lines * q; item += FFT_THREADS) {
    const int l = item / q, j = item - l * q;
    const int k = j % Ns;
    const float2* s = d0 + l * ls + j;
    float2* o = d1 + l * ls + (j - k) * R + k;
    for (int rp = 0; rp < R; ++rp) {
      float2 acc = make_float2(0.f, 0.f);
      for (int r = 0; r < R; ++r) {
        float2 t = tw[(int)(((long long)r * k * tstep + (long long)r * rp * q) % N)];
        if (INV) t.y = -t.y;
        const float2 v = s[r * q];
        acc.x += v.x * t.x - v.y * t.y;
        acc.y += v.x * t.y + v.y * t.x;
      }
      o[rp * Ns] = acc;
    }
  }
}

template <bool INV>
__device__ inline float2* lds_fft_any(float2* d0, float2* d1, int N, int nrad, const unsigned short* rad, int lines,
                                      const float2* __restrict__ tw) {
  int Ns = 1;
  __syncthreads();
  for (int st = 0; st < nrad; ++st) {
    lds_fft_stage_any<INV>(d0, d1, N, rad[st], Ns, lines, tw);
    Ns *= rad[st];
    __syncthreads();
    float2* t = d0;
    d0 = d1;
    d1 = t;
  }
  return d0;
}

struct PassGeom {
  int H, W;          // image size
  int logN;          // log2 of the transform length (W for rows, H for columns); -1 when it is not a power of two
  int nrad;          // general sizes: Stockham stages of radix rad[0..nrad) (any factorisation of N)
  unsigned short rad[12];
  int odd;           // centered transform of odd length: explicit ifftshift/fftshift index rolls instead of sign flips
  int lines;         // lines per workgroup tile
  int n_img;
  float scale;       // 1/sqrt(N) (orthonormal)
  int centered;      // apply the (-1)^n / (-1)^(k+N/2) signs
  const float2* tw;
  int affine;        // 1: XCD-affine block -> image mapping (xcd_affine_decode); set by the launchers when it applies
};

// XCD-affine work mapping for chains of passes over the same images.  Workgroup `id` runs on XCD id % 8 (round-robin
// dispatch); with `per_img` workgroups per image, images are dealt to XCDs in groups of eight: image 8g + x is processed
// entirely by XCD x in EVERY pass of the chain, so what one pass wrote (0.5 MB of k-space per 256^2 image, 3 MB per
// XCD and 48-image batch) is still in that XCD's 4 MiB L2 when the next pass reads it -- the round trip between row and
// column passes stops at the L2 instead of the fabric.  A trailing partial group of images is mapped plainly.
__device__ __forceinline__ void xcd_affine_decode(int id, int per_img, int n_img, int affine, int* b, int* part) {
  const int group = 8 * per_img;
  const int gidx = id / group;
  if (!affine || (gidx + 1) * 8 > n_img) {
    *b = id / per_img;
    *part = id - *b * per_img;
    return;
  }
  const int loc = id - gidx * group;
  *b = gidx * 8 + (loc & 7);
  *part = loc >> 3;
}

// ---- row pass: lines = image rows (contiguous).  grid.x = ceil(n_img*H / lines)
template <bool INV, class Load, class Store>
__global__ __launch_bounds__(FFT_THREADS) void fft_rows_kernel(PassGeom g, Load ld, Store st) {
  extern __shared__ __attribute__((aligned(16))) float2 smem[];
  const int N = g.W, ls = N + 1;
  float2* d0 = smem;
  float2* d1 = smem + g.lines * ls;
  int line0 = blockIdx.x * g.lines;
  if (g.affine) {     // (launcher guarantees H % lines == 0: a workgroup's lines belong to one image)
    int b_, part_;
    xcd_affine_decode(blockIdx.x, g.H / g.lines, g.n_img, 1, &b_, &part_);
    line0 = (b_ * (g.H / g.lines) + part_) * g.lines;
  }
  const int total = g.n_img * g.H;
  const int logW = g.logN;
  const bool flip = g.centered && !g.odd;             // even length: shifts are sign flips
  const int rin = g.odd ? (N + 1) / 2 : 0, rout = g.odd ? N / 2 : 0;   // odd length: explicit rolls
  for (int idx = threadIdx.x; idx < g.lines * N; idx += FFT_THREADS) {
    const int l = logW >= 0 ? idx >> logW : idx / N, x = idx - l * N;
    const int gl = line0 + l;
    float2 v = make_float2(0.f, 0.f);
    if (gl < total) {
      const int b = gl / g.H, y = gl - b * g.H;
      v = ld(b, y, g.odd ? (x + N - rin) % N : x);
      if (flip && (x & 1)) { v.x = -v.x; v.y = -v.y; }
    }
    d0[l * ls + x] = v;
  }
  float2* r = logW >= 0 ? lds_fft<INV>(d0, d1, N, logW, g.lines, g.tw)
                        : lds_fft_any<INV>(d0, d1, N, g.nrad, g.rad, g.lines, g.tw);
  for (int idx = threadIdx.x; idx < g.lines * N; idx += FFT_THREADS) {
    const int l = logW >= 0 ? idx >> logW : idx / N, k = idx - l * N;
    const int gl = line0 + l;
    if (gl < total) {
      const int b = gl / g.H, y = gl - b * g.H;
      float2 v = r[l * ls + k];
      const float sc = (flip && ((k + (N >> 1)) & 1)) ? -g.scale : g.scale;
      st(b, y, g.odd ? (k + rout) % N : k, make_float2(v.x * sc, v.y * sc));
    }
  }
}

// ---- column pass: a tile is `lines` adjacent columns of one image.  grid = (W/lines, n_img).
// FUSED: forward FFT -> mid(b, ky, kx, value) -> inverse FFT (one HBM round trip for both transforms).
struct MidNone {
  __device__ float2 operator()(int, int, int, float2 v) const { return v; }
};

template <bool INV, bool FUSED, class Load, class Mid, class Store>
__global__ __launch_bounds__(FFT_THREADS) void fft_cols_kernel(PassGeom g, Load ld, Mid mid, Store st) {
  extern __shared__ __attribute__((aligned(16))) float2 smem[];
  const int N = g.H, ls = N + 1, C = g.lines;
  float2* d0 = smem;
  float2* d1 = smem + C * ls;
  int b, xt;
  xcd_affine_decode(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, g.n_img, g.affine, &b, &xt);
  const int x0 = xt * C;
  const int half = N >> 1;
  const bool flip = g.centered && !g.odd;
  const int rin = g.odd ? (N + 1) / 2 : 0, rout = g.odd ? N / 2 : 0;
  for (int idx = threadIdx.x; idx < C * N; idx += FFT_THREADS) {
    const int y = idx / C, c = idx - y * C;
    float2 v = (x0 + c < g.W) ? ld(b, g.odd ? (y + N - rin) % N : y, x0 + c) : make_float2(0.f, 0.f);
    if (flip && (y & 1)) { v.x = -v.x; v.y = -v.y; }
    d0[c * ls + y] = v;
  }
  float2* r = g.logN >= 0 ? lds_fft<INV>(d0, d1, N, g.logN, C, g.tw) : lds_fft_any<INV>(d0, d1, N, g.nrad, g.rad, C, g.tw);
  if (FUSED) {
    // For odd N the natural FFT index ky sits at centered position (ky + N/2) % N and, because
    // N/2 + (N+1)/2 == N, is already in ifftshift order for the inverse transform: the value stays in place.
    for (int idx = threadIdx.x; idx < C * N; idx += FFT_THREADS) {
      const int ky = idx / C, c = idx - ky * C;
      float2 v = r[c * ls + ky];
      const float sc = (flip && ((ky + half) & 1)) ? -g.scale : g.scale;
      if (x0 + c < g.W) v = mid(b, g.odd ? (ky + rout) % N : ky, x0 + c, make_float2(v.x * sc, v.y * sc));
      if (flip && (ky & 1)) { v.x = -v.x; v.y = -v.y; }
      r[c * ls + ky] = v;
    }
    float2* other = (r == d0) ? d1 : d0;
    r = g.logN >= 0 ? lds_fft<!INV>(r, other, N, g.logN, C, g.tw) : lds_fft_any<!INV>(r, other, N, g.nrad, g.rad, C, g.tw);
  }
  for (int idx = threadIdx.x; idx < C * N; idx += FFT_THREADS) {
    const int y = idx / C, c = idx - y * C;
    float2 v = r[c * ls + y];
    const float sc = (flip && ((y + half) & 1)) ? -g.scale : g.scale;
    if (x0 + c < g.W) st(b, g.odd ? (y + rout) % N : y, x0 + c, make_float2(v.x * sc, v.y * sc));
  }
}

// ---- common functors
struct LoadC {  // complex [n_img, H, W, 2]
  const float2* p;
  int H, W;
  __device__ float2 operator()(int b, int y, int x) const { return p[((size_t)b * H + y) * W + x]; }
};
struct StoreC {
  float2* p;
  int H, W;
  __device__ void operator()(int b, int y, int x, float2 v) const { p[((size_t)b * H + y) * W + x] = v; }
};

inline int ilog2_exact(int n) {
  if (n < 2 || n > 2048 || (n & (n - 1))) return -1;
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

struct FftPlan2D {
  PassGeom rows, cols;
  dim3 grid_rows, grid_cols;
  size_t lds_rows, lds_cols;
};

// Fills the launch geometry for an H x W transform over n_img images; tw tables come from the context.
int make_fft_plan(pnpx_ctx* ctx, int n_img, int H, int W, bool centered, FftPlan2D* out);

template <bool INV, class Load, class Store>
int launch_rows(const FftPlan2D& P, Load ld, Store st, hipStream_t s) {
  hipLaunchKernelGGL((fft_rows_kernel<INV, Load, Store>), P.grid_rows, dim3(FFT_THREADS), P.lds_rows, s, P.rows, ld,
                     st);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}
template <bool INV, bool FUSED, class Load, class Mid, class Store>
int launch_cols(const FftPlan2D& P, Load ld, Mid mid, Store st, hipStream_t s) {
  hipLaunchKernelGGL((fft_cols_kernel<INV, FUSED, Load, Mid, Store>), P.grid_cols, dim3(FFT_THREADS), P.lds_cols, s,
                     P.cols, ld, mid, st);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

}  // namespace pnpx
