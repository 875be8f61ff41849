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
#include <type_traits>
#include <utility>

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
struct MidNonePre {   // placeholder type for functors without a fetch / apply split
  struct Pre {};
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


// ------------------------------------------------------------------------------------------------ N = 256 fast path (r4)
// 256 = 16 x 16: a thread holds 16 points of one line in registers, so a 256-point transform is two radix-16 register
// butterflies with ONE exchange through LDS in between (the generic Stockham path: four radix-4 stages, each a full LDS
// round trip + barrier).  Index split  n = n1 + 16 n2,  k = 16 k1 + k2:
//     X[16 k1 + k2] = sum_n1 W16^(n1 k1) * [ W256^(n1 k2) * sum_n2 x[n1 + 16 n2] W16^(n2 k2) ]
// step 1: thread n1 loads x[n1 + 16 n2] (16 consecutive lanes = 128 contiguous bytes per n2), 16-point FFT over n2, twiddle;
// exchange; step 2: thread k2 holds the 16 values over n1, 16-point FFT -> X[16 k1 + k2] (again 128-byte runs per k1).
// A tile is 16 lines (4096 points, 34 KiB of LDS): rows = 16 image rows, columns = 16 ADJACENT columns, whose loads are the
// same 128-byte runs (the generic column tile is 4 columns = 32-byte pieces).  In the fused column pass the forward result
// X[k2 + 16 k1] sits in exactly the (n1, n2) register layout the inverse transform starts from: forward FFT -> k-space
// functor -> inverse FFT without another re-layout.  Centering signs: (-1)^n = (-1)^n1 and (-1)^(k + 128) = (-1)^k2 are
// per-thread constants.
template <bool INV>
__device__ __forceinline__ float2 cmulw(float2 v, float c, float s) {   // v * (c - i s)  (INV: v * (c + i s))
  return INV ? make_float2(v.x * c - v.y * s, v.y * c + v.x * s) : make_float2(v.x * c + v.y * s, v.y * c - v.x * s);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2 a0, float2 a1, float2 a2, float2 a3, float2& o0, float2& o1, float2& o2, float2& o3) {
  const float2 s02 = make_float2(a0.x + a2.x, a0.y + a2.y), d02 = make_float2(a0.x - a2.x, a0.y - a2.y);
  const float2 s13 = make_float2(a1.x + a3.x, a1.y + a3.y), d13 = make_float2(a1.x - a3.x, a1.y - a3.y);
  const float2 j = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);   // (+-i) * d13
  o0 = make_float2(s02.x + s13.x, s02.y + s13.y);
  o2 = make_float2(s02.x - s13.x, s02.y - s13.y);
  o1 = make_float2(d02.x + j.x, d02.y + j.y);
  o3 = make_float2(d02.x - j.x, d02.y - j.y);
}
// in-place 16-point DFT, natural order in and out:  v[k] <- sum_n v[n] W16^(n k),  W16 = exp(-+ 2 pi i / 16)
template <bool INV>
__device__ __forceinline__ void fft16_reg(float2 (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
  float2 T[4][4];   // T[na][kb], n = na + 4 nb, k = 4 ka + kb
#pragma unroll
  for (int na = 0; na < 4; ++na) dft4<INV>(v[na], v[na + 4], v[na + 8], v[na + 12], T[na][0], T[na][1], T[na][2], T[na][3]);
  // W16^(na kb): exponents 1, 2, 3 | 2, 4, 6 | 3, 6, 9
  T[1][1] = cmulw<INV>(T[1][1], C1, S1);
  T[1][2] = cmulw<INV>(T[1][2], C2, C2);
  T[1][3] = cmulw<INV>(T[1][3], S1, C1);
  T[2][1] = cmulw<INV>(T[2][1], C2, C2);
  T[2][2] = cmulw<INV>(T[2][2], 0.f, 1.f);
  T[2][3] = cmulw<INV>(T[2][3], -C2, C2);
  T[3][1] = cmulw<INV>(T[3][1], S1, C1);
  T[3][2] = cmulw<INV>(T[3][2], -C2, C2);
  T[3][3] = cmulw<INV>(T[3][3], -C1, -S1);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) dft4<INV>(T[0][kb], T[1][kb], T[2][kb], T[3][kb], v[kb], v[4 + kb], v[8 + kb], v[12 + kb]);
}

constexpr int FFT256_LINES = 16;
constexpr int FFT256_LDS_F2 = 16 * 16 * 17;   // float2 words of the exchange buffer (34 KiB)

// one 256-point transform per 16 threads.  `ex`: the workgroup's exchange buffer; `row_major`: threads of a line are the 16
// consecutive lanes (row tiles: lane = index within the line) or strided by 16 (column tiles: lane = column).  `lin`: line
// of the tile (0..15), `i`: this thread's n1 (then k2).  tw: W256^(i * m), m = 0..15 (forward sign).
template <bool INV, bool ROWS>
__device__ __forceinline__ void fft256_reg(float2 (&v)[16], float2* ex, int lin, int i, const float2 (&tw)[16]) {
  fft16_reg<INV>(v);
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) v[k2] = cmulw<INV>(v[k2], tw[k2].x, -tw[k2].y);   // table holds exp(-i t): c = x, s = -y
  // exchange: written as [n1 = i][k2], read as [n1][k2 = i].  ROWS: word = (lin * 16 + n1) * 17 + k2 (lanes = n1: stride 17
  // words = 34 dwords, conflict-free; reads contiguous over k2).  Columns: word = (n1 * 17 + k2) * 16 + lin (lanes = lin:
  // contiguous both ways; the wave's four n1 / k2 values are 16 * 17 resp. 16 words apart = 32 mod 64 dwords).
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) ex[ROWS ? (lin * 16 + i) * 17 + k2 : (i * 17 + k2) * 16 + lin] = v[k2];
  __syncthreads();
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) v[n1] = ex[ROWS ? (lin * 16 + n1) * 17 + i : (n1 * 17 + i) * 16 + lin];
  fft16_reg<INV>(v);
}

// (the split functors of the fused passes: see MidHasFetch below; a Store with fetch(b, y, x) -> Pre / apply(pre, b, y, x, v)
// has its global READS issued beside the tile's loads instead of behind the transform)
template <class M, class = void>
struct FunctorHasFetch : std::false_type {};
template <class M>
struct FunctorHasFetch<M, std::void_t<decltype(std::declval<const M&>().fetch(0, 0, 0))>> : std::true_type {};
struct NoPre {
  struct Pre {};
};

template <bool INV, class Load, class Store>
__global__ __launch_bounds__(FFT_THREADS) void fft256_rows_kernel(PassGeom g, Load ld, Store st) {
  __shared__ float2 ex[FFT256_LDS_F2];
  const int i = threadIdx.x & 15, l = threadIdx.x >> 4;
  int b, part;
  xcd_affine_decode(blockIdx.x, g.H / FFT256_LINES, g.n_img, g.affine, &b, &part);
  const int y = part * FFT256_LINES + l;
  float2 tw[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) tw[m] = g.tw[(i * m) & 255];
  const float sg = (g.centered && (i & 1)) ? -1.f : 1.f;     // (-1)^n on the way in, (-1)^(k + 128) on the way out
  float2 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float2 t = ld(b, y, i + 16 * j);
    v[j] = make_float2(t.x * sg, t.y * sg);
  }
  constexpr bool PREFETCH = FunctorHasFetch<Store>::value;
  [[maybe_unused]] typename std::conditional<PREFETCH, Store, NoPre>::type::Pre pf[PREFETCH ? 16 : 1];
  if constexpr (PREFETCH) {
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) pf[k1] = st.fetch(b, y, 16 * k1 + i);
  }
  fft256_reg<INV, true>(v, ex, l, i, tw);
  const float sc = g.scale * sg;
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const float2 o = make_float2(v[k1].x * sc, v[k1].y * sc);
    if constexpr (PREFETCH) st.apply(pf[k1], b, y, 16 * k1 + i, o);
    else st(b, y, 16 * k1 + i, o);
  }
}

// Row pass over GROUPS of images (r5; the PR solver's inverse row pass): a workgroup transforms the same 16 rows of the S images of one
// group one after the other and hands every result to an accumulator functor -- acc.add(state, group, s, y, x, value) -- whose per-pixel
// state stays in registers; acc.finish(state, group, y, x) stores once per pixel.  For the coded-diffraction adjoint (mean over the S
// masks of conj(mask_s) F^-1, transforms.py:304-320) this removes the S image-space fields' round trip through memory: the row pass wrote
// them (S x 8 N bytes) and the update kernel read them back.  g.n_img = number of groups.
template <bool INV, class Load, class Acc>
__global__ __launch_bounds__(FFT_THREADS) void fft256_rows_group_kernel(PassGeom g, int S, Load ld, Acc acc) {
  __shared__ float2 ex[FFT256_LDS_F2];
  const int i = threadIdx.x & 15, l = threadIdx.x >> 4;
  int b, part;
  xcd_affine_decode(blockIdx.x, g.H / FFT256_LINES, g.n_img, g.affine, &b, &part);
  const int y = part * FFT256_LINES + l;
  float2 tw[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) tw[m] = g.tw[(i * m) & 255];
  const float sg = (g.centered && (i & 1)) ? -1.f : 1.f;
  const float sc = g.scale * sg;
  typename Acc::State st[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) st[k1] = acc.init();
  for (int s = 0; s < S; ++s) {
    const int img = b * S + s;
    float2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 t = ld(img, y, i + 16 * j);
      v[j] = make_float2(t.x * sg, t.y * sg);
    }
    typename Acc::Pre pf[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) pf[k1] = acc.fetch(img, y, 16 * k1 + i);
    fft256_reg<INV, true>(v, ex, l, i, tw);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) acc.add(st[k1], pf[k1], make_float2(v[k1].x * sc, v[k1].y * sc));
  }
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) acc.finish(st[k1], b, y, 16 * k1 + i);
}

// A k-space functor may split itself into fetch(b, ky, kx) -> Pre (its global reads: measurements, mask ...) and
// apply(pre, v): the fused column kernel then issues those reads up front, beside the tile's own loads, instead of behind the
// forward transform (a second exposed memory latency per workgroup: 31 -> 2x us on the CS-MRI blend pass).
template <class M>
using MidHasFetch = FunctorHasFetch<M>;

template <bool INV, bool FUSED, class Load, class Mid, class Store>
__global__ __launch_bounds__(FFT_THREADS) void fft256_cols_kernel(PassGeom g, Load ld, Mid mid, Store st) {
  __shared__ float2 ex[FFT256_LDS_F2];
  const int c = threadIdx.x & 15, i = threadIdx.x >> 4;
  int b, xt;
  xcd_affine_decode(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, g.n_img, g.affine, &b, &xt);
  const int x = xt * FFT256_LINES + c;
  float2 tw[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) tw[m] = g.tw[(i * m) & 255];
  const float sg = (g.centered && (i & 1)) ? -1.f : 1.f;
  float2 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float2 t = ld(b, i + 16 * j, x);
    v[j] = make_float2(t.x * sg, t.y * sg);
  }
  constexpr bool PREFETCH = FUSED && MidHasFetch<Mid>::value;
  [[maybe_unused]] typename std::conditional<PREFETCH, Mid, MidNonePre>::type::Pre pf[PREFETCH ? 16 : 1];
  if constexpr (PREFETCH) {
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) pf[k1] = mid.fetch(b, 16 * k1 + i, x);
  }
  fft256_reg<INV, false>(v, ex, c, i, tw);
  const float sc = g.scale * sg;
  if (FUSED) {
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      const float2 in = make_float2(v[k1].x * sc, v[k1].y * sc);
      float2 t;
      if constexpr (PREFETCH) t = mid.apply(pf[k1], b, in);
      else t = mid(b, 16 * k1 + i, x, in);
      v[k1] = make_float2(t.x * sg, t.y * sg);               // (-1)^ky of the inverse transform's input
    }
    __syncthreads();                                          // everyone has read the first exchange
    fft256_reg<!INV, false>(v, ex, c, i, tw);
  }
#pragma unroll
  for (int y1 = 0; y1 < 16; ++y1) st(b, 16 * y1 + i, x, make_float2(v[y1].x * sc, v[y1].y * sc));
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
  // N = 256 register-radix-16 kernels (fft256_*_kernel) instead of the generic LDS Stockham passes
  bool fast256_rows = false, fast256_cols = false;
};

// Fills the launch geometry for an H x W transform over n_img images; tw tables come from the context.
int make_fft_plan(pnpx_ctx* ctx, int n_img, int H, int W, bool centered, FftPlan2D* out);

template <bool INV, class Load, class Store>
int launch_rows(const FftPlan2D& P, Load ld, Store st, hipStream_t s) {
  if (P.fast256_rows) {
    hipLaunchKernelGGL((fft256_rows_kernel<INV, Load, Store>), dim3(P.rows.n_img * (P.rows.H / FFT256_LINES)), dim3(FFT_THREADS), 0, s,
                       P.rows, ld, st);
    PNPX_LAUNCH_CHECK();
    return PNPX_OK;
  }
  hipLaunchKernelGGL((fft_rows_kernel<INV, Load, Store>), P.grid_rows, dim3(FFT_THREADS), P.lds_rows, s, P.rows, ld,
                     st);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}
// (fast 256-point path only: the caller checks P.fast256_rows and P.rows.n_img == groups * S)
template <bool INV, class Load, class Acc>
int launch_rows_group(const FftPlan2D& P, int groups, int S, Load ld, Acc acc, hipStream_t s) {
  PassGeom g = P.rows;
  g.n_img = groups;
  hipLaunchKernelGGL((fft256_rows_group_kernel<INV, Load, Acc>), dim3(groups * (g.H / FFT256_LINES)), dim3(FFT_THREADS), 0, s, g, S, ld, acc);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}
template <bool INV, bool FUSED, class Load, class Mid, class Store>
int launch_cols(const FftPlan2D& P, Load ld, Mid mid, Store st, hipStream_t s) {
  if (P.fast256_cols) {
    hipLaunchKernelGGL((fft256_cols_kernel<INV, FUSED, Load, Mid, Store>), dim3(P.cols.W / FFT256_LINES, P.cols.n_img), dim3(FFT_THREADS),
                       0, s, P.cols, ld, mid, st);
    PNPX_LAUNCH_CHECK();
    return PNPX_OK;
  }
  hipLaunchKernelGGL((fft_cols_kernel<INV, FUSED, Load, Mid, Store>), P.grid_cols, dim3(FFT_THREADS), P.lds_cols, s,
                     P.cols, ld, mid, st);
  PNPX_LAUNCH_CHECK();
  return PNPX_OK;
}

}  // namespace pnpx
