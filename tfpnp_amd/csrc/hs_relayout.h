// Space-to-depth / depth-to-space re-layouts of HS8 tensors (record copies; shared by drunet.hip and policy.hip; `static`:
// one copy per translation unit).  Channel order is phase-major: group (dy*2+dx)*G + g.
#pragma once
#include "hs_rec.h"

namespace pnpx {

// [B][G][H+2][W+2] -> [B][4G][H/2+2][W/2+2]; output group = (dy*2+dx)*G + g holds input pixel (2y+dy, 2x+dx)
static __global__ __launch_bounds__(256) void hs_s2d_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int G, int H,
                                                      int W, size_t n) {   // n = B*4G*(H/2)*(W/2)*2 16-byte pieces
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int piece = (int)(i & 1);
  size_t t = i >> 1;
  const int Wo = W / 2, Ho = H / 2;
  const int x = (int)(t % Wo);
  t /= Wo;
  const int y = (int)(t % Ho);
  t /= Ho;
  const int go = (int)(t % (4 * G));
  const size_t b = t / (4 * G);
  const int ph = go / G, g = go - ph * G;
  const size_t s = ((b * G + g) * (H + 2) + (2 * y + (ph >> 1) + 1)) * (W + 2) + 2 * x + (ph & 1) + 1;
  const size_t d = ((b * 4 * G + go) * (Ho + 2) + (y + 1)) * (Wo + 2) + x + 1;
  dst[d * 2 + piece] = src[s * 2 + piece];
}

// [B][4G][h+2][w+2] -> [B][G][2h+2][2w+2]; input group (dy*2+dx)*G + g at (y, x) lands at (2y+dy, 2x+dx) of group g
static __global__ __launch_bounds__(256) void hs_d2s_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int G, int h,
                                                      int w, size_t n) {   // n = B*G*(2h)*(2w)*2 pieces
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int piece = (int)(i & 1);
  size_t t = i >> 1;
  const int W = 2 * w, H = 2 * h;
  const int X = (int)(t % W);
  t /= W;
  const int Y = (int)(t % H);
  t /= H;
  const int g = (int)(t % G);
  const size_t b = t / G;
  const int ph = (Y & 1) * 2 + (X & 1);
  const size_t s = ((b * 4 * G + ph * G + g) * (h + 2) + (Y / 2 + 1)) * (w + 2) + X / 2 + 1;
  const size_t d = ((b * G + g) * (H + 2) + (Y + 1)) * (W + 2) + X + 1;
  dst[d * 2 + piece] = src[s * 2 + piece];
}

}  // namespace pnpx
