// The 32-byte record of the half-split (HS8) activation layout: 8 channels of one pixel as hi[8] | lo[8] f16, the value
// (times HS_ASCALE) being hi + lo.  Tensors are [B][C/8][H+2][W+2] records with a zero border (conv_hs.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace pnpx {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
struct HsRec {
  h8v hi, lo;
};
__device__ __forceinline__ void hs_unpack(const HsRec& r, float v[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)r.hi[e] + (float)r.lo[e];
}
__device__ __forceinline__ HsRec hs_pack(const float v[8]) {
  HsRec r;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    r.hi[e] = (_Float16)v[e];
    r.lo[e] = (_Float16)(v[e] - (float)r.hi[e]);
  }
  return r;
}

}  // namespace pnpx
