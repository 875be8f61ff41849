// Pieces shared by the two Winograd F(2x2, 3x3) fp32-MFMA kernels: conv3x3_wino.hip (4 waves, 16 positions per wave) and
// conv3x3_wino8.hip (8 waves, the 16 positions split over the two waves of a SIMD).
#pragma once
#include <utility>

#include "common.h"

namespace pnpx {
namespace wino {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct WinoArgs {
  const float* in0;
  const float* in1;
  const float* u;      // [cout/CT][cin/CK][a 4][b 4][kg 2][half][m CT][4] fp32
  const float* bias;
  const float* res;    // optional: added after the activation (same geometry as out); res_mask != 0 (8-wave kernel, adjoint convolutions):
                       // the saved forward activation instead -- the result is multiplied by LeakyReLU'(res) = res > 0 ? 1 : mask_slope
  // FUSE_OUTC instances (32-cout layer = the UNet's last): the 1x1 out-conv + residual + clamp of models/unet.py:63-66,124-131 and
  // denoiser/base.py:32 in the epilogue; `out` (the 32-channel tensor) is then neither written nor read again
  const float* outc_w;   // [32]
  const float* outc_b;   // [1]
  const float* x_img;    // [B][H][W] the network's input image (residual)
  float* img;            // [B][H][W] clamped result
  float* img_pre;        // [B][H][W] pre-clamp result (== img when the caller wants none: the clamped store lands second)
  float* pool;         // optional: MaxPool2d(2) of the activated output, padded planar [B][Cout][H/2 + 2][W/2 + 2 PADL] (a 2x2 tile = one lane)
  float* out;
  int B, H, W, Hp, Wp, C0, C1, Cout, nct, nch, rx, ry;
  float slope;
  // UPS instances of the 8-wave kernel (the UNet's decoder entries): in1 is the LOW-resolution tensor [B][C1][ups_h + 2][ups_w + 2 PADL]
  // and the kernel interpolates its bilinear x2 (align_corners) up-sampling into the halo buffer itself (models/unet.py:92-121)
  int ups_h, ups_w;
  float ups_sy, ups_sx;
  int res_mask;
  float mask_slope;
  // KSPLIT instances of the 8-wave kernel (r6): the channel chunks of a tile are dealt to `ksplit` work items (nch = chunks per item,
  // nch_all = per tile); every item writes its output-transformed partial sums to `part` ([tile][piece][wave][row 8][lane 64] f32x4) and
  // a second launch (wino8_ksplit_finish_kernel) adds the pieces in piece order, then bias + activation + stores
  int ksplit, nch_all;
  float* part;
};

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the builtin widens every lane offset to a
// 64-bit VGPR pair (21 + 4 pairs live across the whole kernel here), which is what pushed this kernel into scratch
// (the base is wave-uniform by construction; the explicit readfirstlane is free when the compiler already knows it and keeps the "s"
// operand valid when its divergence analysis gives up)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void glds4(const void* base, unsigned voff, unsigned lds_addr) {
  base = uniform_ptr(base);
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void glds16(const void* base, unsigned voff, unsigned lds_addr) {
  base = uniform_ptr(base);
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

}  // namespace wino
}  // namespace pnpx
