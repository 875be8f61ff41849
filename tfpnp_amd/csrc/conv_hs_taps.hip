// Half-split convolution, sparse-tap instances (plain activation epilogue): 1x1 layers (tap mask 0x010: DRUNet's strided /
// transposed 2x2 convolutions as space-to-depth 1x1 layers, the policy ResNet's stride-2 1x1 shortcuts) and the 2x2-window
// form of a stride-2 3x3 convolution over its space-to-depth input (0x01B: policy ResNet stage entries,
// tfpnp/policy/network.py:33-58).  Only the layer's taps are stored, copied and multiplied.  Kernel template: conv_hs_kernel.h.
#include "conv_hs_kernel.h"

namespace pnpx {

int launch_conv_hs_taps(const ConvHsArgs& a, int mt, int taps, int B, hipStream_t s) {
  if (taps == 0x010) return mt == 64 ? launch_hs_mt<64, EPI_ACT, 0x010>(a, B, s) : launch_hs_mt<32, EPI_ACT, 0x010>(a, B, s);
  if (taps == 0x01B) return mt == 64 ? launch_hs_mt<64, EPI_ACT, 0x01B>(a, B, s) : launch_hs_mt<32, EPI_ACT, 0x01B>(a, B, s);
  set_error("conv_hs: no instance for tap mask 0x%x", taps);
  return PNPX_ERR_SHAPE;
}

}  // namespace pnpx
