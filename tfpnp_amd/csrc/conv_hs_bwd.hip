// Half-split 3x3 convolution, input-gradient epilogue (LeakyReLU' from the saved forward activation): the adjoint
// convolutions of the denoiser VJP (unet_bwd.hip).  Kernel template: conv_hs_kernel.h.
#include "conv_hs_kernel.h"

namespace pnpx {

int launch_conv_hs_dmask(const ConvHsArgs& a, int mt, int B, hipStream_t s) {
  if (mt == 64) return launch_hs_mt<64, EPI_DMASK>(a, B, s);
  return launch_hs_mt<32, EPI_DMASK>(a, B, s);
}

}  // namespace pnpx
