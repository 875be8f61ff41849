// Half-split 3x3 convolution, residual-block epilogue act(conv + bias + res): the stride-1 convolutions that close
// the BasicBlocks of the policy ResNet (policy.hip; tfpnp/policy/network.py:33-58).  Kernel template: conv_hs_kernel.h.
#include "conv_hs_kernel.h"

namespace pnpx {

int launch_conv_hs_res(const ConvHsArgs& a, int mt, int B, hipStream_t s) {
  if (mt == 64) return launch_hs_mt<64, EPI_RES>(a, B, s);
  return launch_hs_mt<32, EPI_RES>(a, B, s);
}

}  // namespace pnpx
