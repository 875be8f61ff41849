import sys
import torch
sys.path.insert(0, ".")
from tfpnp_amd import synth, ops
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(1))
B, H = 48, 256
x = torch.rand(B, 1, H, H, device=dev); s = torch.full((B,), 0.1, device=dev); g = torch.randn(B, 1, H, H, device=dev)
for _ in range(2):
    ops.unet_denoise_backward(den.context(dev), x, s, g)
torch.cuda.synchronize()
