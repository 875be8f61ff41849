"""n plain denoiser forwards (for rocprofv3 passes).  usage: run_denoiser.py [B] [H] [n] [conv_mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=int(sys.argv[4]) if len(sys.argv) > 4 else None)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
for _ in range(n):
    y = den(x, s)
torch.cuda.synchronize()
print("ok", float(y.mean()))
