"""Per-launch table of one denoiser forward (HIP events): name, ms, TFLOP/s.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import ops, synth
from tfpnp_amd.pnp import UNetDenoiser2D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=MODE)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
ctx = den.context(dev)
ops.unet_profile(ctx, x, s)
acc = None
R = 5
for _ in range(R):
    r = ops.unet_profile(ctx, x, s)
    acc = r if acc is None else [(a[0], a[1] + b[1], a[2]) for a, b in zip(acc, r)]
tot = 0
for i, (n, ms, fl) in enumerate(acc):
    ms /= R
    tot += ms
    print(f"{i:2d} {n:22s} {ms:8.3f} ms  {fl/ms/1e9 if fl else 0:8.1f} TF/s  {fl/1e9:10.1f} GF")
print("total", tot)
