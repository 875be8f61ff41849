"""Denoiser forward time over a (batch, size) grid, as ns per pixel: outliers = launch-shape pathologies.  usage: size_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
sizes = [(64, 64), (96, 96), (128, 128), (160, 128), (192, 192), (256, 256), (320, 256), (384, 384), (512, 512)]
print("B \\ HxW " + " ".join(f"{h}x{w:<4d}" for h, w in sizes))
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
    row = []
    for (H, W) in sizes:
        if B * H * W > 48 * 512 * 512 // 2:
            row.append("   -   ")
            continue
        x = torch.rand(B, 1, H, W, device=dev); s = torch.full((B,), 0.1, device=dev)
        for _ in range(5): den(x, s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): den(x, s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        row.append(f"{ms:6.3f}/{ms * 1e6 / (B * H * W):5.1f}")
    print(f"{B:3d}      " + " ".join(row), flush=True)
print("(ms per forward / ns per pixel)")
