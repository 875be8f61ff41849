"""Denoiser forward time for every batch size 1..48 at 256x256 with the library PNPX_LIB selects (default: the production one): the
before / after table of a launch-table change.  usage: batch_sweep.py [conv_mode] [Bmax]   (prints "B ms" lines)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bmax = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=mode)
for B in range(1, bmax + 1):
    x = torch.rand(B, 1, 256, 256, device=dev); s = torch.full((B,), 0.1, device=dev)
    best = 1e9
    for _ in range(3):
        for _ in range(5): den(x, s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(25): den(x, s)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 25 * 1e3)
    print(f"{B} {best:.4f}", flush=True)
