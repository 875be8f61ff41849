"""Forward time at small batches (the per-rank batches of an 8 / 4-way strong split of env_batch 48) against the number of launch
chains, both convolution families.  usage: chains_small.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device("cuda:0")
for mode, opt in ((1, "chains"), (0, "fp32_chains")):
    den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=mode)
    ctx = den.context(dev)
    for B in (6, 12):
        x = torch.rand(B, 1, 256, 256, device=dev); s = torch.full((B,), 0.1, device=dev)
        row = []
        for c in (1, 2, 3, 4, 6):
            ctx.set_option(opt, c)
            best = 1e9
            for _ in range(3):
                for _ in range(5): den(x, s)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(40): den(x, s)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
            row.append(f"{c}: {best:.3f}")
        print(f"conv_mode {mode} B={B}  ms per forward by chains  " + "  ".join(row), flush=True)
