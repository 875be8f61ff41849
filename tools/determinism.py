"""Repeat denoiser forwards and ADMM iterations at several batch sizes and compare every output with the first one bit for bit.
usage: determinism.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI

R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
sol = ADMMSolver_CSMRI(den)
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
for (B, H) in [(1, 64), (1, 128), (1, 256), (2, 256), (3, 256), (4, 128), (6, 256), (8, 256), (12, 256), (24, 256), (48, 256), (5, 96)]:
    x = torch.rand(B, 1, H, H, generator=torch.Generator().manual_seed(B * 1000 + H)).to(dev)
    s = torch.full((B,), 0.1, device=dev)
    d = synth.make_csmri_batch(B, H, H, seed=7)
    a = synth.make_actions(B)[0]
    v0 = sol.reset({"x0": g(d["x0"])}); y0, m, sg, mu = g(d["y0"]), g(d["mask"]), g(a["sigma_d"]), g(a["mu"])
    ref = den(x, s).clone(); ref2 = sol((v0, (y0, m)), (sg, mu)).clone()
    n = 0
    for _ in range(R):
        n += int(not torch.equal(den(x, s), ref)) + int(not torch.equal(sol((v0, (y0, m)), (sg, mu)), ref2))
    bad += n
    print(f"B={B} {H}^2: {n} mismatches in {2 * R} repeats", flush=True)
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)
