"""Wall-clock A/B of context options on whole denoiser forwards and ADMM iterations (no per-launch events).
usage: ab_wall.py "opt=v,opt=v;opt=v" [B] [H]   (configurations separated by ';')"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D

cfgs = sys.argv[1].split(";")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 48
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 1, H, H, generator=g).to(dev)
s = torch.full((B,), 0.1, device=dev)
ctx = den.context(dev)
ref = None
for rnd in range(3):
    for cfg in cfgs:
        for kv in [c for c in cfg.split(",") if c]:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        y = den(x, s)
        if ref is None:
            ref = y.clone()
        same = bool(torch.equal(y, ref))
        for _ in range(20):
            den(x, s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 60
        for _ in range(N):
            den(x, s)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / N
        print(f"[{cfg}] forward {ms:.3f} ms  bit-equal {same}", flush=True)
