"""A/B of a context option on one denoiser forward: bit-equality of the outputs and per-class launch times.
usage: ab_option.py option v0,v1,... [B] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import ops, synth
from tfpnp_amd.pnp import UNetDenoiser2D

opt = sys.argv[1]
vals = [int(v) for v in sys.argv[2].split(",")]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 48
H = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 1, H, H, generator=g).to(dev)
s = torch.full((B,), 0.1, device=dev)
ctx = den.context(dev)
ref = None
for rnd in range(2):
    for v in vals:
        ctx.set_option(opt, v)
        y, pre = den.forward_preclamp(x, s)
        if ref is None:
            ref = pre.clone()
        same = bool(torch.equal(pre, ref))
        ops.unet_profile(ctx, x, s)
        R = 8
        acc = None
        for _ in range(R):
            r = ops.unet_profile(ctx, x, s)
            acc = r if acc is None else [(a[0], a[1] + b[1], a[2]) for a, b in zip(acc, r)]
        conv = sum(ms for n, ms, fl in acc if fl) / R
        tot = sum(ms for n, ms, fl in acc) / R
        fl = sum(fl for n, ms, fl in acc if fl)
        per = " ".join(f"{ms / R:.3f}" for n, ms, fl in acc if fl)
        print(f"{opt}={v}: bit-equal to first: {same}  total {tot:.3f} ms conv {conv:.3f} ms = {fl / conv / 1e9:.1f} TF/s | {per}")
