"""Sweep conv_hs launch configurations (env hook PNPX_HS_<MT>_<W>="nbw,nstage,per_cu") and report per-class time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import ops, synth
from tfpnp_amd.pnp import UNetDenoiser2D

B, H = 48, 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
ctx = den.context(dev)
ref = den(x, s).clone()

CLASSES = {  # key -> layer indices in the profile list
    "32_256": [2, 3, 33, 34, 35], "64_128": [5, 6, 7, 29, 30, 31], "64_64": [9, 10, 11, 25, 26, 27],
    "64_32": [13, 14, 15, 21, 22, 23], "64_16": [17, 18, 19]}

def measure(R=3):
    ops.unet_profile(ctx, x, s)
    acc = None
    for _ in range(R):
        r = ops.unet_profile(ctx, x, s)
        acc = [a[1] for a in r] if acc is None else [u + a[1] for u, a in zip(acc, r)]
    return [v / R for v in acc]

base = measure()
print("baseline total conv", sum(base[i] for c in CLASSES.values() for i in c))
for key, cfgs in {
    "32_256": ["4,2,1", "2,2,2", "2,2,1", "2,1,2", "4,1,1", "1,2,2", "1,1,2"],
    "64_128": ["4,2,1", "2,2,1", "2,1,2", "1,2,1", "1,1,2", "4,1,1"],
    "64_64": ["4,2,1", "2,2,1", "2,1,2", "1,1,2", "1,2,1"],
    "64_32": ["2,2,1", "2,1,2", "1,2,1", "1,1,2", "4,2,1"],
    "64_16": ["1,2,1", "1,1,2", "2,2,1", "2,1,2"],
}.items():
    for cfg in cfgs:
        os.environ["PNPX_HS_" + key] = cfg
        try:
            t = measure()
            ok = bool(torch.allclose(den(x, s), ref, atol=1e-5))
            print(f"{key:7s} {cfg:6s} {sum(t[i] for i in CLASSES[key]):7.3f} ms  per-layer " +
                  " ".join(f"{t[i]:.3f}" for i in CLASSES[key]) + ("" if ok else "  MISMATCH"))
        except Exception as e:
            print(key, cfg, "ERR", str(e)[:100])
    os.environ.pop("PNPX_HS_" + key)
