"""Forward time of the fp32 family per K-split rule (fp32_ksplit = 2: split at every batch size; fp32_ksplit_rule: 1 default = 4 pieces at <= 8
tiles per image and 2 at <= 16, else pieces for the two classes in bits 0-3 / 4-7; column 0 = fp32_ksplit 0) and batch size, two launch chains.  usage: python tools/ab_ksplit_rules.py [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
ctx = den.context(dev)
rules = [0, 1, 0x04, 0x02, 0x22, 0x12, 0x14]
print("| B | " + " | ".join(f"rule {r:#04x}" for r in rules) + " |\n|---|" + "---|" * len(rules))
for B in (3, 6, 12, 24, 48):
    x = torch.rand(B, 1, H, H, device=dev)
    s = torch.full((B,), 0.1, device=dev)
    row = []
    for rep in range(2):
        row = []
        for r in rules:
            ctx.set_option("fp32_ksplit", 2 if r else 0)
            if r:
                ctx.set_option("fp32_ksplit_rule", r)
            for _ in range(3):
                den(x, s)
            torch.cuda.synchronize()
            n = 20 if B <= 12 else 10
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(n):
                    den(x, s)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n * 1e3)
            row.append(best)
    print(f"| {B} | " + " | ".join(f"{t:.3f}" for t in row) + " |")
