"""8-wave Winograd kernel (conv3x3_wino8.hip) against the 4-wave one, the direct fp32 kernel and the fp64 oracle.  GPU box only.
usage: check_wino8.py [--big]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pnp_oracle as O
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D

dev = torch.device("cuda:0")
params = synth.make_unet_params(0)
den = UNetDenoiser2D(state_dict=params, conv_mode=0)
ctx = den.context(dev)
ALL = (1 << 27) - 1
p64 = {k: torch.as_tensor(v).double() for k, v in params.items()}
rel = lambda a, b: float((a - b).norm() / b.norm())
cases = [(2, 64, 64), (1, 128, 96), (3, 32, 64), (1, 32, 32), (5, 16, 32), (2, 256, 256)]
if "--big" in sys.argv:
    cases.append((48, 256, 256))
bad = 0
for B, H, W in cases:
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.rand(B, 1, H, W, generator=g)
    s = torch.rand(B, generator=g) * 0.2 + 0.02
    xt, st = x.to(dev), s.to(dev)
    outs = {}
    for name, (wg, w8, fu) in {"direct": (0, 0, 0), "wino4": (1, 0, 0), "wino8_nofuse": (1, ALL, 0), "wino8": (1, ALL, 1)}.items():
        ctx.set_option("fp32_winograd", wg)
        ctx.set_option("fp32_wino8_layers", w8)
        ctx.set_option("fp32_fuse_up", fu)
        post, pre = den.forward_preclamp(xt, st)
        torch.cuda.synchronize()
        outs[name] = (pre.double().cpu(), post.double().cpu())
        again = den.forward_preclamp(xt, st)[1].double().cpu()
        assert torch.equal(again, outs[name][0]), (name, "not deterministic")
    line = f"B={B} {H}x{W}: wino8 vs wino4 {rel(outs['wino8'][0], outs['wino4'][0]):.2e}  vs direct {rel(outs['wino8'][0], outs['direct'][0]):.2e}"
    if B * H * W <= 3 * 256 * 256:
        with torch.no_grad():
            sig = s.double().view(B, 1, 1, 1).expand(B, 1, H, W)
            ref = O.unet_forward(torch.cat([x.double(), sig], 1), p64)
        e8, e4, ed = rel(outs["wino8"][0], ref), rel(outs["wino4"][0], ref), rel(outs["direct"][0], ref)
        line += f"  | vs fp64: wino8 {e8:.2e} wino4 {e4:.2e} direct {ed:.2e}"
        ok = e8 < 3e-6 and torch.equal(outs["wino8"][1], outs["wino8"][0].clamp(0, 1))
    else:
        ok = rel(outs["wino8"][0], outs["wino4"][0]) < 3e-6
    fuse_same = torch.equal(outs["wino8"][0], outs["wino8_nofuse"][0])
    line += f"  | fused up-sampling vs separate kernel: {'bit-identical' if fuse_same else '%.2e' % rel(outs['wino8'][0], outs['wino8_nofuse'][0])}"
    same = torch.equal(outs["wino8"][0], outs["wino4"][0])
    print(line, "OK" if ok else "FAIL", "(identical to wino4: kernel did not run?)" if same else "", flush=True)
    bad += not ok
print("check_wino8:", "PASS" if not bad else f"{bad} FAILED")
sys.exit(1 if bad else 0)
