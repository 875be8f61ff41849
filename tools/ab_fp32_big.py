"""fp32 family at full batch against the half-split family, per layer class of the 8-wave Winograd kernel (GPU box).  usage: ab_fp32_big.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device("cuda:0")
params = synth.make_unet_params(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
g = torch.Generator().manual_seed(7)
x = torch.rand(B, 1, 256, 256, generator=g).to(dev); s = (torch.rand(B, generator=g) * 0.2 + 0.02).to(dev)
hs = UNetDenoiser2D(state_dict=params, conv_mode=1)
ref = hs.forward_preclamp(x, s)[1].double().cpu()
den = UNetDenoiser2D(state_dict=params, conv_mode=0)
ctx = den.context(dev)
ALL = (1 << 27) - 1
rel = lambda a, b: float((a - b).norm() / b.norm())
def run(name, **opts):
    for k, v in opts.items(): ctx.set_option(k, v)
    o = den.forward_preclamp(x, s)[1].double().cpu()
    bad = [b for b in range(B) if not torch.isfinite(o[b]).all() or rel(o[b], ref[b]) > 1e-4]
    print(f"{name}: rel {rel(o, ref):.3e} nan {int(torch.isnan(o).sum())} bad images {bad[:12]}", flush=True)
run("default")
run("no fuse_up", fp32_fuse_up=0)
run("no fuse_first", fp32_fuse_up=1, fuse_first=0)
run("wino4", fuse_first=1, fp32_wino8_layers=0)
run("wino8 only CT64 layers", fp32_wino8_layers=ALL & ~((1<<1)|(1<<2)|(1<<24)|(1<<25)|(1<<26)))
run("wino8 only CT32 plain", fp32_wino8_layers=(1<<1)|(1<<2)|(1<<25)|(1<<26))
run("wino8 only li24", fp32_wino8_layers=(1<<24))
run("default again", fp32_wino8_layers=ALL)
