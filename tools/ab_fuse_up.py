"""Half-split fuse_up 0 / 1 / 2: output difference and distance to the fp64 oracle, forward times (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from oracle import pnp_oracle as O
dev = torch.device("cuda:0")
params = synth.make_unet_params(0)
den = UNetDenoiser2D(state_dict=params)
ctx = den.context(dev)
p64 = {k: torch.as_tensor(v).double() for k, v in params.items()}
rel = lambda a, b: float((a - b).norm() / b.norm())
for B, H, W in [(2, 256, 256), (3, 64, 64), (1, 128, 96)]:
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 1, H, W, generator=g); s = torch.rand(B, generator=g) * 0.2 + 0.02
    xt, st = x.to(dev), s.to(dev)
    ctx.set_option("fuse_up", 0); a = den.forward_preclamp(xt, st)[1].double().cpu()
    ctx.set_option("fuse_up", 1); b = den.forward_preclamp(xt, st)[1].double().cpu()
    ctx.set_option("fuse_up", 2); c = den.forward_preclamp(xt, st)[1].double().cpu()
    with torch.no_grad():
        sig = s.double().view(B, 1, 1, 1).expand(B, 1, H, W)
        ref = O.unet_forward(torch.cat([x.double(), sig], 1), p64)
    print(f"{B}x{H}x{W}: fuse_up 1 vs 0: {rel(b, a):.2e}, 2 vs 0: {rel(c, a):.2e}; vs fp64: 0 {rel(a, ref):.2e} 1 {rel(b, ref):.2e} 2 {rel(c, ref):.2e}", flush=True)
for B in (48, 24, 6):
    x = torch.rand(B, 1, 256, 256, device=dev); s = torch.full((B,), 0.1, device=dev)
    row = []
    for rep in range(2):
        for f in (0, 1, 2):
            ctx.set_option("fuse_up", f)
            for _ in range(5): den(x, s)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): den(x, s)
            torch.cuda.synchronize(); row.append(f"{f}: {(time.perf_counter() - t0) / 30 * 1e3:.3f}")
    print(f"B={B} 256x256 forward ms by fuse_up  " + "  ".join(row), flush=True)
