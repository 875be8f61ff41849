"""Time the denoiser forward (HS), the VJP (fp32 recompute + adjoint convs) and one differentiable ADMM step."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from tfpnp_amd import synth, ops
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks import csmri

dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(1))
for (B, H, W) in [(48, 128, 128), (16, 256, 256), (48, 256, 256)]:
    x = torch.rand(B, 1, H, W, device=dev)
    s = torch.full((B,), 0.1, device=dev)
    gr = torch.randn(B, 1, H, W, device=dev)
    ctx = den.context(dev)
    for name, fn in [("fwd_hs", lambda: ops.unet_denoise(ctx, x, s)), ("vjp", lambda: ops.unet_denoise_backward(ctx, x, s, gr))]:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 5
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"B={B} {H}x{W} {name}: {(time.time() - t0) / n * 1e3:.2f} ms", flush=True)
    d = synth.make_csmri_batch(B, H, W, seed=3)
    sol = csmri.ADMMSolver_CSMRI(den)
    tg = lambda a: torch.from_numpy(a).to(dev)
    v0 = sol.reset({"x0": tg(d["x0"])})
    sig = torch.full((B, 5), 0.1, device=dev, requires_grad=True)
    mu = torch.full((B, 5), 0.5, device=dev, requires_grad=True)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        out = sol((v0, (tg(d["y0"]), tg(d["mask"]))), (sig, mu))
        out.sum().backward()
        torch.cuda.synchronize()
        print(f"B={B} {H}x{W} admm 5 iters fwd+bwd: {(time.time() - t0) * 1e3:.1f} ms  mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB torch", flush=True)
