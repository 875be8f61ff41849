"""Denoiser VJP alone (activation ring on): ms per call at B=48, 256^2.  usage: [PNPX_LIB=...] time_vjp.py [B] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import ops, synth
from tfpnp_amd.pnp import UNetDenoiser2D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
ctx = den.context(dev)
x = torch.rand(B, 1, H, H, device=dev); s = torch.full((B,), 0.1, device=dev); g = torch.randn(B, 1, H, H, device=dev)
out, ticket = ops.unet_denoise_train(ctx, x, s)
ref = None
for rnd in range(3):
    for _ in range(3):
        r = ops.unet_denoise_backward(ctx, x, s, g, ticket=ticket)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        r = ops.unet_denoise_backward(ctx, x, s, g, ticket=ticket)
    torch.cuda.synchronize()
    ref = r[0].clone() if ref is None else ref
    print(f"VJP B={B} {H}^2: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms  checksum {float(r[0].double().sum()):.10e} bit-equal {bool(torch.equal(r[0], ref))}", flush=True)
