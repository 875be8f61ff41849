"""Diagnostic: denoiser VJP error vs oracle autograd over a list of sizes (fresh context per size optional)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.golden_inputs import denoiser_inputs
from tfpnp_amd import synth, ops
from tfpnp_amd.pnp import UNetDenoiser2D
from oracle import pnp_oracle as O

params = synth.make_unet_params(20240607)
od = O.Denoiser(params)
dev = torch.device("cuda:0")
fresh = "--fresh" in sys.argv
den = UNetDenoiser2D(state_dict=params)
for (B, H, W) in [(2, 32, 32), (1, 48, 80), (2, 48, 80), (2, 32, 64), (2, 64, 32), (2, 48, 48), (2, 80, 80), (1, 50, 39), (2, 64, 64)]:
    if fresh:
        den = UNetDenoiser2D(state_dict=params)
    x, s = denoiser_inputs(B, H, W, 62)
    wts = np.random.RandomState(1).standard_normal((B, 1, H, W)).astype(np.float32)
    xt, st = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(s).requires_grad_(True)
    (od(xt, st) * torch.from_numpy(wts)).sum().backward()
    gx, gs = ops.unet_denoise_backward(den.context(dev), torch.from_numpy(x).to(dev), torch.from_numpy(s).to(dev),
                                       torch.from_numpy(wts).to(dev))
    d = (gx.cpu() - xt.grad)
    e = float(d.norm() / xt.grad.norm())
    idx = np.unravel_index(int(d.abs().argmax()), d.shape)
    print(f"{B}x{H}x{W}: grad_x rel {e:.2e}  max|d| {float(d.abs().max()):.3e} at {idx}  (|g|max {float(xt.grad.abs().max()):.3e})"
          f"  gs rel {float((gs.cpu()-st.grad).norm()/st.grad.norm()):.2e}", flush=True)
