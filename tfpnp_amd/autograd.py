"""Differentiable wrappers of the native ops -- building blocks of the reference's training path.

The reference trains its policy through the solver: PnPEnv.forward (tfpnp/env/base.py:193-206) runs
solver.forward under autograd and back-propagates the critic's value / the PSNR reward into the policy's
(sigma_d, mu, tau) outputs (tfpnp/trainer/mddpg/trainer.py:171-192).  The solvers in tfpnp_amd/tasks do that with
FUSED native training loops (pnpx_*_train keeps what its pnpx_*_backward needs; one call each way per solver call).
This module is the other way to differentiate the path -- the same iteration composed step by step from ops that each
carry a native VJP -- used by user code that changes an iteration, and by the tests as the independent check of the
fused loops (tests/composed_solvers.py):
  * denoise(x, sigma)      native forward, native VJP (pnpx_unet_denoise_backward: activations from the training ring or
                           re-computed, transposed MFMA convolutions), dispatcher-registered ops (torch_ops.py)
  * fft2 / ifft2 / plain FFTs   native; the transforms are unitary, so the VJP is the inverse transform
  * radon forward / backprojection   native; each is the other's VJP (the unmatched pair torch_radon also uses)
and ordinary PyTorch pointwise autograd for the O(N) glue (masks, blends, dual updates).
"""
import torch

from . import torch_ops as T


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def denoise(native_ctx, x, sigma):
    if needs_grad(x, sigma):      # activations parked in the context's training ring for the registered VJP
        return T.call("unet_denoise_train", x, sigma, native_ctx.cid)[0]
    return T.call("unet_denoise", x, sigma, native_ctx.cid)


def fft2(x, inverse=False, centered=True):
    return T.call("fft2", x, bool(inverse), bool(centered))          # unitary: the VJP is the inverse transform


def radon_forward(img, n_view):
    return T.call("radon_forward", img, int(n_view))                 # VJP = backprojection


def radon_backprojection(sino, R):
    return T.call("radon_backprojection", sino, int(R))              # VJP = forward projection


# ---- complex helpers (pure layout / pointwise, autograd by PyTorch) -- tfpnp/utils/transforms.py:12-17,260-274
def r2c(x):
    return torch.stack([x, torch.zeros_like(x)], dim=-1)


def c2r(x):
    return x[..., 0]


def cmul(a, b):
    return torch.stack((a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1], a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0]), -1)


def conj(x):
    return torch.stack([x[..., 0], -x[..., 1]], -1)
