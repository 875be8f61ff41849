"""Differentiable wrappers of the native ops -- the training path of the reference.

The reference trains its policy through the solver: PnPEnv.forward (tfpnp/env/base.py:193-206) runs
solver.forward under autograd and back-propagates the critic's value / the PSNR reward into the policy's
(sigma_d, mu, tau) outputs (tfpnp/trainer/mddpg/trainer.py:171-192).  The fused native solver loops are
inference-only; when gradients are required the solvers in tfpnp_amd/tasks fall back to the reference's own
iteration written with these differentiable building blocks:
  * denoise(x, sigma)      native forward, native VJP (pnpx_unet_denoise_backward: fp32 re-computation + transposed
                           MFMA convolutions)
  * fft2 / ifft2 / plain FFTs   native; the transforms are unitary, so the VJP is the inverse transform
  * radon forward / backprojection   native; each is the other's VJP (the unmatched pair torch_radon also uses)
and ordinary PyTorch pointwise autograd for the O(N) glue (masks, blends, dual updates).
"""
import torch

from . import ops


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


class _Denoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sigma, native_ctx):
        ctx.native = native_ctx
        ctx.save_for_backward(x, sigma)
        return ops.unet_denoise(native_ctx, x, sigma)

    @staticmethod
    def backward(ctx, grad_out):
        x, sigma = ctx.saved_tensors
        gx, gs = ops.unet_denoise_backward(ctx.native, x, sigma.reshape(-1), grad_out)
        return gx, gs.view_as(sigma), None


def denoise(native_ctx, x, sigma):
    return _Denoise.apply(x, sigma, native_ctx)


class _FFT2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inverse, centered):
        ctx.inverse, ctx.centered = inverse, centered
        return ops.fft2(x, inverse=inverse, centered=centered)

    @staticmethod
    def backward(ctx, g):   # orthonormal transform (+ index permutations): adjoint == inverse
        return ops.fft2(g, inverse=not ctx.inverse, centered=ctx.centered), None, None


def fft2(x, inverse=False, centered=True):
    return _FFT2.apply(x, inverse, centered)


class _RadonForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, n_view):
        ctx.R = img.shape[-1]
        return ops.radon_forward(img, n_view)

    @staticmethod
    def backward(ctx, g):
        return ops.radon_backprojection(g, ctx.R), None


class _RadonBackprojection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sino, R):
        ctx.V = sino.shape[2]
        return ops.radon_backprojection(sino, R)

    @staticmethod
    def backward(ctx, g):
        return ops.radon_forward(g, ctx.V), None


def radon_forward(img, n_view):
    return _RadonForward.apply(img, n_view)


def radon_backprojection(sino, R):
    return _RadonBackprojection.apply(sino, R)


# ---- complex helpers (pure layout / pointwise, autograd by PyTorch) -- tfpnp/utils/transforms.py:12-17,260-274
def r2c(x):
    return torch.stack([x, torch.zeros_like(x)], dim=-1)


def c2r(x):
    return x[..., 0]


def cmul(a, b):
    return torch.stack((a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1], a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0]), -1)


def conj(x):
    return torch.stack([x[..., 0], -x[..., 1]], -1)
