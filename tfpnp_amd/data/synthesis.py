"""Measurement synthesis on the GPU -- SURVEY 8(f) rank 3: the step immediately before the hot path.

The reference builds every training item on the CPU inside Dataset.__getitem__ (one image at a time, DataLoader
workers) and, for CT, on the GPU through torch_radon.  Here a whole resident batch of ground-truth images is turned
into the solver's input dictionary in a few batched launches (native FFT / CDP / Radon, PyTorch device RNG), with the
same keys, shapes and value conventions, so `env.reset(data)` can be fed directly:

    csmri_measure   tasks/csmri/dataset.py:27-75   y0 = mask * (fft2(gt) + sigma * N(0,1)),  x0 = ATy0 = ifft2(y0)
    pr_measure      tasks/pr/dataset.py:25-71      y0 = Poisson-like(|CDP(gt)|),              x0 = ones
    spi_measure     tasks/spi/dataset.py:24-66     y0 = [Poisson(K^2 * kron(gt, 1_KxK) / K^2) >= 1],  x0 = avg_pool(y0, K)
    ct_measure      tasks/ct/dataset.py:31-71      y0 = Radon(gt) (+ percent noise),  ATy0 = A^T y0 / |A|^2,  x0 = FBP(y0)

Noise draws use the device generator: parity of the DRAWS with the reference is statistical (same distributions), not
bitwise.  The noise models mirror tfpnp/utils/noise.py and take an optional torch.Generator; every model (and
spi_forward) also accepts the pre-drawn random field itself (`noise=` standard-normal tensor, `counts=` Poisson counts,
`sigma=` / `alpha=` the per-item level) -- with those injected the result is a deterministic function that
tests/test_gpu_synthesis.py holds to fixtures produced by the reference's own noise.py / Dataset.__getitem__ code.
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..utils import transforms


# ------------------------------------------------------------------------------------------- noise models
def _per_item(values, like):
    """[N] host values -> device tensor broadcastable over `like` ([N,1,...])."""
    t = torch.as_tensor(np.asarray(values, dtype=np.float32), device=like.device)
    return t.view(like.shape[0], *([1] * (like.dim() - 1)))


def _randn(like, generator, noise):
    """Standard-normal field shaped like `like`: drawn on the device, or the caller's pre-drawn one."""
    if noise is not None:
        return noise.to(like.device, like.dtype).reshape(like.shape)
    return torch.randn(like.shape, device=like.device, generator=generator)


class GaussianModelD:
    """tfpnp/utils/noise.py:20-33 -- additive N(0, (sigma/255)^2).  The reference draws sigma inside
    Dataset.__getitem__, i.e. ONE PER ITEM; so does the batched form here (sigma returned as [N])."""

    def __init__(self, sigmas):
        self.sigmas = sigmas

    def __call__(self, x, idx=None, generator=None, noise=None):
        N = x.shape[0]
        sig = (np.full(N, self.sigmas[idx]) if idx is not None else np.random.choice(self.sigmas, size=N)) / 255.
        sigma = _per_item(sig, x)
        return x + _randn(x, generator, noise) * sigma, sigma.reshape(N)


class GaussianModelC:
    """tfpnp/utils/noise.py:5-17 -- sigma ~ U(low, high) / 255, one draw per item."""

    def __init__(self, low_sigma=0, high_sigma=55):
        self.low_sigma, self.high_sigma = low_sigma, high_sigma

    def __call__(self, x, generator=None, noise=None, sigma=None):
        N = x.shape[0]
        sigma = _per_item(np.random.uniform(self.low_sigma, self.high_sigma, size=N) / 255. if sigma is None else sigma, x)
        return x + _randn(x, generator, noise) * sigma, sigma.reshape(N)


class GaussianModelP:
    """tfpnp/utils/noise.py:36-58 (batch mode) -- noise std = percentage * mean|x| per item."""

    def __init__(self, sigmas_p):
        self.sigmas_p = sigmas_p

    def __call__(self, x, generator=None, noise=None, sigma=None):
        N = x.shape[0]
        sigma = np.random.choice(self.sigmas_p, size=N) if sigma is None else np.asarray(sigma)
        sigma = torch.from_numpy(sigma.astype(np.float32)).to(x.device)
        sigma = sigma.view(N, *([1] * (x.dim() - 1)))
        x_mean = x.abs().reshape(N, -1).mean(dim=1).view_as(sigma)
        return (x + _randn(x, generator, noise) * x_mean * sigma).float(), sigma


class PoissonModel:
    """tfpnp/utils/noise.py:61-84 -- intensity noise alpha/255 * |z| * N(0,1) on z^2, per-item alpha and residual std."""

    def __init__(self, alphas):
        self.alphas = alphas

    def __call__(self, z, idx=None, generator=None, noise=None):
        N = z.shape[0]   # one alpha per item, as Dataset.__getitem__ draws it
        alpha = _per_item(np.full(N, self.alphas[idx]) if idx is not None else np.random.choice(self.alphas, size=N), z)
        noise = alpha / 255 * z.abs() * _randn(z, generator, noise)
        y = torch.sqrt(torch.clamp(z ** 2 + noise, min=0))
        sigma = (y - z.abs()).reshape(z.shape[0], -1).std(dim=1)
        return y, sigma


# ------------------------------------------------------------------------------------------- per-task synthesis
def _item_map(value, like):
    """Per-item scalar -> constant map shaped like `like` (the reference's `np.ones_like(x) * sigma_n`)."""
    if not isinstance(value, torch.Tensor):
        value = torch.full((like.shape[0],), float(value), device=like.device)
    return value.to(like.device, torch.float32).reshape(-1, *([1] * (like.dim() - 1))).expand_as(like).contiguous()


def csmri_measure(gt, mask, noise_model=None, generator=None, noise=None):
    """gt [B,1,H,W] in [0,1], mask bool [B,1,H,W] -> dict(y0, x0, ATy0, gt, mask, sigma_n, output, input)."""
    y0 = transforms.fft2(transforms.real2complex(gt))
    sigma_n = 0.0
    if noise_model is not None:
        y0, sigma_n = noise_model(y0, generator=generator, noise=noise)
    y0 = y0 * mask.unsqueeze(-1).to(y0.dtype)
    ATy0 = transforms.ifft2(y0)
    x0 = ATy0.clone()
    return {'y0': y0, 'x0': x0, 'ATy0': ATy0, 'gt': gt, 'mask': mask.bool(), 'sigma_n': _item_map(sigma_n, y0),
            'output': transforms.complex2real(ATy0).clone(), 'input': x0}


def pr_measure(gt, mask, noise_model=None, generator=None, noise=None):
    """gt [B,1,H,W], CDP masks [B,S,H,W,2] (unit modulus) -> dict(y0 [B,S,H,W], x0 = ones, ...)."""
    y0 = transforms.complex_abs(transforms.cdp_forward(transforms.real2complex(gt), mask))
    sigma_n = 0.0
    if noise_model is not None:
        y0, sigma_n = noise_model(y0, generator=generator, noise=noise)
    x0 = torch.ones_like(gt)
    return {'y0': y0, 'x0': x0, 'output': x0.clone(), 'gt': gt, 'mask': mask, 'sigma_n': _item_map(sigma_n, x0)}


def spi_theta(x, K, alpha):
    """Poisson rate of every binary sub-pixel: alpha * kron(x, 1_KxK) / K^2   (transforms.py:395-397)."""
    return alpha * x.repeat_interleave(K, dim=-2).repeat_interleave(K, dim=-1) / (K ** 2)


def spi_forward(x, K, alpha, q, generator=None, counts=None):
    """transforms.py:395-401: K x K binary sub-pixels per pixel, each firing iff Poisson(alpha * x / K^2) >= q.
    counts: pre-drawn Poisson counts of spi_theta(x, K, alpha) (reproducible runs / fixtures)."""
    if counts is None:
        counts = torch.poisson(spi_theta(x, K, alpha), generator=generator)
    return (counts.to(x.device) >= q).float()


def spi_measure(gt, K, generator=None, counts=None):
    """gt [B,1,H,W], K int -> dict(x0 = fraction of fired sub-pixels, output, gt, K map = K/10)."""
    y0 = spi_forward(gt, K, K ** 2, 1, generator=generator, counts=counts)
    x0 = F.avg_pool2d(y0, K)
    return {'x0': x0, 'output': x0.clone(), 'gt': gt, 'K': torch.full_like(gt, K / 10)}


def ct_measure(gt, view, radon_generator=None, noise_model=None, generator=None):
    """gt [B,1,R,R], view = number of projection angles -> dict(y0 sinogram, ATy0, output, x0 = FBP, gt, view map,
    sigma_n map)."""
    radon_generator = radon_generator or transforms.RadonGenerator()
    radon = radon_generator(gt.shape[-1], view, device=gt.device)
    y0 = radon.forward(gt)
    sigma_n = 0.0
    if noise_model is not None:
        y0, sigma_n = noise_model(y0, generator=generator)
    ATy0 = radon.backprojection_norm(y0)
    return {'y0': y0, 'ATy0': ATy0, 'output': ATy0.clone(), 'x0': radon.filter_backprojection(y0), 'gt': gt,
            'view': torch.full_like(gt, view / 120), 'sigma_n': _item_map(sigma_n, gt)}
