"""Mirror of the hot-path part of tfpnp/utils/transforms.py (same function names and argument meaning).

Layout helpers (real2complex ...) are plain tensor views/stacks -- pure data movement, exactly as in the
reference.  Everything that computes (fft2, ifft2, cdp_*, spi_inverse, Radon) is a libpnpx.so call.
"""
import numpy as np
import torch

from .. import torch_ops as T


def real2complex(x):
    """transforms.py:12-13"""
    return torch.stack([x, torch.zeros_like(x)], dim=4)


def complex2real(x):
    """transforms.py:16-17"""
    return x[..., 0]


def complex2channel(x):
    """transforms.py:20-26"""
    N, C, H, W, _ = x.shape
    return x.permute(0, 1, 4, 2, 3).contiguous().view(N, C * 2, H, W)


def fft2(data):
    """Centered orthonormal 2-D FFT over dims (-3,-2) of [...,H,W,2].  transforms.py:68-84"""
    assert data.size(-1) == 2
    return T.call("fft2", data, False, True)


def ifft2(data):
    """transforms.py:87-103"""
    assert data.size(-1) == 2
    return T.call("fft2", data, True, True)


def complex_abs(data):
    """transforms.py:106-118 (layout glue used outside the solver loops)."""
    assert data.size(-1) == 2
    return (data ** 2).sum(dim=-1).sqrt()


def complex_mul(x1, x2):
    """transforms.py:260-270"""
    assert x1.size(-1) == 2 and x2.size(-1) == 2
    return torch.stack((x1[..., 0] * x2[..., 0] - x1[..., 1] * x2[..., 1],
                        x1[..., 0] * x2[..., 1] + x1[..., 1] * x2[..., 0]), -1)


def conjugate(x):
    """transforms.py:273-274"""
    return torch.stack([x[..., 0], -x[..., 1]], -1)


def cdp_forward(data, mask):
    """transforms.py:282-301"""
    assert mask.size(-1) == 2
    return T.call("cdp_forward", data, mask)


def cdp_backward(data, mask):
    """transforms.py:304-320"""
    assert mask.size(-1) == 2
    return T.call("cdp_backward", data, mask)


def spi_inverse(ztilde, K1, K, mu):
    """transforms.py:404-439"""
    return T.call("spi_inverse", ztilde, K1, K, mu)


# ------------------------------------------------------------------------------------------------ CT
class Radon_norm:
    """Stands in for transforms.py:465-484 (a torch_radon.Radon subclass): parallel beam,
    angles = linspace(0, 179/180*pi, view), det_count = ceil(sqrt(2)*resolution) (transforms.py:487-491).
    torch_radon is CUDA-only and absent; the discretisation is this project's own."""

    def __init__(self, resolution, view, opnorm=None, device=None, seed=0):
        self.resolution = resolution
        self.view = view
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if opnorm is None:
            # power_method_opnorm (transforms.py:447-462); seeded start vector instead of an unseeded randn
            x = torch.from_numpy(np.random.RandomState(seed).standard_normal((1, 1, resolution, resolution))
                                 .astype(np.float32)).to(self.device)
            x = x / x.norm()
            v = 1.0
            for _ in range(10):
                nx = self.backprojection(self.forward(x))
                v = float(nx.norm())
                x = nx / v
            opnorm = v ** 0.5
        self.opnorm = opnorm

    def forward(self, x):
        return T.call("radon_forward", x, int(self.view))

    def backprojection(self, sinogram):
        return T.call("radon_backprojection", sinogram, int(self.resolution))

    backward = backprojection

    def backprojection_norm(self, sinogram):
        return self.backprojection(sinogram) / self.opnorm ** 2

    def normal_operator(self, x):
        return self.backprojection_norm(self.forward(x))

    def filter_sinogram(self, sinogram, filter_name='ramp'):
        """Ramp-filter every projection along the detector axis (torch_radon.Radon.filter_sinogram as used by
        transforms.py:479-481): zero-pad to max(64, next power of two >= 2*det), multiply the spectrum by the
        Kak-Slaney ramp (the DFT of the band-limited spatial kernel, not |f|), crop, scale by pi / (2 * n_views).
        The 1-D FFTs are native (H = 1 images)."""
        if filter_name != 'ramp':
            raise NotImplementedError(filter_name)
        B, C, V, D = sinogram.shape
        L = max(64, 1 << int(np.ceil(np.log2(2 * D))))
        x = torch.zeros(B * C * V, 1, L, 2, device=sinogram.device, dtype=torch.float32)
        x[:, 0, :D, 0] = sinogram.reshape(-1, D)
        spec = T.call("fft2", x, False, False)
        filt = torch.from_numpy(ramp_filter(L)).to(sinogram.device)
        out = T.call("fft2", spec * filt.view(1, 1, L, 1), True, False)
        return (out[:, 0, :D, 0] * (np.pi / (2 * V))).reshape(B, C, V, D).contiguous()

    def filter_backprojection(self, sinogram):
        """transforms.py:479-481"""
        return self.backprojection(self.filter_sinogram(sinogram, filter_name='ramp'))


def ramp_filter(size):
    """Frequency response of the discrete ramp filter (Kak & Slaney eq. 61; the construction skimage / torch_radon
    use): spatial kernel f[0] = 1/4, f[odd n] = -1/(pi n)^2, response = 2 Re FFT(f).  float32 [size]."""
    n = np.concatenate((np.arange(1, size // 2 + 1, 2), np.arange(size // 2 - 1, 0, -2)))
    f = np.zeros(size)
    f[0] = 0.25
    f[1::2] = -1.0 / (np.pi * n) ** 2
    return (2 * np.real(np.fft.fft(f))).astype(np.float32)


class RadonGenerator:
    """transforms.py:494-508: caches the operator norm per (resolution, view)."""

    def __init__(self):
        self.opnorms = {}

    def __call__(self, resolution, view, device=None):
        key = (resolution, view)
        radon = Radon_norm(resolution, view, opnorm=self.opnorms.get(key), device=device)
        self.opnorms[key] = radon.opnorm
        return radon
