"""ORACLE TOOLING -- build-container only (needs /root/reference; never runs on the GPU box).

Imports the real reference (Vandermode/TFPnP, read-only at /root/reference) on a modern
PyTorch by emulating the two legacy functions it calls (tfpnp/utils/transforms.py:4-5):
    torch.fft(x[...,2], signal_ndim, normalized)   and   torch.ifft(...)
Nothing from the reference is copied: modules are imported in place.
"""
import importlib.util
import os
import sys
import types

import torch

REF = os.environ.get("TFPNP_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "tfpnp"))


def _legacy(fn_n):
    def f(x, signal_ndim, normalized=False):
        dims = tuple(range(-signal_ndim, 0))
        c = torch.view_as_complex(x.contiguous())
        return torch.view_as_real(fn_n(c, dim=dims, norm="ortho" if normalized else "backward"))
    return f


class _CallableFFT(types.ModuleType):
    """Stands in for the torch.fft *module* while also being callable like the legacy function."""

    def __init__(self, real):
        super().__init__("torch.fft")
        self.__dict__.update(real.__dict__)
        self._legacy = _legacy(real.fftn)

    def __call__(self, *a, **k):
        return self._legacy(*a, **k)


_installed = False


def install():
    global _installed
    if _installed:
        return
    real = torch.fft
    torch.ifft = _legacy(real.ifftn)
    torch.fft = _CallableFFT(real)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _installed = True


def load_task_module(task, name):
    """tasks/<task>/<name>.py are scripts, not packages: load them by path."""
    install()
    path = os.path.join(REF, "tasks", task, f"{name}.py")
    spec = importlib.util.spec_from_file_location(f"ref_tasks_{task}_{name}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_denoiser(params, tmpdir):
    """Reference UNetDenoiser2D with the synthetic weights loaded through load_state_dict."""
    install()
    from tfpnp.pnp.denoiser import UNetDenoiser2D
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    path = os.path.join(tmpdir, "unet-synth.pt")
    torch.save(sd, path)
    return UNetDenoiser2D(ckpt_path=path)


def make_drunet(params, in_nc=2, out_nc=1, nc=(64, 128, 256, 512), nb=4):
    """DRUNet assembled from the REFERENCE'S OWN building blocks (tfpnp/pnp/denoiser/models/basicblock.py: conv :61-101,
    ResBlock :211-227, upsample_convtranspose :413-419, downsample_strideconv :437-446, sequential :15-36), following the
    published topology of KAIR's UNetRes (the reference ships the blocks but no model that uses them).  The synthetic
    weights go in through load_state_dict(strict=True) under KAIR's key names."""
    install()
    import torch.nn as nn
    from tfpnp.pnp.denoiser.models import basicblock as B

    class UNetRes(nn.Module):
        def __init__(self):
            super().__init__()
            rb = lambda c: B.ResBlock(c, c, bias=False, mode="CRC")
            self.m_head = B.conv(in_nc, nc[0], bias=False, mode="C")
            self.m_down1 = B.sequential(*[rb(nc[0]) for _ in range(nb)], B.downsample_strideconv(nc[0], nc[1], bias=False, mode="2"))
            self.m_down2 = B.sequential(*[rb(nc[1]) for _ in range(nb)], B.downsample_strideconv(nc[1], nc[2], bias=False, mode="2"))
            self.m_down3 = B.sequential(*[rb(nc[2]) for _ in range(nb)], B.downsample_strideconv(nc[2], nc[3], bias=False, mode="2"))
            self.m_body = B.sequential(*[rb(nc[3]) for _ in range(nb)])
            self.m_up3 = B.sequential(B.upsample_convtranspose(nc[3], nc[2], bias=False, mode="2"), *[rb(nc[2]) for _ in range(nb)])
            self.m_up2 = B.sequential(B.upsample_convtranspose(nc[2], nc[1], bias=False, mode="2"), *[rb(nc[1]) for _ in range(nb)])
            self.m_up1 = B.sequential(B.upsample_convtranspose(nc[1], nc[0], bias=False, mode="2"), *[rb(nc[0]) for _ in range(nb)])
            self.m_tail = B.conv(nc[0], out_nc, bias=False, mode="C")

        def forward(self, x0):
            x1 = self.m_head(x0)
            x2 = self.m_down1(x1)
            x3 = self.m_down2(x2)
            x4 = self.m_down3(x3)
            x = self.m_body(x4)
            x = self.m_up3(x + x4)
            x = self.m_up2(x + x3)
            x = self.m_up1(x + x2)
            return self.m_tail(x + x1)

    net = UNetRes()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    net.eval()
    for prm in net.parameters():
        prm.requires_grad_(False)
    return net
