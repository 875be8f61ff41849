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
