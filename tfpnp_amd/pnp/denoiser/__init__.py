"""Denoiser priors: UNetDenoiser2D (tfpnp/pnp/denoiser/__init__.py) and the DRUNet of BASELINE config #5."""
from .base import UNetDenoiser2D, DRUNetDenoiser2D  # noqa: F401
