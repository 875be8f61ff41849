from .base import UNetDenoiser2D  # noqa: F401
