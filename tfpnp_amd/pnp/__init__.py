"""Mirror of tfpnp.pnp (reference tfpnp/pnp/__init__.py:1-13): PnPSolver + denoiser factory."""
from .solver.base import PnPSolver  # noqa: F401
from .denoiser import UNetDenoiser2D  # noqa: F401


def create_denoiser(opt, **kw):
    """Reference: tfpnp/pnp/__init__.py:5-13 -- only 'unet' exists; anything else raises NotImplementedError."""
    if opt.denoiser == 'unet':
        return UNetDenoiser2D(**kw)
    raise NotImplementedError
