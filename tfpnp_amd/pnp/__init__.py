"""Mirror of tfpnp.pnp (reference tfpnp/pnp/__init__.py:1-13): PnPSolver + denoiser factory."""
from .solver.base import PnPSolver  # noqa: F401
from .denoiser import UNetDenoiser2D, DRUNetDenoiser2D  # noqa: F401


def create_denoiser(opt, **kw):
    """Reference: tfpnp/pnp/__init__.py:5-13 -- 'unet' is the only name the reference knows; 'drunet' (BASELINE config #5)
    is this package's addition; anything else raises NotImplementedError as in the reference."""
    if opt.denoiser == 'unet':
        return UNetDenoiser2D(**kw)
    if opt.denoiser == 'drunet':
        return DRUNetDenoiser2D(**kw)
    raise NotImplementedError
