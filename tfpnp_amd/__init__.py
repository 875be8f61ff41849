"""tfpnp_amd -- MI355X-native PnP proximal-solver inner loop behind TFPnP's solver/denoiser surface.

    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    solver = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=sd))
    state = solver((variables, (y0, mask)), (sigma_d, mu))          # one native call, all inner iterations

The compute path is libpnpx.so (hand-written gfx950 HIP kernels, C ABI in include/pnpx.h).  There is no CPU or
PyTorch fallback: importing this package works anywhere, calling an op without the library or a GPU raises.
"""
__version__ = "0.1.0"
