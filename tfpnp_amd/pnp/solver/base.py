"""Solver state containers -- the surface of tfpnp/pnp/solver/base.py:5-232 (same class names, same
reset / get_output / num_var / filter_hyperparameter / prox_mapping contract).

The reference spells the packing out once per solver family; here a family is described by data -- how many state
variables it carries, how each is initialised from the input dictionary, which policy outputs drive it -- and one
implementation does the tensor bookkeeping (cat / split / clone on whatever device the data lives on).  The iteration
loops themselves are native and live in tfpnp_amd/tasks/*.
"""
import numpy as np
import torch
import torch.nn as nn


class PnPSolver(nn.Module):
    """tfpnp/pnp/solver/base.py:5-84.  Sub-classes set `init_vars` and `hyper_keys`, or override the methods."""

    # state variables in packing order: 'x0' = copy of data['x0'], 'zero' = zeros like x0, 'y0' = copy of data['y0']
    init_vars = None
    # policy outputs consumed by forward(), in order
    hyper_keys = None

    def __init__(self, denoiser):
        super().__init__()
        self.denoiser = denoiser

    # ---- contract used by PnPEnv ---------------------------------------------------------------------------
    @property
    def num_var(self):
        if self.init_vars is None:
            raise NotImplementedError
        return len(self.init_vars)

    def reset(self, data):
        if self.init_vars is None:
            raise NotImplementedError
        parts = []
        for kind in self.init_vars:
            if kind == 'zero':
                parts.append(torch.zeros_like(data['x0']))
            else:
                parts.append(data[kind].clone().detach())
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

    def get_output(self, state):
        if self.init_vars is None:
            raise NotImplementedError
        n = len(self.init_vars)
        return state if n == 1 else state[:, :state.shape[1] // n]

    def filter_hyperparameter(self, action):
        if self.hyper_keys is None:
            raise NotImplementedError
        picked = tuple(action[k] for k in self.hyper_keys)
        return picked[0] if len(picked) == 1 else picked

    def filter_aux_inputs(self, state):
        raise NotImplementedError

    def forward(self, inputs, parameters, iter_num):
        raise NotImplementedError

    def prox_mapping(self, x, sigma):
        return self.denoiser(x, sigma)

    # native context of the denoiser for the device the state lives on
    def _ctx(self, t):
        """The fused native loops (tfpnp_amd/tasks/*) run the denoiser prox INSIDE one libpnpx call, so they need a native
        denoiser and cannot honour a Python override of prox_mapping: both cases fail here with a clear message instead of
        an opaque native error (or a silently ignored override)."""
        if type(self).prox_mapping is not PnPSolver.prox_mapping:
            raise NotImplementedError(
                f'{type(self).__name__} overrides prox_mapping(), but its forward() is one fused native call that applies the '
                'native denoiser itself; compose the iteration from tfpnp_amd.autograd building blocks instead '
                '(tests/composed_solvers.py shows every loop written that way)')
        ctx_of = getattr(self.denoiser, 'context', None)
        if ctx_of is None:
            raise NotImplementedError(
                f'{type(self).__name__} needs a native denoiser (tfpnp_amd.pnp.UNetDenoiser2D / DRUNetDenoiser2D); got '
                f'{type(self.denoiser).__name__}, which has no native context')
        return ctx_of(t.device)


class ADMMSolver(PnPSolver):
    """base.py:87-107 -- (x, z, u) = (x0, x0, 0); hyper-parameters (sigma_d, mu)"""
    init_vars = ('x0', 'x0', 'zero')
    hyper_keys = ('sigma_d', 'mu')


class IADMMSolver(ADMMSolver):
    """base.py:110-116 -- inexact ADMM adds the step size tau"""
    hyper_keys = ('sigma_d', 'mu', 'tau')


class HQSSolver(PnPSolver):
    """base.py:118-138 -- (x, z) = (x0, x0)"""
    init_vars = ('x0', 'x0')
    hyper_keys = ('sigma_d', 'mu')


class PGSolver(PnPSolver):
    """base.py:140-158 -- x = x0"""
    init_vars = ('x0',)
    hyper_keys = ('sigma_d', 'tau')


class APGSolver(PnPSolver):
    """base.py:160-186 -- (x, s) = (x0, x0); `qs` is the FISTA momentum sequence q_{k+1} = (1 + sqrt(1 + 4 q_k^2)) / 2
    the reference tabulates (unused by its forward, kept for attribute parity)."""
    init_vars = ('x0', 'x0')
    hyper_keys = ('sigma_d', 'tau', 'beta')

    def __init__(self, denoiser):
        super().__init__(denoiser)
        qs = [1.0]
        while len(qs) < 30:
            qs.append((1 + (1 + 4 * qs[-1] ** 2) ** 0.5) / 2)
        self.qs = np.asarray(qs)


class REDADMMSolver(PnPSolver):
    """base.py:189-209 -- (x, z, u) = (x0, x0, 0); hyper-parameters (sigma_d, mu, lamda)"""
    init_vars = ('x0', 'x0', 'zero')
    hyper_keys = ('sigma_d', 'mu', 'lamda')


class AMPSolver(PnPSolver):
    """base.py:212-232 -- (x, z) = (0, y0).  State packing only: the reference's AMPSolver_CSMRI.forward calls an
    undefined self.prox_fun (tasks/csmri/solver.py:238) and cannot run, so no AMP loop exists here either."""
    init_vars = ('zero', 'y0')
    hyper_keys = ('sigma_d',)
