"""Sparse-view CT solvers -- drop-in for tasks/ct/solver.py (own Radon pair instead of torch_radon)."""
import torch

from .. import autograd as A
from .. import torch_ops as T
from ..env.base import PnPEnv
from ..pnp.solver.base import IADMMSolver, PGSolver
from ..utils.transforms import RadonGenerator


class CTMixin:
    """tasks/ct/solver.py:7-9"""

    def filter_aux_inputs(self, state):
        return (state['y0'], state['view'])


class IADMMSolver_CT(CTMixin, IADMMSolver):
    """tasks/ct/solver.py:12-53"""

    def __init__(self, denoiser):
        super().__init__(denoiser)
        self.radon_generator = RadonGenerator()

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, view) = inputs
        sigma_d, mu, tau = parameters
        # The reference recovers the view count from the observation map, int(view[0,0,0,0] * 120) (:26): a host sync,
        # and the float32 round trip truncates many counts by one (50 -> 49, 100 -> 99), which torch_radon then rejects
        # with a shape error.  The sinogram itself carries the count, exactly and without a sync.
        n_view = int(y0.shape[2])
        radon = self.radon_generator(variables.shape[-1], n_view, device=variables.device)
        if A.needs_grad(variables, sigma_d, mu, tau):      # training path: native forward + fused native VJP (tasks.hip)
            return T.call("ct_iadmm_train", variables, y0, n_view, float(radon.opnorm), sigma_d, mu, tau,
                          -1 if iter_num is None else iter_num, self._ctx(variables).cid)[0]
        return T.call("ct_iadmm", variables, y0, n_view, float(radon.opnorm), sigma_d, mu, tau, -1 if iter_num is None else iter_num,
                      self._ctx(variables).cid)

class PGSolver_CT(CTMixin, PGSolver):
    """tasks/ct/solver.py:56-87"""

    def __init__(self, denoiser):
        super().__init__(denoiser)
        self.radon_generator = RadonGenerator()

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, view) = inputs
        sigma_d, tau = parameters
        n_view = int(y0.shape[2])                        # see IADMMSolver_CT.forward
        radon = self.radon_generator(variables.shape[-1], n_view, device=variables.device)
        if A.needs_grad(variables, sigma_d, tau):          # training path: native forward + fused native VJP (tasks.hip)
            return T.call("ct_pg_train", variables, y0, n_view, float(radon.opnorm), sigma_d, tau,
                          -1 if iter_num is None else iter_num, self._ctx(variables).cid)[0]
        return T.call("ct_pg", variables, y0, n_view, float(radon.opnorm), sigma_d, tau, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

_solver_map = {'iadmm': IADMMSolver_CT, 'pg': PGSolver_CT}


def create_solver_ct(opt, denoiser):
    """tasks/ct/solver.py:95-103"""
    if opt.solver in _solver_map:
        return _solver_map[opt.solver](denoiser)
    raise NotImplementedError


class CTEnv(PnPEnv):
    """tasks/ct/env.py:6-55.  Observation: variables, ATy0, view, T, sigma_n."""
    ob_base_dim = 4
    ob_keys = ('y0', 'ATy0', 'view', 'sigma_n')
    policy_layout = (('variables', 'raw'), ('ATy0', 'raw'), ('view', 'raw'), ('T', 'raw'), ('sigma_n', 'raw'))
    input_key = 'ATy0'
    aux_keys = ('y0', 'view')
