"""Phase-retrieval solvers -- drop-in for tasks/pr/solver.py."""
import torch

from .. import autograd as A
from .. import torch_ops as T
from ..env.base import PnPEnv
from ..pnp.solver.base import IADMMSolver, PGSolver
from ..utils.transforms import complex2real, real2complex


class PRMixin:
    """tasks/pr/solver.py:15-21"""

    def get_output(self, state):
        return complex2real(super().get_output(state))

    def filter_aux_inputs(self, state):
        return (state['y0'], state['mask'])


class IADMMSolver_PR(PRMixin, IADMMSolver):
    """tasks/pr/solver.py:24-76"""

    def reset(self, data):
        x = real2complex(data['x0'].clone().detach())
        z = x.clone().detach()
        u = torch.zeros_like(x)
        return torch.cat([x, z, u], dim=1)

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu, tau = parameters
        if A.needs_grad(variables, sigma_d, mu, tau):      # training path: native forward + fused native VJP (tasks.hip)
            return T.call("pr_iadmm_train", variables, y0, mask, sigma_d, mu, tau, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("pr_iadmm", variables, y0, mask, sigma_d, mu, tau, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

class PGSolver_PR(PRMixin, PGSolver):
    """tasks/pr/solver.py:79-112: the reference applies `~mask` to the float CDP mask and the CS-MRI fft2 to PR
    data (:102-103), which raises on any input; it is not runnable there, so no native loop exists."""

    def reset(self, data):
        return real2complex(data['x0'].clone().detach())

    def forward(self, inputs, parameters, iter_num=None):
        raise NotImplementedError('PGSolver_PR.forward is broken in the reference (tasks/pr/solver.py:102-103)')


_solver_map = {'iadmm': IADMMSolver_PR, 'pg': PGSolver_PR}


def create_solver_pr(opt, denoiser):
    """tasks/pr/solver.py:120-128"""
    if opt.solver in _solver_map:
        return _solver_map[opt.solver](denoiser)
    raise NotImplementedError


class PREnv(PnPEnv):
    """tasks/pr/env.py:7-56.  Observation: Re(variables), y0 [S], mask as 2S channels, T, sigma_n  (S = 4 -> 14 + 3)."""
    ob_base_dim = 14
    ob_keys = ('y0', 'x0', 'mask', 'sigma_n')
    float_keys = ('mask',)
    policy_layout = (('variables', 'real'), ('y0', 'raw'), ('mask', 'channel'), ('T', 'raw'), ('sigma_n', 'raw'))
    input_key = 'x0'
    aux_keys = ('y0', 'mask')
