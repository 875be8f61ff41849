"""Single-photon-imaging solver -- drop-in for tasks/spi/solver.py."""
from .. import autograd as A
from .. import torch_ops as T
from ..env.base import PnPEnv
from ..pnp.solver.base import ADMMSolver


class SPIMixin:
    """tasks/spi/solver.py:8-10"""

    def filter_aux_inputs(self, state):
        return (state['x0'], state['K'])


class ADMMSolver_SPI(SPIMixin, ADMMSolver):
    """tasks/spi/solver.py:13-52"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (x0, K) = inputs
        sigma_d, mu = parameters
        if A.needs_grad(variables, sigma_d, mu):      # training path: native forward + fused native VJP (tasks.hip)
            return T.call("spi_admm_train", variables, x0, K, sigma_d, mu, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("spi_admm", variables, x0, K, sigma_d, mu, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

_solver_map = {'admm_spi': ADMMSolver_SPI}


def create_solver_spi(opt, denoiser):
    """tasks/spi/solver.py:58-66"""
    if opt.solver in _solver_map:
        return _solver_map[opt.solver](denoiser)
    raise NotImplementedError


class SPIEnv(PnPEnv):
    """tasks/spi/env.py:7-52.  Observation: variables, x0, K, T."""
    ob_base_dim = 3
    ob_keys = ('x0', 'K')
    policy_layout = (('variables', 'raw'), ('x0', 'raw'), ('K', 'raw'), ('T', 'raw'))
    input_key = 'x0'
    aux_keys = ('x0', 'K')
