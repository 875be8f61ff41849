"""Single-photon-imaging solver -- drop-in for tasks/spi/solver.py."""
import torch

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


    def _forward_autograd(self, variables, x0, K, sigma_d, mu, iter_num):
        """The reference's loop (tasks/spi/solver.py:36-47) from differentiable building blocks: what the fused native VJP
        (pnpx_spi_admm_backward) is tested against."""
        x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
        B = x.shape[0]
        Kv = K[:, 0, 0, 0].reshape(B, 1, 1, 1) * 10
        K1 = x0 * (Kv ** 2)
        for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
            _mu = mu[:, i].reshape(B, 1, 1, 1)
            zt = x + u
            # spi_inverse (transforms.py:404-439): the bisection result carries no gradient in the reference
            # either (bmin/bmax/bave are built by masked assignment of constants); only the K1 == 0 branch does.
            bis = T.call("spi_inverse", zt.detach(), K1, Kv, _mu.detach())
            z = torch.clamp(torch.where(K1 == 0, zt - (Kv ** 2 - K1) / _mu, bis), 0.0, 1.0)
            u = u + x - z
            x = self.prox_mapping(z - u, sigma_d[:, i])
        return torch.cat([x, z, u], dim=1)


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
