"""Single-photon-imaging solver -- drop-in for tasks/spi/solver.py."""
from .. import ops
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
        return ops.spi_admm(self._ctx(variables), variables, x0, K, sigma_d, mu, iter_num)


_solver_map = {'admm_spi': ADMMSolver_SPI}


def create_solver_spi(opt, denoiser):
    """tasks/spi/solver.py:58-66"""
    if opt.solver in _solver_map:
        return _solver_map[opt.solver](denoiser)
    raise NotImplementedError
