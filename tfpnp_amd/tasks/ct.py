"""Sparse-view CT solvers -- drop-in for tasks/ct/solver.py (own Radon pair instead of torch_radon)."""
from .. import ops
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
        n_view = int(view[0, 0, 0, 0].item() * 120)     # host sync, as in the reference (:26)
        radon = self.radon_generator(variables.shape[-1], n_view, device=variables.device)
        return ops.ct_iadmm(self._ctx(variables), variables, y0, n_view, radon.opnorm, sigma_d, mu, tau, iter_num)


class PGSolver_CT(CTMixin, PGSolver):
    """tasks/ct/solver.py:56-87"""

    def __init__(self, denoiser):
        super().__init__(denoiser)
        self.radon_generator = RadonGenerator()

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, view) = inputs
        sigma_d, tau = parameters
        n_view = int(view[0, 0, 0, 0].item() * 120)
        radon = self.radon_generator(variables.shape[-1], n_view, device=variables.device)
        return ops.ct_pg(self._ctx(variables), variables, y0, n_view, radon.opnorm, sigma_d, tau, iter_num)


_solver_map = {'iadmm': IADMMSolver_CT, 'pg': PGSolver_CT}


def create_solver_ct(opt, denoiser):
    """tasks/ct/solver.py:95-103"""
    if opt.solver in _solver_map:
        return _solver_map[opt.solver](denoiser)
    raise NotImplementedError
