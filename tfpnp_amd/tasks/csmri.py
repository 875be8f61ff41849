"""CS-MRI solvers on MI355X -- drop-in for tasks/csmri/solver.py (same class names / call surface).

solver(inputs, parameters, iter_num=None) with inputs = (variables, (y0, mask)), parameters the tuple from
filter_hyperparameter(action); returns the next state tensor.  Each forward is ONE native call that runs all
iter_num inner iterations (denoiser prox + masked-FFT data prox + dual update) on the caller's stream.
"""
from .. import ops
from ..pnp.solver.base import ADMMSolver, HQSSolver, PGSolver, APGSolver, REDADMMSolver, AMPSolver
from ..utils import transforms


class CSMRIMixin:
    """tasks/csmri/solver.py:15-21"""

    def get_output(self, state):
        return transforms.complex2real(super().get_output(state))

    def filter_aux_inputs(self, state):
        return (state['y0'], state['mask'])


class ADMMSolver_CSMRI(CSMRIMixin, ADMMSolver):
    """tasks/csmri/solver.py:24-57"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu = parameters
        return ops.csmri_admm(self._ctx(variables), variables, y0, mask, sigma_d, mu, iter_num)


class HQSSolver_CSMRI(CSMRIMixin, HQSSolver):
    """tasks/csmri/solver.py:60-89"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu = parameters
        return ops.csmri_hqs(self._ctx(variables), variables, y0, mask, sigma_d, mu, iter_num)


class PGSolver_CSMRI(CSMRIMixin, PGSolver):
    """tasks/csmri/solver.py:92-120"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, tau = parameters
        return ops.csmri_pg(self._ctx(variables), variables, y0, mask, sigma_d, tau, iter_num)


class APGSolver_CSMRI(CSMRIMixin, APGSolver):
    """tasks/csmri/solver.py:123-165"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, tau, beta = parameters
        return ops.csmri_apg(self._ctx(variables), variables, y0, mask, sigma_d, tau, beta, iter_num)


class REDADMMSolver_CSMRI(CSMRIMixin, REDADMMSolver):
    """tasks/csmri/solver.py:168-204"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu, lamda = parameters
        return ops.csmri_redadmm(self._ctx(variables), variables, y0, mask, sigma_d, mu, lamda, iter_num)


class AMPSolver_CSMRI(CSMRIMixin, AMPSolver):
    """tasks/csmri/solver.py:207-250: broken in the reference (undefined self.prox_fun at :238)."""

    def forward(self, inputs, parameters, iter_num=None):
        raise NotImplementedError('AMPSolver_CSMRI.forward cannot run in the reference either (undefined prox_fun)')


_solver_map = {
    'admm': ADMMSolver_CSMRI,
    'hqs': HQSSolver_CSMRI,
    'pg': PGSolver_CSMRI,
    'apg': APGSolver_CSMRI,
    'redadmm': REDADMMSolver_CSMRI,
    'amp': AMPSolver_CSMRI,
}


def create_solver_csmri(opt, denoiser):
    """tasks/csmri/solver.py:262-270"""
    if opt.solver in _solver_map:
        return _solver_map[opt.solver](denoiser)
    raise NotImplementedError
