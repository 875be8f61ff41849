"""CS-MRI solvers on MI355X -- drop-in for tasks/csmri/solver.py (same class names / call surface).

solver(inputs, parameters, iter_num=None) with inputs = (variables, (y0, mask)), parameters the tuple from
filter_hyperparameter(action); returns the next state tensor.  Each forward is ONE native call that runs all
iter_num inner iterations (denoiser prox + masked-FFT data prox + dual update) on the caller's stream.
"""
from .. import autograd as A
from .. import torch_ops as T
from ..env.base import PnPEnv
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
        if A.needs_grad(variables, sigma_d, mu):      # training path: native forward + native VJP (csmri.hip)
            return T.call("csmri_admm_train", variables, y0, mask, sigma_d, mu, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("csmri_admm", variables, y0, mask, sigma_d, mu, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

class HQSSolver_CSMRI(CSMRIMixin, HQSSolver):
    """tasks/csmri/solver.py:60-89"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu = parameters
        if A.needs_grad(variables, sigma_d, mu):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_hqs_train", variables, y0, mask, sigma_d, mu, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("csmri_hqs", variables, y0, mask, sigma_d, mu, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

class PGSolver_CSMRI(CSMRIMixin, PGSolver):
    """tasks/csmri/solver.py:92-120"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, tau = parameters
        if A.needs_grad(variables, sigma_d, tau):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_pg_train", variables, y0, mask, sigma_d, tau, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("csmri_pg", variables, y0, mask, sigma_d, tau, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

class APGSolver_CSMRI(CSMRIMixin, APGSolver):
    """tasks/csmri/solver.py:123-165"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, tau, beta = parameters
        if A.needs_grad(variables, sigma_d, tau, beta):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_apg_train", variables, y0, mask, sigma_d, tau, beta, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("csmri_apg", variables, y0, mask, sigma_d, tau, beta, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

class REDADMMSolver_CSMRI(CSMRIMixin, REDADMMSolver):
    """tasks/csmri/solver.py:168-204"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu, lamda = parameters
        if A.needs_grad(variables, sigma_d, mu, lamda):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_redadmm_train", variables, y0, mask, sigma_d, mu, lamda,
                          -1 if iter_num is None else iter_num, self._ctx(variables).cid)[0]
        return T.call("csmri_redadmm", variables, y0, mask, sigma_d, mu, lamda, -1 if iter_num is None else iter_num, self._ctx(variables).cid)

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


class CSMRIEnv(PnPEnv):
    """tasks/csmri/env.py:7-56.  Policy observation (9 channels for 3-variable solvers): Re(variables), y0 as 2
    channels, Re(ATy0), mask, T, Re(sigma_n)."""
    ob_base_dim = 6
    ob_keys = ('y0', 'ATy0', 'mask', 'sigma_n')
    float_keys = ('mask',)
    policy_layout = (('variables', 'real'), ('y0', 'channel'), ('ATy0', 'real'), ('mask', 'raw'), ('T', 'raw'),
                     ('sigma_n', 'real'))
    input_key = 'ATy0'
    aux_keys = ('y0', 'mask')
    aux_bool = ('mask',)
