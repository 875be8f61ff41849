"""CS-MRI solvers on MI355X -- drop-in for tasks/csmri/solver.py (same class names / call surface).

solver(inputs, parameters, iter_num=None) with inputs = (variables, (y0, mask)), parameters the tuple from
filter_hyperparameter(action); returns the next state tensor.  Each forward is ONE native call that runs all
iter_num inner iterations (denoiser prox + masked-FFT data prox + dual update) on the caller's stream.
"""
import torch

from .. import autograd as A
from .. import torch_ops as T
from ..env.base import PnPEnv
from ..pnp.solver.base import ADMMSolver, HQSSolver, PGSolver, APGSolver, REDADMMSolver, AMPSolver
from ..utils import transforms


def _v5(t, B):
    return t.reshape(B, 1, 1, 1, 1)


def _blend(k, y0, m, mu):
    """k[mask] = ((mu*k) + y0)[mask] / (1 + mu)   (tasks/csmri/solver.py:49-51) without in-place writes."""
    return torch.where(m, (mu * k + y0) / (1 + mu), k)


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

    def _forward_autograd(self, variables, y0, mask, sigma_d, mu, iter_num):
        """The reference's loop (tasks/csmri/solver.py:43-55) from differentiable building blocks: the composition the
        fused native VJP (pnpx_csmri_admm_backward) is tested against; the other solvers' training paths look like this."""
        x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
        B = x.shape[0]
        m = (mask != 0).unsqueeze(-1)
        for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
            x = A.r2c(self.prox_mapping(A.c2r(z - u), sigma_d[:, i]))
            z = A.fft2(_blend(A.fft2(x + u), y0, m, _v5(mu[:, i], B)), inverse=True)
            u = u + x - z
        return torch.cat((x, z, u), dim=1)


class HQSSolver_CSMRI(CSMRIMixin, HQSSolver):
    """tasks/csmri/solver.py:60-89"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu = parameters
        if A.needs_grad(variables, sigma_d, mu):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_hqs_train", variables, y0, mask, sigma_d, mu, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("csmri_hqs", variables, y0, mask, sigma_d, mu, -1 if iter_num is None else iter_num, self._ctx(variables).cid)


    def _forward_autograd(self, variables, y0, mask, sigma_d, mu, iter_num):
        """The reference's loop (tasks/csmri/solver.py:76-85) from differentiable building blocks: what the fused native VJP
        (pnpx_csmri_hqs_backward) is tested against."""
        x, z = torch.split(variables, variables.shape[1] // 2, dim=1)
        B, m = x.shape[0], (mask != 0).unsqueeze(-1)
        for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
            x = A.r2c(self.prox_mapping(A.c2r(z), sigma_d[:, i]))
            z = A.fft2(_blend(A.fft2(x), y0, m, _v5(mu[:, i], B)), inverse=True)
        return torch.cat([x, z], dim=1)


class PGSolver_CSMRI(CSMRIMixin, PGSolver):
    """tasks/csmri/solver.py:92-120"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, tau = parameters
        if A.needs_grad(variables, sigma_d, tau):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_pg_train", variables, y0, mask, sigma_d, tau, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("csmri_pg", variables, y0, mask, sigma_d, tau, -1 if iter_num is None else iter_num, self._ctx(variables).cid)


    def _forward_autograd(self, variables, y0, mask, sigma_d, tau, iter_num):
        """The reference's loop (tasks/csmri/solver.py:107-116) from differentiable building blocks: what the fused native
        VJP (pnpx_csmri_pg_backward) is tested against."""
        x, B, m = variables, variables.shape[0], (mask != 0).unsqueeze(-1)
        for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
            temp = torch.where(m, A.fft2(x) - y0, torch.zeros_like(y0))
            z = x - _v5(tau[:, i], B) * A.fft2(temp, inverse=True)
            x = A.r2c(self.prox_mapping(A.c2r(z), sigma_d[:, i]))
        return x


class APGSolver_CSMRI(CSMRIMixin, APGSolver):
    """tasks/csmri/solver.py:123-165"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, tau, beta = parameters
        if A.needs_grad(variables, sigma_d, tau, beta):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_apg_train", variables, y0, mask, sigma_d, tau, beta, -1 if iter_num is None else iter_num,
                          self._ctx(variables).cid)[0]
        return T.call("csmri_apg", variables, y0, mask, sigma_d, tau, beta, -1 if iter_num is None else iter_num, self._ctx(variables).cid)


    def _forward_autograd(self, variables, y0, mask, sigma_d, tau, beta, iter_num):
        """The reference's loop (tasks/csmri/solver.py:141-159) from differentiable building blocks: what the fused native VJP
        (pnpx_csmri_apg_backward) is tested against."""
        x, s = torch.split(variables, variables.shape[1] // 2, dim=1)
        B, m = x.shape[0], (mask != 0).unsqueeze(-1)
        for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
            temp = torch.where(m, A.fft2(s) - y0, torch.zeros_like(y0))
            z = s - _v5(tau[:, i], B) * A.fft2(temp, inverse=True)
            x_prev = x
            x = A.r2c(self.prox_mapping(A.c2r(z), sigma_d[:, i]))
            s = x + _v5(beta[:, i], B) * (x - x_prev)
        return torch.cat([x, s], dim=1)


class REDADMMSolver_CSMRI(CSMRIMixin, REDADMMSolver):
    """tasks/csmri/solver.py:168-204"""

    def forward(self, inputs, parameters, iter_num=None):
        variables, (y0, mask) = inputs
        sigma_d, mu, lamda = parameters
        if A.needs_grad(variables, sigma_d, mu, lamda):      # training path: native forward + fused native VJP (csmri.hip)
            return T.call("csmri_redadmm_train", variables, y0, mask, sigma_d, mu, lamda,
                          -1 if iter_num is None else iter_num, self._ctx(variables).cid)[0]
        return T.call("csmri_redadmm", variables, y0, mask, sigma_d, mu, lamda, -1 if iter_num is None else iter_num, self._ctx(variables).cid)


    def _forward_autograd(self, variables, y0, mask, sigma_d, mu, lamda, iter_num):
        """The reference's loop (tasks/csmri/solver.py:183-200) from differentiable building blocks: what the fused native VJP
        (pnpx_csmri_redadmm_backward) is tested against."""
        x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
        B, m = x.shape[0], (mask != 0).unsqueeze(-1)
        for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
            _mu, _la = _v5(mu[:, i], B), _v5(lamda[:, i], B)
            x_half = A.r2c(self.prox_mapping(A.c2r(x), sigma_d[:, i]))
            x = (_la * x_half + _mu * (z - u)) / (_mu + _la)
            z = A.fft2(_blend(A.fft2(x + u), y0, m, _mu), inverse=True)
            u = u + x - z
        return torch.cat([x, z, u], dim=1)


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
