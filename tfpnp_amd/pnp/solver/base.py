"""Solver state containers -- mirror of tfpnp/pnp/solver/base.py:5-232 (same class names, same
reset / get_output / num_var / filter_hyperparameter / prox_mapping surface).

State packing is pure tensor bookkeeping (cat / split / clone on whatever device the data lives on); the
iteration loops themselves are native and live in tfpnp_amd/tasks/*.
"""
import torch
import torch.nn as nn


class PnPSolver(nn.Module):
    """tfpnp/pnp/solver/base.py:5-84"""

    def __init__(self, denoiser):
        super().__init__()
        self.denoiser = denoiser

    def reset(self, data):
        raise NotImplementedError

    def forward(self, inputs, parameters, iter_num):
        raise NotImplementedError

    def get_output(self, state):
        raise NotImplementedError

    def prox_mapping(self, x, sigma):
        return self.denoiser(x, sigma)

    @property
    def num_var(self):
        raise NotImplementedError

    def filter_aux_inputs(self, state):
        raise NotImplementedError

    def filter_hyperparameter(self, action):
        raise NotImplementedError

    # native context of the denoiser for the device the state lives on
    def _ctx(self, t):
        return self.denoiser.context(t.device)


def _first(state, n):
    return torch.split(state, state.shape[1] // n, dim=1)[0]


class ADMMSolver(PnPSolver):
    """tfpnp/pnp/solver/base.py:87-107"""

    @property
    def num_var(self):
        return 3

    def reset(self, data):
        x = data['x0'].clone().detach()
        z = x.clone().detach()
        u = torch.zeros_like(x)
        return torch.cat((x, z, u), dim=1)

    def get_output(self, state):
        return _first(state, 3)

    def filter_hyperparameter(self, action):
        return action['sigma_d'], action['mu']


class IADMMSolver(ADMMSolver):
    """tfpnp/pnp/solver/base.py:110-116"""

    def filter_hyperparameter(self, action):
        return action['sigma_d'], action['mu'], action['tau']


class HQSSolver(PnPSolver):
    """tfpnp/pnp/solver/base.py:118-138"""

    @property
    def num_var(self):
        return 2

    def reset(self, data):
        x = data['x0'].clone().detach()
        z = x.clone().detach()
        return torch.cat([x, z], dim=1)

    def get_output(self, state):
        return _first(state, 2)

    def filter_hyperparameter(self, action):
        return action['sigma_d'], action['mu']


class PGSolver(PnPSolver):
    """tfpnp/pnp/solver/base.py:140-158"""

    @property
    def num_var(self):
        return 1

    def reset(self, data):
        return data['x0'].clone().detach()

    def get_output(self, state):
        return state

    def filter_hyperparameter(self, action):
        return action['sigma_d'], action['tau']


class APGSolver(PnPSolver):
    """tfpnp/pnp/solver/base.py:160-186"""

    def __init__(self, denoiser):
        super().__init__(denoiser)
        import numpy as np
        self.qs = np.zeros(30)
        q = 1
        for i in range(30):
            self.qs[i] = q
            q_prev = q
            q = (1 + (1 + 4 * q_prev ** 2) ** 0.5) / 2

    @property
    def num_var(self):
        return 2

    def reset(self, data):
        x = data['x0'].clone().detach()
        s = x.clone().detach()
        return torch.cat([x, s], dim=1)

    def get_output(self, state):
        return _first(state, 2)

    def filter_hyperparameter(self, action):
        return action['sigma_d'], action['tau'], action['beta']


class REDADMMSolver(PnPSolver):
    """tfpnp/pnp/solver/base.py:189-209"""

    @property
    def num_var(self):
        return 3

    def reset(self, data):
        x = data['x0'].clone().detach()
        z = x.clone().detach()
        u = torch.zeros_like(x)
        return torch.cat([x, z, u], dim=1)

    def get_output(self, state):
        return _first(state, 3)

    def filter_hyperparameter(self, action):
        return action['sigma_d'], action['mu'], action['lamda']


class AMPSolver(PnPSolver):
    """tfpnp/pnp/solver/base.py:212-232.  State packing only: the reference's AMPSolver_CSMRI.forward calls an
    undefined self.prox_fun (tasks/csmri/solver.py:238) and cannot run, so no AMP loop exists here either."""

    @property
    def num_var(self):
        return 2

    def reset(self, data):
        z = data['y0'].clone().detach()
        x = torch.zeros_like(data['x0'])
        return torch.cat([x, z], dim=1)

    def get_output(self, state):
        return _first(state, 2)

    def filter_hyperparameter(self, action):
        return action['sigma_d']
