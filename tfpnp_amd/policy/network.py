"""Policy actors on MI355X -- drop-in for the inference side of tfpnp/policy/network.py (same class names, same
constructor arguments, same `forward(state, idx_stop, train, hidden)` contract and action ranges).

The ResNet-18 encoder and both heads run natively (pnpx_policy_forward: BatchNorm folded, stride-2 convolutions on a
space-to-depth grid with tap masks, fp32 MFMA); what stays here is O(B) scalar work: sampling / arg-max of idx_stop,
log-probability, entropy and the action-range mapping (network.py:149-175).  Eval-mode BatchNorm only -- that is how
the reference runs the actor in rollouts (trainer.py:216-221) and in evaluation (evaluator.py:23); training the actor
itself (batch statistics, weight gradients) is out of scope.
"""
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .. import torch_ops as T


class ResNetActorBase(nn.Module):
    spi_head = False

    def __init__(self, num_inputs, action_bundle, num_actions, state_dict=None):
        super().__init__()
        self.in_dim = num_inputs
        self.num_actions = num_actions
        self.action_range = None
        self.action_bundle = action_bundle
        self._state = None
        self._ctx = {}
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # weights: the reference's own state_dict (torch.load of actor.pkl, trainer.py:254-261)
    def load_state_dict(self, state_dict, strict=True):
        self._state = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in state_dict.items()}
        self._ctx = {}

    def context(self, device):
        device = torch.device(device)
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        if key not in self._ctx:
            if self._state is None:
                raise ValueError('actor weights were not loaded (load_state_dict)')
            ctx = ops.Context(device)
            ctx.load_policy(self._state, self.in_dim, self.action_bundle * self.num_actions, self.spi_head)
            self._ctx[key] = ctx
        return self._ctx[key]

    def forward(self, state, idx_stop, train, hidden):
        """-> (action dict incl. 'idx_stop', log-prob of idx_stop [B,1], entropy of the stop head [B,1], hidden)"""
        p_stop, det = T.call("policy_forward", state, self.context(state.device).cid)   # [B,2] softmax, [B,n_det] sigmoid
        logp = torch.log(p_stop.clamp_min(torch.finfo(p_stop.dtype).eps))     # Categorical's own clamp
        entropy = -torch.special.xlogy(p_stop, p_stop).sum(dim=1, keepdim=True)
        if idx_stop is None:      # stochastic while training, greedy otherwise (network.py:149-156)
            idx_stop = torch.multinomial(p_stop, 1).squeeze(1) if train else p_stop.argmax(dim=1)
        action = self.action_mapping(det)
        action['idx_stop'] = idx_stop
        return action, logp.gather(1, idx_stop.view(-1, 1)), entropy, hidden

    def action_mapping(self, action_deterministic):
        """Sigmoid outputs [B, num_actions * bundle] -> {name: [B, bundle] in the action's range} (network.py:163-175)."""
        per_action = action_deterministic.shape[1] // self.num_actions
        return OrderedDict(
            (name, action_deterministic[:, i * per_action:(i + 1) * per_action] * rng['scale'] + rng['shift'])
            for i, (name, rng) in enumerate(self.action_range.items()))

    def init_state(self, B):      # no recurrent state (the reference returns a dummy as well)
        return torch.zeros(B)


def _actor(name, extra_inputs, num_actions, default_range, spi=False):
    def __init__(self, num_aux_inputs, action_bundle, action_range: Optional[OrderedDict] = None, state_dict=None):
        ResNetActorBase.__init__(self, num_aux_inputs + extra_inputs, action_bundle, num_actions, state_dict)
        self.action_range = OrderedDict(default_range) if action_range is None else action_range
    return type(name, (ResNetActorBase,), {"__init__": __init__, "spi_head": spi, "__doc__":
                                           f"tfpnp/policy/network.py {name}: same inputs / action ranges."})


_R = lambda s, sh=0: {'scale': s, 'shift': sh}  # noqa: E731
ResNetActor_ADMM = _actor("ResNetActor_ADMM", 3, 2, [('sigma_d', _R(70 / 255)), ('mu', _R(1))])
ResNetActor_HQS = _actor("ResNetActor_HQS", 2, 2, [('sigma_d', _R(70 / 255)), ('mu', _R(1))])
ResNetActor_PG = _actor("ResNetActor_PG", 1, 2, [('sigma_d', _R(70 / 255)), ('tau', _R(2))])
ResNetActor_APG = _actor("ResNetActor_APG", 2, 3, [('sigma_d', _R(70 / 255)), ('tau', _R(2)), ('beta', _R(2))])
ResNetActor_RED = _actor("ResNetActor_RED", 3, 3, [('sigma_d', _R(70 / 255)), ('mu', _R(1)), ('lamda', _R(2))])
ResNetActor_IADMM = _actor("ResNetActor_IADMM", 3, 3, [('sigma_d', _R(70 / 255)), ('mu', _R(1)), ('tau', _R(2))])
ResNetActor_AMP = _actor("ResNetActor_AMP", 2, 1, [('sigma_d', _R(2))])
ResNetActor_SPI = _actor("ResNetActor_SPI", 3, 2, [('sigma_d', _R(55 / 255, 15 / 255)), ('mu', _R(70, 50))], spi=True)
