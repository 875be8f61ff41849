"""The caller contract of the hot path: a minimal PnPEnv (tfpnp/env/base.py:121-191, 237-242).

Only what SURVEY.md section 8(b) lists is reproduced: live-row gather (`idx_left`), the solver call, state /
output write-back, delta-PSNR reward and the idx_left shrink, plus the differentiable `forward` used by the
actor/critic update (base.py:193-206).  Observation packing for the RL policy, the policy itself and the trainer are
out of scope.  `metric_fn` runs natively (pnpx_psnr; analytic VJP under autograd).
"""
import torch

from .. import ops


class _Psnr(torch.autograd.Function):
    """Native PSNR forward; analytic VJP: d psnr_b / d out = -(20 / ln 10) * (clamp(out) - gt) / (N * mse_b) inside (0,1)."""

    @staticmethod
    def forward(ctx, output, gt):
        ctx.save_for_backward(output, gt)
        return ops.psnr(output, gt)

    @staticmethod
    def backward(ctx, g):
        output, gt = ctx.saved_tensors
        B = output.shape[0]
        o = output.reshape(B, -1)
        diff = o.clamp(0, 1) - gt.reshape(B, -1)
        sse = (diff * diff).sum(dim=1, keepdim=True)
        inside = (o > 0) & (o < 1)
        go = (-20.0 / 2.302585092994046) * g.reshape(B, 1) * diff / sse * inside
        return go.view_as(output), None


def torch_psnr(output, gt):
    """tfpnp/env/base.py:237-242 -> [B,1]"""
    if torch.is_grad_enabled() and output.requires_grad:
        return _Psnr.apply(output, gt)
    return ops.psnr(output, gt)


class PnPEnv:
    def __init__(self, solver, max_episode_step, aux_keys=None):
        self.solver = solver
        self.max_episode_step = max_episode_step
        self.cur_step = 0
        self.state = None
        self.idx_left = None
        self.last_metric = 0
        self.metric_fn = torch_psnr
        self.aux_keys = aux_keys

    def reset(self, data):
        """data: dict of device tensors with at least x0, gt, output + the solver's aux inputs.  base.py:121-155"""
        self.cur_step = 0
        data = dict(data)
        data['solver'] = self.solver.reset(data)
        data['output'] = data['output'].clone()
        self.state = data
        B = data['gt'].shape[0]
        self.idx_left = torch.arange(0, B, device=data['gt'].device)
        self.last_metric = self._compute_metric()
        return self.state

    def step(self, action):
        """base.py:157-191.  action: dict of [n_live, action_pack] tensors + 'idx_stop' [n_live]."""
        self.cur_step += 1
        il = self.idx_left
        with torch.no_grad():
            aux = tuple(a[il, ...] for a in self.solver.filter_aux_inputs(self.state))
            inputs = (self.state['solver'][il, ...], aux)
            parameters = self.solver.filter_hyperparameter(action)
            solver_state = self.solver(inputs, parameters)
        self.state['output'][il, ...] = self.solver.get_output(solver_state)
        self.state['solver'][il, ...] = solver_state
        reward = self._compute_reward()
        idx_stop = action['idx_stop']
        self.idx_left = il[idx_stop == 0]
        all_done = len(self.idx_left) == 0
        done = idx_stop.detach()
        if self.cur_step == self.max_episode_step:
            all_done = True
            done = torch.ones_like(idx_stop)
        return reward, all_done, {'done': done}

    def forward(self, state, action):
        """Differentiable one-step model of the environment used by the actor/critic update
        (tfpnp/env/base.py:193-206; called from tfpnp/trainer/mddpg/trainer.py:171-192).  `state` is the dict form of
        the observation (keys 'solver', 'output', 'gt' + the solver's aux inputs; the reference packs the same
        tensors channel-wise, base.py:208-231); returns (next solver state, delta-PSNR reward [B,1]).  Gradients
        flow into whatever `action` / `state` tensors require them."""
        inputs = (state['solver'], tuple(self.solver.filter_aux_inputs(state)))
        parameters = self.solver.filter_hyperparameter(action)
        solver_state = self.solver(inputs, parameters)
        output2 = self.solver.get_output(solver_state)
        reward = self.metric_fn(output2, state['gt']) - self.metric_fn(state['output'], state['gt'])
        return solver_state, reward

    def _compute_metric(self):
        return self.metric_fn(self.state['output'].detach(), self.state['gt'])

    def _compute_reward(self):
        metric = self._compute_metric()
        reward = metric - self.last_metric
        self.last_metric = metric
        return reward
