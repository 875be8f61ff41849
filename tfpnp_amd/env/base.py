"""PnPEnv -- the caller of the hot path, mirror of tfpnp/env/base.py:43-242 (same constructor, `reset` / `step` /
`forward` / `get_images` signatures and return values) so the reference's trainer / evaluator loops drive it unchanged.

What happens per `step` (base.py:157-191): gather the live rows `idx_left`, run the native solver under no_grad, write
state / output back, delta-PSNR reward over the whole batch (native pnpx_psnr), build the observation of the rows that
were live, shrink `idx_left` by `idx_stop`, build the observation of the rows still live.  `forward` (base.py:193-206) is
the differentiable one-step model used by the actor / critic update (native VJPs, see tfpnp_amd/autograd.py).

The four task environments of the reference (tasks/*/env.py) differ only in which state entries make up the
observation and how they are packed for the policy; here that is data (class attributes), not four copies of the code.
"""
import numpy as np
import torch

from .. import ops
from ..data.batch import Batch
from ..utils import transforms


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


def torch2img255(img):
    """tfpnp/utils/misc.py:9-12"""
    return np.clip(img.detach().cpu().numpy(), 0, 1) * 255


class PnPEnv:
    # ---- task description (overridden by tasks/*: CSMRIEnv, PREnv, CTEnv, SPIEnv)
    ob_base_dim = 0          # channels of the policy observation besides the solver variables
    ob_keys = ()             # state entries copied into the observation (besides gt / variables / T)
    float_keys = ()          # ... of which these are cast to float (bool masks)
    policy_layout = ()       # (key, 'real' | 'channel' | 'raw') in channel order of get_policy_ob
    input_key = None         # observation entry shown as the "input" image
    aux_keys = ()            # solver aux inputs, in order
    aux_bool = ()            # ... of which these are handed over as bool

    def __init__(self, data_loader, solver, max_episode_step, data_transform=None):
        self.data_loader = data_loader
        self.data_iterator = iter(data_loader) if data_loader is not None else None
        self.device = torch.device('cpu')
        self.data_transform = data_transform
        self.solver = solver
        self.max_episode_step = max_episode_step
        self.cur_step = 0
        self.state = None
        self.idx_left = None
        self.last_metric = 0
        self.metric_fn = torch_psnr

    # ------------------------------------------------------------------ observation plumbing (tasks/*/env.py)
    def get_policy_ob(self, ob):
        parts = []
        for key, kind in self.policy_layout:
            v = ob[key]
            if kind == 'real':
                v = transforms.complex2real(v)
            elif kind == 'channel':
                v = transforms.complex2channel(v)
            parts.append(v)
        return torch.cat(parts, 1)

    def get_eval_ob(self, ob):
        return self.get_policy_ob(ob)

    def _get_attribute(self, ob, key):
        if key == 'gt':
            return ob.gt
        if key == 'output':
            return self.solver.get_output(ob.variables)
        if key == 'input':
            return ob[self.input_key]
        if key == 'solver_input':
            return ob.variables, tuple(ob[k].bool() if k in self.aux_bool else ob[k] for k in self.aux_keys)
        raise NotImplementedError('key is not supported, ' + str(key))

    def _build_next_ob(self, ob, solver_state):
        nxt = Batch({k: ob[k] for k in ('gt',) + tuple(self.ob_keys)})
        nxt.update(variables=solver_state, T=ob.T + 1 / self.max_episode_step)
        return nxt

    def _observation(self):
        il = self.idx_left
        ob = Batch(gt=self.state['gt'][il, ...], variables=self.state['solver'][il, ...], T=self.state['T'][il, ...])
        for k in self.ob_keys:
            v = self.state[k][il, ...]
            ob[k] = v.float() if k in self.float_keys else v
        return ob

    # ------------------------------------------------------------------ basic API (tfpnp/env/base.py:121-206)
    def reset(self, data=None):
        self.cur_step = 0
        if data is None:
            try:
                data = next(self.data_iterator)
            except StopIteration:
                self.data_iterator = iter(self.data_loader)
                data = next(self.data_iterator)
        if self.data_transform is not None:
            data = self.data_transform(data)
        data = {k: (v.to(self.device) if isinstance(v, torch.Tensor) and self.device.type != 'cpu' else v)
                for k, v in dict(data).items()}
        data['solver'] = self.solver.reset(data)
        if 'output' in data:
            data['output'] = data['output'].clone()
        B, _, H, W = data['gt'].shape
        dev = data['gt'].device
        data['T'] = torch.full((B, 1, H, W), self.cur_step / self.max_episode_step, dtype=torch.float32, device=dev)
        self.state = data
        self.idx_left = torch.arange(0, B, device=dev)
        self.last_metric = self._compute_metric()
        return self._observation()

    def step(self, action):
        self.cur_step += 1
        il = self.idx_left
        with torch.no_grad():
            aux = tuple(a[il, ...] for a in self.solver.filter_aux_inputs(self.state))
            inputs = (self.state['solver'][il, ...], aux)
            parameters = self.solver.filter_hyperparameter(action)
            solver_state = self.solver(inputs, parameters)
        self.state['T'] = torch.full_like(self.state['T'], self.cur_step / self.max_episode_step)
        self.state['output'][il, ...] = self.solver.get_output(solver_state)
        self.state['solver'][il, ...] = solver_state
        reward = self._compute_reward()
        ob = self._observation()
        idx_stop = action['idx_stop']
        self.idx_left = il[idx_stop == 0]
        all_done = len(self.idx_left) == 0
        done = idx_stop.detach()
        if self.cur_step == self.max_episode_step:
            all_done = True
            done = torch.ones_like(idx_stop)
        ob_masked = self._observation()
        return ob, ob_masked, reward, all_done, {'done': done}

    def forward(self, ob, action):
        """Differentiable one-step model (base.py:193-206): gradients flow into `action` / `ob` tensors that require
        them.  -> (next observation, delta-PSNR reward [B,1])"""
        output = self._get_attribute(ob, 'output')
        gt = self._get_attribute(ob, 'gt')
        inputs = self._get_attribute(ob, 'solver_input')
        parameters = self.solver.filter_hyperparameter(action)
        solver_state = self.solver(inputs, parameters)
        output2 = self.solver.get_output(solver_state)
        reward = self.metric_fn(output2, gt) - self.metric_fn(output, gt)
        return self._build_next_ob(ob, solver_state), reward

    def get_images(self, ob, pre_process=torch2img255):
        return tuple(pre_process(self._get_attribute(ob, k)) for k in ('input', 'output', 'gt'))

    def to(self, device):
        if not isinstance(device, torch.device):
            raise TypeError('device must be torch.device, but got {}'.format(type(device)))
        self.device = device
        return self

    def _compute_metric(self):
        return self.metric_fn(self.state['output'].detach(), self.state['gt'])

    def _compute_reward(self):
        metric = self._compute_metric()
        reward = metric - self.last_metric
        self.last_metric = metric
        return reward
