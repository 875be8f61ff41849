"""PnPEnv -- the caller of the hot path: same constructor, `reset` / `step` / `forward` / `get_images` signatures and
return values as tfpnp/env/base.py:43-242, so rollout and evaluation loops written against the reference
(eval/evaluator.py, the rollout half of trainer/mddpg/trainer.py) drive it unchanged.  Scope: inference / evaluation
episodes and the differentiable one-step model `forward` (gradients wrt actions through the native VJPs,
tfpnp_amd/autograd.py).  A trainable actor / critic and the MDDPG trainer are NOT part of this package.

One `step` (base.py:157-191), orchestrated on the device (csrc/env.hip):
  live-row gather of solver state + aux inputs (ONE launch)  ->  native solver loop  ->  write-back of state and output
  (ONE launch)  ->  delta-PSNR reward over the whole batch (native)  ->  observation of the rows that were live (ONE
  gather launch)  ->  device-side compaction of the live set by `idx_stop`, whose survivor count is the step's single
  host read (`all_done` is a Python bool in the contract)  ->  observation of the rows still live.
No boolean-mask indexing, no `len(idx_left)` sync, no aten gather / index_put kernels.

The four task environments of the reference (tasks/*/env.py) differ only in which state entries make up the
observation and how they are packed for the policy; here that is data (class attributes), not four copies of the code.
"""
import numpy as np
import torch

from .._lib import PnpxError

from .. import ops
from ..data.batch import Batch
from ..utils import transforms


# ---- row bookkeeping.  Device tensors go through the native ops; CPU tensors (host-logic tests with stub solvers and
# the gloo process group -- the native solvers themselves refuse CPU tensors) use plain indexing.
def _take_rows(tensors, rows, n):
    if rows.is_cuda:
        return ops.rows_gather(tensors, rows, n)
    sel = rows[:n]
    return [t[sel] for t in tensors]


def _put_rows(values, targets, rows, n):
    if rows.is_cuda:
        ops.rows_scatter(values, targets, rows, n)
        return
    sel = rows[:n]
    for v, t in zip(values, targets):
        t[sel] = v


def _surviving_rows(rows, idx_stop, n):
    """(rows that continue, their count as a Python int) -- the step's one host read."""
    idx_stop = idx_stop.reshape(-1).to(torch.int64)
    if rows.is_cuda:
        return ops.live_compact(rows, idx_stop, n)
    keep = rows[:n][idx_stop[:n] == 0]
    return keep, int(keep.numel())


def torch_psnr(output, gt):
    """tfpnp/env/base.py:237-242 -> [B,1].  One dispatcher op (torch.ops.pnpx.psnr) with an analytic VJP."""
    from .. import torch_ops as T
    return T.call("psnr", output, gt)


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
        self._rows = None        # int64 [B]: the first _n_live entries are the batch rows still being iterated
        self._n_live = 0
        self.last_metric = 0
        self.metric_fn = torch_psnr
        self.range_redone_steps = 0     # steps repeated in exact fp32 because the half-split range guard tripped

    @property
    def idx_left(self):
        """Rows still live (the reference's attribute of the same name): a view, no copy, no sync."""
        return self._rows[:self._n_live]

    # ------------------------------------------------------------------ observation plumbing (tasks/*/env.py)
    def get_policy_ob(self, ob):
        parts = [(ob[key], kind) for key, kind in self.policy_layout]
        if parts and parts[0][0].is_cuda:
            return ops.policy_ob_pack(parts)          # one launch: views + channel concat fused
        views = {'real': transforms.complex2real, 'channel': transforms.complex2channel, 'raw': lambda v: v}
        return torch.cat([views[kind](v) for v, kind in parts], 1)

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

    def _observation(self, rows=None, n=None):
        """Batch of the live rows (default: the currently live ones): gt, variables, T and the task's ob_keys (bool entries
        listed in float_keys as float)."""
        names = ('gt', 'solver', 'T') + tuple(self.ob_keys)
        if rows is None:
            rows, n = self._rows, self._n_live
        got = _take_rows([self.state[k] for k in names], rows, n)
        ob = Batch()
        for k, v in zip(names, got):
            ob['variables' if k == 'solver' else k] = v.float() if k in self.float_keys else v
        return ob

    # ------------------------------------------------------------------ basic API (tfpnp/env/base.py:121-206)
    def _next_batch(self):
        try:
            return next(self.data_iterator)
        except StopIteration:
            self.data_iterator = iter(self.data_loader)
            return next(self.data_iterator)

    def reset(self, data=None):
        self.cur_step = 0
        data = dict(self._next_batch() if data is None else data)
        if self.data_transform is not None:
            data = dict(self.data_transform(data))
        if self.device.type != 'cpu':
            data = {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}
        data['solver'] = self.solver.reset(data)
        if 'output' in data:
            data['output'] = data['output'].clone()
        gt = data['gt']
        B, _, H, W = gt.shape
        data['T'] = torch.zeros((B, 1, H, W), dtype=torch.float32, device=gt.device)
        self.state = {k: (v.contiguous() if isinstance(v, torch.Tensor) else v) for k, v in data.items()}
        self._rows = torch.arange(B, device=gt.device)
        self._n_live = B
        self.last_metric = self._compute_metric()
        return self._observation()

    def _native_context(self, like):
        """The denoiser's native context for the device of `like` (None for CPU stubs / foreign solvers)."""
        den = getattr(self.solver, 'denoiser', None)
        if like is None or not like.is_cuda or not hasattr(den, 'context'):
            return None
        return den.context(like.device)

    def step(self, action):
        self.cur_step += 1
        st, rows, n = self.state, self._rows, self._n_live
        with torch.no_grad():
            live = _take_rows([st['solver'], *self.solver.filter_aux_inputs(st)], rows, n)
            hyper = self.solver.filter_hyperparameter(action)
            prev_metric = self.last_metric

            def advance():
                solver_state = self.solver((live[0], tuple(live[1:])), hyper)
                st['T'].fill_(self.cur_step / self.max_episode_step)
                _put_rows([self.solver.get_output(solver_state), solver_state], [st['output'], st['solver']], rows, n)
                self.last_metric = prev_metric
                # observation: the rows that were live DURING this step (also when the step is repeated after the compaction)
                return self._compute_reward(), self._observation(rows, n)

            reward, ob = advance()
            idx_stop = action['idx_stop']
            self._rows, self._n_live = _surviving_rows(rows, idx_stop, n)
            # The compaction above is the step's one host read: the stream has drained, so the half-split range guard's
            # host-mapped flag is final for this step.  If an activation left the f16 hi/lo range the context has just
            # latched itself to the exact-fp32 convolutions; the solver call is functional (its gathered inputs are still
            # here), so the step is simply run again -- nothing invalid ever leaves step().
            ctx = self._native_context(rows) if n else None
            if ctx is not None and ctx.get_option('conv_mode') == 1 and ctx.range_tripped():
                import warnings
                self.range_redone_steps += 1
                if getattr(ctx, 'is_drunet', False):
                    # DRUNet: no fp32 family; the tripped guard has moved the context's passes 16x further inside the range
                    # (positively homogeneous network, csrc/drunet.hip).  Repeat until a pass stays in range.
                    guard = ctx.get_option('range_guard')
                    for _ in range(4):
                        ctx.set_option('range_guard', guard)           # acknowledge (synchronises); the raised shift stays
                        reward, ob = advance()
                        torch.cuda.synchronize(rows.device)
                        if not ctx.range_tripped():
                            break
                    else:      # still out of range at the largest shift: put the rows back and fail loudly
                        _put_rows([self.solver.get_output(live[0]), live[0]], [st['output'], st['solver']], rows, n)
                        self.last_metric = prev_metric
                        self._rows, self._n_live = rows, n
                        self.cur_step -= 1
                        raise PnpxError('half-split range guard: DRUNet activations exceed the f16 hi/lo range even on inputs '
                                        'scaled by 2^-16; the step was rolled back')
                    warnings.warn('half-split range guard tripped in PnPEnv.step: the step was repeated on down-scaled inputs '
                                  f"(DRUNet context, drunet_shift = {ctx.get_option('drunet_shift')})")
                else:
                    warnings.warn('half-split range guard tripped in PnPEnv.step: the step was repeated with the exact-fp32 '
                                  'convolutions and the denoiser context stays in conv_mode 0 (re-arm with '
                                  "context.set_option('range_guard', 1) + set_option('conv_mode', 1))")
                    reward, ob = advance()
            done = idx_stop.detach()
            all_done = self._n_live == 0
            if self.cur_step == self.max_episode_step:
                all_done, done = True, torch.ones_like(idx_stop)
            ob_masked = self._observation()               # rows still live
        return ob, ob_masked, reward, all_done, {'done': done}

    def forward(self, ob, action):
        """Differentiable one-step model (base.py:193-206): gradients flow into `action` / `ob` tensors that require
        them.  -> (next observation, delta-PSNR reward [B,1])"""
        gt = self._get_attribute(ob, 'gt')
        before = self.metric_fn(self._get_attribute(ob, 'output'), gt)
        solver_state = self.solver(self._get_attribute(ob, 'solver_input'), self.solver.filter_hyperparameter(action))
        after = self.metric_fn(self.solver.get_output(solver_state), gt)
        return self._build_next_ob(ob, solver_state), after - before

    def get_images(self, ob, pre_process=torch2img255):
        return tuple(pre_process(self._get_attribute(ob, k)) for k in ('input', 'output', 'gt'))

    def to(self, device):
        if not isinstance(device, torch.device):
            raise TypeError(f'PnPEnv.to expects a torch.device, got {type(device).__name__}')
        self.device = device
        return self

    def _compute_metric(self):
        return self.metric_fn(self.state['output'].detach(), self.state['gt'])

    def _compute_reward(self):
        metric = self._compute_metric()
        reward, self.last_metric = metric - self.last_metric, metric
        return reward
