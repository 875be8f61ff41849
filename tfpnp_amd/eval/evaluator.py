"""Evaluation loop over the native path -- mirror of tfpnp/eval/evaluator.py (Evaluator.eval :21-73, eval_single
:75-118) and of the PSNR it reports (tfpnp/utils/metric.py:23-25: skimage peak_signal_noise_ratio, data_range 255,
averaged over channels).  Policy -> actions -> solver -> env all run on the MI355X; this file is host-side bookkeeping
(per-sample PSNR trace, action sequences, optional image dump)."""
import json
import os
import time

import numpy as np
import torch


def psnr_qrnn3d(X, Y, data_range=255):
    """Band-wise PSNR of [C,H,W] arrays in the 0..255 range (float64, as skimage computes it)."""
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    vals = []
    for ch in range(X.shape[-3]):
        mse = np.mean((X[ch] - Y[ch]) ** 2)
        vals.append(10 * np.log10(data_range ** 2 / mse))
    return float(np.mean(vals))


def _rollout(env, data, policy, max_episode_step, metric, trace):
    """Policy-driven episode over a batch of any size with per-item bookkeeping.  Items stop individually (`idx_stop`),
    exactly as in training rollouts.  trace=True additionally records each item's PSNR after every step and its
    action sequence (one device read-back per step, as the reference's eval_single does)."""
    from ..env.base import torch2img255
    ob = env.reset(data=data)
    B = ob.shape[0]
    hidden = policy.init_state(B)
    first = env.get_images(ob)                                     # (input, output, gt), 0..255 arrays
    rec = [{'psnr': [metric(first[1][b], first[2][b])], 'steps': 0, 'actions': {}} for b in range(B)]
    live = list(range(B))
    t0 = time.time()
    for _ in range(max_episode_step):
        action, _, _, hidden = policy(env.get_policy_ob(ob), idx_stop=None, train=False, hidden=hidden)
        stop = action['idx_stop'].detach().cpu().numpy()
        ob_stepped, ob, _, all_done, _ = env.step(action)
        if trace:
            _, out, gt = env.get_images(ob_stepped)
            values = {k: v.detach().cpu().numpy() for k, v in action.items() if k != 'idx_stop'}
        for row, b in enumerate(live):
            rec[b]['steps'] += 1
            if trace:
                rec[b]['psnr'].append(float(metric(out[row], gt[row])))
                for k, v in values.items():
                    rec[b]['actions'].setdefault(k, []).extend(v[row].tolist())
        live = [b for b, s_ in zip(live, stop) if s_ == 0]
        if all_done:
            break
    if env.state['gt'].is_cuda:
        torch.cuda.synchronize(env.state['gt'].device)
    run_time = time.time() - t0
    final = tuple(torch2img255(env.state[k]) for k in (env.input_key, 'output', 'gt'))
    for b in range(B):
        rec[b]['psnr_final'] = metric(final[1][b], final[2][b])
    return rec, first, final, run_time


def eval_single(env, data, policy, max_episode_step, metric=psnr_qrnn3d):
    """The reference's per-sample evaluation (eval/evaluator.py:75-118) = the B = 1 case of the batched rollout; same
    return layout: psnr_init, psnr_finished, (steps, psnr trace, action sequences, wall time), (input, output_init,
    output, gt)."""
    rec, first, final, run_time = _rollout(env, data, policy, max_episode_step, metric, trace=True)
    r = rec[0]
    return (r['psnr'][0], r['psnr_final'], (r['steps'], r['psnr'], r['actions'], run_time),
            (final[0][0], first[1][0], final[1][0], final[2][0]))


def eval_batch(env, data, policy, max_episode_step, metric=psnr_qrnn3d):
    """A whole loader batch at once (the native path is ~10x more efficient per image at env_batch 48 than at 1).
    Returns per-item lists (psnr_init, psnr_finished, episode_steps) and the wall time of the batch."""
    rec, _, _, run_time = _rollout(env, data, policy, max_episode_step, metric, trace=False)
    return [r['psnr'][0] for r in rec], [r['psnr_final'] for r in rec], [r['steps'] for r in rec], run_time


class Evaluator:
    def __init__(self, env, val_loaders, savedir=None, metric=psnr_qrnn3d):
        self.env = env
        self.val_loaders = val_loaders      # {name: iterable of batch-1 data dicts}
        self.savedir = savedir
        self.metric = metric
        self.history = []

    @torch.no_grad()
    def eval(self, policy, step):
        if hasattr(policy, 'eval'):
            policy.eval()
        total = 0.0
        for name, loader in self.val_loaders.items():
            rows = []
            for index, data in enumerate(loader):
                data = dict(data)
                if data['gt'].shape[0] > 1:       # batched evaluation: metrics only, no per-sample image dump
                    data.pop('name', None)
                    p0, p1, steps, run_time = eval_batch(self.env, data, policy, self.env.max_episode_step, self.metric)
                    rows += [{'iters': st, 'psnr_init': a, 'psnr': b, 'time': run_time / len(p0)}
                             for a, b, st in zip(p0, p1, steps)]
                    continue
                sample = data.pop('name', 'case' + str(index))
                sample = sample[0] if isinstance(sample, (list, tuple)) else sample
                psnr_init, psnr_finished, info, imgs = eval_single(self.env, data, policy,
                                                                   max_episode_step=self.env.max_episode_step,
                                                                   metric=self.metric)
                episode_steps, psnr_seq, action_seqs, run_time = info
                rows.append({'iters': episode_steps, 'psnr_init': psnr_init, 'psnr': psnr_finished, 'time': run_time})
                if self.savedir is not None:
                    base = os.path.join(self.savedir, name, str(sample), str(step))
                    os.makedirs(base, exist_ok=True)
                    for tag, img in zip(('input', 'output_init', 'output', 'gt'), imgs):
                        _save_img(img, os.path.join(base, tag + '.png'))
                    json.dump({'psnr': [float(p) for p in psnr_seq],
                               **{str(k): [float(x) for x in v] for k, v in action_seqs.items()}},
                              open(os.path.join(base, 'action_seqs.json'), 'w'))
            summary = {k: float(np.mean([r[k] for r in rows])) for k in rows[0]} if rows else {}
            self.history.append((step, name, summary))
            total += summary.get('psnr', 0.0)
        return total / max(1, len(self.val_loaders))


def _save_img(img, path):
    from PIL import Image
    a = np.asarray(img)
    a = a[..., 0] if a.ndim == 4 else a           # complex [C,H,W,2] inputs: real part
    a = a[0] if a.ndim == 3 else a
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(path)
