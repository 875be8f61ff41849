"""Training bridge: a plain PyTorch actor optimised THROUGH the native solver path.

The reference learns its tuning-free policy by differentiating the one-step model (PnPEnv.forward,
tfpnp/env/base.py:193-206) inside the MDDPG update (tfpnp/trainer/mddpg/trainer.py:171-200): the deterministic-policy
term of the actor loss is  -(gamma * V(ob2) + reward)  with (ob2, reward) = env.forward(ob, actor(policy_ob)).
tfpnp_amd ships the native, eval-mode ResNet actors for rollouts and evaluation, not a trainer; what it does provide is
that differentiable one-step model on the native kernels (fused ADMM forward + VJP, csrc/csmri.hip).  Any nn.Module that
maps the policy observation to the action dict can therefore be trained against it with ordinary PyTorch optimisers --
this file is that bridge in its smallest form (reward term only, a small CNN actor, Adam), and
tests/test_gpu_train_bridge.py runs it.

usage (GPU box):  python examples/train_bridge.py [steps] [B] [H]
"""
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks import csmri


class TinyActor(nn.Module):
    """policy_ob [B, C, H, W] -> {'sigma_d': [B, T] in (0, 70/255), 'mu': [B, T] in (0, 1), 'idx_stop': [B]} -- the
    contract of ResNetActor_ADMM.forward / action_mapping (tfpnp/policy/network.py:149-175), with trainable weights."""

    def __init__(self, in_ch, action_pack):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(in_ch, 16, 3, stride=2, padding=1), nn.ReLU(),
                                  nn.Conv2d(16, 32, 3, stride=2, padding=1), nn.ReLU(), nn.AdaptiveAvgPool2d(1))
        self.head = nn.Linear(32, 2 * action_pack)
        self.action_pack = action_pack

    def forward(self, policy_ob):
        a = torch.sigmoid(self.head(self.body(policy_ob).flatten(1)))
        T = self.action_pack
        return OrderedDict(sigma_d=a[:, :T] * (70 / 255), mu=a[:, T:],
                           idx_stop=torch.zeros(policy_ob.shape[0], dtype=torch.int64, device=policy_ob.device))


def train(steps=12, B=4, H=64, action_pack=3, lr=3e-2, seed=0, log=print):
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
    env = csmri.CSMRIEnv(None, csmri.ADMMSolver_CSMRI(den), max_episode_step=6)
    data = synth.make_csmri_batch(B, H, H, ratio=4, sigma_n=15.0, seed=seed + 1)
    ob = env.reset({k: torch.from_numpy(v).to(dev) for k, v in data.items() if hasattr(v, "dtype")})
    policy_ob = env.get_policy_ob(ob)
    actor = TinyActor(policy_ob.shape[1], action_pack).to(dev)
    opt = torch.optim.Adam(actor.parameters(), lr=lr)
    history = []
    for it in range(steps):
        action = actor(policy_ob)
        _, reward = env.forward(ob, action)            # native forward; backward() below runs the native VJP
        loss = -reward.mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        history.append(float(reward.detach().mean()))
        log(f"step {it:2d}  mean delta-PSNR reward {history[-1]:+.4f} dB   sigma_d[0] "
            f"{[round(v * 255, 1) for v in action['sigma_d'][0].tolist()]}  mu[0] {[round(v, 3) for v in action['mu'][0].tolist()]}")
    return history


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    train(*a)
