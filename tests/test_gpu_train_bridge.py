"""A plain PyTorch actor learns through the native differentiable one-step model (examples/train_bridge.py): the
gradients PnPEnv.forward produces on the native path are consumable by an ordinary optimiser and do improve the
reward -- the capability the reference's MDDPG update builds on (tfpnp/trainer/mddpg/trainer.py:171-200)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_tiny_actor_improves_reward_through_native_vjp():
    import train_bridge
    hist = train_bridge.train(steps=10, B=3, H=48, action_pack=3, log=lambda *_: None)
    assert all(map(lambda v: v == v, hist))                      # finite
    assert max(hist[-3:]) > hist[0] + 0.05, hist                 # dB of delta-PSNR gained by gradient steps on the actions


def test_batch_stack_and_convert2batch_contract():
    """tfpnp/data/batch.py Batch.stack as trainer.convert2batch uses it (tfpnp/trainer/mddpg/trainer.py:225-228)."""
    from tfpnp_amd.data.batch import Batch
    dev = torch.device("cuda:0")
    items = [Batch(x=torch.full((2, 3), float(i), device=dev), T=torch.tensor([i], device=dev)) for i in range(4)]
    b = Batch.stack(items)
    assert tuple(b.x.shape) == (4, 2, 3) and tuple(b.T.shape) == (4, 1) and float(b.x[2].mean()) == 2.0
    assert len(b) == 4 and b[1:3].x.shape[0] == 2
    c = Batch.cat([b, b])
    assert tuple(c.x.shape) == (8, 2, 3)
