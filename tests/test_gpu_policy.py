"""GPU tests of SURVEY 8(f) rank 2: the native policy actor (pnpx_policy_forward) and the policy-driven episode
(observation packing -> actor -> native solver -> env step), against goldens produced by the real reference and
against the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.golden_inputs import POLICY_SEED, ROLLOUT_CONTINUE_BIAS, policy_obs
from tests.test_oracle_golden import POLICY_CASES
from tfpnp_amd import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


ACTORS = {"admm": ("ResNetActor_ADMM", 6), "admm_rect": ("ResNetActor_ADMM", 6), "iadmm_pr": ("ResNetActor_IADMM", 14),
          "spi": ("ResNetActor_SPI", 3)}


def make_actor(name, nin, n_det, spi, continue_bias=0.0):
    from tfpnp_amd import policy
    cls, num_aux = ACTORS[name]
    P = synth.make_policy_params(nin, n_det, spi, seed=POLICY_SEED)
    P["fc_softmax.0.bias"] = P["fc_softmax.0.bias"] + np.array([continue_bias, 0], np.float32)
    actor = getattr(policy, cls)(num_aux, 5)
    actor.load_state_dict(P)
    return actor, P


@pytest.mark.parametrize("name,nin,n_det,spi,shape", POLICY_CASES)
def test_policy_actor_golden(name, nin, n_det, spi, shape):
    from tfpnp_amd import ops
    gd = golden("policy_actor")
    actor, _ = make_actor(name, nin, n_det, spi)
    ob = g(policy_obs(shape[0], nin, shape[1], shape[2], 81))
    probs, det = ops.policy_forward(actor.context(dev()), ob)
    ep, ed = np.abs(probs.cpu().numpy() - gd[f"{name}_probs"]).max(), np.abs(det.cpu().numpy() - gd[f"{name}_det"]).max()
    print(f"{name}: max|d probs| {ep:.2e}  max|d det| {ed:.2e}")
    assert ep < 2e-5 and ed < 2e-5
    action, logp, ent, hidden = actor(ob, None, False, None)
    for k in action:
        ref = gd[f"{name}_action_{k}"]
        if k == "idx_stop":
            assert np.array_equal(action[k].cpu().numpy(), ref)
        else:
            assert np.allclose(action[k].cpu().numpy(), ref, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    assert np.allclose(logp.cpu().numpy(), gd[f"{name}_logp"], atol=1e-4)
    assert np.allclose(ent.cpu().numpy(), gd[f"{name}_entropy"], atol=1e-4)
    # given stop decisions (training-time replay): log-prob of those
    forced = torch.ones(shape[0], dtype=torch.int64, device=dev())
    a2, lp2, _, _ = actor(ob, forced, True, None)
    assert torch.equal(a2["idx_stop"], forced)
    assert np.allclose(lp2.cpu().numpy()[:, 0], np.log(gd[f"{name}_probs"][:, 1]), atol=1e-4)


def test_policy_actor_full_size_vs_oracle_and_determinism():
    from oracle import pnp_oracle as O
    from tfpnp_amd import ops
    actor, P = make_actor("admm", 9, 10, False)
    ob = policy_obs(6, 9, 256, 256, 83)
    probs, det = ops.policy_forward(actor.context(dev()), g(ob))
    po, do = O.policy_forward(P, torch.from_numpy(ob[:3]), False)
    assert np.abs(probs[:3].cpu().numpy() - po.numpy()).max() < 2e-5
    assert np.abs(det[:3].cpu().numpy() - do.numpy()).max() < 2e-5
    # batch independence, repeatability, workspace reuse with a smaller batch and another size in between
    p1, d1 = ops.policy_forward(actor.context(dev()), g(ob[4:5]))
    assert torch.equal(p1, probs[4:5]) and torch.equal(d1, det[4:5])
    ops.policy_forward(actor.context(dev()), g(policy_obs(2, 9, 64, 96, 84)))
    p2, d2 = ops.policy_forward(actor.context(dev()), g(ob))
    assert torch.equal(p2, probs) and torch.equal(d2, det)


def test_policy_launch_chains_are_bit_identical():
    """The actor forward sliced into independent launch chains (option `chains`; automatic between the round boundaries of its
    deep stages): per-observation outputs do not depend on the slicing, nor on the batch they arrive in."""
    from tfpnp_amd import ops
    actor, _ = make_actor("admm", 9, 10, False)
    ctx = actor.context(dev())
    try:
        for (B, H, W) in [(9, 64, 64), (17, 64, 96), (5, 128, 128), (33, 32, 32)]:
            ob = g(policy_obs(B, 9, H, W, 85))
            ctx.set_option("chains", 1)
            ref = [t_.clone() for t_ in ops.policy_forward(ctx, ob)]
            for c in (2, 3, 0):
                ctx.set_option("chains", c)
                out = ops.policy_forward(ctx, ob)
                assert all(torch.equal(a, b) for a, b in zip(out, ref)), (B, H, W, c)
            one = ops.policy_forward(ctx, ob[B // 2:B // 2 + 1])
            assert all(torch.equal(a, b[B // 2:B // 2 + 1]) for a, b in zip(one, ref))
    finally:
        ctx.set_option("chains", 0)


def test_policy_rejects_bad_input():
    from tfpnp_amd import ops, policy
    from tfpnp_amd._lib import PnpxError
    actor, P = make_actor("admm", 9, 10, False)
    ctx = actor.context(dev())
    with pytest.raises(PnpxError):
        ops.policy_forward(ctx, g(policy_obs(1, 9, 48, 64, 1)))       # H not a multiple of 32
    with pytest.raises(PnpxError):
        ops.policy_forward(ctx, g(policy_obs(1, 8, 64, 64, 1)))       # wrong channel count
    with pytest.raises(PnpxError):
        ops.policy_forward(ctx, torch.from_numpy(policy_obs(1, 9, 64, 64, 1)))   # CPU tensor
    bad = dict(P)
    bad.pop("actor_encoder.layer3.0.shortcut.0.weight")
    a = policy.ResNetActor_ADMM(6, 5)
    a.load_state_dict(bad)
    with pytest.raises(PnpxError):
        a.context(dev())
    with pytest.raises(ValueError):
        policy.ResNetActor_ADMM(6, 5).context(dev())


def test_policy_driven_rollout_golden(unet_params):
    """evaluator.py:85-100: ob -> get_policy_ob -> actor (arg-max stop) -> env.step, three policy steps."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
    gd = golden("policy_rollout_csmri")
    B, H, W, seed = 3, 64, 64, 91
    d = synth.make_csmri_batch(B, H, W, ratio=4, sigma_n=15.0, seed=seed)
    env = CSMRIEnv(None, ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params)), max_episode_step=3)
    actor, _ = make_actor("admm", 9, 10, False, continue_bias=ROLLOUT_CONTINUE_BIAS)
    ob = env.reset({k: g(v) for k, v in d.items()})
    hidden = actor.init_state(ob.shape[0])
    for s in range(int(gd["n_steps"])):
        action, _, _, hidden = actor(env.get_policy_ob(ob), idx_stop=None, train=False, hidden=hidden)
        assert np.array_equal(action["idx_stop"].cpu().numpy(), gd[f"idx_stop{s}"])
        assert np.allclose(action["sigma_d"].cpu().numpy(), gd[f"sigma_d{s}"], atol=2e-5)
        assert np.allclose(action["mu"].cpu().numpy(), gd[f"mu{s}"], atol=2e-5)
        _, ob, reward, all_done, _ = env.step(action)
        assert np.allclose(reward.cpu().numpy(), gd[f"reward{s}"], atol=5e-3)
        assert rel(env.state["output"], gd[f"output{s}"]) < 1e-4
    assert all_done
    inp, out, gt = env.get_images(ob) if len(ob) else (None, None, None)


def test_evaluator_over_mat_items(unet_params, tmp_path):
    """evaluator.py:21-118 on the native path: .mat items -> Evaluator.eval -> per-sample PSNR traces and images."""
    from tfpnp_amd.data.eval_datasets import CSMRIEvalDataset, collate, save_eval_item
    from tfpnp_amd.env import torch_psnr
    from tfpnp_amd.eval import Evaluator
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
    d = synth.make_csmri_batch(2, 64, 64, ratio=4, sigma_n=15.0, seed=101)
    for b in range(2):
        save_eval_item(str(tmp_path / f"item{b}.mat"), {k: v[b] for k, v in d.items()}, task='csmri')
    ds = CSMRIEvalDataset(str(tmp_path))
    loader = [collate([ds[i]], device=dev()) for i in range(len(ds))]
    env = CSMRIEnv(None, ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params)), max_episode_step=3)
    actor, _ = make_actor("admm", 9, 10, False, continue_bias=ROLLOUT_CONTINUE_BIAS)
    ev = Evaluator(env, {"synthetic": loader}, savedir=str(tmp_path / "out"))
    mean_psnr = ev.eval(actor, step=1)
    step, name, summary = ev.history[-1]
    assert name == "synthetic" and summary["iters"] == 3 and np.isfinite(mean_psnr)
    assert abs(summary["psnr"] - mean_psnr) < 1e-9
    # the reported PSNR is the env's own metric on the final output of the last sample
    last = float(torch_psnr(env.state["output"], env.state["gt"])[0, 0])
    from tfpnp_amd.eval import eval_single
    p0, p1, info, imgs = eval_single(env, {k: v for k, v in loader[1].items() if k != "name"}, actor, 3)
    assert abs(p1 - last) < 1e-3 and len(info[1]) == info[0] + 1 and abs(info[1][0] - p0) < 1e-9
    assert (tmp_path / "out" / "synthetic" / "item1" / "1" / "output.png").exists()
    # psnr_init equals the PSNR of the zero-filled reconstruction stored in the item
    zf = np.clip(d["ATy0"][1][..., 0], 0, 1)
    assert abs(p0 - 10 * np.log10(1.0 / np.mean((zf - d["gt"][1]) ** 2))) < 1e-3


def test_batched_evaluation_matches_single(unet_params):
    """eval_batch over 3 samples == three eval_single runs (same policy decisions, same PSNRs)."""
    from tfpnp_amd.eval import eval_batch, eval_single, Evaluator
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
    d = synth.make_csmri_batch(3, 64, 64, ratio=4, sigma_n=15.0, seed=111)
    env = CSMRIEnv(None, ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params)), max_episode_step=3)
    actor, _ = make_actor("admm", 9, 10, False, continue_bias=1.0)
    batch = {k: g(v) for k, v in d.items()}
    p0, p1, steps, _ = eval_batch(env, batch, actor, 3)
    for b in range(3):
        one = {k: v[b:b + 1] for k, v in batch.items()}
        q0, q1, info, _ = eval_single(env, one, actor, 3)
        assert abs(q0 - p0[b]) < 1e-6 and abs(q1 - p1[b]) < 2e-3 and info[0] == steps[b]
    ev = Evaluator(env, {"set": [batch]})
    mean_psnr = ev.eval(actor, step=0)
    assert abs(mean_psnr - float(np.mean(p1))) < 2e-3 and ev.history[-1][2]["iters"] == float(np.mean(steps))
