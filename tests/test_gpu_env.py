"""GPU tests of the device-side episode orchestration (csrc/env.hip): row gather / scatter, live-set compaction, fused
policy-observation packing -- the caller contract of PnPEnv.step (tfpnp/env/base.py:157-191) -- and that a whole env step
issues no host synchronisation besides its one survivor-count read."""
import numpy as np
import pytest
import torch

from tfpnp_amd import synth

pytestmark = pytest.mark.gpu

from tests import composed_solvers   # the reference's loops from differentiable ops (test infrastructure)
composed_solvers.install()


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(7, 3, 16, 16, 2), (5, 1, 50, 39), (9, 2, 5, 3), (48, 1, 64, 64)])
def test_rows_gather_scatter_match_indexing(shape):
    from tfpnp_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    B = shape[0]
    a = torch.rand(shape, generator=g).to(dev())
    m = (torch.rand((B, 1) + shape[2:4], generator=g) > 0.5).to(dev())       # bool rows (1 byte per pixel)
    perm = torch.randperm(B, generator=g).to(dev())
    for n in (B, B // 2, 1, 0):
        ga, gm = ops.rows_gather([a, m], perm, n)
        assert ga.dtype == a.dtype and gm.dtype == torch.bool
        assert torch.equal(ga, a[perm[:n]]) and torch.equal(gm, m[perm[:n]])
        ta, tm = torch.zeros_like(a), torch.zeros_like(m)
        ops.rows_scatter([ga, gm], [ta, tm], perm, n)
        ra, rm = torch.zeros_like(a), torch.zeros_like(m)
        ra[perm[:n]] = ga
        rm[perm[:n]] = gm
        assert torch.equal(ta, ra) and torch.equal(tm, rm)


def test_live_compact_matches_boolean_indexing():
    from tfpnp_amd import ops
    g = torch.Generator().manual_seed(3)
    for n in (1, 2, 63, 64, 65, 300, 1000):
        rows = torch.randperm(2 * n, generator=g)[:n].to(dev())
        stop = (torch.rand(n, generator=g) > 0.6).long().to(dev())
        out, k = ops.live_compact(rows, stop, n)
        ref = rows[stop == 0]
        assert k == ref.numel() and torch.equal(out[:k], ref)
    out, k = ops.live_compact(rows, torch.ones_like(stop), n)
    assert k == 0
    out, k = ops.live_compact(rows, stop, 0)
    assert k == 0


@pytest.mark.parametrize("H,W", [(32, 32), (50, 39), (16, 18)])
def test_policy_ob_pack_matches_views_and_cat(H, W):
    from tfpnp_amd import ops
    from tfpnp_amd.utils import transforms as T
    g = torch.Generator().manual_seed(H * W)
    B = 6
    v = torch.rand(B, 3, H, W, 2, generator=g).to(dev())
    y = torch.rand(B, 1, H, W, 2, generator=g).to(dev())
    m = (torch.rand(B, 1, H, W, generator=g) > 0.5).to(dev())
    t = torch.rand(B, 1, H, W, generator=g).to(dev())
    ref = torch.cat([T.complex2real(v), T.complex2channel(y), m.float(), t], 1)
    got = ops.policy_ob_pack([(v, "real"), (y, "channel"), (m, "raw"), (t, "raw")])
    assert got.shape == ref.shape and torch.equal(got, ref)
    rows = torch.tensor([4, 0, 5], device=dev())
    got = ops.policy_ob_pack([(v, "real"), (y, "channel"), (m, "raw"), (t, "raw")], idx=rows, n_rows=2)
    assert torch.equal(got, ref[rows[:2]])


def test_env_step_has_one_host_read(unet_params):
    """torch's sync detector in 'error' mode: a step may not call .item(), nonzero, boolean-mask indexing, ...; the
    survivor count is read by pnpx_live_compact (its documented, single stream synchronisation)."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
    B, H, W = 5, 32, 32
    d = synth.make_csmri_batch(B, H, W, ratio=4, sigma_n=15.0, seed=3)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    env = CSMRIEnv(None, ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params)), max_episode_step=4)
    ob = env.reset({k: g(v) for k, v in d.items()})
    acts = synth.make_actions(B, n_steps=4, pack=2)
    stops = [[0, 1, 0, 0, 1], [0, 0, 1], [1, 0], [0]]
    live = list(range(B))
    # warm-up step on a second env so that workspace growth (which synchronises once) is out of the way
    env2 = CSMRIEnv(None, env.solver, max_episode_step=4)
    env2.reset({k: g(v) for k, v in d.items()})
    env2.step({"sigma_d": g(acts[0]["sigma_d"]), "mu": g(acts[0]["mu"]), "idx_stop": g(np.zeros(B, np.int64))})
    torch.cuda.synchronize()
    for s, stop in enumerate(stops):
        n = len(live)
        action = {"sigma_d": g(acts[s]["sigma_d"][:n]), "mu": g(acts[s]["mu"][:n]), "idx_stop": g(np.array(stop, np.int64))}
        prev = torch.cuda.get_sync_debug_mode()
        torch.cuda.set_sync_debug_mode("error")
        try:
            ob, ob_masked, reward, all_done, info = env.step(action)
            pol = env.get_policy_ob(ob_masked)
        finally:
            torch.cuda.set_sync_debug_mode(prev)
        live = [r for r, st in zip(live, stop) if st == 0]
        assert env.idx_left.cpu().tolist() == live
        assert ob.variables.shape[0] == n and ob_masked.variables.shape[0] == len(live) and pol.shape[0] == len(live)
        assert isinstance(all_done, bool) and all_done == (len(live) == 0 or s == 3)
        assert reward.shape == (B, 1) and torch.isfinite(reward).all()


@pytest.mark.parametrize("task", ["pr", "spi", "ct"])
def test_other_task_envs_vs_reference_golden(unet_params, task):
    """PREnv / SPIEnv / CTEnv (tasks/{pr,spi,ct}/env.py) against the REAL reference's environments on the same inputs
    (tests/golden/env_other_tasks.npz, oracle/make_goldens.py::env_goldens): observation packing after reset and, for PR
    and SPI, two env steps in which items stop -- policy observation of the rows that were live, reward, done flags,
    written-back solver state / output and the surviving row list."""
    from tests.conftest import golden
    from tests.golden_inputs import env_case
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.pnp.solver.base import IADMMSolver
    from tfpnp_amd.tasks import pr, spi, ct
    gold = golden("env_other_tasks")
    den = UNetDenoiser2D(state_dict=unet_params)
    data, acts = env_case(task)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    rel = lambda a, b: float((a.double().cpu() - torch.from_numpy(b).double()).norm() / max(float(torch.from_numpy(b).double().norm()), 1e-30))
    env = {"pr": lambda: pr.PREnv(None, pr.IADMMSolver_PR(den), max_episode_step=3),
           "spi": lambda: spi.SPIEnv(None, spi.ADMMSolver_SPI(den), max_episode_step=3),
           "ct": lambda: ct.CTEnv(None, IADMMSolver(den), max_episode_step=3)}[task]()
    ob = env.reset({k: t(v) for k, v in data.items()})
    pob = env.get_policy_ob(ob)
    assert tuple(pob.shape) == gold[f"{task}_policy_ob_reset"].shape
    assert rel(pob, gold[f"{task}_policy_ob_reset"]) < 1e-6
    if task == "ct":
        assert rel(ob.variables, gold["ct_variables_reset"]) == 0.0
        return
    # SPI: the 10-step Poisson bisection is discontinuous (one quantum = 1.1 / 2**10 per flipped bracket), see
    # test_gpu_parity.py::test_spi_golden; PR is smooth
    tol_state, tol_ob, tol_reward = (1e-4, 1e-4, 2e-3) if task == "pr" else (5e-3, 5e-3, 5e-2)
    for s, a in enumerate(acts):
        action = {k: (torch.from_numpy(v).to(dev()) if k == "idx_stop" else t(v)) for k, v in a.items()}
        ob, ob_masked, reward, all_done, info = env.step(action)
        assert rel(env.get_policy_ob(ob), gold[f"{task}_policy_ob{s}"]) < tol_ob
        assert tuple(env.get_policy_ob(ob_masked).shape) == tuple(gold[f"{task}_policy_ob_masked_shape{s}"])
        assert np.allclose(reward.cpu().numpy(), gold[f"{task}_reward{s}"], atol=tol_reward)
        assert np.array_equal(info["done"].cpu().numpy(), gold[f"{task}_done{s}"])
        assert rel(env.state["solver"], gold[f"{task}_solver{s}"]) < tol_state
        assert rel(env.state["output"], gold[f"{task}_output{s}"]) < tol_state
        assert np.array_equal(env.idx_left.cpu().numpy(), gold[f"{task}_idx_left{s}"])
        assert not all_done


@pytest.mark.parametrize("task", ["pr", "spi", "ct"])
def test_other_task_env_forward_differentiates_through_the_fused_solver(unet_params, task, monkeypatch):
    """PnPEnv.forward (tfpnp/env/base.py:193-206) of the PR / SPI / CT environments under autograd: the delta-PSNR reward
    differentiated wrt the actions runs through the solver's fused native training path; its action gradients equal those of
    the same env with the solver's composed loop (`_forward_autograd`) put in place of `forward`."""
    from tests.golden_inputs import env_case
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks import pr, spi, ct
    den = UNetDenoiser2D(state_dict=unet_params)
    data, acts = env_case(task)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    make = {"pr": lambda: pr.PREnv(None, pr.IADMMSolver_PR(den), max_episode_step=3),
            "spi": lambda: spi.SPIEnv(None, spi.ADMMSolver_SPI(den), max_episode_step=3),
            "ct": lambda: ct.CTEnv(None, ct.IADMMSolver_CT(den), max_episode_step=3)}[task]
    if task == "ct":                               # the golden CT case is the reference's geometry (47 bins) and covers reset only:
        from tests.golden_inputs import csmri_actions   # measure with this package's own projector instead
        from tfpnp_amd import synth
        from tfpnp_amd.data.synthesis import ct_measure
        gt = t(synth.phantom_batch(3, 32, 32, 512))
        data = {k: v.cpu().numpy() for k, v in ct_measure(gt, 20).items()}
        acts = [dict(csmri_actions(3, 3, 511, ("sigma_d", "mu", "tau")))]
    keys = [k for k in acts[0] if k != "idx_stop"]

    def run(composed):
        env = make()
        sol = env.solver
        if composed:
            if task == "ct":
                monkeypatch.setattr(sol, "forward", lambda inputs, parameters, iter_num=None:
                                    sol._forward_autograd(inputs[0], inputs[1][0], *parameters, iter_num))
            else:
                monkeypatch.setattr(sol, "forward", lambda inputs, parameters, iter_num=None:
                                    sol._forward_autograd(inputs[0], *inputs[1], *parameters, iter_num))
        ob = env.reset({k: t(v) for k, v in data.items()})
        action = {k: t(acts[0][k]).clone().requires_grad_(True) for k in keys}
        _, reward = env.forward(ob, action)
        assert reward.shape[1] == 1 and torch.isfinite(reward).all()
        reward.sum().backward()
        return reward.detach(), [action[k].grad for k in keys]

    r_f, g_f = run(False)
    r_c, g_c = run(True)
    tol = 5e-2 if task == "spi" else 2e-2          # SPI: bisection flips between two forward variants (test_spi_golden)
    assert rel(r_f, r_c) < tol
    for k, a_, b_ in zip(keys, g_f, g_c):
        assert a_ is not None and torch.isfinite(a_).all(), k
        if float(b_.abs().max()) > 0:
            assert rel(a_, b_) < tol, (task, k, rel(a_, b_))
