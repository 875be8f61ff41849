"""GPU tests of the denoiser's execution options: conv kernel family, sub-batching, workspace reuse."""
import os
import warnings

import numpy as np
import pytest
import torch

from tests.golden_inputs import denoiser_inputs
from tfpnp_amd import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    return float((a - b).norm() / b.norm())


def test_conv_modes_agree_at_bench_size(unet_params):
    """Half-split f16 MFMA path vs plain fp32 MFMA path at the headline geometry (B=48, 256^2)."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    x, s = denoiser_inputs(48, 256, 256, 7)
    x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
    hs = UNetDenoiser2D(state_dict=unet_params, conv_mode=1)
    f32 = UNetDenoiser2D(state_dict=unet_params, conv_mode=0)
    a, ap = hs.forward_preclamp(x, s)
    b, bp = f32.forward_preclamp(x, s)
    assert rel(ap, bp) < 1e-5 and rel(a, b) < 1e-5
    # deterministic: a second call is bit-identical
    assert torch.equal(hs(x, s), a)


def test_subbatching_is_bit_identical(unet_params):
    from tfpnp_amd.pnp import UNetDenoiser2D
    x, s = denoiser_inputs(20, 128, 128, 8)
    x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
    den = UNetDenoiser2D(state_dict=unet_params, conv_mode=1)      # `subbatch` is an option of the half-split (fast) family
    outs = []
    ctx = den.context(dev())
    for sb in [0, 1, 3, 24]:
        ctx.set_option("subbatch", sb)
        assert ctx.get_option("subbatch") == sb
        outs.append(den(x, s).clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_set_option_switches_layout_safely(unet_params):
    """Switching the conv mode on a live context re-initialises the arena (zero borders) for the new layout."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    x, s = denoiser_inputs(3, 64, 64, 9)
    x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
    den = UNetDenoiser2D(state_dict=unet_params)
    ctx = den.context(dev())
    assert ctx.get_option("conv_mode") == 0        # r6: the fp32 family (the reference's arithmetic) is the default
    a = den(x, s)
    ctx.set_option("conv_mode", 1)                 # the fast half-split mode is one option away ...
    b = den(x, s)
    ctx.set_option("conv_mode", 0)                 # ... and back
    c = den(x, s)
    assert torch.equal(a, c) and rel(b, a) < 1e-5 and not torch.equal(a, b)
    from tfpnp_amd._lib import PnpxError
    with pytest.raises(PnpxError):
        ctx.set_option("no_such_option", 1)


def test_csmri_episode_drift_not_worse_than_fp32(unet_params):
    """30 inner iterations (6 x 5) on the EXPANSIVE (He-scaled) synthetic UNet, which amplifies fp32 round-off chaotically (x1.5-2
    per solver call): ANY two fp32-class implementations drift apart by several 1e-5 over an episode.  Three distances per seed and
    convolution family (the full 12-seed table is profiles/r5_drift_seeds.md, tools/drift_seeds.py):
      e64  = HIP vs the fp64 oracle            (the yardstick: distance to the true trajectory),
      e32  = HIP vs the fp32 CPU oracle        (the distance north_star's 1e-4 is stated on),
      ecpu = fp32 CPU oracle vs fp64           (what the reference's own arithmetic drifts by: 3.6e-5 .. 6.9e-5).
    r6: the DEFAULT family (conv_mode 0, fp32 arithmetic) must meet the stated tolerance on EVERY seed:
    e32 <= 1e-4, and e64 <= 8e-5 (table max 5.7e-5).  The opt-in fast mode (conv_mode 1, half-split f16 x 3) is held to 1e-4 on every
    seed except the two documented outliers of its table (seed 31: 1.003e-4, seed 36: 1.235e-4 -- INTEGRATION.md section 4 states this
    drift beside the mode's speed), which are capped at 1.6e-4; e64 <= 1.3e-4; and both families to e64 <= 1.5 ecpu + 3e-5 per seed.
    The non-expansive weight set below holds 1e-5 over the same 30 iterations in both."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    B, H, W = 2, 64, 64
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    acts = synth.make_actions(B)
    FAST_MODE_OUTLIERS = {31, 36}
    sols = {0: ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params)),                     # the default family
            1: ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params, conv_mode=1))}        # the fast mode
    assert sols[0].denoiser.context(dev()).get_option("conv_mode") == 0
    for seed in (31, 32, 35, 36):      # incl. the table's worst seeds of both families
        d = synth.make_csmri_batch(B, H, W, ratio=4, seed=seed)

        def run_oracle(dtype):
            den = O.Denoiser(unet_params, dtype=dtype)
            c = lambda a: t(a).to(dtype) if a.dtype != np.bool_ else t(a)
            v = O.admm_reset(c(d["x0"]))
            with torch.no_grad():
                for a in acts:
                    v = O.csmri_admm(den, v, c(d["y0"]), t(d["mask"]), c(a["sigma_d"]), c(a["mu"]))
            return O.complex2real(v[:, :1]).double()

        ref64, ref32 = run_oracle(torch.float64), run_oracle(torch.float32)
        e_cpu32 = rel(ref32, ref64)
        for mode in (0, 1):
            sol = sols[mode]
            g = lambda a: t(a).to(dev())
            v = sol.reset({"x0": g(d["x0"])})
            for a in acts:
                v = sol((v, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
            out = sol.get_output(v).double().cpu()
            e64, e32 = rel(out, ref64), rel(out, ref32)
            print(f"seed {seed} conv_mode {mode}: HIP vs fp64 {e64:.3e}  HIP vs fp32 CPU oracle {e32:.3e}  (CPU fp32 oracle vs fp64 {e_cpu32:.3e})")
            assert e64 <= 1.5 * e_cpu32 + 3e-5, (seed, mode, e64, e_cpu32)
            if mode == 0:
                assert e32 <= 1e-4 and e64 <= 8e-5, (seed, mode, e64, e32)
            else:
                assert e32 <= (1.6e-4 if seed in FAST_MODE_OUTLIERS else 1e-4) and e64 <= 1.3e-4, (seed, mode, e64, e32)


@pytest.mark.parametrize("config", ["#1: B=1 128x128", "#2: B=48 256x256 (4 items checked)"])
def test_default_scale_weights_30_iterations_within_1e5(config):
    """VERDICT r2 'what's weak' #1 / next #3b: with the He-scaled synthetic UNet the 30-iteration figures sit at 7-8e-5 of
    the 1e-4 bar only because that network is EXPANSIVE (round-off x1.5-2 per call).  On a denoiser-like, non-expansive
    weight set (PyTorch's default init scale, what SURVEY section 6 probed) both convolution families must stay within
    1e-5 of the CPU oracle after the full 6 x 5 = 30 iterations of BASELINE configs #1 and #2."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    params = synth.make_unet_params_default(0)
    if config.startswith("#1"):
        B, H, W, ncheck, seed = 1, 128, 128, 1, 99
    else:
        B, H, W, ncheck, seed = 48, 256, 256, 4, 1234
    d = synth.make_csmri_batch(B, H, W, ratio=4, sigma_n=15.0, seed=seed)
    acts = synth.make_actions(B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    torch.set_num_threads(16)
    oden = O.Denoiser(params)
    with torch.no_grad():
        v = O.admm_reset(t(d["x0"][:ncheck]))
        for a in acts:
            v = O.csmri_admm(oden, v, t(d["y0"][:ncheck]), t(d["mask"][:ncheck]), t(a["sigma_d"][:ncheck]),
                             t(a["mu"][:ncheck]))
    want = O.complex2real(v[:, :1])
    assert float(want.std()) > 0.05                      # a real image came out, not a collapsed constant
    for mode in (1, 0):
        sol = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=params, conv_mode=mode))
        g = lambda a: t(a).to(dev())
        gv = sol.reset({"x0": g(d["x0"])})
        for a in acts:
            gv = sol((gv, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
        e = rel(sol.get_output(gv)[:ncheck].cpu(), want)
        print(f"default-scale weights, config {config}, conv_mode {mode}: 30-iteration rel-L2 vs CPU oracle = {e:.2e}")
        assert e <= 1e-5


def test_fused_upsample_option_matches_separate_kernel(unet_params):
    """Option fuse_up = 1 (default since r5; producer waves of the conv kernel interpolate the full-resolution decoder entry's second source
    on the fly, conv_hs_kernel.h UPS): same arithmetic as the separate up-sampling kernel -- per call and over a
    5-iteration solver call -- at even sizes, silently the separate kernel at sizes the fused instance does not cover,
    deterministic, and within the golden tolerance of the oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    den = UNetDenoiser2D(state_dict=unet_params, conv_mode=1)
    ctx = den.context(dev())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    try:
        for (B, H, W) in [(2, 64, 64), (3, 48, 80), (2, 256, 256), (2, 50, 39), (5, 128, 96)]:
            x = torch.rand(B, 1, H, W, device=dev(), generator=torch.Generator(dev()).manual_seed(H + W))
            s = torch.full((B,), 0.08, device=dev())
            ctx.set_option("fuse_up", 0)
            ref = den(x, s).clone()
            ctx.set_option("fuse_up", 1)
            out = den(x, s).clone()
            assert rel(out, ref) < 1e-6 and torch.equal(out, den(x, s))
            # opt-in value 2: every decoder entry whose geometry allows it (64-cout tiles: 8-row instance), measured slower than the
            # separate kernel at those levels (profiles/r5_wino8.md's neighbour r5_hs_fuse_up.md) -- same bits
            ctx.set_option("fuse_up", 2)
            assert rel(den(x, s), ref) < 1e-6
        oden = O.Denoiser(unet_params)
        d = synth.make_csmri_batch(2, 64, 64, ratio=4, seed=41)
        a = synth.make_actions(2)[0]
        sol = ADMMSolver_CSMRI(den)
        v0 = sol.reset({"x0": t(d["x0"])})
        got = sol((v0, (t(d["y0"]), t(d["mask"]))), (t(a["sigma_d"]), t(a["mu"])))
        c = lambda k: torch.from_numpy(np.ascontiguousarray(d[k]))
        want = O.csmri_admm(oden, O.admm_reset(c("x0")), c("y0"), c("mask"), torch.from_numpy(a["sigma_d"]),
                            torch.from_numpy(a["mu"]))
        assert rel(got.cpu(), want) < 1e-5
    finally:
        ctx.set_option("fuse_up", 1)      # the default since r5


def _hot_params(unet_params, scale=3e3):
    """The synthetic UNet with its first convolution scaled up so that activations leave the f16 hi/lo range (|v| >= 4095)."""
    p = {k: np.array(v, copy=True) for k, v in unet_params.items()}
    for k in ("inc.conv.conv-0.conv2d.weight", "inc.conv.conv-0.conv2d.bias"):
        p[k] = (p[k] * scale).astype(np.float32)
    return p


def test_range_guard_strict_returns_exact_result(unet_params):
    """range_guard = 2: a call whose activations exceed the half-split range is repeated in exact fp32 before it
    returns -- correct output, never inf/NaN -- and the context stays in conv_mode 0 afterwards."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    x, s = denoiser_inputs(3, 64, 64, 11)
    x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
    hot = _hot_params(unet_params)
    ref_den = UNetDenoiser2D(state_dict=hot, conv_mode=0)
    ref, ref_pre = ref_den.forward_preclamp(x, s)
    assert torch.isfinite(ref_pre).all() and float(ref_pre.abs().max()) > 100.0   # the network really is "hot"
    den = UNetDenoiser2D(state_dict=hot, conv_mode=1)
    ctx = den.context(dev())
    ctx.set_option("range_guard", 2)
    out, pre = den.forward_preclamp(x, s)
    assert torch.isfinite(pre).all() and torch.equal(pre, ref_pre) and torch.equal(out, ref)
    assert ctx.get_option("conv_mode") == 0
    ctx.status()   # nothing invalid escaped: no error pending
    # a well-scaled network never trips the guard and stays on the fast path
    ok = UNetDenoiser2D(state_dict=unet_params, conv_mode=1)
    ok.context(dev()).set_option("range_guard", 2)
    a = ok(x, s)
    assert ok.context(dev()).get_option("conv_mode") == 1 and torch.isfinite(a).all()


def test_range_guard_default_reports_clean_error(unet_params):
    """range_guard = 1 (default, no synchronisation): the tripping call cannot be repaired, but the error is surfaced by
    Context.status() at the next synchronisation point, and every later call runs the exact fp32 kernels."""
    from tfpnp_amd._lib import PnpxError
    from tfpnp_amd.pnp import UNetDenoiser2D
    x, s = denoiser_inputs(3, 64, 64, 12)
    x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
    hot = _hot_params(unet_params)
    ref = UNetDenoiser2D(state_dict=hot, conv_mode=0)(x, s)
    den = UNetDenoiser2D(state_dict=hot, conv_mode=1)
    ctx = den.context(dev())
    assert ctx.get_option("range_guard") == 1
    den(x, s)
    torch.cuda.synchronize()
    with pytest.raises(PnpxError, match="range guard"):
        ctx.status()
    assert ctx.get_option("conv_mode") == 0
    out = den(x, s)                     # latched to the exact kernels: correct from here on
    assert torch.equal(out, ref) and torch.isfinite(out).all()
    with pytest.raises(PnpxError):      # the error stays visible until the guard is re-armed
        ctx.status()
    ctx.set_option("range_guard", 1)
    ctx.status()
    # the solver entries are guarded the same way
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    d = synth.make_csmri_batch(2, 64, 64, ratio=4, seed=5)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    acts = synth.make_actions(2)[0]
    sols = {}
    for mode, guard in ((0, 1), (1, 2)):
        sol = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=hot, conv_mode=mode))
        sol.denoiser.context(dev()).set_option("range_guard", guard)
        v = sol.reset({"x0": g(d["x0"])})
        sols[mode] = sol((v, (g(d["y0"]), g(d["mask"]))), (g(acts["sigma_d"]), g(acts["mu"])))
    assert torch.isfinite(sols[1]).all() and torch.equal(sols[0], sols[1])


def test_env_step_redoes_a_tripped_step_in_exact_fp32(unet_params):
    """with the default (non-synchronising) range guard a tripped call used to return invalid state and reward
    from PnPEnv.step without anyone noticing.  The step now looks at the guard right after its one host read and, if it
    tripped, repeats itself with the exact-fp32 convolutions: state, reward and observation equal a conv_mode-0 run."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
    hot = _hot_params(unet_params)
    d = synth.make_csmri_batch(3, 64, 64, ratio=4, seed=8)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    data = {k: g(v) for k, v in d.items()}
    acts = [{k: g(v) for k, v in a.items()} for a in synth.make_actions(3)[:2]]
    # item 1 stops IN the tripped step (the repeated step must still report the rows that were live during it)
    acts[0]["idx_stop"] = torch.tensor([0, 1, 0], dtype=torch.int64, device=dev())
    acts[1] = {k: (v[:2] if k != "idx_stop" else v) for k, v in acts[1].items()}
    acts[1]["idx_stop"] = torch.zeros(2, dtype=torch.int64, device=dev())
    got = {}
    for mode in (0, 1):
        den = UNetDenoiser2D(state_dict=hot, conv_mode=mode)
        env = CSMRIEnv(None, ADMMSolver_CSMRI(den), max_episode_step=6)
        env.reset(data)
        rows = []
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for i, a in enumerate(acts):
                ob, ob_masked, reward, _, info = env.step(a)
                # ob: the rows live DURING the step (= the action batch); reward: every item of the batch (reference: whole state)
                assert ob.variables.shape[0] == info["done"].shape[0] == a["sigma_d"].shape[0] == (3, 2)[i]
                assert reward.shape[0] == 3 and ob_masked.variables.shape[0] == 2
                rows.append((ob.variables.clone(), reward.clone()))
        got[mode] = (rows, env.range_redone_steps, [str(x.message) for x in w], den.context(dev()).get_option("conv_mode"))
    assert got[0][1] == 0 and got[1][1] == 1                 # the first hot step tripped, was redone; the second ran exact
    assert any("range guard" in m for m in got[1][2]) and got[1][3] == 0
    for (v0, r0), (v1, r1) in zip(got[0][0], got[1][0]):
        assert torch.isfinite(v1).all() and torch.equal(v0, v1) and torch.equal(r0, r1)


def test_fold_first_option_is_bit_identical(unet_params):
    """Option fold_first = 1 (conv_hs WREG == 2: the network's first convolution evaluated inside the tile loader of the
    second one, its output tensor never written): same arithmetic in the same order -- bit-identical outputs at the bench
    geometry and on a ragged one; geometries the instance does not cover silently take the separate kernels."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    den = UNetDenoiser2D(state_dict=unet_params, conv_mode=1)
    ctx = den.context(dev())
    try:
        for (B, H, W) in [(24, 256, 256), (9, 160, 224), (2, 64, 64), (3, 50, 39)]:
            x, s = denoiser_inputs(B, H, W, 77)
            x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
            ctx.set_option("fold_first", 0)
            ref = den.forward_preclamp(x, s)[1].clone()
            ctx.set_option("fold_first", 1)
            assert torch.equal(den.forward_preclamp(x, s)[1], ref), (B, H, W)
    finally:
        ctx.set_option("fold_first", 0)


def test_round3_execution_options_are_bit_identical(unet_params):
    """wreg (weights-in-registers instances), chains (independent launch chains over slices of the batch), fft_affine /
    fft_tile (FFT pass mapping and tile size) only change HOW the work is scheduled: outputs must not move by a bit."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    den = UNetDenoiser2D(state_dict=unet_params, conv_mode=1)
    ctx = den.context(dev())
    defaults = {k: ctx.get_option(k) for k in ("wreg", "chains", "fft_affine", "fft_tile")}
    try:
        for (B, H, W) in [(24, 256, 256), (7, 128, 160), (6, 256, 256), (1, 64, 48), (3, 32, 32)]:
            x, s = denoiser_inputs(B, H, W, 91)
            x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
            ref = None
            for wreg in (2, 0, 1):
                for chains in (1, 2, 3):
                    ctx.set_option("wreg", wreg)
                    ctx.set_option("chains", chains)
                    pre = den.forward_preclamp(x, s)[1]
                    ref = pre.clone() if ref is None else ref
                    assert torch.equal(pre, ref), (B, H, W, wreg, chains)
        ctx.set_option("wreg", defaults["wreg"])
        ctx.set_option("chains", defaults["chains"])
        d = synth.make_csmri_batch(9, 256, 256, ratio=4, seed=17)      # 9 images: one full XCD group + a plain-mapped image
        g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
        a = synth.make_actions(9)[0]
        sol = ADMMSolver_CSMRI(den)
        v0 = sol.reset({"x0": g(d["x0"])})
        ref = None
        for aff in (1, 0):
            for tile in (0, 512, 2048, 4096):
                ctx.set_option("fft_affine", aff)
                ctx.set_option("fft_tile", tile)
                out = sol((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
                ref = out.clone() if ref is None else ref
                assert torch.equal(out, ref), (aff, tile)
    finally:
        for k, v in defaults.items():
            ctx.set_option(k, v)


def test_fp32_winograd_layers_match_direct_kernel_and_oracle(unet_params):
    """conv_mode 0 runs its layers with cout % 32 == 0 as Winograd F(2x2,3x3) on the fp32 MFMA (conv3x3_wino.hip, option
    fp32_winograd, default on; two tile shapes: 64 couts / W % 16 and 32 couts / W % 32 -- the sizes below exercise both, mixed
    with per-layer fall-backs).  Same network as the direct kernel (tfpnp/pnp/denoiser/models/unet.py:8-31) in a different summation order:
    both must sit within fp32 rounding of the fp64 oracle, and of each other; geometries the Winograd kernel does not take
    (sizes not divisible by 16 at some level) fall back per layer and stay bit-identical when every layer falls back."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.pnp import UNetDenoiser2D
    den = UNetDenoiser2D(state_dict=unet_params, conv_mode=0)
    ctx = den.context(dev())
    assert ctx.get_option("fp32_winograd") == 1
    all_layers = (1 << 27) - 1
    assert ctx.get_option("fp32_wino8_layers") == all_layers      # default: every Winograd layer on the 8-wave kernel
    p64 = {k: torch.as_tensor(v).double() for k, v in unet_params.items()}
    for B, H, W in [(2, 64, 64), (1, 128, 96), (3, 32, 64), (1, 32, 32), (5, 16, 32), (1, 48, 64)]:   # incl. one-region grids
        x, s = denoiser_inputs(B, H, W, 11 + B)
        xt, st = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
        ctx.set_option("fp32_winograd", 0)
        _, direct = den.forward_preclamp(xt, st)
        direct = direct.double().cpu()
        ctx.set_option("fp32_winograd", 1)
        _, wino = den.forward_preclamp(xt, st)
        wino = wino.double().cpu()
        # the 4-wave kernel (conv3x3_wino.hip) on the same layers: same algebra, the output transform summed in the other order
        ctx.set_option("fp32_wino8_layers", 0)
        _, wino4 = den.forward_preclamp(xt, st)
        wino4 = wino4.double().cpu()
        ctx.set_option("fp32_wino8_layers", all_layers)
        with torch.no_grad():
            sig = torch.from_numpy(s).double().view(B, 1, 1, 1).expand(B, 1, H, W)
            ref = O.unet_forward(torch.cat([torch.from_numpy(x).double(), sig], 1), p64)
        assert not torch.equal(wino, direct), "the Winograd layers did not run"
        assert rel(direct, ref) < 3e-6 and rel(wino, ref) < 3e-6, (rel(direct, ref), rel(wino, ref))
        assert rel(wino, direct) < 4e-6
        assert not torch.equal(wino, wino4), "the 8-wave kernel did not run"
        assert rel(wino4, ref) < 3e-6 and rel(wino, wino4) < 3e-6, (rel(wino4, ref), rel(wino, wino4))
    # 24 x 24: no level is a multiple of 16, every layer falls back
    x, s = denoiser_inputs(2, 24, 24, 5)
    xt, st = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
    ctx.set_option("fp32_winograd", 0)
    a = den.forward_preclamp(xt, st)[1].clone()
    ctx.set_option("fp32_winograd", 1)
    b = den.forward_preclamp(xt, st)[1]
    assert torch.equal(a, b)


def test_fp32_decoder_entries_upsample_inside_the_kernel(unet_params):
    """r5: in conv_mode 0 the decoder-entry convolutions interpolate the bilinear x2 (align_corners) up-sampling of their second
    source themselves (conv3x3_wino8.hip UPS instances, option fp32_fuse_up, default on): the up-sampled tensor of
    tfpnp/pnp/denoiser/models/unet.py:92-121 never exists.  Same interpolation arithmetic as the separate kernel (unet.hip
    upsample2x_v4_kernel), so the network output must agree with the un-fused path to rounding noise at most, and with the fp64
    oracle as before.  Sizes: both tile shapes fused (256^2: all four entries), partly fused (64^2, 128 x 96: small levels fall back)."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.pnp import UNetDenoiser2D
    den = UNetDenoiser2D(state_dict=unet_params, conv_mode=0)
    ctx = den.context(dev())
    assert ctx.get_option("fp32_fuse_up") == 1
    p64 = {k: torch.as_tensor(v).double() for k, v in unet_params.items()}
    try:
        for B, H, W in [(2, 256, 256), (3, 64, 64), (1, 128, 96), (1, 128, 256)]:
            x, s = denoiser_inputs(B, H, W, 40 + B)
            xt, st = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
            ctx.set_option("fp32_fuse_up", 1)
            post, fused = den.forward_preclamp(xt, st)
            assert torch.equal(den.forward_preclamp(xt, st)[1], fused)          # deterministic
            assert torch.equal(post, fused.clamp(0, 1))
            ctx.set_option("fp32_fuse_up", 0)
            _, plain = den.forward_preclamp(xt, st)
            fused, plain = fused.double().cpu(), plain.double().cpu()
            with torch.no_grad():
                sig = torch.from_numpy(s).double().view(B, 1, 1, 1).expand(B, 1, H, W)
                ref = O.unet_forward(torch.cat([torch.from_numpy(x).double(), sig], 1), p64)
            print(f"{B}x{H}x{W}: fused vs separate up-sampling {rel(fused, plain):.2e}; vs fp64 oracle fused {rel(fused, ref):.2e} separate {rel(plain, ref):.2e}")
            assert rel(fused, plain) < 1e-6 and rel(fused, ref) < 3e-6
    finally:
        ctx.set_option("fp32_fuse_up", 1)


def test_fp32_launch_chains_are_bit_identical(unet_params):
    """r5: the fp32 family's forward (and the VJP's adjoint chain) as launch chains over slices of the batch (option fp32_chains: 0 one chain, 1 only the bottom level
    forks, n >= 2 whole-forward chains; default 2).  Per-image results must not depend on the slicing, for odd batch sizes, batches
    smaller than the chain count, sizes where only some layers run on the Winograd kernels, and inside a solver call."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    den = UNetDenoiser2D(state_dict=unet_params, conv_mode=0)
    ctx = den.context(dev())
    assert ctx.get_option("fp32_chains") == 2
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    try:
        for (B, H, W) in [(5, 64, 64), (1, 128, 96), (9, 32, 48), (7, 256, 256), (3, 50, 39)]:
            x, s = denoiser_inputs(B, H, W, 500 + B)
            x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
            ctx.set_option("fp32_chains", 0)
            ref = den.forward_preclamp(x, s)[1].clone()
            for c in (1, 2, 3):
                ctx.set_option("fp32_chains", c)
                assert torch.equal(den.forward_preclamp(x, s)[1], ref), (B, H, W, c)
            # the VJP's adjoint chain is sliced the same way (two chains when fp32_chains >= 2)
            from tfpnp_amd import ops
            g = torch.from_numpy(np.random.default_rng(B).standard_normal((B, 1, H, W)).astype(np.float32)).to(dev())
            ctx.set_option("fp32_chains", 0)
            gx0, gs0 = [v.clone() for v in ops.unet_denoise_backward(ctx, x, s.reshape(-1), g)]
            ctx.set_option("fp32_chains", 2)
            gx2, gs2 = ops.unet_denoise_backward(ctx, x, s.reshape(-1), g)
            assert torch.equal(gx0, gx2) and torch.equal(gs0, gs2), (B, H, W)
        # r5: the half-split family's VJP slices its adjoint chain the same way (option chains; the gradient scale is the whole batch's)
        hden = UNetDenoiser2D(state_dict=unet_params, conv_mode=1)
        hctx = hden.context(dev())
        for (B, H, W) in [(5, 64, 64), (7, 128, 96), (3, 50, 39)]:
            x, s = denoiser_inputs(B, H, W, 600 + B)
            x, s = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
            g = torch.from_numpy(np.random.default_rng(B).standard_normal((B, 1, H, W)).astype(np.float32)).to(dev())
            res = []
            for c in (1, 2, 3):
                hctx.set_option("chains", c)
                res.append([v.clone() for v in ops.unet_denoise_backward(hctx, x, s.reshape(-1), g)] + [hden(x, s).clone()])
            hctx.set_option("chains", 0)
            for r in res[1:]:
                assert all(torch.equal(u, v) for u, v in zip(r, res[0])), (B, H, W)
        d = synth.make_csmri_batch(5, 64, 64, ratio=4, seed=43)
        a = synth.make_actions(5)[0]
        sol = ADMMSolver_CSMRI(den)
        v0 = sol.reset({"x0": t(d["x0"])})
        outs = []
        for c in (0, 2):
            ctx.set_option("fp32_chains", c)
            outs.append(sol((v0, (t(d["y0"]), t(d["mask"]))), (t(a["sigma_d"]), t(a["mu"]))).clone())
        assert torch.equal(outs[0], outs[1])
    finally:
        ctx.set_option("fp32_chains", 2)


def test_fp32_ksplit_is_deterministic_and_batch_independent(unet_params):
    """r6: the layers of the two deepest levels split their input-channel chunks over 2 / 4 workgroups of the 8-wave
    Winograd kernel (conv3x3_wino8.hip KSPLIT: every piece writes its output-transformed partial sums to a scratch slab, a second small
    launch adds the pieces in piece order, then bias + activation).  Option fp32_ksplit: 2 = at every batch size -- per-image results
    bit-identical across batch sizes and launch chains; 1 (default) = only in calls whose unsplit tiles cannot fill the chip -- a small
    call reproduces the always-split bits, a large one the unsplit bits; 0 = never.  All within fp32 rounding of the fp64 oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.pnp import UNetDenoiser2D
    den = UNetDenoiser2D(state_dict=unet_params)
    ctx = den.context(dev())
    assert ctx.get_option("fp32_ksplit") == 1 and ctx.get_option("conv_mode") == 0
    p64 = {k: torch.as_tensor(v).double() for k, v in unet_params.items()}
    try:
        for (B, H, W) in [(7, 128, 128), (3, 256, 256), (5, 64, 64)]:
            x, s = denoiser_inputs(B, H, W, 700 + B)
            xt, st = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
            outs = {}
            for k in (0, 2, 1):
                ctx.set_option("fp32_ksplit", k)
                outs[k] = den.forward_preclamp(xt, st)[1].clone()
                for _ in range(3):                                           # which workgroup runs which piece varies; the bits must not
                    assert torch.equal(den.forward_preclamp(xt, st)[1], outs[k]), (B, H, W, k)
            assert not torch.equal(outs[2], outs[0]), "the K-split instances did not run"
            assert torch.equal(outs[1], outs[2]), "a small call must take the split path"
            with torch.no_grad():
                sig = torch.from_numpy(s).double().view(B, 1, 1, 1).expand(B, 1, H, W)
                ref = O.unet_forward(torch.cat([torch.from_numpy(x).double(), sig], 1), p64)
            assert rel(outs[2].cpu().double(), ref) < 3e-6 and rel(outs[0].cpu().double(), ref) < 3e-6 and rel(outs[2], outs[0]) < 3e-6
            ctx.set_option("fp32_ksplit", 2)
            for chains in (0, 2, 3):
                ctx.set_option("fp32_chains", chains)
                for n in (1, 2, B - 1):
                    part = den.forward_preclamp(xt[:n].contiguous(), st[:n].contiguous())[1]
                    assert torch.equal(part, outs[2][:n]), (B, H, W, chains, n)
            ctx.set_option("fp32_chains", 2)
        # a call large enough that no layer is short of tiles (128 x 128: >= 64 images) takes the unsplit path in the default mode
        x, s = denoiser_inputs(64, 128, 128, 9)
        xt, st = torch.from_numpy(x).to(dev()), torch.from_numpy(s).to(dev())
        ctx.set_option("fp32_ksplit", 0)
        a = den(xt, st).clone()
        ctx.set_option("fp32_ksplit", 1)
        assert torch.equal(den(xt, st), a)
        ctx.set_option("fp32_ksplit", 2)
        b = den(xt, st)
        assert not torch.equal(b, a) and rel(b, a) < 3e-6
    finally:
        ctx.set_option("fp32_ksplit", 1)
        ctx.set_option("fp32_chains", 2)


@pytest.mark.parametrize("mode", [1, 0])
def test_solver_call_replays_from_a_hip_graph(unet_params, mode):
    """A solver call launches on the caller's stream (+ side streams forked and joined by events for the launch chains) and, once its
    arenas exist, neither allocates nor synchronises: it can be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed,
    bit-identical to the eager call.  (No speed-up at the bench's batch sizes -- the iteration is GPU-bound down to B = 6,
    tools/graph_capture.py -- but integrators that capture their whole episode need the property.)"""
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    sol = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params, conv_mode=mode))
    d = synth.make_csmri_batch(5, 64, 64, ratio=4, seed=44)
    a = synth.make_actions(5)[0]
    v0 = sol.reset({"x0": t(d["x0"])})
    aux, par = (t(d["y0"]), t(d["mask"])), (t(a["sigma_d"]), t(a["mu"]))
    ref = sol((v0, aux), par).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm-up on the capture stream's pool, as torch.cuda.graph asks
        sol((v0, aux), par)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = sol((v0, aux), par)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
