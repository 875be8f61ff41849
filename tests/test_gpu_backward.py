"""GPU tests of the training path: the native denoiser VJP (pnpx_unet_denoise_backward) and the differentiable
solver iterations, against torch.autograd through the CPU oracle.

Yardstick: LeakyReLU / max-pool / clamp have discontinuous derivatives, so two correct fp32 evaluations that round
differently disagree at the few elements whose pre-activation sits within round-off of a kink.  The tolerance is
therefore tied to what the CPU oracle itself shows between fp32 and fp64 (a structural error is O(1))."""
import numpy as np
import pytest
import torch

from tests.golden_inputs import denoiser_inputs, csmri_actions, _KinkProbe, solver_kink_margin, GRAD_CASE
from tfpnp_amd import synth

pytestmark = pytest.mark.gpu

from tests import composed_solvers   # the reference's loops from differentiable ops (test infrastructure)
composed_solvers.install()


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def dev():
    return torch.device("cuda:0")


def g(a, grad=False):
    x = t(a).to(dev())
    return x.requires_grad_(True) if grad else x


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module", params=[1, 0], ids=["hs_f16x3", "f32_mfma"])
def den(unet_params, request):
    """Both kernel families: the backward re-computes the forward with the context's own family."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    return UNetDenoiser2D(state_dict=unet_params, conv_mode=request.param)


@pytest.fixture(scope="module")
def oden64(unet_params):
    from oracle import pnp_oracle as O
    return O.Denoiser(unet_params, dtype=torch.float64)


@pytest.fixture(scope="module")
def oden32(unet_params):
    from oracle import pnp_oracle as O
    return O.Denoiser(unet_params)


def oracle_grads(fn, inputs, wts, dtype):
    leaves = [t(a).to(dtype).requires_grad_(True) for a in inputs]
    out = fn(*leaves)
    (out * t(wts).to(dtype)).sum().backward()
    return out.detach(), [l.grad for l in leaves]


def kink_margin(oden64, x, s):
    from oracle import pnp_oracle as O
    probe, keep = _KinkProbe(), O.F
    O.F = probe
    try:
        with torch.no_grad():
            xt, st = t(x).double(), t(s).double()
            N, _, H, W = xt.shape
            pre = O.unet_forward(torch.cat([xt, torch.ones(N, 1, H, W, dtype=xt.dtype) * st.view(N, 1, 1, 1)], 1),
                                 oden64.params)
    finally:
        O.F = keep
    return min(probe.margin, float(pre.abs().min()), float((pre - 1).abs().min()))


def kink_free_inputs(oden64, B, H, W, seed, tau=1e-5, tries=200):
    """Seeded inputs at which the denoiser is differentiable with a safety margin: no activation, pooling decision or
    clamp within `tau` (relative) of its kink in the fp64 oracle.  The derivative there does not depend on fp32
    round-off, so the VJP can be compared tightly."""
    for k in range(tries):
        x, s = denoiser_inputs(B, H, W, seed + 1000 * k)
        if kink_margin(oden64, x, s) > tau:
            return x, s, k
    raise AssertionError("no kink-free input found")


@pytest.mark.parametrize("B,H,W,seed", [(1, 16, 16, 61), (2, 16, 16, 62), (1, 17, 23, 63), (1, 24, 40, 64), (2, 19, 33, 65),
                                         (1, 32, 32, 66)])
def test_denoiser_vjp_tight_at_kink_free_inputs(den, oden64, B, H, W, seed):
    x, s, k = kink_free_inputs(oden64, B, H, W, seed)
    wts = np.random.RandomState(seed).standard_normal((B, 1, H, W)).astype(np.float32)
    _, (gx64, gs64) = oracle_grads(lambda a, b: oden64(a, b), (x, s), wts, torch.float64)
    xd, sd = g(x, True), g(s, True)
    out = den(xd, sd)
    assert out.requires_grad
    (out * g(wts)).sum().backward()
    ex, es = rel(xd.grad, gx64), rel(sd.grad, gs64)
    print(f"vjp {B}x{H}x{W} (try {k}): grad_x {ex:.2e}  grad_sigma {es:.2e}")
    assert ex < 2e-5 and es < 5e-5


@pytest.mark.parametrize("B,H,W,seed", [(2, 48, 80, 62), (1, 50, 39, 63), (3, 64, 64, 64), (1, 128, 128, 65)])
def test_denoiser_vjp_vs_oracle_autograd(den, oden32, oden64, B, H, W, seed):
    """Arbitrary inputs: a handful of the ~1e6 activations sit within fp32 round-off of a LeakyReLU / pooling kink, and
    each such element legitimately takes either one-sided derivative (observed effect 1e-3 .. 5e-3 relative)."""
    x, s = denoiser_inputs(B, H, W, seed)
    wts = np.random.RandomState(seed).standard_normal((B, 1, H, W)).astype(np.float32)
    _, (gx64, gs64) = oracle_grads(lambda a, b: oden64(a, b), (x, s), wts, torch.float64)
    xd, sd = g(x, True), g(s, True)
    (den(xd, sd) * g(wts)).sum().backward()
    ex, es = rel(xd.grad, gx64), rel(sd.grad, gs64)
    print(f"vjp {B}x{H}x{W}: grad_x {ex:.2e}  grad_sigma {es:.2e}  kink margin {kink_margin(oden64, x, s):.1e}")
    assert ex < 2e-2 and es < 2e-2


def test_denoiser_vjp_is_linear_and_batch_independent(den):
    from tfpnp_amd import ops
    x, s = denoiser_inputs(3, 48, 48, 66)
    rs = np.random.RandomState(66)
    g1, g2 = (rs.standard_normal((3, 1, 48, 48)).astype(np.float32) for _ in range(2))
    ctx = den.context(dev())
    a, sa = ops.unet_denoise_backward(ctx, g(x), g(s), g(g1))
    b, sb = ops.unet_denoise_backward(ctx, g(x), g(s), g(g2))
    c, sc = ops.unet_denoise_backward(ctx, g(x), g(s), g(2 * g1 - 3 * g2))
    # grad_sigma is a signed sum over all pixels (cancellation): its round-off is relative to sum |terms|, not to the sum
    assert rel(c, 2 * a - 3 * b) < 1e-5 and rel(sc, 2 * sa - 3 * sb) < 1e-4
    one, sone = ops.unet_denoise_backward(ctx, g(x[1:2]), g(s[1:2]), g(g1[1:2]))
    assert rel(one, a[1:2]) < 2e-6 and rel(sone, sa[1:2]) < 1e-4
    # the forward pass in between still works and is unchanged by the backward workspaces
    ref = den(g(x), g(s)).clone()
    ops.unet_denoise_backward(ctx, g(x), g(s), g(g1))
    assert torch.equal(den(g(x), g(s)), ref)


def _check(names, got, want64, want32, floor=2e-2):   # floor: see test_denoiser_vjp_vs_oracle_autograd
    for n, a, b64, b32 in zip(names, got, want64, want32):
        e, y = rel(a, b64), rel(b32, b64)
        print(f"  d/d{n}: {e:.2e} (cpu fp32 {y:.2e})")
        assert e < max(3 * y, floor), n


@pytest.mark.parametrize("name,keys", [("admm", ("sigma_d", "mu")), ("hqs", ("sigma_d", "mu")),
                                       ("pg", ("sigma_d", "tau")), ("apg", ("sigma_d", "tau", "beta")),
                                       ("redadmm", ("sigma_d", "mu", "lamda"))])
def test_csmri_solver_gradients(den, oden32, oden64, name, keys):
    """PnPEnv.forward under autograd (tfpnp/env/base.py:193-206): d loss / d (state, policy actions).

    The derivative of a T-iteration solve jumps whenever ANY LeakyReLU / pooling / clamp decision of ANY denoiser call
    flips, and one flip in the 2x2 bottleneck of a 32x32 image moves these gradients by several per cent (measured on one
    input: 1e-3 with the MFMA first convolution, 5e-2 with the exact-fp32 one, the CPU fp32 oracle itself 1e-3 from
    fp64).  So (a) the tight comparison is made on seeded inputs whose whole fp64 trajectory keeps a safety margin from
    every kink (found within a few tries at B=1, 16x16, T=2); (b) arbitrary inputs (B=2, 32x32, T=3) check the forward
    values tightly and the gradients against a structural-error bound (a wrong VJP is O(1))."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import csmri
    sol = {"admm": csmri.ADMMSolver_CSMRI, "hqs": csmri.HQSSolver_CSMRI, "pg": csmri.PGSolver_CSMRI,
           "apg": csmri.APGSolver_CSMRI, "redadmm": csmri.REDADMMSolver_CSMRI}[name](den)
    ofn = getattr(O, "csmri_" + name)

    def case(B, H, W, T, k):
        d = synth.make_csmri_batch(B, H, W, seed=71 + 100 * k)
        a = csmri_actions(B, T, 72 + 100 * k, keys)
        if "beta" in a:
            a["beta"] = (0.3 * a["beta"]).astype(np.float32)
        return d, [a[key] for key in keys], sol.reset({"x0": g(d["x0"])}).cpu().numpy()

    def compare(d, acts, v0, floor):
        wts = np.random.RandomState(73).standard_normal(v0.shape).astype(np.float32)

        def run_oracle(oden, dtype):
            y0, m = t(d["y0"]).to(dtype), t(d["mask"])
            return oracle_grads(lambda v, *p: ofn(oden, v, y0, m, *p), [v0] + acts, wts, dtype)

        out64, g64 = run_oracle(oden64, torch.float64)
        _, g32 = run_oracle(oden32, torch.float32)
        leaves = [g(v0, True)] + [g(p, True) for p in acts]
        out = sol((leaves[0], (g(d["y0"]), g(d["mask"]))), tuple(leaves[1:]))
        assert rel(out, out64) < 1e-4
        with torch.no_grad():      # same values as the fused inference loop
            fused = sol((g(v0), (g(d["y0"]), g(d["mask"]))), tuple(g(p) for p in acts))
        assert rel(out, fused) < 1e-5
        (out * g(wts)).sum().backward()
        _check(["variables"] + list(keys), [l.grad for l in leaves], g64, g32, floor=floor)

    # (a) kink-free, tight
    for k in range(60):
        d, acts, v0 = case(1, 16, 16, 2, k)
        margin = solver_kink_margin(lambda: ofn(oden64, t(v0).double(), t(d["y0"]).double(), t(d["mask"]),
                                               *[t(p).double() for p in acts]))
        if margin > 1e-5:
            break
    else:
        raise AssertionError("no kink-free CS-MRI case found")
    print(f"  {name}: kink-free case at try {k} (margin {margin:.1e})")
    compare(d, acts, v0, floor=1e-4)
    # (b) arbitrary inputs
    d, acts, v0 = case(2, 32, 32, 3, 0)
    compare(d, acts, v0, floor=0.25)
    if name != "admm":    # (c) the same arbitrary case against the REAL reference's autograd (tests/golden/solver_grads.npz)
        from tests.conftest import golden
        gold = golden("solver_grads")
        leaves = [g(v0, True)] + [g(p, True) for p in acts]
        out = sol((leaves[0], (g(d["y0"]), g(d["mask"]))), tuple(leaves[1:]))
        (out * g(np.random.RandomState(73).standard_normal(v0.shape).astype(np.float32))).sum().backward()
        assert rel(out, t(gold[name + "_out"])) < 1e-4
        for key, leaf in zip(("variables",) + tuple(keys), leaves):
            e = rel(leaf.grad, t(gold[f"{name}_grad_{key}"]))
            print(f"  {name} vs reference autograd d/d{key}: {e:.2e}")
            assert e < 0.25, key          # kink-flip bound of (b); a structural error is O(1)


def test_training_path_vs_reference_autograd_golden(den):
    """The native training path against gradients computed by the REAL reference under torch.autograd
    (tests/golden/solver_grads.npz; generator: oracle/make_goldens.py::gradient_goldens): the fused ADMM VJP on the
    kink-free case, and d reward / d policy logits through PnPEnv.forward."""
    from tests.conftest import golden
    from tfpnp_amd.tasks import csmri
    C, gold = GRAD_CASE, golden("solver_grads")
    sol = csmri.ADMMSolver_CSMRI(den)
    d = synth.make_csmri_batch(C.B, C.H, C.W, seed=C.data_seed)
    a = csmri_actions(C.B, C.T, C.action_seed, ("sigma_d", "mu"))
    v0 = sol.reset({"x0": g(d["x0"])})
    wts = np.random.RandomState(C.wts_seed).standard_normal(tuple(v0.shape)).astype(np.float32)
    leaves = [v0.clone().requires_grad_(True), g(a["sigma_d"], True), g(a["mu"], True)]
    out = sol((leaves[0], (g(d["y0"]), g(d["mask"]))), tuple(leaves[1:]))
    (out * g(wts)).sum().backward()
    assert rel(out, t(gold["admm_out"])) < 1e-5
    for leaf, key in zip(leaves, ("admm_grad_variables", "admm_grad_sigma_d", "admm_grad_mu")):
        e = rel(leaf.grad, t(gold[key]))
        print(f"  vs reference autograd {key}: {e:.2e}")
        assert e < 1e-4, key
    d2 = synth.make_csmri_batch(C.env_B, C.env_H, C.env_W, seed=C.env_data_seed)
    env = csmri.CSMRIEnv(None, sol, max_episode_step=6)
    ob = env.reset({k: g(v) for k, v in d2.items() if isinstance(v, np.ndarray)})
    raw = g(np.random.RandomState(C.env_raw_seed).standard_normal((C.env_B, 10)).astype(np.float32), True)
    _, reward = env.forward(ob, {"sigma_d": torch.sigmoid(raw[:, :5]) * 70 / 255, "mu": torch.sigmoid(raw[:, 5:])})
    reward.sum().backward()
    assert np.allclose(reward.detach().cpu().numpy(), gold["env_reward"], atol=5e-4)
    e = rel(raw.grad, t(gold["env_grad_raw"]))
    print(f"  vs reference autograd d reward / d logits: {e:.2e}")
    assert e < 5e-3


def test_csmri_admm_fused_vjp_vs_composed_autograd(den):
    """pnpx_csmri_admm_backward (one native call: masked-FFT adjoints, d/d mu reduction, denoiser VJPs) against the
    same loop composed from differentiable building blocks (native denoiser VJP + unitary FFT ops + PyTorch pointwise
    autograd, ADMMSolver_CSMRI._forward_autograd) -- both run the same denoiser kernels, so they agree to rounding
    unless a kink decision flips on the 1e-7 difference of their forward values (bounded below); also the
    iter_num < action_pack case: the unused hyper-parameter columns get exactly zero gradient, and determinism."""
    from tfpnp_amd.tasks import csmri
    sol = csmri.ADMMSolver_CSMRI(den)
    B, H, W, T = 3, 64, 48, 4
    d = synth.make_csmri_batch(B, H, W, seed=171)
    a = csmri_actions(B, T + 1, 172, ("sigma_d", "mu"))
    v0 = sol.reset({"x0": g(d["x0"])})
    v0 = v0 + 0.05 * torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(3))
    wts = torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(4))
    y0, m = g(d["y0"]), g(d["mask"])

    def grads(fn):
        leaves = [v0.clone().requires_grad_(True), g(a["sigma_d"], True), g(a["mu"], True)]
        out = fn(*leaves)
        (out * wts).sum().backward()
        return out.detach(), [l.grad for l in leaves]

    out_f, gf = grads(lambda v, s_, mu: sol((v, (y0, m)), (s_, mu), iter_num=T))
    out_c, gc = grads(lambda v, s_, mu: sol._forward_autograd(v, y0, m, s_, mu, T))
    with torch.no_grad():     # training forward == inference forward up to the un-fused network tail's summation order
        assert rel(out_f, sol((v0, (y0, m)), (g(a["sigma_d"]), g(a["mu"])), iter_num=T)) < 1e-6
    assert rel(out_f, out_c) < 1e-5
    for n, x, y in zip(("variables", "sigma_d", "mu"), gf, gc):
        print(f"  fused vs composed d/d{n}: {rel(x, y):.2e}")
        assert rel(x, y) < 2e-2, n
        assert x.shape == y.shape
    assert float(gf[0][:, 0].abs().max()) == 0.0             # x of the incoming state is never read by an iteration
    assert float(gf[1][:, T:].abs().max()) == 0.0 and float(gf[2][:, T:].abs().max()) == 0.0
    assert float(gf[1][:, :T].abs().min()) > 0 and float(gf[2][:, :T].abs().min()) > 0
    _, gf2 = grads(lambda v, s_, mu: sol((v, (y0, m)), (s_, mu), iter_num=T))
    assert all(torch.equal(x, y) for x, y in zip(gf, gf2))   # deterministic reductions


def test_csmri_hqs_fused_vjp_vs_composed_autograd(den):
    """HQSSolver_CSMRI under autograd runs ONE native call each way as well (pnpx_csmri_hqs_train / _backward): same
    checks as for ADMM against the composed path (HQSSolver_CSMRI._forward_autograd)."""
    from tfpnp_amd.tasks import csmri
    sol = csmri.HQSSolver_CSMRI(den)
    B, H, W, T = 3, 48, 64, 3
    d = synth.make_csmri_batch(B, H, W, seed=171)
    a = csmri_actions(B, 5, 172, ("sigma_d", "mu"))
    v0 = sol.reset({"x0": g(d["x0"])})
    v0 = v0 + 0.05 * torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(3))
    wts = torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(4))
    y0, m = g(d["y0"]), g(d["mask"])

    def grads(fn):
        leaves = [v0.clone().requires_grad_(True), g(a["sigma_d"], True), g(a["mu"], True)]
        out = fn(*leaves)
        (out * wts).sum().backward()
        return out.detach(), [l.grad for l in leaves]

    out_f, gf = grads(lambda v, s_, mu: sol((v, (y0, m)), (s_, mu), iter_num=T))
    out_c, gc = grads(lambda v, s_, mu: sol._forward_autograd(v, y0, m, s_, mu, T))
    with torch.no_grad():
        assert rel(out_f, sol((v0, (y0, m)), (g(a["sigma_d"]), g(a["mu"])), iter_num=T)) < 1e-6
    assert rel(out_f, out_c) < 1e-5
    for n, x, y in zip(("variables", "sigma_d", "mu"), gf, gc):
        print(f"  HQS fused vs composed d/d{n}: {rel(x, y):.2e}")
        assert rel(x, y) < 2e-2 and x.shape == y.shape, n
    assert float(gf[0][:, 0].abs().max()) == 0.0             # x of the incoming state is never read by an iteration
    assert float(gf[1][:, T:].abs().max()) == 0.0 and float(gf[2][:, T:].abs().max()) == 0.0
    assert float(gf[1][:, :T].abs().min()) > 0 and float(gf[2][:, :T].abs().min()) > 0
    _, gf2 = grads(lambda v, s_, mu: sol((v, (y0, m)), (s_, mu), iter_num=T))
    assert all(torch.equal(x, y) for x, y in zip(gf, gf2))   # deterministic reductions
    out0 = sol((v0.clone().requires_grad_(True), (y0, m)), (g(a["sigma_d"], True), g(a["mu"], True)), iter_num=0)
    assert torch.equal(out0, v0)


def test_csmri_pg_fused_vjp_vs_composed_autograd(den):
    """PGSolver_CSMRI under autograd: pnpx_csmri_pg_train / _backward (one native call each way) against the composed path,
    incl. the complex first iterate (x0 = ATy0 has an imaginary part; later iterates are real)."""
    from tfpnp_amd.tasks import csmri
    sol = csmri.PGSolver_CSMRI(den)
    B, H, W, T = 3, 64, 48, 3
    d = synth.make_csmri_batch(B, H, W, seed=173)
    a = csmri_actions(B, 4, 174, ("sigma_d", "tau"))
    v0 = sol.reset({"x0": g(d["x0"])})
    v0 = v0 + 0.05 * torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(3))
    wts = torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(4))
    y0, m = g(d["y0"]), g(d["mask"])

    def grads(fn):
        leaves = [v0.clone().requires_grad_(True), g(a["sigma_d"], True), g(a["tau"], True)]
        out = fn(*leaves)
        (out * wts).sum().backward()
        return out.detach(), [l.grad for l in leaves]

    out_f, gf = grads(lambda v, s_, t_: sol((v, (y0, m)), (s_, t_), iter_num=T))
    out_c, gc = grads(lambda v, s_, t_: sol._forward_autograd(v, y0, m, s_, t_, T))
    with torch.no_grad():
        assert rel(out_f, sol((v0, (y0, m)), (g(a["sigma_d"]), g(a["tau"])), iter_num=T)) < 1e-6
    assert rel(out_f, out_c) < 1e-5
    for n, x, y in zip(("x", "sigma_d", "tau"), gf, gc):
        print(f"  PG fused vs composed d/d{n}: {rel(x, y):.2e}")
        assert rel(x, y) < 2e-2 and x.shape == y.shape, n
    assert float(gf[0][..., 1].abs().max()) > 0              # the complex first iterate receives an imaginary cotangent
    assert float(gf[1][:, T:].abs().max()) == 0.0 and float(gf[2][:, T:].abs().max()) == 0.0
    _, gf2 = grads(lambda v, s_, t_: sol((v, (y0, m)), (s_, t_), iter_num=T))
    assert all(torch.equal(x, y) for x, y in zip(gf, gf2))


def test_csmri_apg_fused_vjp_vs_composed_autograd(den):
    """APGSolver_CSMRI under autograd: pnpx_csmri_apg_train / _backward against the composed path (three hyper-parameters:
    sigma_d, tau and the momentum beta, whose gradient needs x' - x_prev of every iteration)."""
    from tfpnp_amd.tasks import csmri
    sol = csmri.APGSolver_CSMRI(den)
    B, H, W, T = 3, 48, 48, 3
    d = synth.make_csmri_batch(B, H, W, seed=175)
    a = csmri_actions(B, 4, 176, ("sigma_d", "tau", "beta"))
    v0 = sol.reset({"x0": g(d["x0"])})
    v0 = v0 + 0.05 * torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(3))
    wts = torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(4))
    y0, m = g(d["y0"]), g(d["mask"])

    def grads(fn):
        leaves = [v0.clone().requires_grad_(True), g(a["sigma_d"], True), g(a["tau"], True), g(a["beta"], True)]
        out = fn(*leaves)
        (out * wts).sum().backward()
        return out.detach(), [l.grad for l in leaves]

    out_f, gf = grads(lambda v, s_, t_, b_: sol((v, (y0, m)), (s_, t_, b_), iter_num=T))
    out_c, gc = grads(lambda v, s_, t_, b_: sol._forward_autograd(v, y0, m, s_, t_, b_, T))
    with torch.no_grad():
        assert rel(out_f, sol((v0, (y0, m)), (g(a["sigma_d"]), g(a["tau"]), g(a["beta"])), iter_num=T)) < 1e-6
    assert rel(out_f, out_c) < 1e-5
    for n, x, y in zip(("variables", "sigma_d", "tau", "beta"), gf, gc):
        print(f"  APG fused vs composed d/d{n}: {rel(x, y):.2e}")
        assert rel(x, y) < 2e-2 and x.shape == y.shape, n
    for k in (1, 2, 3):
        assert float(gf[k][:, T:].abs().max()) == 0.0 and float(gf[k][:, :T].abs().min()) > 0
    _, gf2 = grads(lambda v, s_, t_, b_: sol((v, (y0, m)), (s_, t_, b_), iter_num=T))
    assert all(torch.equal(x, y) for x, y in zip(gf, gf2))


def test_csmri_redadmm_fused_vjp_vs_composed_autograd(den):
    """REDADMMSolver_CSMRI under autograd: pnpx_csmri_redadmm_train / _backward against the composed path (mu enters both the
    x-update and the k-space blend; lamda only the x-update; the state is complex with a non-zero imaginary part)."""
    from tfpnp_amd.tasks import csmri
    sol = csmri.REDADMMSolver_CSMRI(den)
    B, H, W, T = 3, 48, 48, 3
    d = synth.make_csmri_batch(B, H, W, seed=181)
    a = csmri_actions(B, 4, 182, ("sigma_d", "mu", "lamda"))
    v0 = sol.reset({"x0": g(d["x0"])})
    v0 = v0 + 0.05 * torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(5))
    wts = torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(6))
    y0, m = g(d["y0"]), g(d["mask"])

    def grads(fn):
        leaves = [v0.clone().requires_grad_(True), g(a["sigma_d"], True), g(a["mu"], True), g(a["lamda"], True)]
        out = fn(*leaves)
        (out * wts).sum().backward()
        return out.detach(), [l.grad for l in leaves]

    out_f, gf = grads(lambda v, s_, m_, l_: sol((v, (y0, m)), (s_, m_, l_), iter_num=T))
    out_c, gc = grads(lambda v, s_, m_, l_: sol._forward_autograd(v, y0, m, s_, m_, l_, T))
    with torch.no_grad():
        assert rel(out_f, sol((v0, (y0, m)), (g(a["sigma_d"]), g(a["mu"]), g(a["lamda"])), iter_num=T)) < 1e-6
    assert rel(out_f, out_c) < 1e-5
    for n, x, y in zip(("variables", "sigma_d", "mu", "lamda"), gf, gc):
        print(f"  RED-ADMM fused vs composed d/d{n}: {rel(x, y):.2e}")
        assert rel(x, y) < 2e-2 and x.shape == y.shape, n
    for k in (1, 2, 3):
        assert float(gf[k][:, T:].abs().max()) == 0.0 and float(gf[k][:, :T].abs().min()) > 0
    _, gf2 = grads(lambda v, s_, m_, l_: sol((v, (y0, m)), (s_, m_, l_), iter_num=T))
    assert all(torch.equal(x, y) for x, y in zip(gf, gf2))


def test_csmri_admm_train_degenerate_calls(den):
    """iter_num = 0 (identity: gradient passes straight through, hyper-parameters get zeros) and an empty batch."""
    from tfpnp_amd.tasks import csmri
    sol = csmri.ADMMSolver_CSMRI(den)
    B, H, W = 2, 32, 32
    d = synth.make_csmri_batch(B, H, W, seed=191)
    a = csmri_actions(B, 3, 192, ("sigma_d", "mu"))
    y0, m = g(d["y0"]), g(d["mask"])
    v = sol.reset({"x0": g(d["x0"])}).requires_grad_(True)
    sg, mu = g(a["sigma_d"], True), g(a["mu"], True)
    out = sol((v, (y0, m)), (sg, mu), iter_num=0)
    assert torch.equal(out, v)
    wts = torch.randn_like(out)
    (out * wts).sum().backward()
    assert torch.equal(v.grad, wts) and float(sg.grad.abs().max()) == 0.0 and float(mu.grad.abs().max()) == 0.0
    ve = v.detach()[:0].requires_grad_(True)
    oe = sol((ve, (y0[:0], m[:0])), (sg.detach()[:0].requires_grad_(True), mu.detach()[:0]))
    assert oe.shape == (0, 3, H, W, 2)
    oe.sum().backward()
    assert ve.grad.shape == ve.shape


def test_every_fused_solver_degenerate_calls(den):
    """iter_num = 0 and an empty batch through every fused training path (CS-MRI family, PR, SPI, CT): identity forward, the
    gradient passes straight through, hyper-parameter gradients are zeros of the right shape."""
    from tfpnp_amd.tasks import csmri, pr, spi, ct
    B, H = 2, 32
    dm = synth.make_csmri_batch(B, H, H, seed=201)
    a = csmri_actions(B, 3, 202, ("sigma_d", "mu", "tau", "beta", "lamda"))
    dp = synth.make_pr_batch(B, H, H, S=2, alpha=9.0, seed=203)
    ds = synth.make_spi_batch(B, H, H, K=6, seed=204)
    y0c, x0c, view = _ct_case(B, H, 12, 205)
    cases = [(cls(den), lambda sol: sol.reset({"x0": g(dm["x0"])}), (g(dm["y0"]), g(dm["mask"])), keys)
             for cls, keys in ((csmri.ADMMSolver_CSMRI, ("sigma_d", "mu")), (csmri.HQSSolver_CSMRI, ("sigma_d", "mu")),
                               (csmri.PGSolver_CSMRI, ("sigma_d", "tau")), (csmri.APGSolver_CSMRI, ("sigma_d", "tau", "beta")),
                               (csmri.REDADMMSolver_CSMRI, ("sigma_d", "mu", "lamda")))]
    cases += [(pr.IADMMSolver_PR(den), lambda sol: sol.reset({"x0": g(dp["x0"])}), (g(dp["y0"]), g(dp["mask"])),
               ("sigma_d", "mu", "tau")),
              (spi.ADMMSolver_SPI(den), lambda sol: sol.reset({"x0": g(ds["x0"])}), (g(ds["x0"]), g(ds["K"])), ("sigma_d", "mu")),
              (ct.IADMMSolver_CT(den), lambda sol: sol.reset({"x0": x0c}), (y0c, view), ("sigma_d", "mu", "tau")),
              (ct.PGSolver_CT(den), lambda sol: sol.reset({"x0": x0c}), (y0c, view), ("sigma_d", "tau"))]
    for sol, reset, aux, keys in cases:
        name = type(sol).__name__
        v = reset(sol).requires_grad_(True)
        hp = [g(a[k], True) for k in keys]
        out = sol((v, aux), tuple(hp), iter_num=0)
        assert torch.equal(out, v), name
        wts = torch.randn_like(out)
        (out * wts).sum().backward()
        assert torch.equal(v.grad, wts), name
        assert all(h.grad is not None and h.grad.shape == h.shape and float(h.grad.abs().max()) == 0.0 for h in hp), name
        ve = v.detach()[:0].requires_grad_(True)
        oe = sol((ve, tuple(t_[:0] for t_ in aux)), tuple(h.detach()[:0].requires_grad_(True) for h in hp))
        assert oe.shape == ve.shape, name
        oe.sum().backward()
        assert ve.grad.shape == ve.shape, name


def test_csmri_admm_activation_cache(den):
    """The training path parks the denoiser activations of training forwards in a ring (tickets):
    (a) gradients with the cache == gradients by re-computation (bit for bit for the same forward: same kernels);
    (b) a backward whose forward has been overwritten by a later training forward falls back to re-computation -- it
        must never read the newer activations; (c) the memory is accounted for and released by train_cache_gb = 0."""
    from tfpnp_amd.tasks import csmri
    sol = csmri.ADMMSolver_CSMRI(den)
    ctx = den.context(dev())
    B, H, W, T = 2, 48, 32, 3
    y0s, ms, v0s, acts = [], [], [], []
    for k in range(2):
        d = synth.make_csmri_batch(B, H, W, seed=181 + k)
        a = csmri_actions(B, T, 182 + k, ("sigma_d", "mu"))
        y0s.append(g(d["y0"])); ms.append(g(d["mask"])); v0s.append(sol.reset({"x0": g(d["x0"])}))
        acts.append((a["sigma_d"], a["mu"]))
    wts = torch.randn(v0s[0].shape, device=dev(), generator=torch.Generator(dev()).manual_seed(5))

    def forward(k):
        leaves = [v0s[k].clone().requires_grad_(True), g(acts[k][0], True), g(acts[k][1], True)]
        return leaves, sol((leaves[0], (y0s[k], ms[k])), tuple(leaves[1:]))

    def backward(leaves, out):
        (out * wts).sum().backward()
        return [l.grad.clone() for l in leaves]

    ctx.set_option("train_cache_gb", 0)                             # also drops what earlier tests left in the cache
    ref = [backward(*forward(k)) for k in range(2)]                 # re-computation only
    base = ctx.bytes()
    ctx.set_option("train_cache_gb", 96)
    try:
        l0, o0 = forward(0)
        held = ctx.bytes()
        assert held > base + T * B * H * W * 4 * 32                 # T ring slots are resident now
        cached0 = backward(l0, o0)                                  # (a) served from the cache
        # (the un-cached forward runs the fused network tail: forward values differ in the last bit, gradients follow)
        assert all(rel(x, y) < 2e-2 for x, y in zip(cached0, ref[0]))
        l0, o0 = forward(0)
        for _ in range(64 // T + 1):                                # enough later forwards to re-use every slot of the
            l1, o1 = forward(1)                                     # ring (at most 64 slots, T per forward)
        cached1 = backward(l1, o1)
        stale0 = backward(l0, o0)                                   # (b) stale ticket -> re-computation, not wrong data:
        assert all(torch.equal(x, y) for x, y in zip(stale0, cached0))   # same kernels on the same inputs, bit for bit
        assert all(rel(x, y) < 2e-2 for x, y in zip(cached1, ref[1]))
        assert rel(cached1[0], cached0[0]) > 0.1                    # (the two cases do have different gradients)
        ctx.set_option("train_cache_gb", 0)                         # (c)
        assert ctx.bytes() < held
    finally:
        ctx.set_option("train_cache_gb", 96)


def test_pr_solver_gradients(den, oden32, oden64):
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import pr
    B, H, W, S, T = 2, 32, 32, 4, 3
    d = synth.make_pr_batch(B, H, W, S=S, alpha=9.0, seed=75)
    a = csmri_actions(B, T, 76, ("sigma_d", "mu", "tau"))
    a["tau"] = (a["tau"] * 0.5).astype(np.float32)
    sol = pr.IADMMSolver_PR(den)
    v0 = sol.reset({"x0": g(d["x0"])}).cpu().numpy()
    wts = np.random.RandomState(77).standard_normal(v0.shape).astype(np.float32)
    acts = [a["sigma_d"], a["mu"], a["tau"]]

    def run_oracle(oden, dtype):
        y0, m = t(d["y0"]).to(dtype), t(d["mask"]).to(dtype)
        return oracle_grads(lambda v, *p: O.pr_iadmm(oden, v, y0, m, *p), [v0] + acts, wts, dtype)

    out64, g64 = run_oracle(oden64, torch.float64)
    _, g32 = run_oracle(oden32, torch.float32)
    leaves = [g(v0, True)] + [g(p, True) for p in acts]
    out = sol((leaves[0], (g(d["y0"]), g(d["mask"]))), tuple(leaves[1:]))
    assert rel(out, out64) < 1e-4
    (out * g(wts)).sum().backward()
    _check(["variables", "sigma_d", "mu", "tau"], [l.grad for l in leaves], g64, g32)
    from tests.conftest import golden      # and against the REAL reference's autograd on the same inputs
    gold = golden("solver_grads")
    assert rel(out, t(gold["pr_out"])) < 1e-4
    for key, leaf in zip(("variables", "sigma_d", "mu", "tau"), leaves):
        e = rel(leaf.grad, t(gold[f"pr_grad_{key}"]))
        print(f"  pr vs reference autograd d/d{key}: {e:.2e}")
        assert e < 2e-2, key


def test_pr_fused_vjp_vs_composed_autograd(den):
    """IADMMSolver_PR under autograd: pnpx_pr_iadmm_train / _backward against the composed path (the |Az| Jacobian of the
    data step evaluated at the saved k-space images; three hyper-parameters)."""
    from tfpnp_amd.tasks import pr
    sol = pr.IADMMSolver_PR(den)
    B, H, W, S, T = 3, 32, 32, 4, 3
    d = synth.make_pr_batch(B, H, W, S=S, alpha=9.0, seed=81)
    a = csmri_actions(B, 4, 82, ("sigma_d", "mu", "tau"))
    a["tau"] = (a["tau"] * 0.5).astype(np.float32)
    v0 = sol.reset({"x0": g(d["x0"])})
    v0 = v0 + 0.05 * torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(7))
    wts = torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(8))
    y0, m = g(d["y0"]), g(d["mask"])

    def grads(fn):
        leaves = [v0.clone().requires_grad_(True), g(a["sigma_d"], True), g(a["mu"], True), g(a["tau"], True)]
        out = fn(*leaves)
        (out * wts).sum().backward()
        return out.detach(), [l.grad for l in leaves]

    out_f, gf = grads(lambda v, s_, m_, t_: sol((v, (y0, m)), (s_, m_, t_), iter_num=T))
    out_c, gc = grads(lambda v, s_, m_, t_: sol._forward_autograd(v, y0, m, s_, m_, t_, T))
    with torch.no_grad():
        assert rel(out_f, sol((v0, (y0, m)), (g(a["sigma_d"]), g(a["mu"]), g(a["tau"])), iter_num=T)) < 1e-6
    assert rel(out_f, out_c) < 1e-5
    for n, x, y in zip(("variables", "sigma_d", "mu", "tau"), gf, gc):
        print(f"  PR fused vs composed d/d{n}: {rel(x, y):.2e}")
        assert rel(x, y) < 2e-2 and x.shape == y.shape, n
    assert float(gf[0][:, 0].abs().max()) == 0.0          # an iteration never reads its x
    for k in (1, 2, 3):
        assert float(gf[k][:, T:].abs().max()) == 0.0 and float(gf[k][:, :T].abs().min()) > 0
    _, gf2 = grads(lambda v, s_, m_, t_: sol((v, (y0, m)), (s_, m_, t_), iter_num=T))
    assert all(torch.equal(x, y) for x, y in zip(gf, gf2))


def test_spi_solver_gradients(den, oden32):
    """One iteration from the same state (the bisection prox is discontinuous, see test_spi_golden)."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import spi
    B, H, W = 2, 32, 32
    d = synth.make_spi_batch(B, H, W, K=6, seed=78)
    rs = np.random.RandomState(79)
    sg = rs.uniform(15 / 255.0, 70 / 255.0, (B, 1)).astype(np.float32)
    m = rs.uniform(50, 120, (B, 1)).astype(np.float32)
    sol = spi.ADMMSolver_SPI(den)
    v0 = sol.reset({"x0": g(d["x0"])}).cpu().numpy()
    v0[:, 2] = 0.02 * rs.standard_normal(v0[:, 2].shape).astype(np.float32)
    wts = rs.standard_normal(v0.shape).astype(np.float32)
    out32, g32 = oracle_grads(lambda v, s_, m_: O.spi_admm(oden32, v, t(d["x0"]), t(d["K"]), s_, m_), [v0, sg, m], wts,
                              torch.float32)
    leaves = [g(v0, True), g(sg, True), g(m, True)]
    out = sol((leaves[0], (g(d["x0"]), g(d["K"]))), tuple(leaves[1:]))
    assert rel(out, out32) < 5e-3
    (out * g(wts)).sum().backward()
    for n, a_, b_ in zip(["variables", "sigma_d", "mu"], [l.grad for l in leaves], g32):
        print(f"  spi d/d{n}: {rel(a_, b_):.2e}")
        assert rel(a_, b_) < 2e-2, n
    from tests.conftest import golden      # and against the REAL reference's autograd on the same inputs
    gold = golden("solver_grads")
    for n, leaf in zip(["variables", "sigma_d", "mu"], leaves):
        ref = t(gold[f"spi_grad_{n}"])
        if float(ref.abs().max()) == 0.0:      # the bisection prox carries no gradient wrt mu in the reference either
            assert leaf.grad is None or float(leaf.grad.abs().max()) == 0.0, n
            continue
        e = rel(leaf.grad, ref)
        print(f"  spi vs reference autograd d/d{n}: {e:.2e}")
        assert e < 2e-2, n


def _fused_vs_composed(sol, v0, aux, acts, composed, T, names, tol=2e-2, zero_slot=None):
    """Gradients of the fused native training path of `sol` against `composed` (the reference's loop from differentiable ops)."""
    wts = torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(12))

    def grads(fn):
        leaves = [v0.clone().requires_grad_(True)] + [g(p, True) for p in acts]
        out = fn(*leaves)
        (out * wts).sum().backward()
        return out.detach(), [l.grad for l in leaves]

    out_f, gf = grads(lambda v, *p: sol((v, aux), tuple(p), iter_num=T))
    out_c, gc = grads(lambda v, *p: composed(v, *p))
    with torch.no_grad():
        assert rel(out_f, sol((v0, aux), tuple(g(p) for p in acts), iter_num=T)) < 1e-6
    assert rel(out_f, out_c) < 1e-5
    for n, x, y in zip(names, gf, gc):
        e = rel(x, y) if float(y.abs().max()) > 0 else float(x.abs().max())
        print(f"  {type(sol).__name__} fused vs composed d/d{n}: {e:.2e}")
        assert e < tol and x.shape == y.shape, n
    if zero_slot is not None:
        assert float(gf[0][:, zero_slot].abs().max()) == 0.0
    for k in range(1, len(gf)):
        assert float(gf[k][:, T:].abs().max()) == 0.0
    _, gf2 = grads(lambda v, *p: sol((v, aux), tuple(p), iter_num=T))
    assert all(torch.equal(x, y) for x, y in zip(gf, gf2))


def test_spi_fused_vjp_vs_composed_autograd(den):
    """ADMMSolver_SPI under autograd (pnpx_spi_admm_train / _backward): one and two iterations; K1 == 0 pixels (the only ones whose
    prox carries a gradient, also wrt mu) are present in the batch."""
    from tfpnp_amd.tasks import spi
    sol = spi.ADMMSolver_SPI(den)
    B, H, W = 3, 32, 32
    d = synth.make_spi_batch(B, H, W, K=6, seed=83)
    assert float((g(d["x0"]) == 0).float().mean()) > 0.01
    rs = np.random.RandomState(84)
    acts = [rs.uniform(15 / 255.0, 70 / 255.0, (B, 3)).astype(np.float32), rs.uniform(50, 120, (B, 3)).astype(np.float32)]
    v0 = sol.reset({"x0": g(d["x0"])})
    v0[:, 2] = 0.02 * torch.randn(v0[:, 2].shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(9))
    aux = (g(d["x0"]), g(d["K"]))
    for T in (1, 2):
        _fused_vs_composed(sol, v0, aux, acts, lambda v, s_, m_: sol._forward_autograd(v, aux[0], aux[1], s_, m_, T), T,
                           ("variables", "sigma_d", "mu"), tol=3e-2, zero_slot=1)


def _ct_case(B=2, R=32, V=20, seed=95):
    from tfpnp_amd.utils import transforms as Tr
    gt = synth.phantom_batch(B, R, R, seed)
    radon = Tr.Radon_norm(R, V, device=dev())
    rs = np.random.RandomState(seed + 1)
    sino = radon.forward(g(gt))
    y0 = sino * (1 + 0.05 * g(rs.standard_normal(tuple(sino.shape)).astype(np.float32)))
    x0 = radon.backprojection_norm(y0)
    view = g(np.full((B, 1, R, R), V / 120.0, np.float32))
    return y0, x0, view


def test_ct_fused_vjp_vs_composed_autograd(den):
    """IADMMSolver_CT and PGSolver_CT under autograd: pnpx_ct_{iadmm,pg}_train / _backward against the composed paths (the data
    step's adjoint is A^T A / opnorm^2 with the projector pair standing in for each other's transpose)."""
    from tfpnp_amd.tasks import ct
    B, T = 2, 3
    y0, x0, view = _ct_case(B)
    a = csmri_actions(B, 4, 97, ("sigma_d", "mu", "tau"))
    sol = ct.IADMMSolver_CT(den)
    v0 = sol.reset({"x0": x0})
    v0 = v0 + 0.02 * torch.randn(v0.shape, device=v0.device, generator=torch.Generator(v0.device).manual_seed(10))
    _fused_vs_composed(sol, v0, (y0, view), [a["sigma_d"], a["mu"], a["tau"]],
                       lambda v, s_, m_, t_: sol._forward_autograd(v, y0, s_, m_, t_, T), T,
                       ("variables", "sigma_d", "mu", "tau"), zero_slot=0)
    pg = ct.PGSolver_CT(den)
    pg.radon_generator.opnorms = dict(sol.radon_generator.opnorms)
    _fused_vs_composed(pg, pg.reset({"x0": x0}), (y0, view), [a["sigma_d"], a["tau"]],
                       lambda v, s_, t_: pg._forward_autograd(v, y0, s_, t_, T), T, ("variables", "sigma_d", "tau"))


def test_fused_vjps_with_and_without_the_activation_ring(unet_params):
    """Every fused training path gives the same gradients whether the denoiser VJPs find their activations in the training ring
    or re-compute them (to the level at which the two forward variants differ: the ring-less forward fuses the out-conv) -- the
    re-computing VJP uses (and may re-allocate) the context's scratch buffer, which the solver loops share with it: a FRESH
    context per solver and small shapes, so that the scratch has to grow inside the backward."""
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks import csmri, pr, spi, ct

    def grads(sol, v0, aux, acts, T):
        leaves = [v0.clone().requires_grad_(True)] + [g(p, True) for p in acts]
        out = sol((leaves[0], aux), tuple(leaves[1:]), iter_num=T)
        wts = torch.randn(out.shape, device=out.device, generator=torch.Generator(out.device).manual_seed(21))
        (out * wts).sum().backward()
        return [out.detach()] + [l.grad for l in leaves]

    def check(make_sol, make_case, T=2):
        res = []
        for cache_gb in (-1, 0):
            d_ = UNetDenoiser2D(state_dict=unet_params)           # fresh context: its scratch starts empty
            d_.context(dev()).set_option("train_cache_gb", cache_gb)
            sol = make_sol(d_)
            v0, aux, acts = make_case(sol)
            res.append(grads(sol, v0, aux, acts, T))
        for k, (a_, b_) in enumerate(zip(*res)):
            e = rel(a_, b_) if float(b_.abs().max()) > 0 else float(a_.abs().max())
            assert e < (1e-5 if k == 0 else 2e-2), (type(sol).__name__, k, e)

    B, H = 1, 16
    dm = synth.make_csmri_batch(B, H, H, seed=31)
    am = csmri_actions(B, 3, 32, ("sigma_d", "mu", "tau", "beta", "lamda"))
    mri_aux = (g(dm["y0"]), g(dm["mask"]))
    for cls, keys in ((csmri.ADMMSolver_CSMRI, ("sigma_d", "mu")), (csmri.HQSSolver_CSMRI, ("sigma_d", "mu")),
                      (csmri.PGSolver_CSMRI, ("sigma_d", "tau")), (csmri.APGSolver_CSMRI, ("sigma_d", "tau", "beta")),
                      (csmri.REDADMMSolver_CSMRI, ("sigma_d", "mu", "lamda"))):
        check(cls, lambda sol: (sol.reset({"x0": g(dm["x0"])}), mri_aux, [am[k] for k in keys]))
    dp = synth.make_pr_batch(B, H, H, S=1, alpha=9.0, seed=33)     # S = 1: the k-space scratch is SMALLER than the VJP's
    ap = csmri_actions(B, 3, 34, ("sigma_d", "mu", "tau"))
    ap["tau"] = (ap["tau"] * 0.5).astype(np.float32)
    check(pr.IADMMSolver_PR, lambda sol: (sol.reset({"x0": g(dp["x0"])}), (g(dp["y0"]), g(dp["mask"])),
                                          [ap["sigma_d"], ap["mu"], ap["tau"]]))
    ds = synth.make_spi_batch(B, H, H, K=6, seed=35)
    rs = np.random.RandomState(36)
    spi_acts = [rs.uniform(15 / 255.0, 70 / 255.0, (B, 3)).astype(np.float32), rs.uniform(50, 120, (B, 3)).astype(np.float32)]
    check(spi.ADMMSolver_SPI, lambda sol: (sol.reset({"x0": g(ds["x0"])}), (g(ds["x0"]), g(ds["K"])), spi_acts),
          T=1)                                                          # (later iterations: bisection flips on 1e-7 differences)
    y0, x0, view = _ct_case(B, 16, 12, 37)
    ac = csmri_actions(B, 3, 38, ("sigma_d", "mu", "tau"))
    check(ct.IADMMSolver_CT, lambda sol: (sol.reset({"x0": x0}), (y0, view), [ac["sigma_d"], ac["mu"], ac["tau"]]), T=3)
    check(ct.PGSolver_CT, lambda sol: (sol.reset({"x0": x0}), (y0, view), [ac["sigma_d"], ac["tau"]]), T=3)


def test_ct_solver_gradients(den, oden32, oden64, monkeypatch):
    """The Radon pair is unmatched and each operator is used as the other's VJP (torch_radon's convention); the
    oracle is given the same convention through autograd.Function wrappers for this test."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import ct
    from tfpnp_amd.utils import transforms as Tr
    B, R, V, T = 2, 32, 20, 3
    fwd, bwd = O.radon_forward, O.radon_backprojection

    class F_(torch.autograd.Function):
        @staticmethod
        def forward(ctx, img, angles, det):
            ctx.a, ctx.R = angles, img.shape[-1]
            return fwd(img, angles, det)

        @staticmethod
        def backward(ctx, gr):
            return bwd(gr, ctx.a, ctx.R), None, None

    class B_(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sino, angles, res):
            ctx.a, ctx.det = angles, sino.shape[-1]
            return bwd(sino, angles, res)

        @staticmethod
        def backward(ctx, gr):
            return fwd(gr, ctx.a, ctx.det), None, None

    monkeypatch.setattr(O, "radon_forward", lambda img, a, det: F_.apply(img, a, det))
    monkeypatch.setattr(O, "radon_backprojection", lambda s, a, res: B_.apply(s, a, res))
    angles, det = O.radon_geometry(R, V)
    gt = synth.phantom_batch(B, R, R, 91)
    rs = np.random.RandomState(92)
    sino = fwd(t(gt), angles, det).numpy()
    y0 = (sino * (1 + 0.05 * rs.standard_normal(sino.shape))).astype(np.float32)
    radon = Tr.Radon_norm(R, V, device=dev())
    x0 = radon.backprojection_norm(g(y0)).cpu().numpy()
    view = np.full((B, 1, R, R), V / 120.0, np.float32)
    a = csmri_actions(B, T, 93, ("sigma_d", "mu", "tau"))
    sol = ct.IADMMSolver_CT(den)
    v0 = sol.reset({"x0": g(x0)}).cpu().numpy()
    wts = rs.standard_normal(v0.shape).astype(np.float32)
    acts = [a["sigma_d"], a["mu"], a["tau"]]

    def run_oracle(oden, dtype):
        return oracle_grads(lambda v, *p: O.ct_iadmm(oden, v, t(y0).to(dtype), V, radon.opnorm, *p), [v0] + acts, wts,
                            dtype)

    out64, g64 = run_oracle(oden64, torch.float64)
    _, g32 = run_oracle(oden32, torch.float32)
    leaves = [g(v0, True)] + [g(p, True) for p in acts]
    out = sol((leaves[0], (g(y0), g(view))), tuple(leaves[1:]))
    assert rel(out, out64) < 1e-4
    (out * g(wts)).sum().backward()
    _check(["variables", "sigma_d", "mu", "tau"], [l.grad for l in leaves], g64, g32)
    # PG variant: values + gradients flow
    pg = ct.PGSolver_CT(den)
    pg.radon_generator.opnorms = dict(sol.radon_generator.opnorms)
    leaves = [g(x0, True), g(a["sigma_d"], True), g(a["tau"], True)]
    out = pg((leaves[0], (g(y0), g(view))), (leaves[1], leaves[2]))
    _, gp64 = oracle_grads(lambda v, s_, t_: O.ct_pg(oden64, v, t(y0).double(), V, radon.opnorm, s_, t_),
                           [x0, a["sigma_d"], a["tau"]], wts[:, :1], torch.float64)
    _, gp32 = oracle_grads(lambda v, s_, t_: O.ct_pg(oden32, v, t(y0), V, radon.opnorm, s_, t_),
                           [x0, a["sigma_d"], a["tau"]], wts[:, :1], torch.float32)
    (out * g(wts[:, :1])).sum().backward()
    _check(["variables", "sigma_d", "tau"], [l.grad for l in leaves], gp64, gp32)


def test_env_forward_trains_through_the_solver(den, oden64):
    """PnPEnv.forward (tfpnp/env/base.py:193-206): delta-PSNR reward differentiated wrt policy-like actions."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import csmri
    B, H, W = 2, 32, 32
    d = synth.make_csmri_batch(B, H, W, seed=95)
    sol = csmri.ADMMSolver_CSMRI(den)
    env = csmri.CSMRIEnv(None, sol, max_episode_step=6)
    ob = env.reset({k: g(v) for k, v in d.items() if isinstance(v, np.ndarray)})
    raw0 = np.random.RandomState(96).standard_normal((B, 10)).astype(np.float32)

    def act(raw):
        return {"sigma_d": torch.sigmoid(raw[:, :5]) * 70 / 255, "mu": torch.sigmoid(raw[:, 5:])}

    raw = g(raw0, True)
    next_ob, reward = env.forward(ob, act(raw))
    assert abs(float(next_ob.T.mean()) - 1 / 6) < 1e-6
    assert reward.shape == (B, 1)
    reward.sum().backward()
    # oracle: same computation in fp64 on the CPU
    r64 = t(raw0).double().requires_grad_(True)
    a64 = act(r64)
    v0 = O.admm_reset(t(d["x0"]).double())
    st = O.csmri_admm(oden64, v0, t(d["y0"]).double(), t(d["mask"]), a64["sigma_d"], a64["mu"])
    out2 = O.complex2real(st[:, 0:1]) if st.dim() == 5 else st[:, 0:1]
    rew64 = O.torch_psnr(out2, t(d["gt"]).double()) - O.torch_psnr(t(d["output"]).double(), t(d["gt"]).double())
    rew64.sum().backward()
    assert rel(reward, rew64) < 1e-4
    print(f"  env.forward d reward / d policy logits: {rel(raw.grad, r64.grad):.2e}")
    assert rel(raw.grad, r64.grad) < 2e-3


def test_vjp_gradient_scale_invariance(den):
    """The half-split backward normalises grad_out by a power of two: tiny, huge and zero upstream gradients."""
    from tfpnp_amd import ops
    x, s = denoiser_inputs(2, 32, 32, 67)
    g1 = np.random.RandomState(67).standard_normal((2, 1, 32, 32)).astype(np.float32)
    ctx = den.context(dev())
    a, sa = ops.unet_denoise_backward(ctx, g(x), g(s), g(g1))
    for scale in (1e-12, 3e-5, 7e4, 1e20):
        b, sb = ops.unet_denoise_backward(ctx, g(x), g(s), g(g1 * np.float32(scale)))
        assert torch.isfinite(b).all() and rel(b, a * scale) < 1e-5 and rel(sb, sa * scale) < 1e-4, scale
    z, sz = ops.unet_denoise_backward(ctx, g(x), g(s), g(np.zeros_like(g1)))
    assert float(z.abs().max()) == 0.0 and float(sz.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------------------------------
# Native VJPs against the REAL reference's autograd on kink-free cases (tests/golden/solver_grads_kinkfree.npz, generator
# oracle/make_goldens.py::kinkfree_gradient_goldens): every LeakyReLU / ReLU / max-pool / clamp / bisection
# decision of these trajectories is >= 1e-5 from its kink (tests/test_oracle_golden.py re-checks it), so a correct fp32-class
# evaluation shares every decision with the reference and the gradients agree to rounding: bound 1e-3, observed ~1e-5.
@pytest.mark.parametrize("name", ["hqs", "pg", "apg", "redadmm", "pr", "spi"])
def test_fused_solver_vjps_vs_reference_autograd_kinkfree(den, name):
    from tests.conftest import golden
    from tests.golden_inputs import KINKFREE_KEYS, kinkfree_case
    from tfpnp_amd.tasks import csmri, pr, spi
    gold = golden("solver_grads_kinkfree")
    c = kinkfree_case(name)
    if name in ("hqs", "pg", "apg", "redadmm"):
        sol = {"hqs": csmri.HQSSolver_CSMRI, "pg": csmri.PGSolver_CSMRI, "apg": csmri.APGSolver_CSMRI,
               "redadmm": csmri.REDADMMSolver_CSMRI}[name](den)
        aux = (g(c["y0"]), g(c["mask"]))
    elif name == "pr":
        sol, aux = pr.IADMMSolver_PR(den), (g(c["y0"]), g(c["mask"]))
    else:
        sol, aux = spi.ADMMSolver_SPI(den), (g(c["x0"]), g(c["K"]))
    leaves = [g(c["v0"], True)] + [g(p, True) for p in c["acts"]]
    out = sol((leaves[0], aux), tuple(leaves[1:]))
    assert rel(out, t(gold[f"{name}_out"])) < 1e-5
    (out * g(c["wts"])).sum().backward()
    for key, leaf in zip(("variables",) + KINKFREE_KEYS[name], leaves):
        want = t(gold[f"{name}_grad_{key}"])
        if float(want.abs().max()) == 0.0:     # e.g. SPI: the bisection prox carries no gradient wrt mu in the reference either
            assert leaf.grad is None or float(leaf.grad.abs().max()) == 0.0, key
            continue
        e = rel(leaf.grad, want)
        print(f"  {name} (kink-free) vs reference autograd d/d{key}: {e:.2e}")
        assert e < 1e-3, key
