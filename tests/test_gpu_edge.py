"""GPU edge cases: empty and ragged batches, the largest BASELINE configurations (PR 36 x 256^2, CT 32 x 256^2 / 30
views, SPI 64 x 512^2), maximum FFT length -- through size-independent properties and small-slice oracle checks."""
import numpy as np
import pytest
import torch

from tests.golden_inputs import csmri_actions, denoiser_inputs
from tfpnp_amd import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def den(unet_params):
    from tfpnp_amd.pnp import UNetDenoiser2D
    return UNetDenoiser2D(state_dict=unet_params)


@pytest.fixture(scope="module")
def oden(unet_params):
    from oracle import pnp_oracle as O
    return O.Denoiser(unet_params)


def test_empty_batch_everywhere(den):
    """All items stopped (idx_left empty): every entry point returns an empty tensor of the right shape."""
    from tfpnp_amd.env import torch_psnr
    from tfpnp_amd.tasks import csmri, ct, pr, spi
    e = lambda *s: torch.empty(*s, device=dev())
    assert den(e(0, 1, 64, 64), e(0)).shape == (0, 1, 64, 64)
    sol = csmri.ADMMSolver_CSMRI(den)
    out = sol((e(0, 3, 64, 64, 2), (e(0, 1, 64, 64, 2), torch.empty(0, 1, 64, 64, dtype=torch.bool, device=dev()))),
              (e(0, 5), e(0, 5)))
    assert out.shape == (0, 3, 64, 64, 2)
    assert pr.IADMMSolver_PR(den)((e(0, 3, 32, 32, 2), (e(0, 4, 32, 32), e(0, 4, 32, 32, 2))),
                                  (e(0, 5), e(0, 5), e(0, 5))).shape == (0, 3, 32, 32, 2)
    assert spi.ADMMSolver_SPI(den)((e(0, 3, 32, 32), (e(0, 1, 32, 32), e(0, 1, 32, 32))), (e(0, 5), e(0, 5))).shape[0] == 0
    assert torch_psnr(e(0, 1, 8, 8), e(0, 1, 8, 8)).shape == (0, 1)


def test_env_runs_to_the_last_item(den):
    """Ragged compaction 5 -> 3 -> 1 -> 0 live items: stopped rows are frozen, rewards of stopped rows are zero."""
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
    B, H, W = 5, 64, 64
    d = synth.make_csmri_batch(B, H, W, seed=33)
    env = CSMRIEnv(None, ADMMSolver_CSMRI(den), max_episode_step=6)
    env.reset({k: g(v) for k, v in d.items()})
    stops = [[0, 1, 0, 1, 0], [1, 0, 1], [1]]
    frozen = {}
    for s, stop in enumerate(stops):
        live = env.idx_left.clone()
        a = csmri_actions(len(stop), 2, 40 + s)
        before = env.state["solver"].clone()
        ob, ob_masked, reward, all_done, info = env.step({"sigma_d": g(a["sigma_d"]), "mu": g(a["mu"]),
                                                          "idx_stop": g(np.array(stop))})
        dead = [i for i in range(B) if i not in live.tolist()]
        assert torch.equal(env.state["solver"][dead], before[dead])
        assert float(reward[dead].abs().max()) == 0.0 if dead else True
        assert len(ob) == len(stop) and len(ob_masked) == stop.count(0)
    assert all_done and len(env.idx_left) == 0


def test_pr_full_config_slice_vs_oracle(den, oden):
    """BASELINE config #3 (PR, B=36, 256^2, S=4): one iADMM iteration; two items re-checked against the CPU oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import pr
    B, H, W, S = 36, 256, 256, 4
    d = synth.make_pr_batch(B, H, W, S=S, alpha=9.0, seed=77)
    a = csmri_actions(B, 1, 78, ("sigma_d", "mu", "tau"))
    a["tau"] = (0.5 * a["tau"]).astype(np.float32)
    sol = pr.IADMMSolver_PR(den)
    v0 = sol.reset({"x0": g(d["x0"])})
    st = sol((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"]), g(a["tau"])))
    for i in (0, 35):
        sl = slice(i, i + 1)
        ref = O.pr_iadmm(oden, O.pr_reset(t(d["x0"][sl])), t(d["y0"][sl]), t(d["mask"][sl]), t(a["sigma_d"][sl]),
                         t(a["mu"][sl]), t(a["tau"][sl]))
        assert rel(st[sl], ref) < 1e-4
    assert torch.equal(st, sol((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"]), g(a["tau"]))))


def test_spi_full_config_slice_vs_oracle(den, oden):
    """BASELINE config #5 (SPI, B=64, 512^2): one iteration; one item re-checked against the CPU oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import spi
    B, H, W = 64, 512, 512
    d = synth.make_spi_batch(B, H, W, K=6, seed=88)
    sg = np.full((B, 1), 40 / 255.0, np.float32)
    m = np.full((B, 1), 85.0, np.float32)
    sol = spi.ADMMSolver_SPI(den)
    v0 = sol.reset({"x0": g(d["x0"])})
    st = sol((v0, (g(d["x0"]), g(d["K"]))), (g(sg), g(m)))
    ref = O.spi_admm(oden, O.admm_reset(t(d["x0"][63:64])), t(d["x0"][63:64]), t(d["K"][63:64]), t(sg[63:64]), t(m[63:64]))
    assert rel(st[63:64], ref) < 5e-4          # bisection quantum, see test_spi_golden
    x, z, u = torch.split(st, 1, dim=1)
    assert float(z.min()) >= 0.0 and float(z.max()) <= 1.0 and float(x.min()) >= 0.0 and float(x.max()) <= 1.0


def test_ct_full_config_slice_vs_oracle(den, oden):
    """BASELINE config #4 (CT, B=32, 256^2, 30 views): sinogram + one iADMM iteration, one item vs the CPU oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import ct
    from tfpnp_amd.utils import transforms as T
    B, R, V = 32, 256, 30
    gt = synth.phantom_batch(B, R, R, 99)
    radon = T.Radon_norm(R, V, device=dev())
    y0 = radon.forward(g(gt))
    angles, det = O.radon_geometry(R, V)
    assert tuple(y0.shape) == (B, 1, V, det) and det == 363
    assert rel(y0[31:32], O.radon_forward(t(gt[31:32]), angles, det)) < 1e-5
    x0 = radon.backprojection_norm(y0)
    a = csmri_actions(B, 1, 98, ("sigma_d", "mu", "tau"))
    sol = ct.IADMMSolver_CT(den)
    sol.radon_generator.opnorms[(R, V)] = radon.opnorm
    view = torch.full((B, 1, R, R), V / 120.0, device=dev())
    st = sol((sol.reset({"x0": x0}), (y0, view)), (g(a["sigma_d"]), g(a["mu"]), g(a["tau"])))
    ref = O.ct_iadmm(oden, O.admm_reset(x0[31:32].cpu()), y0[31:32].cpu(), V, radon.opnorm, t(a["sigma_d"][31:32]),
                     t(a["mu"][31:32]), t(a["tau"][31:32]))
    assert rel(st[31:32], ref) < 1e-4


def test_fft_maximum_length_roundtrip_and_parseval():
    from tfpnp_amd.utils import transforms as T
    x = torch.randn(2, 1, 2048, 2048, 2, device=dev())
    k = T.fft2(x)
    assert abs(float((k ** 2).sum() / (x ** 2).sum()) - 1) < 1e-5
    assert rel(T.ifft2(k), x) < 1e-5
    from tfpnp_amd._lib import PnpxError
    with pytest.raises(PnpxError):
        T.fft2(torch.randn(1, 1, 4096, 8, 2, device=dev()))


def test_denoiser_ragged_batches_reuse_workspace(den, oden):
    """B = 1 .. capB in arbitrary order on one context (idx_left compaction) -> same per-item results."""
    x, s = denoiser_inputs(7, 96, 96, 5)
    full = den(g(x), g(s)).clone()
    for idx in ([6], [0, 3], [1, 2, 4, 5, 6], [5, 1], list(range(7))):
        assert torch.equal(den(g(x[idx]), g(s[idx])), full[idx])
    assert rel(full[:2], oden(t(x[:2]), t(s[:2]))) < 1e-4


@pytest.mark.parametrize("B,R,V", [(3, 100, 17), (2, 64, 180), (1, 511, 7)])
def test_radon_forward_odd_geometries_vs_oracle(B, R, V):
    """Transposed-copy projector (steep views read the transposed image): odd sizes, many / few views."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.utils import transforms as T
    gt = torch.rand(B, 1, R, R, generator=torch.Generator().manual_seed(R))
    radon = T.Radon_norm(R, V, device=dev(), opnorm=1.0)
    angles, det = O.radon_geometry(R, V)
    assert rel(radon.forward(gt.to(dev())), O.radon_forward(gt, angles, det)) < 1e-5


def test_two_contexts_from_two_threads_and_side_streams(unet_params):
    """Threading / stream contract of the boundary (include/pnpx.h): one ctx per thread, work enqueued on the caller's
    current stream.  Two host threads drive two contexts concurrently on side streams; results equal the serial ones."""
    import threading
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    B, H, W = 3, 64, 64
    jobs = []
    for k in range(2):
        d = synth.make_csmri_batch(B, H, W, seed=200 + k)
        a = csmri_actions(B, 4, 210 + k)
        jobs.append((d, a))
    serial = []
    for d, a in jobs:
        sol = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params))
        v0 = sol.reset({"x0": g(d["x0"])})
        serial.append(sol((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"]))).clone())
    out = [None, None]
    errs = []

    def work(k):
        try:
            d, a = jobs[k]
            sol = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=unet_params))     # its own denoiser -> its own pnpx_ctx
            st = torch.cuda.Stream(device=dev())
            with torch.cuda.stream(st):
                v0 = sol.reset({"x0": g(d["x0"])})
                args = ((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
                for _ in range(3):
                    r = sol(*args)
                out[k] = r.clone()
            st.synchronize()
        except Exception as e:   # surfaced in the main thread
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    assert not errs, errs
    for k in range(2):
        assert torch.equal(out[k], serial[k])


def _spi_yardstick(den, unet_params, d, sg, m, item):
    """Free-running SPI ADMM on one item: HIP vs the CPU fp32 oracle vs an fp64 run of the same oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import spi
    sol = spi.ADMMSolver_SPI(den)
    st = sol((sol.reset({"x0": g(d["x0"])}), (g(d["x0"]), g(d["K"]))), (g(sg), g(m)))
    hip = st[item:item + 1].double().cpu()
    sl = slice(item, item + 1)

    def run(dtype):
        c = lambda a: t(a[sl]).to(dtype)
        return O.spi_admm(O.Denoiser(unet_params, dtype=dtype), O.admm_reset(c(d["x0"])), c(d["x0"]), c(d["K"]), c(sg),
                          c(m)).double()

    o32, o64 = run(torch.float32), run(torch.float64)
    return rel(hip, o32), rel(hip, o64), rel(o32, o64), hip, o32


@pytest.mark.parametrize("B,H,W,T,item", [(2, 64, 64, 6, 1), (64, 512, 512, 5, 63)])
def test_spi_free_running_divergence_is_fp32_class(den, unet_params, B, H, W, T, item):
    """The 10-step bisection prox (tfpnp/utils/transforms.py:404-439) is a DISCONTINUOUS map: an fp32-ulp difference in
    its input flips a bracket decision at a few pixels (output quantum 1.1/2**10) and the UNet spreads each flip over
    its receptive field, so two CORRECT fp32 evaluations diverge when iterated freely.  Yardstick, on the golden SPI
    inputs and at the full BASELINE config #5 (64 x 512^2, >= 5 free iterations): the CPU fp32 oracle itself diverges
    from an fp64 run of the same code; the HIP path must be no further from either than that (it is not a bug hiding
    behind a loose bound), and z stays a valid bisection output."""
    d = synth.make_spi_batch(B, H, W, K=6, seed=51 if H == 64 else 88)
    rs = np.random.RandomState(52)
    if H == 64:
        sg = rs.uniform(15 / 255.0, 70 / 255.0, (B, T)).astype(np.float32)
        m = rs.uniform(50, 120, (B, T)).astype(np.float32)
    else:
        sg = np.full((B, T), 40 / 255.0, np.float32)
        m = np.full((B, T), 85.0, np.float32)
    e_hip32, e_hip64, e_cpu, hip, o32 = _spi_yardstick(den, unet_params, d, sg, m, item)
    frac = float(((hip[:, 1] - o32[:, 1]).abs() > 0).double().mean())
    print(f"SPI {B}x{H}x{W}, {T} free iterations, item {item}: rel-L2  HIP vs CPU-fp32 {e_hip32:.3e}   HIP vs fp64 "
          f"{e_hip64:.3e}   CPU-fp32 vs fp64 (yardstick) {e_cpu:.3e}   z pixels differing HIP/CPU {100 * frac:.3f} %")
    assert e_cpu > 0                                   # the yardstick itself is not zero: fp32 is not reproducible here
    assert e_hip64 <= 3.0 * e_cpu + 1e-6 and e_hip32 <= 3.0 * e_cpu + 1e-6
    assert float(hip[:, 1].min()) >= 0.0 and float(hip[:, 1].max()) <= 1.0


def test_radon_pair_vs_analytic_ellipses_full_config():
    """The HIP Radon pair at BASELINE config #4 (B=32, 256^2, 30 views, 363 detectors) against EXACT ellipse chords
    (the pin that replaces the unavailable torch_radon): forward rel-L2 < 1 %, adjoint mismatch 1e-5 on smooth images."""
    from tests.golden_inputs import ellipse_phantom, ellipse_sinogram
    from tfpnp_amd.utils import transforms as T
    B, R, V = 32, 256, 30
    radon = T.Radon_norm(R, V, device=dev())
    ells = [ellipse_phantom.make(100 + b) for b in range(B)]
    img = np.stack([ellipse_phantom.raster(e, R, ss=2) for e in ells])[:, None]
    sino = radon.forward(g(img))
    assert tuple(sino.shape) == (B, 1, V, 363)
    worst = 0.0
    for b in (0, 7, 31):
        ana = ellipse_sinogram(ells[b], radon.angles if hasattr(radon, "angles") else np.linspace(0, 179 / 180 * np.pi, V), 363)
        err = float(np.linalg.norm(sino[b, 0].cpu().numpy() - ana) / np.linalg.norm(ana))
        worst = max(worst, err)
    print(f"HIP radon forward vs exact ellipse chords (256^2, 30 views): worst rel-L2 {worst:.3e}")
    assert worst < 1e-2            # measured 0.4-0.7 %, like the oracle (tests/test_oracle_golden.py)
    x = g(img)
    lhs = float((sino.double() ** 2).sum())
    rhs = float((x.double() * radon.backprojection(sino).double()).sum())
    print(f"HIP radon adjoint mismatch on smooth images: {abs(lhs - rhs) / abs(lhs):.3e}")
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_two_devices_in_one_process(unet_params):
    """The boundary promises independent contexts "one per GPU / per thread" (include/pnpx.h), the way DataParallel
    would drive the reference (tfpnp/policy/sync_batchnorm/replicate.py:50-75): two host threads, two DEVICES, one
    process -- per-device kernel attributes and workspaces, identical results on both."""
    import threading
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    B, H, W = 3, 64, 64
    d = synth.make_csmri_batch(B, H, W, seed=300)
    a = csmri_actions(B, 4, 310)
    den = UNetDenoiser2D(state_dict=unet_params)          # one module, one native context per device
    out, errs = {}, []

    def work(idx):
        try:
            device = torch.device("cuda", idx)
            torch.cuda.set_device(device)
            to = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(device)
            sol = ADMMSolver_CSMRI(den)
            v0 = sol.reset({"x0": to(d["x0"])})
            for _ in range(3):
                r = sol((v0, (to(d["y0"]), to(d["mask"]))), (to(a["sigma_d"]), to(a["mu"])))
            torch.cuda.synchronize(device)
            out[idx] = r.cpu()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    assert torch.equal(out[0], out[1])


def test_repeated_calls_are_bit_identical(den):
    """Run-to-run determinism at the batch shapes that select different kernel instances (half tiles, 8-row tiles, launch chains,
    plain and XCD-grouped walks): every repeat of a denoiser forward and of a 5-iteration solver call equals the first one bit for
    bit.  (tools/determinism.py is the long version; a pipeline variant with a latent ordering hazard once passed every parity test
    and failed only this.)"""
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    sol = ADMMSolver_CSMRI(den)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    for (B, H) in [(1, 64), (1, 256), (3, 256), (6, 256), (12, 128), (5, 96)]:
        x, s = denoiser_inputs(B, H, H, 300 + B)
        x, s = g(x), g(s)
        d = synth.make_csmri_batch(B, H, H, seed=7)
        a = synth.make_actions(B)[0]
        v0 = sol.reset({"x0": g(d["x0"])})
        y0, m, sg, mu = g(d["y0"]), g(d["mask"]), g(a["sigma_d"]), g(a["mu"])
        ref, ref2 = den(x, s).clone(), sol((v0, (y0, m)), (sg, mu)).clone()
        for _ in range(12):
            assert torch.equal(den(x, s), ref), (B, H)
            assert torch.equal(sol((v0, (y0, m)), (sg, mu)), ref2), (B, H)


@pytest.mark.parametrize("chains", [1, 2])
def test_solver_call_is_hipgraph_capturable(den, chains):
    """One solver call (all inner iterations: denoiser launches, fused FFT passes, range-guard bookkeeping) records
    into a hipGraph (torch.cuda.graph) once its workspaces exist, and a replay reproduces the eager result bit for bit --
    i.e. the native path issues nothing but stream-ordered work (no hidden allocation, host read-back or default-stream
    operation) in steady state.  (Replay is not faster: the path is GPU-bound down to B=1, tools/graph_capture.py.)"""
    from tfpnp_amd.tasks import csmri
    sol = csmri.ADMMSolver_CSMRI(den)
    # chains = 2: the denoiser forward forks onto a side stream and joins back (fork / join events are capturable)
    den.context(dev()).set_option("chains", chains)
    B, H, W, T = 3, 64, 64, 4
    d = synth.make_csmri_batch(B, H, W, seed=311)
    a = csmri_actions(B, T, 312, ("sigma_d", "mu"))
    v0, y0, m = sol.reset({"x0": g(d["x0"])}), g(d["y0"]), g(d["mask"])
    sg, mu = g(a["sigma_d"]), g(a["mu"])
    with torch.no_grad():
        ref = sol((v0, (y0, m)), (sg, mu)).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sol((v0, (y0, m)), (sg, mu))
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = sol((v0, (y0, m)), (sg, mu))
        for _ in range(3):
            out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
        v0.copy_(sol.reset({"x0": g(d["x0"])}) * 0.5)       # new data through the same captured pointers
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, sol((v0, (y0, m)), (sg, mu)))
    den.context(dev()).set_option("chains", 0)
    den.context(dev()).status()


@pytest.mark.parametrize("B,H,W,S,T", [(3, 48, 80, 3, 2), (1, 50, 39, 5, 2), (5, 16, 16, 1, 3)])
def test_pr_ragged_sizes_vs_oracle(den, oden, B, H, W, S, T):
    """IADMMSolver_PR on non-square / non-power-of-two images and mask counts other than 4 (mixed-radix FFT passes,
    arbitrary S in the CDP reductions), every item against the CPU oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import pr
    d = synth.make_pr_batch(B, H, W, S=S, alpha=9.0, seed=500 + H)
    a = csmri_actions(B, T, 501 + H, ("sigma_d", "mu", "tau"))
    a["tau"] = (0.5 * a["tau"]).astype(np.float32)
    sol = pr.IADMMSolver_PR(den)
    st = sol((sol.reset({"x0": g(d["x0"])}), (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"]), g(a["tau"])))
    ref = O.pr_iadmm(oden, O.pr_reset(t(d["x0"])), t(d["y0"]), t(d["mask"]), t(a["sigma_d"]), t(a["mu"]), t(a["tau"]))
    assert st.shape == ref.shape and rel(st, ref) < 1e-4


@pytest.mark.parametrize("B,H,W,K", [(3, 48, 80, 4), (2, 64, 40, 8), (1, 18, 30, 6)])
def test_spi_ragged_sizes_and_K_vs_oracle(den, oden, B, H, W, K):
    """ADMMSolver_SPI with K in {4, 6, 8} on non-square images: one iteration (the bisection prox is discontinuous, see
    test_spi_golden), every item against the CPU oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import spi
    d = synth.make_spi_batch(B, H, W, K=K, seed=600 + K)
    rs = np.random.RandomState(601 + K)
    sg = rs.uniform(15 / 255.0, 70 / 255.0, (B, 1)).astype(np.float32)
    m = rs.uniform(50, 120, (B, 1)).astype(np.float32)
    sol = spi.ADMMSolver_SPI(den)
    st = sol((sol.reset({"x0": g(d["x0"])}), (g(d["x0"]), g(d["K"]))), (g(sg), g(m)))
    ref = O.spi_admm(oden, O.admm_reset(t(d["x0"])), t(d["x0"]), t(d["K"]), t(sg), t(m))
    assert st.shape == ref.shape and rel(st, ref) < 5e-4


@pytest.mark.parametrize("name,keys", [("hqs", ("sigma_d", "mu")), ("pg", ("sigma_d", "tau")),
                                       ("apg", ("sigma_d", "tau", "beta")), ("redadmm", ("sigma_d", "mu", "lamda"))])
def test_csmri_other_solvers_ragged_vs_oracle(den, oden, name, keys):
    """HQS / PG / APG / RED-ADMM on a non-power-of-two rectangle with iter_num < action_pack, against the CPU oracle."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.tasks import csmri
    B, H, W, T = 3, 48, 80, 4
    d = synth.make_csmri_batch(B, H, W, seed=700)
    a = csmri_actions(B, T + 1, 701, keys)
    sol = {"hqs": csmri.HQSSolver_CSMRI, "pg": csmri.PGSolver_CSMRI, "apg": csmri.APGSolver_CSMRI,
           "redadmm": csmri.REDADMMSolver_CSMRI}[name](den)
    v0 = sol.reset({"x0": g(d["x0"])})
    st = sol((v0, (g(d["y0"]), g(d["mask"]))), tuple(g(a[k]) for k in keys), iter_num=T)
    ref = getattr(O, "csmri_" + name)(oden, t(v0.cpu().numpy()), t(d["y0"]), t(d["mask"]), *[t(a[k][:, :T]) for k in keys])
    assert rel(st, ref) < 1e-4


def test_radon_forward_shape_sweep_vs_oracle():
    """The ray-driven projector on zero-bordered copies against the oracle over odd / tiny / large resolutions and view counts,
    with LOUD image borders (an off-by-one in the border handling or the loose sample interval would show at once)."""
    from oracle import pnp_oracle as O
    from tfpnp_amd.utils import transforms as T
    rs = np.random.RandomState(0)
    for (R, V, B) in [(16, 1, 1), (17, 3, 2), (31, 7, 1), (33, 30, 2), (50, 11, 3), (97, 13, 1), (128, 60, 1), (200, 9, 1), (300, 5, 1),
                      (40, 6, 11), (24, 4, 9)]:      # r5: images are dealt to XCDs by b % 8 -- batches that are no multiple of 8
        img = rs.rand(B, 1, R, R).astype(np.float32)
        img[:, :, :2, :], img[:, :, -2:, :], img[:, :, :, :2], img[:, :, :, -2:] = 5.0, -3.0, 7.0, -2.0
        angles, det = O.radon_geometry(R, V)
        ref = O.radon_forward(torch.from_numpy(img), angles, det)
        out = T.Radon_norm(R, V, device=dev()).forward(torch.from_numpy(img).to(dev())).cpu()
        assert float((out - ref).abs().max() / ref.abs().max()) < 2e-6, (R, V, B)


def test_fft_beside_a_running_denoiser_is_not_disturbed(unet_params):
    """Regression (r4): on this pool's MI355X boxes a wave executing packed-fp32 VALU instructions on a CU that also hosts another
    kernel's dense f16 MFMA wave computes wrong values in lanes 48-63 (tools/attic/stress_aggressor.py reproduces it without any library
    code).  The FFT passes of a second context used to be such victims whenever another context's conv_hs ran (60-90 % of the
    transforms wrong on affected boxes).  The library is now built without packed-fp32 instructions and conv_hs keeps LDS-using
    neighbours off its CUs: every transform computed beside 150 denoiser forwards must equal the solo result bit for bit."""
    import threading
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.utils import transforms as T
    for (B, H) in [(6, 256), (3, 64)]:
        den = UNetDenoiser2D(state_dict=unet_params)
        x = torch.rand(B, 1, H, H, device=dev())
        s = torch.full((B,), 0.1, device=dev())
        c = torch.randn(B, 1, H, H, 2, device=dev())
        ref_d, ref_f = den(x, s).clone(), T.fft2(c).clone()
        torch.cuda.synchronize()
        stop, bad, n = threading.Event(), [0, 0], [0, 0]

        def run_denoiser():
            st = torch.cuda.Stream(device=dev())
            with torch.cuda.stream(st):
                for _ in range(150):
                    n[0] += 1
                    bad[0] += int(not torch.equal(den(x, s), ref_d))
            st.synchronize()
            stop.set()

        def run_fft():
            st = torch.cuda.Stream(device=dev())
            with torch.cuda.stream(st):
                while not stop.is_set():
                    n[1] += 1
                    bad[1] += int(not torch.equal(T.fft2(c), ref_f))
            st.synchronize()

        ts = [threading.Thread(target=run_denoiser), threading.Thread(target=run_fft)]
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        print(f"  B={B} {H}x{H}: {bad[0]} of {n[0]} denoiser forwards, {bad[1]} of {n[1]} transforms differ from the solo results")
        assert bad == [0, 0] and n[1] > 20
