"""GPU parity of the DRUNet denoiser (csrc/drunet.hip) -- the only unbuilt item of SURVEY section 8 after round 2
(VERDICT r2 #5).  Fixtures come from a model assembled from the REFERENCE'S OWN basicblock.py parts
(oracle/ref_shim.py::make_drunet, tests/golden/drunet_*.npz); the bar is the north star's 1e-4 relative L2, observed
~1e-6 (fp32-class arithmetic: half-split f16x3 MFMA with fp32 accumulation)."""
import numpy as np
import pytest
import torch

from oracle import pnp_oracle as O          # checker only
from tests.conftest import golden
from tests.golden_inputs import DRUNET_CASES, denoiser_inputs, drunet_spi_case, sha
from tfpnp_amd import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def drunet():
    from tfpnp_amd.pnp import DRUNetDenoiser2D
    return DRUNetDenoiser2D(state_dict=synth.make_drunet_params(0))


@pytest.mark.parametrize("B,H,W,seed", DRUNET_CASES)
def test_drunet_golden(drunet, B, H, W, seed):
    g = golden(f"drunet_B{B}_{H}x{W}")
    x, sigma = denoiser_inputs(B, H, W, seed)
    assert (sha(x, sigma) == g["in_sha"]).all()
    post, pre = drunet.forward_preclamp(t(x).to(dev()), t(sigma).to(dev()))
    e_pre, e_post = rel(pre, g["pre"]), rel(post, g["post"])
    print(f"DRUNet {B}x{H}x{W}: pre-clamp {e_pre:.2e}  clamped {e_post:.2e}")
    assert e_pre < 2e-5 and e_post < 2e-5
    assert torch.equal(drunet(t(x).to(dev()), t(sigma).to(dev())), post)          # deterministic, same entry


def test_drunet_full_size_vs_oracle(drunet):
    """256 x 256 (the headline image size) and a ragged batch against the CPU oracle; batch-order independence."""
    x, sigma = denoiser_inputs(3, 256, 256, 65)
    with torch.no_grad():
        want = O.drunet_forward(torch.cat([t(x), torch.ones(3, 1, 256, 256) * t(sigma).view(3, 1, 1, 1)], 1),
                                synth.make_drunet_params(0))
    post, pre = drunet.forward_preclamp(t(x).to(dev()), t(sigma).to(dev()))
    assert rel(pre, want) < 2e-5
    perm = torch.tensor([2, 0, 1])
    post2, pre2 = drunet.forward_preclamp(t(x)[perm].to(dev()), t(sigma)[perm].to(dev()))
    assert torch.equal(pre2.cpu(), pre.cpu()[perm])
    one, pre1 = drunet.forward_preclamp(t(x)[1:2].to(dev()), t(sigma)[1:2].to(dev()))   # B = 1 after B = 3: arena reuse
    assert torch.equal(pre1.cpu(), pre.cpu()[1:2])


def test_drunet_is_the_prox_of_the_native_solver_loops(drunet):
    """BASELINE config #5's pairing: the native SPI ADMM loop (pnpx_spi_admm) with the DRUNet as its denoiser prox, every
    iteration teacher-forced against the reference's ADMMSolver_SPI run with the reference-assembled DRUNet."""
    from tfpnp_amd.tasks.spi import ADMMSolver_SPI
    g = golden("drunet_spi_B2_64x64")
    d, sg, m = drunet_spi_case()
    assert (sha(d["x0"], sg, m) == g["in_sha"]).all()
    sol = ADMMSolver_SPI(drunet)
    x0, K = t(d["x0"]).to(dev()), t(d["K"]).to(dev())
    v = sol.reset({"x0": x0})
    for i in range(sg.shape[1]):
        prev = t(g[f"admm_step{i}"]).to(dev()) if i else v
        v = sol((prev, (x0, K)), (t(sg[:, i:i + 1]).to(dev()), t(m[:, i:i + 1]).to(dev())))
        want = t(g[f"admm_step{i + 1}"])
        x_err = rel(v[:, :1], want[:, :1])
        zu_exact = float((v[:, 1:].cpu() == want[:, 1:]).float().mean())
        print(f"SPI+DRUNet iteration {i + 1}: x rel {x_err:.2e}, z/u bit-equal fraction {zu_exact:.5f}")
        assert x_err < 1e-4 and zu_exact > 0.995


def test_drunet_full_config5_iteration_runs_and_matches_oracle_slice(drunet):
    """One SPI ADMM iteration at BASELINE config #5's image size (512 x 512; batch 8 of the 64) with the DRUNet prox; two
    items checked against the CPU oracle."""
    from tfpnp_amd.tasks.spi import ADMMSolver_SPI
    B, H, W = 8, 512, 512
    d = synth.make_spi_batch(B, H, W, K=6, seed=81)
    sg = np.full((B, 1), 40 / 255.0, np.float32)
    m = np.full((B, 1), 85.0, np.float32)
    sol = ADMMSolver_SPI(drunet)
    x0, K = t(d["x0"]).to(dev()), t(d["K"]).to(dev())
    v = sol((sol.reset({"x0": x0}), (x0, K)), (t(sg).to(dev()), t(m).to(dev())))
    assert torch.isfinite(v).all()
    with torch.no_grad():
        x0c = t(d["x0"][:2])
        want = O.spi_admm(O.DRUNetDenoiser(synth.make_drunet_params(0)), O.admm_reset(x0c), x0c, t(d["K"][:2]), t(sg[:2]),
                          t(m[:2]))
    assert rel(v[:2, :1], want[:, :1]) < 1e-4


def test_drunet_launch_chains_are_bit_identical(drunet):
    """The DRUNet forward sliced into independent launch chains (option `chains`; automatic for the batch sizes between round
    boundaries, as for the UNet): per-image results do not depend on the slicing."""
    ctx = drunet.context(dev())
    try:
        for (B, H, W) in [(5, 64, 64), (9, 32, 48), (3, 128, 64), (6, 256, 256)]:
            x, s = denoiser_inputs(B, H, W, 77)
            x, s = t(x).to(dev()), t(s).to(dev())
            ctx.set_option("chains", 1)
            ref = drunet(x, s).clone()
            for c in (2, 3, 0):
                ctx.set_option("chains", c)
                assert torch.equal(drunet(x, s), ref), (B, H, W, c)
    finally:
        ctx.set_option("chains", 0)


def test_drunet_contract_errors(drunet):
    from tfpnp_amd._lib import PnpxError
    from tfpnp_amd.pnp import DRUNetDenoiser2D, create_denoiser
    x = torch.rand(1, 1, 36, 40, device=dev())           # 36 is not a multiple of 8
    with pytest.raises(PnpxError, match="multiples of 8"):
        drunet(x, torch.full((1,), 0.1, device=dev()))
    xg = torch.rand(1, 1, 32, 32, device=dev(), requires_grad=True)
    with pytest.raises(NotImplementedError):
        drunet(xg, torch.full((1,), 0.1, device=dev()))
    with pytest.raises(ValueError):
        DRUNetDenoiser2D()
    bad = dict(synth.make_drunet_params(0))
    bad.pop("m_tail.weight")
    with pytest.raises(PnpxError, match="missing"):
        DRUNetDenoiser2D(state_dict=bad).context(dev())

    class Opt:
        denoiser = "drunet"
    assert isinstance(create_denoiser(Opt(), state_dict=synth.make_drunet_params(0)), DRUNetDenoiser2D)


def test_drunet_range_overflow_is_loud():
    """The DRUNet has no exact-fp32 fallback: if an activation leaves the half-split range the guard latches the context
    and every later denoiser call FAILS (PNPX_ERR_RANGE) instead of returning possibly invalid values."""
    from tfpnp_amd._lib import PnpxError
    from tfpnp_amd.pnp import DRUNetDenoiser2D
    hot = {k: np.array(v, copy=True) for k, v in synth.make_drunet_params(0).items()}
    hot["m_head.weight"] = (hot["m_head.weight"] * 1e4).astype(np.float32)
    den = DRUNetDenoiser2D(state_dict=hot)
    x = torch.rand(2, 1, 64, 64, device=dev())
    s = torch.full((2,), 0.1, device=dev())
    den(x, s)                                   # trips the guard (not visible yet: no synchronisation in the default mode)
    torch.cuda.synchronize()
    with pytest.raises(PnpxError, match="range guard"):
        den.context(dev()).status()
    with pytest.raises(PnpxError, match="half-split"):
        den(x, s)
