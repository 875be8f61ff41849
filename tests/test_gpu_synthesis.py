"""GPU tests of SURVEY 8(f) rank 3: batched measurement synthesis (tfpnp_amd/data/synthesis.py) -- deterministic parts
against the CPU oracle, noise against its distribution (RNG parity with the reference is statistical by design)."""
import numpy as np
import pytest
import torch

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


def gen(seed):
    return torch.Generator(device=dev()).manual_seed(seed)


def test_csmri_measure():
    from oracle import pnp_oracle as O
    from tfpnp_amd.data import synthesis as S
    B, H, W = 4, 64, 96
    gt = synth.phantom_batch(B, H, W, 5)
    mask = np.stack([synth.radial_mask(H, W, 4, seed=7 + b) for b in range(B)])[:, None]
    d = S.csmri_measure(g(gt), g(mask))
    assert set(d) == {'y0', 'x0', 'ATy0', 'gt', 'mask', 'sigma_n', 'output', 'input'}
    clean = O.fft2c(O.real2complex(t(gt))) * t(mask).unsqueeze(-1)
    assert rel(d['y0'], clean) < 1e-5
    assert rel(d['ATy0'], O.ifft2c(clean)) < 1e-5 and torch.equal(d['x0'], d['ATy0'])
    assert rel(d['output'], O.ifft2c(clean)[..., 0]) < 1e-5
    assert d['mask'].dtype == torch.bool and float(d['sigma_n'].abs().max()) == 0.0
    assert d['sigma_n'].shape == d['y0'].shape
    # noisy: N(0, (15/255)^2) on the sampled k-space entries only; reproducible from the generator
    n1 = S.csmri_measure(g(gt), g(mask), S.GaussianModelD([15]), generator=gen(3))
    n2 = S.csmri_measure(g(gt), g(mask), S.GaussianModelD([15]), generator=gen(3))
    assert torch.equal(n1['y0'], n2['y0'])
    r = (n1['y0'].cpu() - clean)
    m = t(mask).unsqueeze(-1).expand_as(r)
    assert float(r[~m].abs().max()) == 0.0
    assert abs(float(r[m].std()) / (15 / 255) - 1) < 0.03 and abs(float(r[m].mean())) < 2e-3
    assert np.allclose(n1['sigma_n'].cpu().numpy(), 15 / 255)
    # feeds the env directly
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
    env = CSMRIEnv(None, ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=synth.make_unet_params(1))), 6)
    ob = env.reset(n1)
    assert tuple(env.get_policy_ob(ob).shape) == (B, 9, H, W)


def test_pr_measure():
    from oracle import pnp_oracle as O
    from tfpnp_amd.data import synthesis as S
    B, H, W, Sm = 3, 64, 64, 4
    d0 = synth.make_pr_batch(B, H, W, S=Sm, alpha=9.0, seed=11)
    gt, mask = d0["gt"], d0["mask"]
    d = S.pr_measure(g(gt), g(mask))
    z = O.complex_abs(O.cdp_forward(O.real2complex(t(gt)), t(mask)))
    assert rel(d['y0'], z) < 1e-5 and float((d['x0'] - 1).abs().max()) == 0.0
    n = S.pr_measure(g(gt), g(mask), S.PoissonModel([9]), generator=gen(4))
    y = n['y0'].cpu()
    # y^2 - z^2 = alpha/255 * |z| * N(0,1) (where not clamped): unit variance after normalisation
    u = ((y ** 2 - z ** 2) / (9 / 255 * z.abs()))[z > 0.2]
    assert abs(float(u.std()) - 1) < 0.03 and abs(float(u.mean())) < 0.02
    assert n['sigma_n'].shape == d['x0'].shape
    assert np.allclose(n['sigma_n'][:, 0, 0, 0].cpu().numpy(), (y - z).reshape(B, -1).std(dim=1).numpy(), rtol=1e-4)


def test_spi_measure():
    from tfpnp_amd.data import synthesis as S
    B, H, W, K = 4, 64, 64, 6
    gt = synth.phantom_batch(B, H, W, 21)
    d = S.spi_measure(g(gt), K, generator=gen(5))
    x0 = d['x0'].cpu().numpy()
    assert x0.shape == gt.shape and np.allclose(d['K'].cpu().numpy(), K / 10)
    assert np.allclose(x0 * K * K, np.round(x0 * K * K), atol=1e-4)          # counts of fired sub-pixels / K^2
    # a sub-pixel fires with probability 1 - exp(-gt): compare batch means (B*H*W*K^2 ~ 6e5 Bernoulli draws)
    expect = 1 - np.exp(-gt.astype(np.float64))
    assert abs(x0.mean() - expect.mean()) < 2e-3
    hi = gt > 0.6
    assert abs(x0[hi].mean() - expect[hi].mean()) < 6e-3


def test_ct_measure_and_fbp():
    from oracle import pnp_oracle as O
    from tfpnp_amd.data import synthesis as S
    from tfpnp_amd.utils import transforms as T
    B, R, V = 2, 64, 45
    gt = synth.phantom_batch(B, R, R, 31)
    rg = T.RadonGenerator()
    d = S.ct_measure(g(gt), V, rg)
    angles, det = O.radon_geometry(R, V)
    sino = O.radon_forward(t(gt), angles, det)
    assert rel(d['y0'], sino) < 1e-5
    opn = rg.opnorms[(R, V)]
    assert rel(d['ATy0'], O.radon_backprojection(sino, angles, R) / opn ** 2) < 1e-5
    assert np.allclose(d['view'].cpu().numpy(), V / 120)
    # FBP against a numpy restatement of filter_sinogram + the oracle backprojection
    L = max(64, 1 << int(np.ceil(np.log2(2 * det))))
    pad = np.zeros((B, 1, V, L), np.float64)
    pad[..., :det] = sino.numpy()
    filt = np.real(np.fft.ifft(np.fft.fft(pad, axis=-1) * T.ramp_filter(L).astype(np.float64), axis=-1))[..., :det]
    filt = filt * (np.pi / (2 * V))
    radon = rg(R, V, device=dev())
    assert rel(radon.filter_sinogram(g(sino.numpy())), t(filt)) < 1e-5
    fbp = O.radon_backprojection(t(filt.astype(np.float32)), angles, R)
    assert rel(d['x0'], fbp) < 1e-4
    # the ramp filter is what makes backprojection an approximate inverse: FBP correlates with the phantom far better
    # than the unfiltered backprojection does
    def corr(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    x0, aty = d['x0'].cpu().numpy(), d['ATy0'].cpu().numpy()
    assert corr(x0, gt) > 0.9 and corr(x0, gt) > corr(aty, gt)
    # percent noise, per item
    n = S.ct_measure(g(gt), V, rg, S.GaussianModelP([0.05]), generator=gen(6))
    r = n['y0'].cpu() - sino
    for b in range(B):
        assert abs(float(r[b].std()) / (0.05 * float(sino[b].abs().mean())) - 1) < 0.05
    assert np.allclose(n['sigma_n'].cpu().numpy(), 0.05)


# ------------------------------------------------------------------------------------------------------------------
# Pinned to the reference's own code: tests/golden/synthesis_ref.npz holds what tfpnp/utils/noise.py and
# Dataset.__getitem__ of tasks/{csmri,pr,spi}/dataset.py produced from a temporary image folder under fixed seeds
# (oracle/make_goldens.py::synthesis_goldens).  The CPU random fields are re-drawn here from the same seeds, in the order
# the reference consumes them, and injected into the GPU synthesis: everything else must match to 1e-6.
def _ref():
    from tests.conftest import golden
    from tests.golden_inputs import sha, synthesis_images, synthesis_masks, noise_model_inputs
    gd = golden("synthesis_ref")
    imgs = synthesis_images()
    mask, cdp = synthesis_masks()
    assert (sha(imgs, mask, cdp, *noise_model_inputs()) == gd["in_sha"]).all()
    gt = (imgs.astype(np.float32) / 255.0)[:, None]                     # np.array(PIL 'L' image) / 255  (dataset.py:41)
    return gd, gt, mask, cdp


def _seed(k):
    from tests.golden_inputs import SYNTH_SEED
    np.random.seed(SYNTH_SEED + k)
    torch.manual_seed(SYNTH_SEED + k)


def test_csmri_items_match_reference_dataset():
    from tfpnp_amd.data import synthesis as S
    gd, gt, mask, _ = _ref()
    for i in range(gt.shape[0]):
        d = S.csmri_measure(g(gt[i:i + 1]), g(mask[None, None]))
        for k in ("y0", "x0", "ATy0", "output"):
            assert rel(d[k][0], t(gd[f"csmri_clean{i}_{k}"])) < 1e-6, k
        assert torch.equal(d["gt"][0].cpu(), t(gd[f"csmri_clean{i}_gt"]))
        assert torch.equal(d["mask"][0].cpu(), t(gd[f"csmri_clean{i}_mask"]))
        assert float(d["sigma_n"].abs().max()) == 0.0 and d["sigma_n"][0].shape == gd[f"csmri_clean{i}_sigma_n"].shape
    # noisy item: the reference draws  mask index (numpy), sigma (numpy), then torch.randn(*y0.shape)
    _seed(1)
    np.random.randint(0, 1)
    sigma = np.random.choice([5, 10, 15])
    noise = torch.randn(1, 64, 64, 2)
    d = S.csmri_measure(g(gt[1:2]), g(mask[None, None]), noise_model=S.GaussianModelD([sigma]), noise=noise[None])
    for k in ("y0", "x0", "ATy0", "output"):
        assert rel(d[k][0], t(gd[f"csmri_noisy1_{k}"])) < 1e-6, k
    assert rel(d["sigma_n"][0], t(gd["csmri_noisy1_sigma_n"]).float()) < 1e-6


def test_pr_and_spi_items_match_reference_datasets():
    from tfpnp_amd.data import synthesis as S
    gd, gt, _, cdp = _ref()
    d = S.pr_measure(g(gt[0:1]), g(cdp[None]))
    assert rel(d["y0"][0], t(gd["pr_clean0_y0"])) < 1e-6
    assert torch.equal(d["x0"][0].cpu(), t(gd["pr_clean0_x0"])) and torch.equal(d["output"][0].cpu(), t(gd["pr_clean0_output"]))
    _seed(2)
    np.random.randint(0, 1)
    alpha = np.random.choice([9, 27, 81])
    noise = torch.randn(4, 64, 64)
    d = S.pr_measure(g(gt[1:2]), g(cdp[None]), noise_model=S.PoissonModel([alpha]), noise=noise[None])
    assert rel(d["y0"][0], t(gd["pr_noisy1_y0"])) < 2e-6
    assert rel(d["sigma_n"][0], t(gd["pr_noisy1_sigma_n"]).float()) < 1e-4            # std of the residual (one item)
    # SPI: K index (numpy), then torch.poisson on the CPU generator
    _seed(3)
    K = [4, 6, 8][np.random.randint(0, 3)]
    theta = S.spi_theta(t(gt[0:1]), K, K ** 2)
    counts = torch.poisson(theta)
    d = S.spi_measure(g(gt[0:1]), K, counts=counts)
    assert torch.equal(d["x0"][0].cpu(), t(gd["spi0_x0"])) and torch.equal(d["output"][0].cpu(), t(gd["spi0_output"]))
    assert np.allclose(d["K"][0].cpu().numpy(), gd["spi0_K"]) and float(gd["spi0_K"].flat[0]) == K / 10


def test_noise_models_match_reference_noise_py():
    from tfpnp_amd.data import synthesis as S
    from tests.golden_inputs import noise_model_inputs
    gd, *_ = _ref()
    xk, xm, xs = noise_model_inputs()
    _seed(4)
    s = np.random.uniform(0, 55)
    y, sig = S.GaussianModelC(0, 55)(g(xk), noise=torch.randn(*xk.shape), sigma=np.full(3, s / 255.))
    assert rel(y, t(gd["noiseC_y"])) < 1e-6 and abs(float(sig[0]) - float(gd["noiseC_sigma"])) < 1e-7
    _seed(5)
    s = np.random.choice([5, 10, 15])
    y, sig = S.GaussianModelD([s])(g(xk), idx=0, noise=torch.randn(*xk.shape))
    assert rel(y, t(gd["noiseD_y"])) < 1e-6 and abs(float(sig[0]) - float(gd["noiseD_sigma"])) < 1e-7
    _seed(6)
    a = np.random.choice([9, 27, 81])
    y, _ = S.PoissonModel([a])(g(xm), idx=0, noise=torch.randn(*xm.shape))
    assert rel(y, t(gd["noisePo_y"])) < 2e-6
    assert abs(float((y.cpu() - t(xm).abs()).std()) - float(gd["noisePo_sigma"])) < 1e-5   # the reference's whole-tensor std
    _seed(7)
    sp = np.random.choice([0.05, 0.075, 0.1], size=3)
    y, sig = S.GaussianModelP([0.05])(g(xs), noise=torch.randn(*xs.shape), sigma=sp)
    assert rel(y, t(gd["noiseP_y"])) < 1e-6 and rel(sig.reshape(-1), t(gd["noiseP_sigma"]).reshape(-1)) < 1e-7
