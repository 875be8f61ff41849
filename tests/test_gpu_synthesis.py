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
