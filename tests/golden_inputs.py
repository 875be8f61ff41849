"""Seeded input generators shared by oracle/make_goldens.py (which runs the real reference on them)
and by the tests (which rebuild the same inputs and compare against tests/golden/*.npz)."""
import hashlib
import os

import numpy as np
import torch

from tfpnp_amd import synth

WEIGHT_SEED = 0


def sha(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()

# ---- shared seeded inputs (tests rebuild these with the same helpers) -------------------------------

def denoiser_inputs(B, H, W, seed):
    rs = np.random.RandomState(seed)
    gt = synth.phantom_batch(B, H, W, seed)
    sigma = rs.uniform(5, 50, B).astype(np.float32) / 255.0
    x = (gt + rs.standard_normal(gt.shape).astype(np.float32) * sigma[:, None, None, None]).astype(np.float32)
    return x, sigma

def complex_inputs(shape, seed):
    rs = np.random.RandomState(seed)
    return rs.standard_normal(tuple(shape) + (2,)).astype(np.float32)

def csmri_actions(B, T, seed, keys=("sigma_d", "mu")):
    rs = np.random.RandomState(seed)
    rng = {"sigma_d": (5 / 255.0, 50 / 255.0), "mu": (0.05, 0.95), "tau": (0.1, 1.5), "beta": (0.1, 1.0),
           "lamda": (0.1, 1.5)}
    return {k: rs.uniform(*rng[k], size=(B, T)).astype(np.float32) for k in keys}

def spi_grid(seed=3):
    """ztilde/K1/K/mu grid covering K1==0, interior roots, clamp-low and clamp-high branches."""
    rs = np.random.RandomState(seed)
    B, H, W = 3, 16, 16
    K = np.array([4, 6, 8], np.float32).reshape(B, 1, 1, 1)
    K1 = rs.randint(0, 1 + int(4 ** 2), size=(B, 1, H, W)).astype(np.float32)
    K1[:, :, :2] = 0.0                      # K1 == 0 branch
    K1 = np.minimum(K1, K ** 2)
    K1[1, 0, 5, :] = 36.0                   # saturated pixels (K0 == 0)
    zt = rs.uniform(-0.5, 1.5, size=(B, 1, H, W)).astype(np.float32)
    mu = np.array([50.0, 85.0, 120.0], np.float32).reshape(B, 1, 1, 1)
    return zt, K1, K, mu


# DRUNet fixtures (oracle/make_goldens.py::drunet_goldens): (B, H, W, seed); H, W multiples of 8, one non-square
DRUNET_CASES = [(2, 32, 32, 61), (2, 64, 64, 62), (1, 128, 128, 63), (2, 48, 80, 64)]


def drunet_spi_case():
    B, H, W, seed = 2, 64, 64, 71
    d = synth.make_spi_batch(B, H, W, K=6, seed=seed)
    rs = np.random.RandomState(seed + 1)
    sg = rs.uniform(15 / 255.0, 70 / 255.0, (B, 3)).astype(np.float32)
    m = rs.uniform(50, 120, (B, 3)).astype(np.float32)
    return d, sg, m


POLICY_SEED = 4242
ROLLOUT_CONTINUE_BIAS = 3.0   # added to the "continue" logit of the rollout actor so that the episode runs > 1 step


def policy_obs(B, C, H, W, seed):
    """Synthetic policy observation in [0,1] (float32 [B,C,H,W])."""
    return np.random.RandomState(seed).uniform(0, 1, (B, C, H, W)).astype(np.float32)


# ------------------------------------------------------------------------------------------- CT: analytic phantoms
class ellipse_phantom:
    """Random ellipse phantoms with EXACT line integrals (the chord of an ellipse has a closed form): the best available
    pin for the Radon pair, whose reference implementation (torch_radon) cannot be built here.
    Coordinates are centred pixel units: x = column - (R/2 - 0.5), y = row - (R/2 - 0.5)."""

    @staticmethod
    def make(seed, n=6, R=256):
        rs = np.random.RandomState(seed)
        ells = []
        for _ in range(n):
            a, b = rs.uniform(0.05, 0.2, 2) * R          # r + max(a, b) <= 0.45 R: every ellipse lies inside the image
            r = rs.uniform(0, 0.25) * R
            ph = rs.uniform(0, 2 * np.pi)
            ells.append(dict(cx=r * np.cos(ph), cy=r * np.sin(ph), a=a, b=b, phi=rs.uniform(0, np.pi),
                             rho=rs.uniform(0.2, 1.0)))
        return ells

    @staticmethod
    def raster(ells, R, ss=4):
        """Area-sampled (ss x ss sub-pixels) image [R,R] float32."""
        sub = (np.arange(R * ss) + 0.5) / ss - 0.5 - (R / 2 - 0.5)
        X, Y = np.meshgrid(sub, sub)
        img = np.zeros((R * ss, R * ss), np.float64)
        for e in ells:
            c, s = np.cos(e["phi"]), np.sin(e["phi"])
            u = (X - e["cx"]) * c + (Y - e["cy"]) * s
            v = -(X - e["cx"]) * s + (Y - e["cy"]) * c
            img += e["rho"] * ((u / e["a"]) ** 2 + (v / e["b"]) ** 2 <= 1.0)
        return img.reshape(R, ss, R, ss).mean(axis=(1, 3)).astype(np.float32)


def ellipse_sinogram(ells, angles, det):
    """Exact parallel-beam sinogram [V, det] of the phantom in the geometry of tfpnp/utils/transforms.py:487-491 (unit
    detector spacing, s = k - det/2 + 0.5): chord length 2ab*sqrt(r^2 - t^2)/r^2, r^2 = a^2 cos^2 + b^2 sin^2."""
    s = np.arange(det, dtype=np.float64) - det / 2 + 0.5
    out = np.zeros((len(angles), det), np.float64)
    for v, th in enumerate(np.asarray(angles, np.float64)):
        for e in ells:
            r2 = (e["a"] * np.cos(th - e["phi"])) ** 2 + (e["b"] * np.sin(th - e["phi"])) ** 2
            tt = s - (e["cx"] * np.cos(th) + e["cy"] * np.sin(th))
            out[v] += e["rho"] * 2 * e["a"] * e["b"] * np.sqrt(np.maximum(r2 - tt ** 2, 0.0)) / r2
    return out.astype(np.float32)


# ------------------------------------------------------------------ gradient cases
class GRAD_CASE:
    """The CS-MRI ADMM case whose gradients are pinned to the real reference's autograd (tests/golden/solver_grads.npz):
    B=1, 16x16, T=2 -- picked by the kink-margin search of tests/test_gpu_backward.py::test_csmri_solver_gradients
    (try k=1: no LeakyReLU / max-pool / clamp decision of either denoiser call within 1e-5 of its kink, re-checked by
    tests/test_oracle_golden.py), so that fp32 evaluations agree to rounding."""
    B, H, W, T = 1, 16, 16, 2
    data_seed, action_seed, wts_seed = 171, 172, 73
    # PnPEnv.forward reward gradient wrt policy logits (arbitrary inputs: kinks possible, loose bound)
    env_B, env_H, env_W, env_data_seed, env_raw_seed = 2, 32, 32, 95, 96


class _KinkProbe:
    """Stands in for torch.nn.functional inside the oracle and records how close any LeakyReLU input or max-pool
    decision comes to its kink, relative to the layer's mean magnitude."""

    def __init__(self):
        import torch.nn.functional as F
        self.F, self.margin = F, float("inf")

    def __getattr__(self, name):
        return getattr(self.F, name)

    def leaky_relu(self, x, slope):
        self.margin = min(self.margin, float(x.abs().min() / x.abs().mean()))
        return self.F.leaky_relu(x, slope)

    def relu(self, x):
        self.margin = min(self.margin, float(x.abs().min() / x.abs().mean()))
        return self.F.relu(x)

    def max_pool2d(self, x, k):
        B, C, H, W = x.shape
        win = x[:, :, :H // 2 * 2, :W // 2 * 2].reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5)
        top = win.reshape(B, C, H // 2, W // 2, 4).topk(2, dim=-1).values
        self.margin = min(self.margin, float((top[..., 0] - top[..., 1]).min() / x.abs().mean()))
        return self.F.max_pool2d(x, k)



def solver_kink_margin(run):
    """Smallest distance of any LeakyReLU / ReLU / max-pool / clamp / bisection decision from its kink over a whole fp64
    oracle run (every denoiser call of every inner iteration; UNet or DRUNet prox; the SPI Poisson prox's 10 sign tests)."""
    from oracle import pnp_oracle as O
    probe = _KinkProbe()
    keep = (O.F, O.denoise, O.drunet_denoise, O.spi_inverse)
    other = [float("inf")]

    def note(v):
        other[0] = min(other[0], float(v))

    def clamp01(pre):
        note(pre.abs().min())
        note((pre - 1).abs().min())
        return torch.clamp(pre, 0, 1)

    def noise_cat(x, sigma):
        N, _, H, W = x.shape
        return torch.cat([x, torch.ones(N, 1, H, W, dtype=x.dtype) * sigma.view(N, 1, 1, 1)], 1)

    def spi_inverse_probe(ztilde, K1, K, mu):          # oracle/pnp_oracle.py::spi_inverse with the decisions recorded
        K0 = K ** 2 - K1
        live = (K1 != 0).expand_as(ztilde)
        bmin = 1e-5 * torch.ones_like(ztilde)
        bmax = 1.1 * torch.ones_like(ztilde)
        bave = (bmin + bmax) / 2.0
        for _ in range(10):
            tmp = K1 / (torch.exp(bave) - 1) - mu * bave - K0 + mu * ztilde
            if bool(live.any()):
                note((tmp.abs() / mu)[live].min())      # in units of z
        z_lin = ztilde - (K0 / mu)
        if bool((~live).any()):
            note(z_lin[~live].abs().min())
            note((z_lin[~live] - 1).abs().min())
        return keep[3](ztilde, K1, K, mu)

    O.F = probe
    O.denoise = lambda x, sigma, params: clamp01(O.unet_forward(noise_cat(x, sigma), params))
    O.drunet_denoise = lambda x, sigma, params: clamp01(O.drunet_forward(noise_cat(x, sigma), params))
    O.spi_inverse = spi_inverse_probe
    try:
        with torch.no_grad():
            run()
    finally:
        O.F, O.denoise, O.drunet_denoise, O.spi_inverse = keep
    return min(probe.margin, other[0])


# ---- kink-free gradient cases (tests/golden/solver_grads_kinkfree.npz) -------------------------------------------------
# One small case per differentiable solver / denoiser whose whole fp64 trajectory keeps every non-smooth decision at least
# KINK_MARGIN away from its kink, so that independent fp32-class evaluations (the reference's autograd on the CPU, the native
# VJPs) share every decision and agree to rounding.  KINKFREE_TRY[name] = the first try index that qualifies, found by
# `python tests/golden_inputs.py` (search) and re-verified by tests/test_oracle_golden.py.
KINK_MARGIN = 1e-5
KINKFREE_KEYS = {"hqs": ("sigma_d", "mu"), "pg": ("sigma_d", "tau"), "apg": ("sigma_d", "tau", "beta"),
                 "redadmm": ("sigma_d", "mu", "lamda"), "pr": ("sigma_d", "mu", "tau"), "spi": ("sigma_d", "mu"),
                 "drunet": ("sigma",)}
KINKFREE_TRY = {"hqs": 0, "pg": 2, "apg": 1, "redadmm": 0, "pr": 1, "spi": 5, "drunet": 4}


def kinkfree_case(name, k=None):
    """Inputs of the kink-free gradient case `name` (try index k, default the frozen one): dict with the data tensors, `v0`
    (initial variables; the image for "drunet"), `acts` (list, KINKFREE_KEYS order) and `wts` (output weights of the loss)."""
    from oracle import pnp_oracle as O
    k = KINKFREE_TRY[name] if k is None else k
    keys = KINKFREE_KEYS[name]
    if name in ("hqs", "pg", "apg", "redadmm"):
        d = synth.make_csmri_batch(1, 16, 16, seed=1071 + 100 * k)
        a = csmri_actions(1, 2, 1072 + 100 * k, keys)
        if "beta" in a:
            a["beta"] = (0.3 * a["beta"]).astype(np.float32)
        x0 = d["x0"]
        nvar = {"hqs": 2, "pg": 1, "apg": 2, "redadmm": 3}[name]
        v0 = np.concatenate([x0] * nvar, 1)             # reset: x (and z / s) = x0.clone() ...
        if name == "redadmm":
            v0[:, 2] = 0                                #       ... u = 0 (tfpnp/pnp/solver/base.py:118-209)
        c = {"y0": d["y0"], "mask": d["mask"]}
    elif name == "pr":
        d = synth.make_pr_batch(1, 16, 16, S=4, alpha=9.0, seed=1075 + 100 * k)
        a = csmri_actions(1, 2, 1076 + 100 * k, keys)
        a["tau"] = (0.5 * a["tau"]).astype(np.float32)
        # the solver's own start (x = z = ones) is a constant image, whose denoised value sits ON the clamp's kink at 1: start
        # from a mid-grey estimate instead (the gradient identity under test does not care where the iteration starts)
        rs = np.random.RandomState(1077 + 100 * k)
        z = 0.2 + 0.6 * d["gt"] + 0.02 * rs.standard_normal(d["gt"].shape).astype(np.float32)
        zc = np.stack([z, 0.02 * rs.standard_normal(z.shape).astype(np.float32)], -1)
        v0 = np.concatenate([zc, zc, 0.02 * rs.standard_normal(zc.shape).astype(np.float32)], 1)
        c = {"y0": d["y0"], "mask": d["mask"]}
    elif name == "spi":
        d = synth.make_spi_batch(1, 16, 16, K=6, seed=1078 + 100 * k)
        rs = np.random.RandomState(1079 + 100 * k)
        a = {"sigma_d": rs.uniform(15 / 255.0, 70 / 255.0, (1, 1)).astype(np.float32),
             "mu": rs.uniform(50, 120, (1, 1)).astype(np.float32)}
        v0 = np.concatenate([d["x0"], d["x0"], 0.02 * rs.standard_normal(d["x0"].shape).astype(np.float32)], 1)
        c = {"x0": d["x0"], "K": d["K"]}
    elif name == "drunet":
        x, sigma = denoiser_inputs(1, 16, 16, 1061 + 100 * k)
        a = {"sigma": sigma}
        v0 = x
        c = {}
    else:
        raise KeyError(name)
    c.update(v0=np.ascontiguousarray(v0, np.float32), acts=[a[key] for key in keys],
             wts=np.random.RandomState(1073).standard_normal(v0.shape).astype(np.float32))
    return c


def kinkfree_oracle_run(name, c, dtype, den):
    """The oracle evaluation of case `c` as a function of (v0, *acts) -> output, in `dtype` (den: oracle denoiser of that dtype)."""
    from oracle import pnp_oracle as O
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    if name in ("hqs", "pg", "apg", "redadmm"):
        y0, m = tt(c["y0"]).to(dtype), tt(c["mask"])
        return lambda v, *p: getattr(O, "csmri_" + name)(den, v, y0, m, *p)
    if name == "pr":
        y0, m = tt(c["y0"]).to(dtype), tt(c["mask"]).to(dtype)
        return lambda v, *p: O.pr_iadmm(den, v, y0, m, *p)
    if name == "spi":
        x0, K = tt(c["x0"]).to(dtype), tt(c["K"]).to(dtype)
        return lambda v, *p: O.spi_admm(den, v, x0, K, *p)
    if name == "drunet":
        return lambda v, sigma: den(v, sigma)
    raise KeyError(name)


def kinkfree_margin(name, c, den64):
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    fn = kinkfree_oracle_run(name, c, torch.float64, den64)
    return solver_kink_margin(lambda: fn(tt(c["v0"]), *[tt(p) for p in c["acts"]]))


def env_case(task):
    """Inputs of the PR / SPI / CT environment fixtures (tests/golden/env_other_tasks.npz): (data dict, list of actions).
    3 items, max_episode_step 3; step 0 stops item 1, step 1 stops the first of the two remaining items."""
    B, H, W = 3, 32, 32
    rs = np.random.RandomState({"pr": 401, "spi": 402, "ct": 403}[task])
    stops = [np.array([0, 1, 0]), np.array([1, 0])]
    if task == "pr":
        d = synth.make_pr_batch(B, H, W, S=4, alpha=9.0, seed=411)
        d["sigma_n"] = (np.ones((B, 1, H, W)) * rs.uniform(0.02, 0.2, (B, 1, 1, 1))).astype(np.float32)
        acts = []
        for s, stop in enumerate(stops):
            a = csmri_actions(len(stop), 2, 412 + s, ("sigma_d", "mu", "tau"))
            a["tau"] = (0.5 * a["tau"]).astype(np.float32)
            a["idx_stop"] = stop
            acts.append(a)
        return d, acts
    if task == "spi":
        d = synth.make_spi_batch(B, H, W, K=6, seed=421)
        acts = []
        for s, stop in enumerate(stops):
            n = len(stop)
            acts.append({"sigma_d": rs.uniform(15 / 255.0, 70 / 255.0, (n, 1)).astype(np.float32),
                         "mu": rs.uniform(50, 120, (n, 1)).astype(np.float32), "idx_stop": stop})
        return d, acts
    if task == "ct":
        V, det = 20, 47
        d = {"gt": rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32),
             "y0": rs.uniform(0, 5, (B, 1, V, det)).astype(np.float32),
             "ATy0": rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32),
             "view": np.full((B, 1, H, W), V / 120.0, np.float32),
             "sigma_n": (np.ones((B, 1, H, W)) * rs.uniform(0.01, 0.1, (B, 1, 1, 1))).astype(np.float32)}
        d["x0"] = d["ATy0"].copy()
        d["output"] = d["ATy0"].copy()
        return d, []
    raise KeyError(task)


# ---- measurement-synthesis fixtures (oracle/make_goldens.py::synthesis_goldens) ---------------------------------------
SYNTH_SEED = 9100


def synthesis_images(n=2, H=64, W=64):
    """uint8 grey images written to a temporary folder for the reference's Dataset classes / rebuilt by the tests."""
    return np.round(synth.phantom_batch(n, H, W, 515)[:, 0] * 255).astype(np.uint8)


def synthesis_masks(H=64, W=64):
    """(CS-MRI radial mask [H,W] bool, CDP masks [S,H,W,2] unit modulus)"""
    m = synth.make_csmri_batch(1, H, W, ratio=4, seed=516)["mask"][0, 0].astype(bool)
    ph = np.random.RandomState(517).uniform(0, 2 * np.pi, (4, H, W)).astype(np.float32)
    return m, np.stack([np.cos(ph), np.sin(ph)], -1).astype(np.float32)


def noise_model_inputs():
    rs = np.random.RandomState(518)
    return (rs.standard_normal((3, 1, 16, 24, 2)).astype(np.float32),       # k-space-like (GaussianModelC / D)
            rs.uniform(0, 2, (3, 4, 16, 24)).astype(np.float32),            # magnitudes (PoissonModel)
            rs.uniform(0, 40, (3, 1, 30, 45)).astype(np.float32))           # sinogram-like (GaussianModelP)


if __name__ == "__main__":      # search: first qualifying try per case (paste into KINKFREE_TRY)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pnp_oracle as O
    unet64 = O.Denoiser(synth.make_unet_params(WEIGHT_SEED), torch.float64)
    dru64 = O.DRUNetDenoiser(synth.make_drunet_params(WEIGHT_SEED), torch.float64)
    found = {}
    for name in (sys.argv[1:] or KINKFREE_KEYS):
        for k in range(200):
            m = kinkfree_margin(name, kinkfree_case(name, k), dru64 if name == "drunet" else unet64)
            if m > KINK_MARGIN:
                found[name] = k
                print(f"{name}: try {k} margin {m:.2e}", flush=True)
                break
        else:
            print(f"{name}: none found")
    print("KINKFREE_TRY =", found)
