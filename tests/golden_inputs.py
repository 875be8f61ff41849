"""Seeded input generators shared by oracle/make_goldens.py (which runs the real reference on them)
and by the tests (which rebuild the same inputs and compare against tests/golden/*.npz)."""
import hashlib

import numpy as np

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


POLICY_SEED = 4242
ROLLOUT_CONTINUE_BIAS = 3.0   # added to the "continue" logit of the rollout actor so that the episode runs > 1 step


def policy_obs(B, C, H, W, seed):
    """Synthetic policy observation in [0,1] (float32 [B,C,H,W])."""
    return np.random.RandomState(seed).uniform(0, 1, (B, C, H, W)).astype(np.float32)
