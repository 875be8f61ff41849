"""CPU-only tests: C-ABI surface, host-side logic (state packing, sharding, synthetic data).  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import ROOT
from tfpnp_amd import synth


def _header_functions():
    src = open(os.path.join(ROOT, "include", "pnpx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnpx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tfpnp_amd import _lib
    names = _header_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libpnpx.so does not export {n}"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, "python binding and include/pnpx.h disagree"
    l = _lib.lib()
    assert l.pnpx_unet_num_params() == synth.unet_num_params() == 11773857
    assert b"gfx950" in l.pnpx_version()
    assert l.pnpx_radon_det_count(256) == 363


def test_no_cpu_path():
    """The product path must fail loudly instead of falling back."""
    from tfpnp_amd import ops
    from tfpnp_amd._lib import PnpxError
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.utils import transforms as T
    with pytest.raises(PnpxError):
        T.fft2(torch.zeros(1, 1, 8, 8, 2))
    with pytest.raises(PnpxError):
        ops.psnr(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
    with pytest.raises(PnpxError):
        den(torch.zeros(1, 1, 16, 16), torch.zeros(1))
    with pytest.raises(ValueError):          # reference contract: no default checkpoint -> ValueError
        UNetDenoiser2D()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "tfpnp_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{fn} imports the oracle"


def test_state_packing_matches_reference_contract():
    from oracle import pnp_oracle as O
    from tfpnp_amd.pnp import solver as S
    from tfpnp_amd.tasks import csmri, pr, spi
    x0 = torch.randn(2, 1, 8, 8, 2)
    den = object()
    s = csmri.ADMMSolver_CSMRI(den)
    v = s.reset({"x0": x0})
    assert torch.equal(v, O.admm_reset(x0)) and s.num_var == 3
    assert torch.equal(s.get_output(v), x0[:, :, :, :, 0])
    act = {"sigma_d": 1, "mu": 2, "tau": 3, "beta": 4, "lamda": 5}
    assert s.filter_hyperparameter(act) == (1, 2)
    assert csmri.HQSSolver_CSMRI(den).reset({"x0": x0}).shape[1] == 2
    assert csmri.PGSolver_CSMRI(den).filter_hyperparameter(act) == (1, 3)
    assert csmri.APGSolver_CSMRI(den).filter_hyperparameter(act) == (1, 3, 4)
    assert csmri.REDADMMSolver_CSMRI(den).filter_hyperparameter(act) == (1, 2, 5)
    assert S.IADMMSolver(den).filter_hyperparameter(act) == (1, 2, 3)
    assert csmri.ADMMSolver_CSMRI(den).filter_aux_inputs({"y0": "a", "mask": "b"}) == ("a", "b")
    xr = torch.rand(2, 1, 8, 8)
    p = pr.IADMMSolver_PR(den)
    assert torch.equal(p.reset({"x0": xr}), O.pr_reset(xr))
    sp = spi.ADMMSolver_SPI(den)
    assert sp.reset({"x0": xr}).shape == (2, 3, 8, 8)
    assert sp.filter_aux_inputs({"x0": 1, "K": 2}) == (1, 2)
    with pytest.raises(NotImplementedError):
        csmri.create_solver_csmri(type("o", (), {"solver": "nope"})(), den)
    assert isinstance(csmri.create_solver_csmri(type("o", (), {"solver": "admm"})(), den), csmri.ADMMSolver_CSMRI)
    amp = S.AMPSolver(den)
    assert amp.reset({"x0": x0, "y0": x0 + 1}).shape[1] == 2


def test_synth_is_deterministic():
    a = synth.make_unet_params(0)
    b = synth.make_unet_params(0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert list(a) == [k for k, _ in synth.unet_param_specs()] and len(a) == 56
    assert synth.flatten_params(a).size == synth.unet_num_params()
    d1 = synth.make_csmri_batch(2, 32, 32)
    d2 = synth.make_csmri_batch(2, 32, 32)
    assert all(np.array_equal(d1[k], d2[k]) for k in d1)
    m = synth.radial_mask(128, 128, 4)
    assert 0.2 < m.mean() < 0.32 and m[64, 64]
    acts = synth.make_actions(3)
    assert len(acts) == 6 and acts[0]["sigma_d"].shape == (3, 5)
    assert abs(acts[0]["sigma_d"][0, 0] - 50 / 255) < 1e-6 and abs(acts[-1]["mu"][0, -1] - 0.9) < 1e-6


def test_shard_bounds():
    from tfpnp_amd.dist import shard_bounds, shard_batch
    for n, g in [(48, 8), (36, 8), (32, 4), (64, 8), (5, 2), (3, 4)]:
        cover = []
        for r in range(g):
            lo, hi = shard_bounds(n, g, r)
            cover += list(range(lo, hi))
        assert cover == list(range(n))
        sizes = [shard_bounds(n, g, r)[1] - shard_bounds(n, g, r)[0] for r in range(g)]
        assert max(sizes) - min(sizes) <= 1
    d = shard_batch({"a": np.arange(10), "b": torch.arange(10)}, 3, 1)
    assert list(d["a"]) == [4, 5, 6] and d["b"].tolist() == [4, 5, 6]


def test_fused_solvers_refuse_foreign_denoisers_and_prox_overrides_clearly():
    """every solver forward is one fused native call that applies the native denoiser itself.  A denoiser without
    a native context, or a subclass overriding prox_mapping (which the fused loop could only ignore), gets a clear
    NotImplementedError naming the way out -- not an opaque native error, not a silently skipped override."""
    import pytest
    import torch
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, PGSolver_CSMRI

    class Plain(torch.nn.Module):
        def forward(self, x, sigma):
            return x

    v = torch.zeros(1, 3, 8, 8, 2)
    aux = (torch.zeros(1, 1, 8, 8, 2), torch.ones(1, 1, 8, 8, dtype=torch.bool))
    par = (torch.full((1, 2), 0.1), torch.full((1, 2), 0.5))
    with pytest.raises(NotImplementedError, match="native denoiser"):
        ADMMSolver_CSMRI(Plain())((v, aux), par)

    class Mine(PGSolver_CSMRI):
        def prox_mapping(self, x, sigma):
            return x * 0.5

    with pytest.raises(NotImplementedError, match="overrides prox_mapping"):
        Mine(Plain())((v[:, :1], aux), par)


def test_every_context_option_is_documented_in_the_header():
    """pnpx_ctx_set_option's keys (csrc/api.hip) and the option list in include/pnpx.h must not drift apart."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "tfpnp_amd", "csrc", "api.hip")).read()
    header = open(os.path.join(root, "include", "pnpx.h")).read()
    keys = sorted(set(re.findall(r'is\("([a-z0-9_]+)"\)', api)))
    assert len(keys) >= 20
    missing = [k for k in keys if f'"{k}"' not in header]
    assert not missing, f"options accepted by pnpx_ctx_set_option but not described in include/pnpx.h: {missing}"
