"""Bounded numerics probe (VERDICT r5 next #2d): would Winograd F(4x4,3x3) on the 16^2 / 32^2 levels of the fp32 family stay inside the
1e-4 drift bar?  CPU only -- an fp32 emulation of the transform-domain arithmetic (torch, fp32 tensors; transformed weights made in fp64
and rounded once, as conv3x3_wino8.hip's packer does), no kernel is built before this says yes.

  (1) per-layer error vs an fp64 convolution on the real activations of the He-scaled synthetic UNet at 2 x 256 x 256: direct fp32,
      F(2x2,3x3) (what conv_mode 0 runs today), F(4x4,3x3) with the textbook points (0, +-1, +-2, inf) and with (0, +-1, +-1/2, inf);
  (2) the 30-iteration CS-MRI ADMM drift table of tests/test_gpu_modes.py::test_csmri_episode_drift_not_worse_than_fp32 (B = 2, 64 x 64,
      12 seeds) with every cout % 32 == 0 layer emulated as F(2x2,3x3) -- the stand-in for today's fp32 family, to be compared with
      profiles/r5_drift_seeds.md's conv_mode 0 rows -- and the same with the two deepest levels as F(4x4,3x3).

usage: python tools/wino_f4_probe.py [n_seeds] > profiles/r6_wino_f4_probe.md"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import pnp_oracle as O
from tfpnp_amd import synth


def cook_toom(m, r, pts):
    """1-D F(m, r): y = A^T [(G g) * (B^T d)], finite points `pts` (n - 1 of them) + infinity; fp64 matrices.
    Rows of B^T are scaled to the product of point differences (integer entries for integer points), G rows by its inverse."""
    n = m + r - 1
    assert len(pts) == n - 1
    V = np.zeros((n, n))
    for j, p in enumerate(pts):
        V[j] = [p ** i for i in range(n)]
    V[n - 1, n - 1] = 1.0
    Bt = np.linalg.inv(V).T
    G = V[:, :r].copy()
    G[n - 1] = 0
    G[n - 1, r - 1] = 1.0
    At = V[:, :m].T.copy()
    At[:, n - 1] = 0
    At[m - 1, n - 1] = 1.0
    for j in range(n - 1):
        s = np.prod([pts[j] - q for k, q in enumerate(pts) if k != j])
        Bt[j] *= s
        G[j] /= s
    return At, G, Bt


def wino_conv(x, w, b, m, mats):
    """3x3 pad-1 convolution as Winograd F(m x m, 3x3) in fp32 (x, w fp32; transformed weights from fp64)."""
    At, G, Bt = mats
    n = m + 2
    Bsz, C, H, W = x.shape
    K = w.shape[0]
    assert H % m == 0 and W % m == 0
    U = torch.from_numpy(np.einsum("ia,kcab,jb->ijkc", G, w.double().numpy(), G)).float()        # [n, n, K, C]
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, n, m).unfold(3, n, m)                     # [B, C, th, tw, n, n]
    Btt = torch.from_numpy(Bt).float()
    Att = torch.from_numpy(At).float()
    V = torch.einsum("ia,bcyxaz->bcyxiz", Btt, t)              # fp32 throughout
    V = torch.einsum("jz,bcyxiz->bcyxij", Btt, V)
    M = torch.einsum("ijkc,bcyxij->bkyxij", U, V)
    Y = torch.einsum("pi,bkyxij->bkyxpj", Att, M)
    Y = torch.einsum("qj,bkyxpj->bkyxpq", Att, Y)              # [B, K, th, tw, m, m]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bsz, K, H, W)
    return Y + b.view(1, -1, 1, 1)


MATS = {
    "F2": (2, cook_toom(2, 3, [0.0, 1.0, -1.0])),
    "F4 (0,+-1,+-2,inf)": (4, cook_toom(4, 3, [0.0, 1.0, -1.0, 2.0, -2.0])),
    "F4 (0,+-1,+-1/2,inf)": (4, cook_toom(4, 3, [0.0, 1.0, -1.0, 0.5, -0.5])),
}

rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())


def layer_table(params):
    """Per-layer error on the activations the deep layers actually see (one forward of the fp64 oracle at 2 x 256 x 256)."""
    from tests.golden_inputs import denoiser_inputs
    x, s = denoiser_inputs(2, 256, 256, 7)
    p64 = {k: torch.as_tensor(v).double() for k, v in params.items()}
    acts = {}
    orig = F.conv2d

    def spy(inp, w, b=None, **kw):
        if w.shape[-1] == 3 and inp.shape[-1] in (16, 32):
            acts[len(acts)] = (inp.clone(), w.clone(), b.clone())
        return orig(inp, w, b, **kw)

    F.conv2d = spy
    try:
        with torch.no_grad():
            sig = torch.from_numpy(s).double().view(2, 1, 1, 1).expand(2, 1, 256, 256)
            O.unet_forward(torch.cat([torch.from_numpy(x).double(), sig], 1), p64)
    finally:
        F.conv2d = orig
    print("| layer (Cin -> Cout @ H) | direct fp32 | F(2x2) | " + " | ".join(k for k in MATS if k != "F2") + " |")
    print("|---|---|---|---|---|")
    worst = {k: 0.0 for k in MATS}
    for i, (inp, w, b) in acts.items():
        ref = orig(inp, w, b, padding=1)
        row = [rel(orig(inp.float(), w.float(), b.float(), padding=1), ref)]
        for k, (m, mats) in MATS.items():
            e = rel(wino_conv(inp.float(), w.float(), b.float(), m, mats), ref)
            worst[k] = max(worst[k], e)
            row.append(e)
        print(f"| {w.shape[1]} -> {w.shape[0]} @ {inp.shape[-1]} | " + " | ".join(f"{e:.2e}" for e in row) + " |")
    return worst


def make_den(params, deep):
    """fp32 CPU denoiser whose 3x3 layers with cout % 32 == 0 run as emulated F(2x2); `deep` (a MATS key) replaces the scheme on the two
    deepest levels (spatial size <= 1/8 of the image)."""
    p = {k: torch.as_tensor(v).float() for k, v in params.items()}
    orig = F.conv2d

    class Den:
        def __call__(self, x, sigma):
            H0 = x.shape[-1]

            def patched(inp, w, b=None, **kw):
                if w.shape[-1] == 3 and w.shape[0] % 32 == 0 and w.shape[1] % 8 == 0 and inp.shape[-1] % 2 == 0:
                    key = "F2"
                    if deep and inp.shape[-1] * 8 <= H0 and inp.shape[-1] % 4 == 0:
                        key = deep
                    m, mats = MATS[key]
                    return wino_conv(inp, w, b, m, mats)
                return orig(inp, w, b, **kw)

            F.conv2d = patched
            try:
                return O.denoise(x, sigma, p)
            finally:
                F.conv2d = orig

    return Den()


def drift_table(params, n_seeds):
    B, H, W = 2, 64, 64
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    acts = synth.make_actions(B)
    schemes = {"emulated F(2x2) everywhere (= today's fp32 family)": None}
    for k in MATS:
        if k != "F2":
            schemes[f"{k} on the two deepest levels"] = k
    rows = {k: [] for k in schemes}
    ecpu = []
    for seed in range(31, 31 + n_seeds):
        d = synth.make_csmri_batch(B, H, W, ratio=4, seed=seed)

        def run(den, dtype):
            c = lambda a: t(a).to(dtype) if a.dtype != np.bool_ else t(a)
            v = O.admm_reset(c(d["x0"]))
            with torch.no_grad():
                for a in acts:
                    v = O.csmri_admm(den, v, c(d["y0"]), t(d["mask"]), c(a["sigma_d"]), c(a["mu"]))
            return O.complex2real(v[:, :1]).double()

        ref64 = run(O.Denoiser(params, dtype=torch.float64), torch.float64)
        ref32 = run(O.Denoiser(params, dtype=torch.float32), torch.float32)
        ecpu.append(rel(ref32, ref64))
        for name, deep in schemes.items():
            out = run(make_den(params, deep), torch.float32)
            rows[name].append((seed, rel(out, ref64), rel(out, ref32)))
    print("\n| scheme | vs fp64: max / median | vs fp32 CPU oracle: max / median | seeds above 1e-4 (vs fp32 oracle) |")
    print("|---|---|---|---|")
    for name, r in rows.items():
        e64 = np.array([x[1] for x in r])
        e32 = np.array([x[2] for x in r])
        print(f"| {name} | {e64.max():.2e} / {np.median(e64):.2e} | {e32.max():.2e} / {np.median(e32):.2e} | {int((e32 > 1e-4).sum())} of {len(r)} |")
    print(f"\nfp32 CPU oracle vs fp64 over the same seeds: max {max(ecpu):.2e}, median {np.median(ecpu):.2e}\n")
    print("| seed | " + " | ".join(f"{n}: vs fp64 / vs fp32" for n in rows) + " |")
    print("|---|" + "---|" * len(rows))
    for i in range(n_seeds):
        print(f"| {31 + i} | " + " | ".join(f"{rows[n][i][1]:.2e} / {rows[n][i][2]:.2e}" for n in rows) + " |")


if __name__ == "__main__":
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.set_num_threads(8)
    params = synth.make_unet_params(0)
    # self-check of the generated matrices against an fp64 convolution
    xx, ww, bb = torch.randn(1, 8, 8, 8).double(), torch.randn(4, 8, 3, 3).double(), torch.randn(4).double()
    for k, (m, mats) in MATS.items():
        At, G, Bt = mats
        U = np.einsum("ia,kcab,jb->ijkc", G, ww.numpy(), G)
        xp = F.pad(xx, (1, 1, 1, 1)).unfold(2, m + 2, m).unfold(3, m + 2, m).numpy()
        Vv = np.einsum("ia,bcyxaz,jz->bcyxij", Bt, xp, Bt)
        Y = np.einsum("pi,ijkc,bcyxij,qj->bkyxpq", At, U, Vv, At).transpose(0, 1, 2, 4, 3, 5).reshape(1, 4, 8, 8) + bb.view(1, -1, 1, 1).numpy()
        assert rel(torch.from_numpy(Y), F.conv2d(xx, ww, bb, padding=1)) < 1e-12, k
    print("# r6: numerics probe -- Winograd F(4x4,3x3) on the 16^2 / 32^2 levels of the fp32 family (CPU emulation, `tools/wino_f4_probe.py`)\n")
    print("Per-layer relative L2 error against an fp64 convolution, real activations of the He-scaled UNet at 2 x 256 x 256:\n")
    worst = layer_table(params)
    print("\nworst layer: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    print(f"\n30-iteration CS-MRI ADMM episode (B = 2, 64 x 64, {n_seeds} seeds), relative L2 of the reconstructed image:")
    drift_table(params, n_seeds)
