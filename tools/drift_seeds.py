"""Per-seed drift table of the 30-iteration CS-MRI ADMM episode on the EXPANSIVE (He-scaled) synthetic UNet.
For every seed and both convolution families: HIP vs the fp64 oracle, HIP vs the fp32 CPU oracle (the distance north_star's 1e-4
is stated on) and the fp32 CPU oracle vs fp64 (the yardstick: what fp32 arithmetic itself drifts by on this chaotic map).
Same case as tests/test_gpu_modes.py::test_csmri_episode_drift_not_worse_than_fp32 (B = 2, 64 x 64, 6 x 5 iterations).
usage: drift_seeds.py [n_seeds] > profiles/r6_drift_seeds.md      (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pnp_oracle as O
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.set_num_threads(16)
dev = torch.device("cuda:0")
params = synth.make_unet_params(0)
B, H, W = 2, 64, 64
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
rel = lambda a, b: float((a - b).norm() / b.norm())
acts = synth.make_actions(B)
sols = {m: ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=params, conv_mode=m)) for m in (1, 0)}
rows = []
for seed in range(31, 31 + n_seeds):
    d = synth.make_csmri_batch(B, H, W, ratio=4, seed=seed)

    def run_oracle(dtype):
        den = O.Denoiser(params, dtype=dtype)
        c = lambda a: t(a).to(dtype) if a.dtype != np.bool_ else t(a)
        v = O.admm_reset(c(d["x0"]))
        with torch.no_grad():
            for a in acts:
                v = O.csmri_admm(den, v, c(d["y0"]), t(d["mask"]), c(a["sigma_d"]), c(a["mu"]))
        return O.complex2real(v[:, :1]).double()

    ref64, ref32 = run_oracle(torch.float64), run_oracle(torch.float32)
    e_cpu = rel(ref32, ref64)
    for mode in (1, 0):
        sol = sols[mode]
        g = lambda a: t(a).to(dev)
        v = sol.reset({"x0": g(d["x0"])})
        for a in acts:
            v = sol((v, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
        out = sol.get_output(v).double().cpu()
        rows.append((seed, mode, rel(out, ref64), rel(out, ref32), e_cpu))
print("# r6: 30-iteration CS-MRI ADMM drift on the expansive (He-scaled) synthetic UNet, per seed\n")
print("`python tools/drift_seeds.py %d` on one MI355X (B = 2, 64 x 64, 6 x 5 iterations, radial x4; relative L2 of the reconstructed image)." % n_seeds)
print("conv_mode 0 = fp32 arithmetic (the DEFAULT since r6: 8-wave Winograd incl. the K-split deep levels + direct fp32 MFMA kernels), 1 = the opt-in fast mode (half-split f16 x 3 MFMA).")
print("`cpu32` = the fp32 CPU oracle (= the reference's arithmetic, `tasks/csmri/solver.py:43-55`), `fp64` = the same oracle in double.\n")
print("| seed | conv_mode | HIP vs fp64 | HIP vs cpu32 | cpu32 vs fp64 | HIP-vs-fp64 / cpu32-vs-fp64 |")
print("|---|---|---|---|---|---|")
for seed, mode, e64, e32, ec in rows:
    print(f"| {seed} | {mode} | {e64:.3e} | {e32:.3e} | {ec:.3e} | {e64 / ec:.2f} |")
for mode in (1, 0):
    r = [x for x in rows if x[1] == mode]
    e64 = np.array([x[2] for x in r]); e32 = np.array([x[3] for x in r]); ec = np.array([x[4] for x in r])
    print(f"\nconv_mode {mode}: HIP vs fp64 max {e64.max():.3e} median {np.median(e64):.3e}; HIP vs cpu32 max {e32.max():.3e} median {np.median(e32):.3e} "
          f"({int((e32 > 1e-4).sum())} of {len(r)} seeds above 1e-4); cpu32 vs fp64 max {ec.max():.3e} median {np.median(ec):.3e}; "
          f"max of (HIP-vs-fp64 - 1.5 x cpu32-vs-fp64) = {(e64 - 1.5 * ec).max():.3e}")
