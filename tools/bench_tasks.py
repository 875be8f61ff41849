"""Per-iteration time of the PR / CT / SPI solver loops at the BASELINE config sizes, with the denoiser share.
(`prox+update` = iteration time minus a SEPARATELY timed denoiser forward at the same batch: both are wall-clock loops, so the difference
carries +-0.15 ms of box noise and can come out negative for the 0.06-0.13 ms prox steps; the per-kernel times are in the counter table
profiles/r*_tasks_pmc.md.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfpnp_amd import synth, ops
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks import pr, spi, ct, csmri
from tfpnp_amd.utils import transforms as T
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

def timeit(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def den_ms(B, H):
    x = torch.rand(B, 1, H, H, device=dev); s = torch.full((B,), 0.1, device=dev)
    return timeit(lambda: den(x, s))

Tn = 5
# PR
B, H, S = 36, 256, 4
d = synth.make_pr_batch(B, H, H, S=S)
sol = pr.IADMMSolver_PR(den); v0 = sol.reset({"x0": g(d["x0"])})
par = (torch.full((B, Tn), 0.1, device=dev), torch.full((B, Tn), 0.5, device=dev), torch.full((B, Tn), 0.5, device=dev))
y0, m = g(d["y0"]), g(d["mask"])
t = timeit(lambda: sol((v0, (y0, m)), par)) / Tn; dd = den_ms(B, H)
print(f"PR  iADMM B={B} {H}^2 S={S}: {t:.3f} ms/iter (denoiser {dd:.3f}, prox+update {t-dd:.3f})")
# CT
B, R, V = 32, 256, 30
gt = synth.phantom_batch(B, R, R)
radon = T.Radon_norm(R, V, device=dev)
y0 = radon.forward(g(gt)); x0 = radon.backprojection_norm(y0)
view = torch.full((B, 1, R, R), V / 120.0, device=dev)
sol = ct.IADMMSolver_CT(den); sol.radon_generator.opnorms[(R, V)] = radon.opnorm
v0 = sol.reset({"x0": x0})
t = timeit(lambda: sol((v0, (y0, view)), par[:3] if False else (torch.full((B, Tn), 0.1, device=dev), torch.full((B, Tn), 0.5, device=dev), torch.full((B, Tn), 0.5, device=dev)))) / Tn
dd = den_ms(B, R)
tf = timeit(lambda: radon.forward(x0)); tb = timeit(lambda: radon.backprojection(y0))
print(f"CT  iADMM B={B} {R}^2 V={V}: {t:.3f} ms/iter (denoiser {dd:.3f}, prox+update {t-dd:.3f}; radon fwd {tf:.3f} bwd {tb:.3f})")
# SPI
B, H = 64, 512
d = synth.make_spi_batch(B, H, H, K=6)
sol = spi.ADMMSolver_SPI(den); x0 = g(d["x0"]); v0 = sol.reset({"x0": x0}); K = g(d["K"])
par2 = (torch.full((B, Tn), 40 / 255.0, device=dev), torch.full((B, Tn), 85.0, device=dev))
t = timeit(lambda: sol((v0, (x0, K)), par2)) / Tn; dd = den_ms(B, H)
print(f"SPI ADMM  B={B} {H}^2: {t:.3f} ms/iter (denoiser {dd:.3f}, prox+update {t-dd:.3f})")
# CS-MRI for reference
B, H = 48, 256
d = synth.make_csmri_batch(B, H, H)
sol = csmri.ADMMSolver_CSMRI(den); v0 = sol.reset({"x0": g(d["x0"])})
par3 = (torch.full((B, Tn), 0.1, device=dev), torch.full((B, Tn), 0.5, device=dev))
y0, m = g(d["y0"]), g(d["mask"])
t = timeit(lambda: sol((v0, (y0, m)), par3)) / Tn; dd = den_ms(B, H)
print(f"MRI ADMM  B={B} {H}^2: {t:.3f} ms/iter (denoiser {dd:.3f}, prox+update {t-dd:.3f})")
