import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tfpnp_amd import synth
from tfpnp_amd.pnp import DRUNetDenoiser2D
from tfpnp_amd.tasks.spi import ADMMSolver_SPI
dev = torch.device("cuda:0")
den = DRUNetDenoiser2D(state_dict=synth.make_drunet_params(0))
B, H, T = 64, 512, 5
d = synth.make_spi_batch(B, H, H, K=6, seed=5)
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sol = ADMMSolver_SPI(den)
x0, K = g(d["x0"]), g(d["K"])
v = sol.reset({"x0": x0})
par = (torch.full((B, T), 40 / 255.0, device=dev), torch.full((B, T), 85.0, device=dev))
sol((v, (x0, K)), par); torch.cuda.synchronize()
t0 = time.perf_counter(); out = sol((v, (x0, K)), par); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / T
print(f"config #5 (SPI ADMM + DRUNet, B={B}, {H}x{H}): {dt*1e3:.1f} ms per iteration, {B/dt:.0f} image-iterations/s, finite {bool(torch.isfinite(out).all())}, context {den.context(dev).bytes()/2**30:.1f} GiB")
