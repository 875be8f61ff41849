import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks import csmri
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
if os.environ.get("CHAINS"):
    den.context(dev).set_option("chains", int(os.environ["CHAINS"]))
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for (B, H) in [(1, 128), (1, 256), (4, 128), (4, 256), (6, 256), (12, 256), (24, 256)]:
    d = synth.make_csmri_batch(B, H, H)
    sol = csmri.ADMMSolver_CSMRI(den); v0 = sol.reset({"x0": g(d["x0"])})
    par = (torch.full((B, 5), 0.1, device=dev), torch.full((B, 5), 0.5, device=dev))
    y0, m = g(d["y0"]), g(d["mask"])
    f = lambda: sol((v0, (y0, m)), par)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"B={B} {H}^2: {t*1e3:.3f} ms per 5-iteration call ({t/5*1e3:.3f} ms/iter); host issue time {t_issue/10*1e3:.3f} ms/call")
