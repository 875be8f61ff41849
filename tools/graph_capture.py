"""hipGraph capture of one solver call (torch.cuda.graph around ADMMSolver_CSMRI.forward): eager vs replay time per call at
small batches, and bit equality of the replayed result.  GPU box only.   usage: graph_capture.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks import csmri

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
sol = csmri.ADMMSolver_CSMRI(den)
H, T = 256, 5
for B in [int(v) for v in sys.argv[1:]] or [1, 6, 12, 48]:
    d = synth.make_csmri_batch(B, H, H, seed=1)
    a = synth.make_actions(B, 1, T)[0]
    v0, y0, m = sol.reset({"x0": t(d["x0"])}), t(d["y0"]), t(d["mask"])
    sg, mu = t(a["sigma_d"]), t(a["mu"])
    with torch.no_grad():
        ref = sol((v0, (y0, m)), (sg, mu)).clone()          # warm-up: workspaces, plans, kernel attributes
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sol((v0, (y0, m)), (sg, mu))
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = sol((v0, (y0, m)), (sg, mu))
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))

        def timed(fn, n=30):
            fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        te = timed(lambda: sol((v0, (y0, m)), (sg, mu)))
        tg = timed(g.replay)
    print(f"B={B:2d} {H}x{H} T={T}: eager {te:7.3f} ms  graph replay {tg:7.3f} ms per call  ({te / T:.3f} vs {tg / T:.3f} ms per "
          f"iteration)  replay == eager: {same}")
