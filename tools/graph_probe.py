"""Is a small-batch solver call launch-bound?  Host issue time against GPU time of ADMMSolver_CSMRI calls (T = 5) at B = 6 / 12 / 48, and
the same call replayed from a captured HIP graph (torch.cuda.CUDAGraph around the library's launches).  usage: graph_probe.py [conv_mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sol = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=mode))
for B in (6, 12, 48):
    d = synth.make_csmri_batch(B, 256, 256, ratio=4, sigma_n=15.0, seed=77)
    a = synth.make_actions(B, 6, 5)[0]
    v0 = sol.reset({"x0": t(d["x0"])})
    aux = (t(d["y0"]), t(d["mask"]))
    par = (t(a["sigma_d"]), t(a["mu"]))
    for _ in range(3): ref = sol((v0, aux), par)
    torch.cuda.synchronize()
    N = 40
    t0 = time.perf_counter()
    for _ in range(N): sol((v0, aux), par)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"conv_mode {mode} B={B}: host issue {(t1 - t0) / N / 5 * 1e3:.3f} ms per iteration, issue + drain {(t2 - t0) / N / 5 * 1e3:.3f} ms per iteration", flush=True)
    try:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2): sol((v0, aux), par)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = sol((v0, aux), par)
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(out, ref)
        t0 = time.perf_counter()
        for _ in range(N): g.replay()
        torch.cuda.synchronize()
        print(f"   graph replay {(time.perf_counter() - t0) / N / 5 * 1e3:.3f} ms per iteration, bit-identical {same}", flush=True)
    except Exception as e:
        print("   graph capture failed:", str(e).split("\n")[0][:200], flush=True)
        torch.cuda.synchronize()
