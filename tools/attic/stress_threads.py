"""Stress of the threading / stream contract (tests/test_gpu_edge.py::test_two_contexts_from_two_threads_and_side_streams): two
host threads drive two contexts on side streams; every result is compared with the serial one; prints mismatch statistics."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
from tests.golden_inputs import csmri_actions
dev = torch.device("cuda:0")
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
params = synth.make_unet_params(0)
B, H, W = 3, 64, 64
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
opts = [kv.split("=") for kv in sys.argv[2:] if "=" in kv]
DEN_ONLY = "den" in sys.argv[2:]
SYNC = "sync" in sys.argv[2:]
ONE = "one" in sys.argv[2:]        # only one worker thread (sanity check of the harness)
PRE = "pre" in sys.argv[2:]        # create the native contexts (weights, arenas) before the threads start
NOISE = [a for a in sys.argv[2:] if a.startswith("noise")]   # a third thread hammering device-wide calls
bad = 0
for rep in range(N):
    jobs = [(synth.make_csmri_batch(B, H, W, seed=200 + k), csmri_actions(B, 4, 210 + k)) for k in range(2)]
    serial = []
    for d, a in jobs:
        sol = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=params))
        for k_, v_ in opts: sol.denoiser.context(dev).set_option(k_, int(v_))
        v0 = sol.reset({"x0": g(d["x0"])})
        if DEN_ONLY:
            xin = g(d["x0"])[..., 0].contiguous()
            serial.append(sol.denoiser(xin, g(a["sigma_d"])[:, 0].contiguous()).clone())
        else:
            serial.append(sol((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"]))).clone())
    out, errs = [None, None], []
    pre = []
    if PRE:
        for k in range(2):
            d, a = jobs[k]
            sp = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=params))
            v0 = sp.reset({"x0": g(d["x0"])})
            sp((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
            pre.append(sp)
        torch.cuda.synchronize()
    stop = threading.Event()
    def noise():
        import ctypes
        while not stop.is_set():
            if "noise_sync" in NOISE: torch.cuda.synchronize()
            if "noise_malloc" in NOISE:
                t_ = torch.empty(1 << 20, device=dev); del t_; torch.cuda.empty_cache()
            if "noise_memset" in NOISE:
                torch.zeros(1 << 22, device=dev)
    def work(k):
        try:
            d, a = jobs[k]
            sol = pre[k] if PRE else ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=params))
            for k_, v_ in opts: sol.denoiser.context(dev).set_option(k_, int(v_))
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                v0 = sol.reset({"x0": g(d["x0"])})
                args = ((v0, (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
                if DEN_ONLY:
                    xin = g(d["x0"])[..., 0].contiguous(); sg0 = g(a["sigma_d"])[:, 0].contiguous()
                    rs = []
                    for _ in range(12):
                        rs.append(sol.denoiser(xin, sg0).clone())
                        if SYNC: torch.cuda.synchronize()
                else:
                    rs = [sol(*args).clone() for _ in range(3)]
                out[k] = rs
            st.synchronize()
        except Exception as e:
            errs.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(1 if ONE else 2)]
    nt = threading.Thread(target=noise) if NOISE else None
    if nt: nt.start()
    [t.start() for t in ts]; [t.join() for t in ts]
    stop.set()
    if nt: nt.join()
    assert not errs, errs
    for k in range(1 if ONE else 2):
        for j, r in enumerate(out[k]):
            if not torch.equal(r, serial[k]):
                diff = (r - serial[k]).abs()
                nz = int((diff > 0).sum())
                per_item = [float(diff[b].max()) for b in range(B)]
                per_var = [float(diff[:, v].max()) for v in range(diff.shape[1])]
                print(f"rep {rep} thread {k} call {j}: {nz} of {diff.numel()} differ, max {float(diff.max()):.3e}, per item {per_item}, per var (x,z,u) {per_var}, nan {bool(torch.isnan(r).any())}", flush=True)
                bad += 1
print(f"{bad} mismatching results in {N} repetitions ({' '.join(sys.argv[2:])})")
