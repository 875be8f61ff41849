"""Training-path timing of ADMMSolver_CSMRI (forward + backward of T inner iterations under autograd): the fused native
VJP (pnpx_csmri_admm_train / _backward) vs the same loop composed from differentiable building blocks.  GPU box only.
usage: time_train.py [B] [H] [T] [conv_mode] [--fused-only] [--set=option:value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks import csmri
from tests import composed_solvers
composed_solvers.install()      # the composed (step-by-step autograd) loops live with the tests

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 5
args = [v for v in sys.argv[1:] if not v.startswith("--")]
B = int(args[0]) if len(args) > 0 else 48
H = int(args[1]) if len(args) > 1 else 256
T = int(args[2]) if len(args) > 2 else 5
MODE = int(args[3]) if len(args) > 3 else 1
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=MODE)
for kv in [v.split("=", 1)[1] for v in sys.argv[1:] if v.startswith("--set=")]:
    den.context(dev).set_option(kv.split(":")[0], int(kv.split(":")[1]))
print(f"conv_mode {MODE}" + ("" if MODE else " (fp32 arithmetic: Winograd / direct fp32 MFMA kernels)"))
sol = csmri.ADMMSolver_CSMRI(den)
d = synth.make_csmri_batch(B, H, H, seed=1)
a = synth.make_actions(B, 1, T)[0]
v0 = sol.reset({"x0": t(d["x0"])})
y0, m = t(d["y0"]), t(d["mask"])
w = torch.randn_like(v0)


def step(fn):
    leaves = [v0.clone().requires_grad_(True), t(a["sigma_d"]).requires_grad_(True), t(a["mu"]).requires_grad_(True)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn(*leaves)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (out * w).sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3


ctx = den.context(dev)
cases = (("fused, ring", 96, lambda v, s, mu: sol((v, (y0, m)), (s, mu))),
         ("fused, recompute", 0, lambda v, s, mu: sol((v, (y0, m)), (s, mu))),
         ("composed, ring", 96, lambda v, s, mu: sol._forward_autograd(v, y0, m, s, mu, None)),
         ("composed, recompute", 0, lambda v, s, mu: sol._forward_autograd(v, y0, m, s, mu, None)))
for name, gb, fn in (cases[:2] if "--fused-only" in sys.argv else cases):
    ctx.set_option("train_cache_gb", gb)
    step(fn)
    print(f"  context holds {ctx.bytes() / 2**30:.1f} GiB")
    r = [step(fn) for _ in range(5)]
    f, b = min(x[0] for x in r), min(x[1] for x in r)
    print(f"{name:18s} B={B} {H}x{H} T={T}: forward {f:7.2f} ms  backward {b:7.2f} ms  total {f + b:7.2f} ms "
          f"({(f + b) / T:.2f} ms per iteration)")
with torch.no_grad():
    sol((v0, (y0, m)), (t(a["sigma_d"]), t(a["mu"])))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sol((v0, (y0, m)), (t(a["sigma_d"]), t(a["mu"])))
    torch.cuda.synchronize()
    print(f"inference forward  {(time.perf_counter() - t0) * 1e3:7.2f} ms")
