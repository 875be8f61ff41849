"""Wall time of the denoiser forward (B=48, 256^2) for several values of the "subbatch" context option."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
ctx = den.context(dev)
ctx.set_option("subbatch", 0)
ref = den(x, s).clone()
for sb in ["0", "2", "4", "6", "8", "12", "16", "24"]:
    ctx.set_option("subbatch", int(sb))
    y = den(x, s); torch.cuda.synchronize()
    ok = torch.equal(y, ref)
    t0 = time.perf_counter()
    for _ in range(10): den(x, s)
    torch.cuda.synchronize()
    print(f"subbatch {sb:>3s}: {(time.perf_counter()-t0)/10*1e3:.3f} ms  identical={ok}")
