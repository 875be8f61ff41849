"""Round-1 library (libpnpx_r1.so, built from commit d620153) with its PNPX_HS_<MT>_<W>="nbw,nstage,per_cu" hook: does the
co-resident <32,2> configuration still produce stale tiles on this box?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
R = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cfg = sys.argv[2] if len(sys.argv) > 2 else "2,2,2"
B, H = int(os.environ.get("CHECK_B", "24")), 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
ref = den(x, s).clone()
os.environ["PNPX_HS_32_%d" % H] = cfg
bad = 0
for r in range(R):
    y = den(x, s)
    if not torch.equal(y, ref):
        bad += 1
        d = (y != ref)
        if bad <= 5:
            print(f"  repeat {r}: {int(d.sum())} differing pixels in {int(d.flatten(1).any(1).sum())} images, max |diff| "
                  f"{float((y - ref).abs().max()):.3e}")
print(f"config {cfg}: {bad} of {R} forwards differ from the default-configuration result")
