"""Co-residency check of conv_hs (round-1 open issue: two co-resident workgroups of the <32,2> instance produced rare
stale 1-KiB operand pieces).  Tuning build only:

    PNPX_LIB=tfpnp_amd/libpnpx_tune.so python tools/micro/coresident_check.py [repeats]

Runs the denoiser with the 32-cout layers forced to small tiles (PNPX_HS_32_<W>="1,4": 66 KiB of LDS per workgroup) once
with one workgroup per CU (reference) and then `repeats` times with TWO co-resident workgroups per CU (PNPX_HS_PERCU=2),
and reports every launch whose output differs from the reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D

R = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, H = int(os.environ.get("CHECK_B", "24")), 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
plain = den(x, s).clone()
os.environ["PNPX_HS_32_%d" % H] = "1,4"
ref = den(x, s).clone()
print("small-tile config equals the default config bit for bit:", bool(torch.equal(ref, plain)))
os.environ["PNPX_HS_PERCU"] = "2"
bad = 0
for r in range(R):
    y = den(x, s)
    if not torch.equal(y, ref):
        bad += 1
        d = (y != ref)
        print(f"  repeat {r}: {int(d.sum())} differing output pixels in {int(d.flatten(1).any(1).sum())} images, "
              f"max |diff| {float((y - ref).abs().max()):.3e}")
print(f"{bad} of {R} co-resident forwards differ from the one-workgroup-per-CU result")
