import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tests.golden_inputs import denoiser_inputs
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
for (B, H) in [(2, 64), (3, 128), (48, 256)]:
    x, s = denoiser_inputs(B, H, H, 5)
    x = torch.from_numpy(x).to(dev); s = torch.from_numpy(s).to(dev)
    os.environ.pop("PNPX_UP_FUSE", None)
    ref = den.forward_preclamp(x, s)[1].clone()
    os.environ["PNPX_UP_FUSE"] = "1"
    o = den.forward_preclamp(x, s)[1]
    e = (o - ref).abs()
    print(B, H, "fused vs unfused: rel", float((o - ref).norm() / ref.norm()), "max", float(e.max()), "equal", bool(torch.equal(o, ref)))
    for mode in ["1", None]:
        if mode: os.environ.pop("PNPX_UP_FUSE", None)
        else: os.environ["PNPX_UP_FUSE"] = "1"
        den(x, s); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): den(x, s)
        torch.cuda.synchronize()
        print("   ", "unfused" if mode else "fused  ", f"{(time.perf_counter()-t0)/10*1e3:.3f} ms")
