import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tests.golden_inputs import denoiser_inputs
dev = torch.device("cuda:0")
p = synth.make_unet_params(0)
d0 = UNetDenoiser2D(state_dict=p, conv_mode=0)
d1 = UNetDenoiser2D(state_dict=p, conv_mode=1)
B, H = 16, 256
x, s = denoiser_inputs(B, H, H, 5)
x = torch.from_numpy(x).to(dev); s = torch.from_numpy(s).to(dev)
r = d0.forward_preclamp(x, s)[1]
keys = ["32_256", "64_128", "64_64", "64_32", "64_16"]
def run(env):
    for k in keys: os.environ["PNPX_HS_" + k] = env.get(k, "2,2,1")
    o = d1.forward_preclamp(x, s)[1]
    e = (o - r).abs()
    return float((o - r).norm() / r.norm()), int((e > 1e-3).sum())
print("all 2,2,1:", run({}))
for cfg in ["2,1,2", "2,2,2"]:
    os.environ["PNPX_HS_32_256"] = cfg
    print(cfg, run({"32_256": cfg}), run({"32_256": cfg}), run({"32_256": cfg}))
