"""Forward time with one and two launch chains for every batch size 1..48 at 256x256 (the data behind the automatic rule of
csrc/unet.hip::launch_chains).  usage: chains_table.py [out.txt] [conv_mode]   (conv_mode 0: option fp32_chains)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device("cuda:0")
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 1
OPT = "chains" if MODE == 1 else "fp32_chains"
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=MODE)
ctx = den.context(dev)
out = open(sys.argv[1], "w") if len(sys.argv) > 1 and sys.argv[1] != "-" else sys.stdout
print(f"# B  chains=1 ms  chains=2 ms  ratio  (256x256, conv_mode {MODE}; best of 2 x 30 forwards)", file=out)
for B in range(1, 49):
    x = torch.rand(B, 1, 256, 256, device=dev); s = torch.full((B,), 0.1, device=dev)
    t = {}
    for c in (1, 2):
        ctx.set_option(OPT, c)
        best = 1e9
        for _ in range(2):
            for _ in range(5): den(x, s)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): den(x, s)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
        t[c] = best
    print(f"{B:3d}  {t[1]:8.3f}  {t[2]:8.3f}  {t[2] / t[1]:.3f}", file=out, flush=True)
