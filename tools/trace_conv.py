"""Per-step timeline of conv_hs (workgroup 8, wave 0) from an HS_TRACE build: tools/build_variant.sh trace -DHS_TRACE.

usage: PNPX_LIB=tools/_build/libpnpx_trace.so python tools/trace_conv.py [B] [H] [out.txt]
Stamps (s_memtime, shader clocks): 1 loop top, 2 DMA wait done, 3 barrier passed, 4 chunk multiplied, 5 epilogue done."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/trace.txt"
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
den(x, s); den(x, s)
torch.cuda.synchronize()
if os.path.exists(out):
    os.remove(out)
os.environ["PNPX_HS_TRACE"] = out
den(x, s)
torch.cuda.synchronize()
os.environ.pop("PNPX_HS_TRACE")

names = {(1, 2): "dma wait", (2, 3): "barrier", (3, 4): "multiply", (4, 5): "epilogue", (4, 1): "loop", (5, 1): "loop"}
hdr = None
for line in open(out):
    if line.startswith("#"):
        hdr = line.strip()
        continue
    ev = [tuple(int(v) for v in tok.split(":")) for tok in line.split()]
    tot = {}
    cnt = {}
    for (t0, _), (t1, d) in zip(ev[:-1], ev[1:]):
        k = names.get((t0, t1), f"{t0}->{t1}")
        tot[k] = tot.get(k, 0) + d
        cnt[k] = cnt.get(k, 0) + 1
    allc = sum(tot.values())
    if not allc:
        continue
    print(hdr)
    print("   " + "  ".join(f"{k}: {100.0 * v / allc:.1f}% ({v // max(cnt[k], 1)} clk x {cnt[k]})" for k, v in tot.items()),
          f" total {allc} clk")
