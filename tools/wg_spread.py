"""Per-workgroup start / end spread of every conv_hs launch of one denoiser forward (tuning build, PNPX_HS_WGT).

usage: PNPX_LIB=tools/_build/libpnpx_tune.so python tools/wg_spread.py [B] [H] [out.txt]
Stamps are wall_clock64() (100 MHz).  Per launch: HIP-event time around the launch, span = max(end) - min(start), the
spread of the start stamps, and the distribution of per-workgroup busy time -- span minus the median busy time is what a
perfectly balanced walk with no fill / drain would save."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/wgt.txt"
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
for _ in range(3):
    den(x, s)
torch.cuda.synchronize()
if os.path.exists(out):
    os.remove(out)
os.environ["PNPX_HS_WGT"] = out
den(x, s)
torch.cuda.synchronize()
os.environ.pop("PNPX_HS_WGT")
tot_ev = tot_span = tot_med = 0.0
for line in open(out):
    hdr, body = line.split("|")
    st = np.array([[int(v) for v in tok.split(":")] for tok in body.split()], dtype=np.float64) / 100.0   # us
    t0 = st[:, 0].min()
    busy = st[:, 1] - st[:, 0]
    span = st[:, 1].max() - t0
    ev = float(hdr.split("event_us=")[1])
    tot_ev += ev
    tot_span += span
    tot_med += np.median(busy)
    xcd = np.arange(len(st)) % 8
    per_xcd = " ".join(f"{np.median(busy[xcd == k]):.0f}" for k in range(8)) if len(st) >= 8 else ""
    print(f"{hdr.strip()}  span {span:.1f}  start spread {st[:, 0].max() - t0:.1f}  busy min/med/p90/max "
          f"{busy.min():.1f}/{np.median(busy):.1f}/{np.percentile(busy, 90):.1f}/{busy.max():.1f}  "
          f"end spread {st[:, 1].max() - st[:, 1].min():.1f}  busy med per XCD [{per_xcd}]")
print(f"sum event {tot_ev:.0f} us, sum span {tot_span:.0f} us, sum median busy {tot_med:.0f} us")
