"""Per-launch table of one denoiser forward (HIP events): name, ms, TFLOP/s.  GPU box only.

usage: profile_layers.py [B] [H] [conv_mode] [--brief] [--set=option:value ...]
PNPX_LIB selects an A/B build of the library (tools only); with libpnpx_tune.so the launch table can be overridden
through PNPX_HS_<MT>_<W>="nbw,nw" (see csrc/conv_hs.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import ops, synth
from tfpnp_amd.pnp import UNetDenoiser2D

args = [a for a in sys.argv[1:] if not a.startswith("--")]
brief = "--brief" in sys.argv
B = int(args[0]) if len(args) > 0 else 48
H = int(args[1]) if len(args) > 1 else 256
MODE = int(args[2]) if len(args) > 2 else 1
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=MODE)
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 1, H, H, generator=g).to(dev)
s = torch.full((B,), 0.1, device=dev)
ctx = den.context(dev)
for kv in [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--set=")]:      # --set=key:value context options
    ctx.set_option(kv.split(":")[0], int(kv.split(":")[1]))
y = den(x, s)
abl = [a for a in sys.argv[1:] if a.startswith("--abl=")]
if abl:   # tuning builds: ablation bits take effect AFTER a real forward has filled the arena with real activations
    os.environ["PNPX_HS_ABL"] = abl[0].split("=")[1]
ops.unet_profile(ctx, x, s)
acc = None
R = 5
for _ in range(R):
    r = ops.unet_profile(ctx, x, s)
    acc = r if acc is None else [(a[0], a[1] + b[1], a[2]) for a, b in zip(acc, r)]
tot = conv = fl_tot = 0
cls = {}
for i, (n, ms, fl) in enumerate(acc):
    ms /= R
    tot += ms
    if fl:
        conv += ms
        fl_tot += fl
    if not brief:
        print(f"{i:2d} {n:22s} {ms:8.3f} ms  {fl/ms/1e9 if fl else 0:8.1f} TF/s  {fl/1e9:10.1f} GF")
    else:
        cls.setdefault(n, []).append(ms)
if brief:
    print("conv layers ms:", " ".join(f"{v:.3f}" for v in cls.get("conv3x3", []) + cls.get("conv3x3_wino", [])))
    for n, v in cls.items():
        if n not in ("conv3x3", "conv3x3_wino"):
            print(f"{n}: {sum(v):.3f} ms over {len(v)} launches")
print(f"total {tot:.3f} ms  conv {conv:.3f} ms  = {fl_tot/conv/1e9:.1f} TF/s   checksum {float(y.double().sum()):.10e} "
      f"lib {os.environ.get('PNPX_LIB', 'default')} " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("PNPX_HS")))
