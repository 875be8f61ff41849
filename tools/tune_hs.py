"""Sweep conv_hs launch configurations per (cout tile, level width) with the tuning build (make -C tfpnp_amd/csrc tuning):
PNPX_LIB=tools/_build/libpnpx_tune.so python tools/tune_hs.py [B] [H].  Override hook: PNPX_HS_<MT>_<W>="nbw,nw"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import ops, synth
from tfpnp_amd.pnp import UNetDenoiser2D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
ctx = den.context(dev)
ctx.set_option("subbatch", 0)          # one launch per layer: the profile list is [prep, 3 convs x 5, (ups, 3 convs) x 4]
ref = den(x, s).clone()


def measure(R=3):
    ops.unet_profile(ctx, x, s)
    acc = None
    for _ in range(R):
        r = ops.unet_profile(ctx, x, s)
        acc = [a[1] for a in r] if acc is None else [u + a[1] for u, a in zip(acc, r)]
    return [v / R for v in acc], [a[0] for a in r]


base, names = measure()
conv_idx = [i for i, n in enumerate(names) if n == "conv3x3"]
lvl_of = [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 3, 3, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0]
CLASSES = {}
for k, i in enumerate(conv_idx):
    lvl = lvl_of[k]
    key = ("32" if lvl == 0 else "64") + "_" + str(H >> lvl)
    CLASSES.setdefault(key, []).append(i)
print("baseline conv total", sum(base[i] for i in conv_idx), {k: round(sum(base[i] for i in v), 3) for k, v in CLASSES.items()})
for key, idx in CLASSES.items():
    for cfg in ["4,4", "2,4", "1,4", "2,8", "1,8"]:
        os.environ["PNPX_HS_" + key] = cfg
        try:
            t, _ = measure()
            ok = bool(torch.equal(den(x, s), ref))
            print(f"{key:7s} {cfg:4s} {sum(t[i] for i in idx):7.3f} ms  per-layer " + " ".join(f"{t[i]:.3f}" for i in idx) +
                  ("" if ok else "  MISMATCH"))
        except Exception as e:
            print(key, cfg, "ERR", str(e)[:100])
    os.environ.pop("PNPX_HS_" + key)
