"""Timeline of the 8-wave Winograd kernel (conv3x3_wino8.hip built with -DWINO8_TRACE = libpnpx_trace.so, `make -C tfpnp_amd/csrc trace`).
Waves 0 (transform role) and 4 (DMA role) of workgroup 0 stamp the shader clock behind every stage's barrier and at the phases of the
epilogue; this prints, per layer, the cycles per stage and per epilogue phase.  GPU box only.
usage: PNPX_LIB=tools/_build/libpnpx_trace.so python tools/trace_wino8.py [B] [H] [layer ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfpnp_amd import _lib, ops, synth
from tfpnp_amd.pnp import UNetDenoiser2D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
layers = [int(v) for v in sys.argv[3:]] or [1, 4, 7, 10, 13, 15, 16, 19, 22, 25]
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=0)
ctx = den.context(dev)
x = torch.rand(B, 1, H, H, generator=torch.Generator().manual_seed(1)).to(dev)
s = torch.full((B,), 0.1, device=dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
names = {14: "start", 15: "prologue", 16: "epi:in", 17: "epi:#1", 18: "epi:#2", 19: "epi:stores issued", 20: "epi:#3"}
for li in layers:
    ctx.set_option("fp32_wino8_layers", 1 << li)        # only this layer on the traced kernel: the buffer holds its launch
    den(x, s)
    den(x, s)
    torch.cuda.synchronize()
    buf = np.zeros((2, 2048), np.uint64)
    rc = lib.pnpx_debug_wino8_trace(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    print(f"== layer {li} (B={B}, {H}x{H})")
    for w in range(2):
        n = int(buf[w, 2047])
        t = (buf[w, :n] >> np.uint64(8)).astype(np.int64)
        tag = (buf[w, :n] & np.uint64(255)).astype(np.int64)
        if n < 3:
            print("  (no stamps: the layer did not run on the 8-wave kernel)")
            continue
        # stage stamps: 32 + S before the closing wait, 48 + S behind it, S behind the barrier
        keep = tag < 32
        for S in range(4):
            iw = np.where(tag == 32 + S)[0]
            if len(iw):
                wait = t[iw + 1] - t[iw]
                bar = t[iw + 2] - t[iw + 1]
                print(f"      stage {S}: wait median {int(np.median(wait))} mean {wait.mean():.0f} max {wait.max()}; barrier median {int(np.median(bar))} mean {bar.mean():.0f} max {bar.max()}")
        t, tag = t[keep], tag[keep]
        d = np.diff(t)
        tg = tag[1:]
        stage = d[tg < 14]
        for S in range(4):
            if (tg == S).any():
                print(f"      stage {S}: whole median {int(np.median(d[tg == S]))} mean {d[tg == S].mean():.0f}")
        ends = np.where(tag == 20)[0]
        n = len(t)
        p0 = np.where(tag == 15)[0][0]
        tile_cyc = np.diff(np.concatenate([[t[p0]], t[ends]]))
        print(f"  wave {4 * w} ({'DMA' if w else 'transform'} role): {n} stamps, total {t[-1] - t[0]} clk, prologue {int(d[tg == 15][0])}; "
              f"stage median {int(np.median(stage))} mean {stage.mean():.0f} max {stage.max()} (n={len(stage)}); tiles {len(ends)} x {tile_cyc.mean():.0f} clk")
        for k in (16, 17, 18, 19, 20):
            v = d[tg == k]
            if len(v):
                print(f"      -> {names[k]:18s} median {int(np.median(v)):6d}  mean {v.mean():8.0f}  max {v.max():6d}")
        print("      first deltas:", list(zip([int(q) for q in tg[:14]], [int(v) for v in d[:14]])))
