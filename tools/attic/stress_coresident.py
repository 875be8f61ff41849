"""Who is the victim?  Thread A: HS denoiser forwards (own context, own stream) on a fixed input; thread B: native fft2 /
psnr / other small kernels on a fixed input (own stream).  Every output is compared with the single-threaded reference."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfpnp_amd import synth, ops
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.utils import transforms as T
dev = torch.device("cuda:0")
B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 64
what = sys.argv[3] if len(sys.argv) > 3 else "fft"
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=mode)
for kv in sys.argv[5:]:
    tgt = ops.default_context(dev) if kv.startswith("fft_") else den.context(dev)
    tgt.set_option(kv.split("=")[0], int(kv.split("=")[1]))
x = torch.rand(B, 1, H, H, device=dev); sg = torch.full((B,), 0.1, device=dev)
c = torch.randn(B, 1, H, H, 2, device=dev)
ref_d = den(x, sg).clone()
ref_f = T.fft2(c).clone()
big = torch.randn(64, 1, 256, 256, device=dev)
sp_z = torch.rand(B, 1, H, H, device=dev); sp_k1 = torch.randint(0, 30, (B, 1, H, H), device=dev).float(); sp_k = torch.full((B, 1, 1, 1), 6.0, device=dev); sp_mu = torch.full((B, 1, 1, 1), 85.0, device=dev)
ref_s = T.spi_inverse(sp_z, sp_k1, sp_k, sp_mu).clone()
c1 = torch.randn(B * H, 1, 1, H, 2, device=dev); ref_1 = T.fft2(c1).clone()
cc = torch.view_as_complex(c.clone()); ref_t = torch.view_as_real(torch.fft.fft2(cc)).clone()
c_cpu, ref_f_cpu, x_cpu, ref_d_cpu = c.cpu().clone(), ref_f.cpu().clone(), x.cpu().clone(), ref_d.cpu().clone()
torch.cuda.synchronize()
bad = {"den": 0, "other": 0}; n = {"den": 0, "other": 0}
stop = threading.Event()
def a():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(300):
            y = den(x, sg)
            if not torch.equal(y, ref_d): bad["den"] += 1
            n["den"] += 1
    st.synchronize(); stop.set()
def b():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        while not stop.is_set():
            if what == "fft":
                y = T.fft2(c)
                if not torch.equal(y, ref_f):
                    bad["other"] += 1
                    if bad["other"] <= 0:
                        d = (y != ref_f).any(-1)[:, 0]            # [B, H, W]
                        for bb in range(B):
                            if d[bb].any():
                                cols = d[bb].any(0).nonzero().view(-1).tolist(); rows = d[bb].any(1).nonzero().view(-1).tolist()
                                print(f"  fft mismatch image {bb}: {int(d[bb].sum())} px, cols {cols[:40]}{'...' if len(cols) > 40 else ''} ({len(cols)}), rows {len(rows)}, max err {float((y[bb]-ref_f[bb]).abs().max()):.3e}", flush=True)
            elif what == "spi":
                y = T.spi_inverse(sp_z, sp_k1, sp_k, sp_mu)
                if not torch.equal(y, ref_s): bad["other"] += 1
            elif what == "rows":          # H = 1: only the row pass does work
                y = T.fft2(c1)
                if not torch.equal(y, ref_1): bad["other"] += 1
            elif what == "torchfft":
                y = torch.fft.fft2(cc)
                if not torch.equal(torch.view_as_real(y), ref_t): bad["other"] += 1
            elif what == "torch":
                y = (big * 1.5 + 0.25).sum()      # plain PyTorch kernels (no LDS to speak of)
            n["other"] += 1
    st.synchronize()
ts = [threading.Thread(target=a), threading.Thread(target=b)]
[t.start() for t in ts]; [t.join() for t in ts]
print(f"B={B} {H}^2 mode={mode} other={what} {sys.argv[5:]}: denoiser {bad['den']} bad of {n['den']}; other {bad['other']} bad of {n['other']}")
torch.cuda.synchronize()
print("  after the run: input c intact", bool(torch.equal(c.cpu(), c_cpu)), "| ref_f intact", bool(torch.equal(ref_f.cpu(), ref_f_cpu)),
      "| x intact", bool(torch.equal(x.cpu(), x_cpu)), "| ref_d intact", bool(torch.equal(ref_d.cpu(), ref_d_cpu)),
      "| fresh single-threaded fft2 == ref", bool(torch.equal(T.fft2(c), ref_f)))
