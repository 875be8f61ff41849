"""DRUNet denoiser forward timing (wall clock over N calls) and algorithmic TFLOP/s.  usage: time_drunet.py [B] [H] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import DRUNetDenoiser2D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
den = DRUNetDenoiser2D(state_dict=synth.make_drunet_params(0))
x = torch.rand(B, 1, H, H, device=dev)
s = torch.full((B,), 0.1, device=dev)
for _ in range(3):
    den(x, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    den(x, s)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / N
nc = (64, 128, 256, 512)
macs = H * H * (64 * 18 + 576)                                        # head + tail
macs += sum((16 if l < 3 else 8) * (H >> l) ** 2 * nc[l] ** 2 * 9 for l in range(4))   # ResBlock convolutions
macs += sum(2 * (H >> (l + 1)) ** 2 * nc[l + 1] * 4 * nc[l] for l in range(3))          # strided + transposed 2x2
print(f"DRUNet B={B} {H}x{H}: {ms:.2f} ms per forward, {2 * macs * B / ms / 1e9:.1f} TFLOP/s algorithmic "
      f"({2 * macs / 1e9:.1f} GFLOP per image)")
