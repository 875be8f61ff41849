"""Is a small-batch forward bound by ONE host thread's launch rate?  B images as one call with two launch chains (both issued by the
calling thread, one after the other) against two host threads that each issue B / 2 images on their own context and stream.
usage: two_thread_probe.py [B] [conv_mode]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
params = synth.make_unet_params(0)
x = torch.rand(B, 1, 256, 256, device=dev); s = torch.full((B,), 0.1, device=dev)
den = UNetDenoiser2D(state_dict=params, conv_mode=mode)
N = 200
for c in (1, 2):
    den.context(dev).set_option("chains" if mode == 1 else "fp32_chains", c)
    for _ in range(10): den(x, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): den(x, s)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"one thread, B={B}, {c} chain(s): host issue {(t1 - t0) / N * 1e3:.3f} ms, total {(t2 - t0) / N * 1e3:.3f} ms per forward", flush=True)
dens = [UNetDenoiser2D(state_dict=params, conv_mode=mode) for _ in range(2)]
halves = [(x[: B // 2].contiguous(), s[: B // 2].contiguous()), (x[B // 2:].contiguous(), s[B // 2:].contiguous())]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for k in range(2):
    dens[k].context(dev).set_option("chains" if mode == 1 else "fp32_chains", 1)
    with torch.cuda.stream(streams[k]):
        for _ in range(10): dens[k](*halves[k])
torch.cuda.synchronize()
def work(k):
    with torch.cuda.stream(streams[k]):
        for _ in range(N): dens[k](*halves[k])
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
[t.start() for t in th]; [t.join() for t in th]
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"two threads x B={B // 2} (own contexts, own streams): host {(t1 - t0) / N * 1e3:.3f} ms, total {(t2 - t0) / N * 1e3:.3f} ms per pair of half-forwards", flush=True)
