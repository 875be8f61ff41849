"""Thread A: HS (or fp32) denoiser forwards; thread B: the arithmetic victim (tools/micro/victim.hip v3_kernel), whose output must be
bit-identical on every launch."""
import os, sys, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libvictim.so"))
B, H, mode, vmode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 8
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=mode)
x = torch.rand(B, 1, H, H, device=dev); sg = torch.full((B,), 0.1, device=dev)
den(x, sg)
NWG = 768
out = torch.zeros(NWG * 256 * 16, 2, device=dev)
def run(st):
    lib.v3_launch(C.c_void_p(out.data_ptr()), NWG, iters, vmode, C.c_void_p(st.cuda_stream))
s0 = torch.cuda.current_stream(); run(s0); torch.cuda.synchronize(); ref = out.clone()
run(s0); torch.cuda.synchronize(); assert torch.equal(out, ref)
stop = threading.Event(); stat = [0, 0, 0]
def a():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(200): den(x, sg)
    st.synchronize(); stop.set()
def b():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        while not stop.is_set():
            out.zero_(); run(st)
            d = (out != ref).any(-1)
            nb = int(d.sum())
            stat[0] += 1
            if nb:
                stat[1] += 1; stat[2] += nb
                if stat[1] <= 3:
                    idx = d.nonzero().view(-1)
                    wg = (idx // 4096).unique().tolist(); th = ((idx // 16) % 256).unique().tolist()
                    print(f"  launch {stat[0]}: {nb} wrong values in workgroups {wg[:8]} threads {th[:20]}{'...' if len(th) > 20 else ''}", flush=True)
ts = [threading.Thread(target=a), threading.Thread(target=b)]
[t.start() for t in ts]; [t.join() for t in ts]
print(f"denoiser mode {mode} B={B} {H}^2, victim mode {vmode} ({['packed-fp32 math, registers only', 'packed-fp32 math + LDS exchange', 'LDS exchange only', 'scalar v_fma_f32 math, registers only'][vmode]}), {iters} iterations: {stat[1]} corrupted launches of {stat[0]} ({stat[2]} values)")
