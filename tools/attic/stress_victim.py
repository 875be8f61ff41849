"""Thread A: HS denoiser forwards; thread B: the victim kernel (tools/micro/victim.hip).  Reports corrupted LDS words / registers /
global loads seen by the victim."""
import os, sys, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libvictim.so"))
B, H = int(sys.argv[1]), int(sys.argv[2]); mode = int(sys.argv[3]); lds = int(sys.argv[4]) if len(sys.argv) > 4 else 34816
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=mode)
x = torch.rand(B, 1, H, H, device=dev); sg = torch.full((B,), 0.1, device=dev)
den(x, sg)
table = torch.zeros(256, dtype=torch.int32, device=dev); res = torch.zeros(8, dtype=torch.int32, device=dev)
assert lib.victim_setup(C.c_void_p(table.data_ptr())) == 0
stop = threading.Event()
def a():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(200): den(x, sg)
    st.synchronize(); stop.set()
V2 = len(sys.argv) > 5 and sys.argv[5] == "v2"
buf = torch.zeros(768 * 4096, dtype=torch.int32, device=dev)
def b():
    st = torch.cuda.Stream(device=dev)
    seq = 0
    while V2 and not stop.is_set():
        for _ in range(50):
            seq += 1
            lib.v2_pair(C.c_void_p(buf.data_ptr()), C.c_void_p(res.data_ptr()), 768, seq, C.c_void_p(st.cuda_stream))
        st.synchronize()
    while not stop.is_set():
        lib.victim_launch(C.c_void_p(table.data_ptr()), C.c_void_p(res.data_ptr()), 768, 40, lds, C.c_void_p(st.cuda_stream))
        st.synchronize()
ts = [threading.Thread(target=a), threading.Thread(target=b)]
[t.start() for t in ts]; [t.join() for t in ts]
torch.cuda.synchronize()
r = res.tolist()
print(f"B={B} {H}^2 mode={mode} victim LDS {lds} B: bad LDS words {r[0]}, bad registers {r[1]}, bad global loads {r[2]}, BARRIER LEAKS {r[4]} over {r[3]} workgroups; producer->consumer stale words {r[5]} over {r[6]} reader workgroups")
