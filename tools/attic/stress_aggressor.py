"""Which instruction class disturbs a co-resident wave's packed-fp32 math?  Stream A: a synthetic aggressor kernel (256 persistent
workgroups, 82 KiB of LDS each like conv_hs) in a loop; stream B: the packed-fp32 victim (tools/micro/victim.hip v3_kernel<0>)."""
import os, sys, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libvictim.so"))
names = ["f16 MFMA 32x32x16", "LDS-DMA global_load_lds x4", "v_pk_fma_f16", "v_cvt_pk_f16_f32", "f32 MFMA 32x32x2", "ds_read_b128", "bf16 MFMA 32x32x16"]
NWG = 768
VM = int(os.environ.get('VICTIM_MODE', '0')); ALDS = int(os.environ.get('AGG_LDS', str(84 * 1024)))
out = torch.zeros(NWG * 256 * 16, 2, device=dev); aout = torch.zeros(256 * 256, device=dev); src = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
s0 = torch.cuda.current_stream()
lib.v3_launch(C.c_void_p(out.data_ptr()), NWG, 32, VM, C.c_void_p(s0.cuda_stream)); torch.cuda.synchronize(); ref = out.clone()
lib.aggressor_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float]
cases = [(0, 1.0, 1.0, 'f16 MFMA, normal operands'), (6, 1.0, 1.0, 'bf16 MFMA, normal operands'), (4, 1.0, 1.0, 'f32 MFMA')]
for kind, asc, bsc, label in cases:
    stop = threading.Event(); stat = [0, 0]
    def a():
        st = torch.cuda.Stream(device=dev)
        while not stop.is_set():
            for _ in range(4):
                lib.aggressor_launch(C.c_void_p(aout.data_ptr()), C.c_void_p(src.data_ptr()), kind, 100000 if kind in (0, 4, 6) else 20000, ALDS, C.c_void_p(st.cuda_stream), asc, bsc)
            st.synchronize()
    def b():
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            for _ in range(300):
                out.zero_(); lib.v3_launch(C.c_void_p(out.data_ptr()), NWG, 32, VM, C.c_void_p(st.cuda_stream))
                stat[0] += 1; stat[1] += int(bool((out != ref).any()))
        stop.set()
    ts = [threading.Thread(target=a), threading.Thread(target=b)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print(f"[victim mode {VM}, aggressor LDS {ALDS}] aggressor {kind} ({label}): packed-fp32 victim corrupted in {stat[1]} of {stat[0]} launches", flush=True)
