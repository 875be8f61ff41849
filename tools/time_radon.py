import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tfpnp_amd import synth
from tfpnp_amd.utils import transforms as T
dev = torch.device("cuda:0")
for (B, R, V) in [(32, 256, 30), (32, 256, 120), (8, 512, 60)]:
    gt = torch.from_numpy(synth.phantom_batch(B, R, R)).to(dev)
    radon = T.Radon_norm(R, V, device=dev)
    y = radon.forward(gt)
    for name, f in [("fwd", lambda: radon.forward(gt)), ("bwd", lambda: radon.backprojection(y))]:
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        print(f"B={B} R={R} V={V} radon {name}: {(time.perf_counter()-t0)/10*1e3:.3f} ms", flush=True)
