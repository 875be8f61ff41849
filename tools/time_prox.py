"""Time of the CS-MRI prox + dual update (three FFT-pass kernels) per ADMM iteration: solver call minus denoiser."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tfpnp_amd import synth, ops
from tfpnp_amd.utils import transforms as T
dev = torch.device("cuda:0")
for (B, H) in [(48, 256), (48, 128), (16, 512)]:
    x = torch.randn(B, 1, H, H, 2, device=dev)
    for name, f in [("fft2", lambda: T.fft2(x)), ("ifft2", lambda: T.ifft2(x))]:
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
        print(f"B={B} {H}^2 {name}: {t*1e6:.1f} us  ({2*x.numel()*4/t/1e12:.2f} TB/s in+out)", flush=True)
