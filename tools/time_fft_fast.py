"""A/B of the N = 256 register-radix-16 FFT kernels (option fft_fast) against the generic LDS Stockham passes: plain fft2 /
ifft2 and the CS-MRI ADMM prox (three passes) per iteration, plus the difference between the two paths."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tfpnp_amd import synth, ops
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.utils import transforms as T
dev = torch.device("cuda:0")
ctx = ops.default_context(dev)
B, H = 48, 256
x = torch.randn(B, 1, H, H, 2, device=dev)
outs = {}
for fast in (0, 1, 0, 1):
    ctx.set_option("fft_fast", fast)
    for name, f in [("fft2", lambda: T.fft2(x)), ("ifft2", lambda: T.ifft2(x))]:
        outs[(fast, name)] = f().clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 200
        print(f"fft_fast={fast} B={B} {H}^2 {name}: {t*1e6:.1f} us  ({2*x.numel()*4/t/1e12:.2f} TB/s in+out)", flush=True)
for name in ("fft2", "ifft2"):
    a, b = outs[(0, name)].double(), outs[(1, name)].double()
    ref = torch.view_as_real(torch.fft.fftshift((torch.fft.fft2 if name == "fft2" else torch.fft.ifft2)(
        torch.fft.ifftshift(torch.view_as_complex(x.double()), dim=(-2, -1)), norm="ortho"), dim=(-2, -1)))
    print(f"{name}: fast vs generic {float((a-b).norm()/a.norm()):.2e}; generic vs torch fp64 {float((a-ref).norm()/ref.norm()):.2e}; "
          f"fast vs torch fp64 {float((b-ref).norm()/ref.norm()):.2e}")
# ADMM prox per iteration: a T=30 solver call minus 30 denoiser forwards
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
sol = ADMMSolver_CSMRI(den)
d = synth.make_csmri_batch(B, H, H, ratio=4, sigma_n=15.0, seed=3)
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
v0 = sol.reset({"x0": g(d["x0"])}); aux = (g(d["y0"]), g(d["mask"]))
Tn = 10
par = (torch.full((B, Tn), 0.1, device=dev), torch.full((B, Tn), 0.5, device=dev))
xr = torch.rand(B, 1, H, H, device=dev); sg = torch.full((B,), 0.1, device=dev)
dctx = den.context(dev)
res = {}
for fast in (0, 1, 0, 1):
    dctx.set_option("fft_fast", fast)
    res[fast] = sol((v0, aux), par).clone(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): sol((v0, aux), par)
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / (5 * Tn)
    t0 = time.perf_counter()
    for _ in range(50): den(xr, sg)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 50
    print(f"fft_fast={fast}: ADMM iteration {ts*1e3:.3f} ms, denoiser forward {td*1e3:.3f} ms, prox+update = {1e6*(ts-td):.1f} us "
          f"(algorithmic 179 MB -> {179e6/(ts-td)/1e12:.2f} TB/s)", flush=True)
print(f"ADMM T={Tn}: fast vs generic state {float((res[0].double()-res[1].double()).norm()/res[0].double().norm()):.2e}")
