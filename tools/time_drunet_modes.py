"""DRUNet forward time in both kernel families (conv_mode 1 half-split, conv_mode 0 fp32) at B x H x H."""
import sys, time, torch
sys.path.insert(0, ".")
from tfpnp_amd import synth
from tfpnp_amd.pnp import DRUNetDenoiser2D
B, H = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
p = synth.make_drunet_params(0)
x = torch.rand(B, 1, H, H, device=dev); s = torch.full((B,), 0.1, device=dev)
for mode in (1, 0):
    den = DRUNetDenoiser2D(state_dict=p, conv_mode=mode)
    t0 = time.perf_counter(); den(x, s); torch.cuda.synchronize(); first = time.perf_counter() - t0
    den(x, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): den(x, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2 * 9 * H * H * B * sum(c * c * 8 * (2 if i < 3 else 1) / 4 ** i for i, c in enumerate((64, 128, 256, 512)))
    print(f"conv_mode {mode}: {ms:.2f} ms per forward ({fl / ms / 1e9:.0f} TF/s over the ResBlock convolutions), first call {first:.1f} s", flush=True)
    del den
