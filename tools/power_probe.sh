#!/bin/bash
# Samples package power / clocks (rocm-smi) while a denoiser-only loop and the bench run: is the chip at its power cap?
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
out=${1:-gpurun_out/power_probe.txt}
mkdir -p $(dirname $out)
{
echo "## idle"; rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
python - <<'PY' &
import torch, time, sys
sys.path.insert(0, '.')
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
dev = torch.device('cuda:0')
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0), conv_mode=1)
x = torch.rand(48, 1, 256, 256, device=dev); s = torch.full((48,), 0.1, device=dev)
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(50): den(x, s)
    torch.cuda.synchronize()
PY
pid=$!
sleep 5
for i in 1 2 3 4 5 6; do echo "## denoiser loop sample $i"; rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" ; sleep 1; done
wait $pid
} > $out 2>&1
tail -60 $out
