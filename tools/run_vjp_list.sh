# Timeline of one fused training step (B=48, 256^2, T=1): forward-train launches then the VJP's, in launch order.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/vjp_list; rm -rf $O; mkdir -p $O
cat > /tmp/one_step.py <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from tfpnp_amd import synth
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks import csmri
dev = torch.device("cuda:0"); t = lambda a: torch.from_numpy(a).to(dev)
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0)); sol = csmri.ADMMSolver_CSMRI(den)
B, H, T = 48, 256, 1
d = synth.make_csmri_batch(B, H, H, seed=1); a = synth.make_actions(B, 1, T)[0]
v0 = sol.reset({"x0": t(d["x0"])}); y0, m = t(d["y0"]), t(d["mask"]); w = torch.randn_like(v0)
for _ in range(4):
    leaves = [v0.clone().requires_grad_(True), t(a["sigma_d"]).requires_grad_(True), t(a["mu"]).requires_grad_(True)]
    out = sol((v0 if False else leaves[0], (y0, m)), (leaves[1], leaves[2]))
    (out * w).sum().backward()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace -d $O/t -o t -- python /tmp/one_step.py > $O/log.txt 2>&1
python tools/rocpd_list.py $O/t/t_results.db 110 | cut -c1-170 > $O/list.txt
find $O -name "*.db" -delete
