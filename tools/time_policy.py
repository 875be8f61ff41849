"""Time the native actor forward against the same network in PyTorch-ROCm (MIOpen convs), and one policy-driven env step."""
import sys, time
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from tfpnp_amd import synth, ops, policy

dev = torch.device("cuda:0")
P = synth.make_policy_params(9, 10, False, seed=1)
actor = policy.ResNetActor_ADMM(6, 5)
actor.load_state_dict(P)
pt = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}


def torch_forward(state):   # plain PyTorch restatement (GPU) for timing only
    def bn(x, pre):
        return F.batch_norm(x, pt[pre + ".running_mean"], pt[pre + ".running_var"], pt[pre + ".weight"], pt[pre + ".bias"], False, 0.1, 1e-5)
    x = F.relu(bn(F.conv2d(state, pt["actor_encoder.conv1.weight"], stride=2, padding=1), "actor_encoder.bn1"))
    for li in range(1, 5):
        for blk in range(2):
            pre = f"actor_encoder.layer{li}.{blk}"
            st = 2 if blk == 0 else 1
            out = F.relu(bn(F.conv2d(x, pt[pre + ".conv1.weight"], stride=st, padding=1), pre + ".bn1"))
            out = bn(F.conv2d(out, pt[pre + ".conv2.weight"], padding=1), pre + ".bn2")
            sc = x if blk else bn(F.conv2d(x, pt[pre + ".shortcut.0.weight"], stride=st), pre + ".shortcut.1")
            x = F.relu(out + sc)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return torch.softmax(F.linear(x, pt["fc_softmax.0.weight"], pt["fc_softmax.0.bias"]), 1), torch.sigmoid(F.linear(x, pt["fc_deterministic.0.weight"], pt["fc_deterministic.0.bias"]))


for (B, H, W) in [(48, 128, 128), (48, 256, 256)]:
    ob = torch.rand(B, 9, H, W, device=dev)
    ctx = actor.context(dev)
    res = {}
    for name, fn in [("native", lambda: ops.policy_forward(ctx, ob)), ("torch", lambda: torch_forward(ob))]:
        with torch.no_grad():
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            t0 = time.time()
            n = 10
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize()
        res[name] = out
        print(f"B={B} {H}x{W} policy {name}: {(time.time() - t0) / n * 1e3:.2f} ms", flush=True)
    print("   max diff probs", float((res["native"][0] - res["torch"][0]).abs().max()), "det", float((res["native"][1] - res["torch"][1]).abs().max()))
