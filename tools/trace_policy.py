import sys
import torch
sys.path.insert(0, ".")
from tfpnp_amd import synth, ops, policy
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
actor = policy.ResNetActor_ADMM(6, 5)
actor.load_state_dict(synth.make_policy_params(9, 10, False, seed=1))
ob = torch.rand(48, 9, H, H, device=dev)
for _ in range(3):
    ops.policy_forward(actor.context(dev), ob)
torch.cuda.synchronize()
