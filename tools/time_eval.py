"""End-to-end time of one policy-driven episode (evaluator.py:75-118 on the native path): B=1 and B=48."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tfpnp_amd import synth, policy
from tfpnp_amd.pnp import UNetDenoiser2D
from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI, CSMRIEnv
dev = torch.device("cuda:0")
den = UNetDenoiser2D(state_dict=synth.make_unet_params(0))
P = synth.make_policy_params(9, 10, False, seed=1)
P["fc_softmax.0.bias"] = P["fc_softmax.0.bias"] + np.array([8.0, 0], np.float32)   # never stop: full 6 x 5 iterations
actor = policy.ResNetActor_ADMM(6, 5); actor.load_state_dict(P)
for (B, H) in [(1, 128), (1, 256), (48, 256)]:
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_csmri_batch(B, H, H, seed=3).items()}
    env = CSMRIEnv(None, ADMMSolver_CSMRI(den), max_episode_step=6)
    def episode(profile=None):
        ob = env.reset(d)
        t_pol = t_step = 0.0
        for _ in range(6):
            t0 = time.perf_counter()
            action, _, _, _ = actor(env.get_policy_ob(ob), None, False, None)
            if profile: torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, ob, reward, done, _ = env.step(action)
            if profile: torch.cuda.synchronize()
            t2 = time.perf_counter()
            t_pol += t1 - t0; t_step += t2 - t1
        return t_pol, t_step
    episode(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): episode()
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 3
    tp, ts = episode(profile=True)
    print(f"B={B} {H}^2: episode (6 policy steps x 5 iterations) {tot*1e3:.1f} ms = {B/tot:.1f} images/s; "
          f"policy+ob {tp*1e3:.1f} ms, env.step {ts*1e3:.1f} ms (synchronised breakdown)", flush=True)
