#!/usr/bin/env python
"""Headline benchmark: CS-MRI PnP-ADMM, 256x256, env_batch=48 per GPU, 6 policy steps x 5 inner iterations.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 20 --warmup 5            # self-launching: spawns the 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # ... or under a launcher (RANK / WORLD_SIZE in the env)

One "step" = one episode of the hot path over one resident batch: solver.reset, 6 x ADMMSolver_CSMRI.forward
(5 inner iterations each: UNet denoiser prox + masked-FFT data prox + dual update), 6 x PSNR reward, and -- for
N > 1 -- 6 reward exchanges over RCCL (ONE small all_gather per env step, issued off the compute stream and consumed
after the next step's kernels have been launched: tfpnp_amd/dist.py::StepExchange).  No early stop.  Inputs (y0, mask, x0, gt, the action
schedule, packed weights) are in HBM before the timed region starts.
  --scaling weak   (default) every rank owns its own 48 items; value = inner iterations, each over a 48-image batch,
                   per second, summed over ranks.
  --scaling strong the reference's own split (tasks/csmri/main.py:79-80: DataParallel scatters ONE env batch): the
                   GLOBAL env batch of 48 is sharded contiguously, 48/N items per rank; value = inner iterations over
                   the global 48-image batch per second.

Prints ONE JSON line on rank 0 (fields: see the contract in the task statement) including
  roofline     MFMA roofline of the dominant kernel of the DEFAULT convolution family (conv_mode 0: fp32 arithmetic like the
               reference's, conv3x3_wino8_f32_kernel), measured with HIP events around whole production forwards;
  cpu_baseline the CPU oracle timed on the host cores on a bounded sample of the same workload (N=1 only);
  batch_table  (N=1) one ADMM iteration at B = 6, 12, 24, 48: what an 8/4/2-way strong split would run per rank;
  fast_mode    (N=1) the same episode, roofline and batch table in the opt-in fast mode (conv_mode 1: half-split f16 x 3 MFMA
               convolutions, conv_hs_kernel) -- narrower than fp32 (22-bit significand), hence reported beside the headline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tfpnp_amd import dist as D  # noqa: E402
from tfpnp_amd import ops, synth  # noqa: E402
from tfpnp_amd.tasks.csmri import CSMRIEnv  # noqa: E402
from tfpnp_amd.pnp import UNetDenoiser2D  # noqa: E402
from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)" 157.3 TF/s; "Peak BF16/FP16 MFMA ~2.5 PF dense".
# The default convolution kernel evaluates every fp32 product with THREE f16 MFMAs (half-split operands, fp32
# accumulate), so its roofline in ALGORITHMIC (fp32-equivalent) FLOP/s is the dense f16 MFMA peak / 3.
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
PEAK_HS_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
N_POLICY_STEPS, ACTION_PACK = 6, 5
DTYPE_F32 = "f32 (v_mfma_f32_32x32x2_f32; Winograd F(2x2,3x3) on the layers with cout % 32 == 0)"
DTYPE_HS = "f32 values as f16 hi+lo pairs (22-bit significand); 3 f16 MFMAs per product, f32 accumulate"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=48, help="env_batch per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--ratio", type=int, default=4, help="radial mask undersampling (4 or 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=24, help="items in the bounded CPU-baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-batch-table", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="KEY=VALUE",
                    help="pnpx_ctx_set_option on the denoiser context (A/B experiments, e.g. fuse_up=0)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launch / collective plumbing only: gloo + a CPU stub solver, NOT a measurement (tests/)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))      # one rank per GPU; rank 0 of the child job prints the JSON line
    # PNPX_BENCH_SHARE_GPU=1 (tests only): every rank uses device 0 and the process group is gloo -- the whole N > 1 path
    # (self-launch, sharding, native solver on device tensors, one exchange per env step, max-over-ranks timing) on a box
    # with ONE GPU.  RCCL refuses two ranks on one device, hence gloo; the output is marked and is not a measurement.
    share_gpu = os.environ.get("PNPX_BENCH_SHARE_GPU") == "1"
    rank, world, local_rank = D.init_from_env(backend="gloo" if (args.selftest_cpu or share_gpu) else None)
    if share_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    H, W = args.size, args.size
    strong = args.scaling == "strong"
    if strong:      # one global env batch, contiguous shards (uneven when world does not divide it)
        lo, hi = D.shard_bounds(args.batch, world, rank)
        B, n_global = hi - lo, args.batch
        if B == 0:
            raise SystemExit(f"--scaling strong: rank {rank} of {world} has no items of a batch of {args.batch}")
    else:
        B, n_global = args.batch, args.batch * world

    # ---- resident inputs --------------------------------------------------------------------------
    params = synth.make_unet_params(0)
    den = UNetDenoiser2D(state_dict=params)
    solver = ADMMSolver_CSMRI(den)
    if strong:
        full = synth.make_csmri_batch(n_global, H, W, ratio=args.ratio, sigma_n=15.0, seed=1234)
        data_np = {k: v[lo:hi] for k, v in full.items()}
    else:
        data_np = synth.make_csmri_batch(B, H, W, ratio=args.ratio, sigma_n=15.0, seed=1234 + 1000 * rank)
    data = {k: t(v).to(dev) for k, v in data_np.items()}
    actions = [{k: t(v).to(dev) for k, v in a.items()} for a in synth.make_actions(B, N_POLICY_STEPS, ACTION_PACK)]
    for a in actions:
        a["idx_stop"] = torch.zeros(B, dtype=torch.int64, device=dev)
    for kv in args.ctx_option:
        den.context(dev).set_option(kv.split("=")[0], int(kv.split("=")[1]))
    den.context(dev).reserve(B, H, W)
    env = CSMRIEnv(None, solver, max_episode_step=N_POLICY_STEPS)
    exchange = D.StepExchange(n_global, dev)       # contiguous shards of n_global items: both scaling modes
    episode = make_episode(env, data, actions, exchange)

    for _ in range(args.warmup):
        episode()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    sampler = PowerSampler() if rank == 0 else None          # host thread reading the SMU telemetry, no GPU work
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rewards = episode()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    power = sampler.stop() if sampler else None

    iters = N_POLICY_STEPS * ACTION_PACK * args.steps
    value = (1 if strong else world) * iters / elapsed      # iterations over a 48-image batch per second, whole job
    den.context(dev).status()                                # half-split range guard: raises if any call overflowed
    final_psnr_gain = float(torch.stack(rewards).sum(0).mean().item())

    out = {
        "metric": "PnP-ADMM iters/sec (and images/sec) at env_batch=48, 256x256, 30 inner iters",
        "value": value,
        "unit": "iters/s",
        "n_gpus": world,
        "world_size_seen": D.world_size(),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": (DTYPE_HS if den.context(dev).get_option("conv_mode") == 1 else DTYPE_F32),
        "data": "synthetic",
        "config": {
            "workload": f"CS-MRI ADMM {H}x{W} env_batch=" + (f"{n_global} global ({B} on rank 0)" if strong else f"{B}/GPU") +
                        f" radial x{args.ratio} sigma_n=15, "
                        f"{N_POLICY_STEPS} solver calls x {ACTION_PACK} inner iters + PSNR reward, no early stop",
            "global_batch": n_global,
            "iters_per_step": N_POLICY_STEPS * ACTION_PACK,
            "parallelism": (f"batch-shard x{world}, one all_gather_into_tensor(reward|done|finished) per env step on a "
                            f"side stream (RCCL world size {D.world_size()})") if world > 1 else "single GPU",
        },
        "collectives_per_env_step": exchange.posted / ((args.steps + args.warmup) * N_POLICY_STEPS),
        "images_per_s": n_global * args.steps / elapsed,
        "image_iters_per_s": value * args.batch,
        "psnr_gain_db_random_init_denoiser": final_psnr_gain,
    }
    if share_gpu:
        out["NOT_A_MEASUREMENT"] = f"PNPX_BENCH_SHARE_GPU: {world} ranks time-share device 0 over gloo (plumbing test)"

    if rank == 0 and not args.no_roofline:
        # on the images and noise levels the episode ended with (real operand statistics: the package is power-capped
        # and MFMA power follows operand bit activity)
        out["roofline"] = roofline(den, dev, env.state["output"].detach().clone(), actions[-1]["sigma_d"][:, -1].contiguous())
        out["roofline"]["power"] = power
        if power and power.get("gfx_clk_mhz_avg"):
            # the same peak at the clock the power cap actually allowed during the timed region (nominal 2400 MHz)
            peak_at_clk = out["roofline"]["peak"] * power["gfx_clk_mhz_avg"] / 2400.0
            out["roofline"]["frac_of_peak_at_measured_clock"] = out["roofline"]["achieved"] / peak_at_clk
    if rank == 0 and world == 1 and not args.no_batch_table:
        out["batch_table"] = batch_table(solver, dev, H, W, args.ratio)
    if rank == 0 and world == 1 and not args.no_fast_mode and den.context(dev).get_option("conv_mode") == 0:
        out["fast_mode"] = fast_mode(params, data, actions, dev, B, H, W, args.steps, args.warmup, den,
                                     None if args.no_batch_table else args.ratio)
        # the same metric in the opt-in half-split mode, surfaced beside `value` (NOT the headline: narrower than fp32)
        out["value_fast_mode"] = out["fast_mode"]["value"]
        out["ms_per_step_fast_mode"] = out["fast_mode"]["ms_per_step"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity_rel_l2_vs_cpu"] = cpu_baseline(params, solver, dev, args, value)
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        D.barrier()          # rank 0 may still be in its (non-collective) roofline pass
        torch.distributed.destroy_process_group()


def make_episode(env, data, actions, exchange):
    """One bench step.  The reward exchange of env step k is posted right after the step and resolved only after step
    k+1 has been issued, so the (latency-bound) collective overlaps with the next step's kernels."""

    def episode():
        env.reset(data)
        rewards, pending = [], None
        for a in actions:
            _, _, reward, _, _ = env.step(a)
            nxt = exchange.post(reward, None, False)
            if pending is not None:
                rewards.append(pending.result()[0])                   # [n_global, 1] on every rank
            pending = nxt
        rewards.append(pending.result()[0])
        return rewards

    return episode


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command line under torch.distributed.run with one
    rank per GPU on a free local port and hand its exit status back; rank 0 of that job prints the JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL (see the task's environment notes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def selftest_cpu(args, rank, world):
    """--selftest-cpu: the launch path (self-spawn, rendezvous, sharding, one collective per env step, max-over-ranks
    timing, single JSON line) with gloo and a CPU stub in place of the native solver.  Exists for tests/test_bench_launch.py;
    its output says so and is not a measurement of anything."""
    from tfpnp_amd.env.base import PnPEnv

    class Stub(torch.nn.Module):                      # PnPSolver contract, x <- x + mu (gt - x)
        def reset(self, data):
            return data["x0"].clone()

        def get_output(self, state):
            return state

        def filter_aux_inputs(self, state):
            return (state["gt"],)

        def filter_hyperparameter(self, action):
            return (action["mu"],)

        def forward(self, inputs, parameters):
            return inputs[0] + parameters[0][:, :1].view(-1, 1, 1, 1) * (inputs[1][0] - inputs[0])

    strong = args.scaling == "strong"
    n_global = args.batch if strong else args.batch * world
    lo, hi = D.shard_bounds(n_global, world, rank)
    gt = torch.from_numpy(np.random.RandomState(0).rand(n_global, 1, 4, 4).astype(np.float32))[lo:hi]
    data = {"gt": gt, "x0": torch.zeros_like(gt), "output": torch.zeros_like(gt)}
    B = hi - lo
    actions = [{"mu": torch.full((B, ACTION_PACK), 0.5), "idx_stop": torch.zeros(B, dtype=torch.int64)}
               for _ in range(N_POLICY_STEPS)]
    env = PnPEnv(None, Stub(), max_episode_step=N_POLICY_STEPS)
    env.metric_fn = lambda out, g: -((out - g) ** 2).reshape(out.shape[0], -1).mean(1, keepdim=True)
    exchange = D.StepExchange(n_global, torch.device("cpu"))
    episode = make_episode(env, data, actions, exchange)
    for _ in range(args.warmup):
        episode()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rewards = episode()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({
            "metric": "SELFTEST -- launch plumbing only (gloo, CPU stub solver); not a measurement",
            "value": (1 if strong else world) * N_POLICY_STEPS * ACTION_PACK * args.steps / elapsed, "unit": "iters/s",
            "n_gpus": world, "world_size_seen": D.world_size(), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "scaling": args.scaling, "data": "selftest-stub",
            "config": {"workload": "stub", "global_batch": n_global},
            "collectives_per_env_step": exchange.posted / ((args.steps + args.warmup) * N_POLICY_STEPS),
            "reward_rows_seen": int(rewards[-1].shape[0]), "reward_sum": float(torch.stack(rewards).sum())}), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


class PowerSampler:
    """Socket power and shader clock of GPU 0 of this process sampled every 50 ms on a host thread (amdsmi) while the
    timed region runs: is the run at the package power cap, and at which clock?  Telemetry only; None if unavailable."""

    def __init__(self, period=0.05):
        import threading
        self.rows, self.cap_w, self._stop = [], None, threading.Event()
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[0]
            self.cap_w = amdsmi.amdsmi_get_power_cap_info(h)["power_cap"] / 1e6
        except Exception:
            self.thread = None
            return

        def run():
            while not self._stop.wait(period):
                try:
                    p = amdsmi.amdsmi_get_power_info(h)
                    c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                    self.rows.append((float(p["current_socket_power"]), float(c["clk"])))
                except Exception:
                    pass

        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def stop(self):
        if self.thread is None:
            return None
        self._stop.set()
        self.thread.join()
        if not self.rows:
            return None
        n = len(self.rows)
        return {"socket_w_avg": sum(r[0] for r in self.rows) / n, "socket_w_max": max(r[0] for r in self.rows),
                "cap_w": self.cap_w, "gfx_clk_mhz_avg": sum(r[1] for r in self.rows) / n, "gfx_clk_mhz_max": 2400,
                "samples": n, "source": "amdsmi current_socket_power / GFX clk, 50 ms period over the timed region"}


def forward_split(den, dev, x, sigma, n_fwd=12, reps=3):
    """Steady-state time of ONE denoiser forward and its split by kernel.
    whole_ms : one HIP-event pair around n_fwd back-to-back production forwards (the launch chains the solver runs,
               hot clocks), divided by n_fwd -- the number every per-kernel figure is scaled to.
    shares   : per-kernel fractions from ops.unet_profile, which brackets EVERY launch with its own event pair (that
               serialises the launch chains and adds a gap per launch, so its absolute sum is a few per cent longer
               than a real forward; only the ratios are used)."""
    ctx = den.context(dev)
    for _ in range(4):          # clocks and caches in the state of a running episode
        den(x, sigma)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n_fwd):
        den(x, sigma)
    e1.record()
    torch.cuda.synchronize()
    whole_ms = e0.elapsed_time(e1) / n_fwd
    per, fl = {}, 0.0
    for _ in range(reps):
        for name, ms, f in ops.unet_profile(ctx, x, sigma):
            per[name] = per.get(name, 0.0) + ms
            if name in ("conv3x3", "conv3x3_wino"):
                fl += f
    tot = sum(per.values())
    return whole_ms, {k: v / tot for k, v in per.items()}, fl / reps, tot / reps


def roofline(den, dev, x, sigma):
    """MFMA roofline of the dominant kernel family (the 27 conv3x3 layers of one denoiser forward) of the family the context runs:
    conv_mode 0 (default) -> roofline_fp32, conv_mode 1 (--ctx-option conv_mode=1, or the fast_mode leg) -> roofline_hs."""
    if den.context(dev).get_option("conv_mode") == 0:
        return roofline_fp32(den, dev, x, sigma, n_fwd=12)
    return roofline_hs(den, dev, x, sigma)


def roofline_hs(den, dev, x, sigma):
    """Half-split family: algorithmic FLOPs (2*9*Cin*Cout*H*W*B per launch) / time of exactly those launches inside a steady-state
    production forward (forward_split: bracketed whole forwards x the per-kernel share) against dense f16 MFMA peak / 3."""
    B, _, H, W = x.shape
    whole_ms, shares, conv_fl, profiled_ms = forward_split(den, dev, x, sigma)
    conv_ms = whole_ms * (shares.get("conv3x3", 0.0) + shares.get("conv3x3_wino", 0.0))
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12
    return {
        "bound": "mfma",
        "kernel": "conv_hs_kernel: the 27 3x3 convolutions of one denoiser forward (half-split f16 MFMA, 3 MFMAs per fp32 "
                  "product; since r5 two launch chains of 30 launches each over halves of the batch at B=48, the K=18 first layer on the vector ALU "
                  "included)",
        "achieved": achieved,
        "peak": PEAK_HS_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / PEAK_HS_TFLOPS,
        "peak_note": "dense f16 MFMA peak 2500 TF/s / 3 MFMAs per product; the exact-fp32 MFMA peak is 157.3 TF/s",
        "executed_mfma_flops_per_forward": 3 * conv_fl,
        "executed_mfma_note": "every algorithmic product is 3 f16 MFMA products (hi*hi, hi*lo, lo*hi); no Winograd / FFT "
                              "reduction (profiles/r4_winograd_probe.md), so executed = 3 x algorithmic",
        "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
        # what the same MFMA stream sustains on this GPU (profiles/r1_mfma_microbench.md, tools/micro/mfma_rate.hip):
        # register-only loop on toggling operands 1740 TF/s f16 (power-limited), with conv_hs's fragment reads 1491
        "frac_of_sustained_mfma_loop": achieved / (1740.0 / 3.0),
        "frac_of_sustained_mfma_loop_with_operand_reads": achieved / (1491.0 / 3.0),
        "traffic": pmc_traffic(B, H, W),
        "flops_per_forward": conv_fl,
        "conv_ms_per_forward": conv_ms,
        "denoiser_ms_per_forward": whole_ms,
        "timing": "one HIP-event pair around 12 back-to-back production forwards on the episode's final images; per-kernel "
                  "split = shares of a per-launch-event pass scaled to it (that pass alone sums to "
                  f"{profiled_ms:.3f} ms: ONE chain, launches serialised; the production forward overlaps two chains)",
        "ms_by_kernel": {k: whole_ms * v for k, v in shares.items()},
    }


def batch_table(solver, dev, H, W, ratio, sizes=(6, 12, 24, 48), T=ACTION_PACK, reps=3):
    """One ADMMSolver_CSMRI call (T inner iterations) at the per-rank batch sizes of an 8 / 4 / 2 / 1-way strong split
    of env_batch 48: ms per iteration, image-iterations/s and per-image efficiency relative to B=48 (= the predicted
    strong-scaling efficiency at 48/B ranks, reward all_gather aside)."""
    rows = []
    for b in sizes:
        d = synth.make_csmri_batch(b, H, W, ratio=ratio, sigma_n=15.0, seed=77)
        a = synth.make_actions(b, N_POLICY_STEPS, ACTION_PACK)[0]
        g = lambda v: t(v).to(dev)
        v0 = solver.reset({"x0": g(d["x0"])})
        aux = (g(d["y0"]), g(d["mask"]))
        par = (g(a["sigma_d"][:, :T]), g(a["mu"][:, :T]))
        solver((v0, aux), par)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            solver((v0, aux), par)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / (reps * T)
        rows.append({"B": b, "ms_per_iter": ms, "image_iters_per_s": b / (ms * 1e-3)})
    ref = rows[-1]["image_iters_per_s"]
    for r in rows:
        r["per_image_efficiency_vs_B48"] = r["image_iters_per_s"] / ref
    return rows


def fast_mode(params, data, actions, dev, B, H, W, steps, warmup, den_default, table_ratio):
    """The opt-in FAST mode (conv_mode 1: csrc/conv_hs_kernel.h, every fp32 value an f16 hi + lo pair, 3 f16 MFMAs per product; 22-bit
    significand, i.e. narrower than the reference's fp32 -- 2 of 12 seeds of the expansive 30-iteration drift table end above 1e-4 from
    the fp32 oracle, profiles/r5_drift_seeds.md) on the same episode with the same --steps / --warmup as the headline, with its own
    roofline against dense f16 MFMA peak / 3 and its own batch table.  Until r5 this was the headline leg."""
    den = UNetDenoiser2D(state_dict=params, conv_mode=1)
    solver = ADMMSolver_CSMRI(den)
    env = CSMRIEnv(None, solver, max_episode_step=N_POLICY_STEPS)

    def episode():
        env.reset(data)
        for a in actions:
            env.step(a)
    for _ in range(max(1, warmup)):
        episode()
    torch.cuda.synchronize()
    sampler = PowerSampler()
    t0 = time.perf_counter()
    for _ in range(steps):
        episode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    power = sampler.stop()
    den.context(dev).status()                                # half-split range guard: raises if any call overflowed
    x = env.state["output"].detach().clone()
    sigma = actions[-1]["sigma_d"][:, -1].contiguous()
    rl = roofline_hs(den, dev, x, sigma)
    rl["power"] = power
    if power and power.get("gfx_clk_mhz_avg"):
        rl["frac_of_peak_at_measured_clock"] = rl["achieved"] / (rl["peak"] * power["gfx_clk_mhz_avg"] / 2400.0)
    # accuracy gate of this leg (the headline's gate is the CPU oracle, cpu_baseline): one forward of both families on the episode's
    # final images -- a timing of wrong results is not a measurement
    with torch.no_grad():
        sg = torch.as_tensor(sigma).to(dev)
        yhs = den(x, sg)
        y32 = den_default(x, sg)
    rel_hs = float((y32.double() - yhs.double()).norm() / y32.double().norm())
    if not (rel_hs < 1e-4):
        raise RuntimeError(f"fast_mode: denoiser forward differs from the default fp32 family by {rel_hs} (must be < 1e-4)")
    out = {"forward_rel_l2_vs_default_family": rel_hs, "value": N_POLICY_STEPS * ACTION_PACK / dt, "unit": "iters/s", "steps": steps,
           "warmup": max(1, warmup), "ms_per_step": 1e3 * dt, "dtype": DTYPE_HS, "iters_per_s": N_POLICY_STEPS * ACTION_PACK / dt,
           "roofline": rl}
    if table_ratio is not None:
        out["batch_table"] = batch_table(solver, dev, H, W, table_ratio)
    return out


def roofline_fp32(den, dev, x, sigma, n_fwd=12):
    """Roofline of the fp32 convolution family (conv_mode 0) against the 157.3 TF/s fp32-MFMA peak: `achieved` counts the FLOPs
    the matrix pipe EXECUTES (Winograd layers: 16 products per 2x2 outputs, not 36); the algorithmic rate is given beside it."""
    whole_ms, shares, conv_fl, _ = forward_split(den, dev, x, sigma, n_fwd=n_fwd)
    conv_ms = whole_ms * (shares.get("conv3x3", 0.0) + shares.get("conv3x3_wino", 0.0))
    tf = conv_fl / (conv_ms * 1e-3) / 1e12
    wino_fl = sum(f for name, _, f in ops.unet_profile(den.context(dev), x, sigma) if name == "conv3x3_wino")
    # the matrix pipe executes 16/36 of the Winograd launches' algorithmic FLOPs and none of the first convolution's (vector ALU, r5)
    B, _, H, W = x.shape
    first_fl = 2.0 * 9 * 2 * 32 * H * W * B if W % 4 == 0 else 0.0
    executed = conv_fl - wino_fl * (1.0 - 16.0 / 36.0) - first_fl
    tf_exec = executed / (conv_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv3x3_wino8_f32_kernel<64|32> (26 launches per denoiser forward, 4 of them with the bilinear x2 "
                                       "up-sampling inside) + conv_first_f32_kernel",
            "achieved": tf_exec, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf_exec / PEAK_FP32_MFMA_TFLOPS,
            "achieved_note": "EXECUTED MFMA FLOPs / time (the honest utilisation of the fp32 matrix pipe); the algorithmic rate "
                             "of the same launches is `algorithmic_tflops`",
            "algorithmic_tflops": tf, "algorithmic_flops_per_forward": conv_fl, "executed_mfma_flops_per_forward": executed,
            "conv_ms_per_forward": conv_ms, "denoiser_ms_per_forward": whole_ms,
            "ms_by_kernel": {k: whole_ms * v for k, v in shares.items()},
            "timing": "one HIP-event pair around back-to-back production forwards (two launch chains) on the episode's final images; "
                      "conv share from a per-launch-event pass",
            "traffic": pmc_traffic(x.shape[0], x.shape[2], x.shape[3], "_fp32")}


def pmc_traffic(B, H, W, suffix=""):
    """HBM bytes of the conv launches of one denoiser forward, from the newest committed rocprofv3 PMC passes
    (profiles/r*_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, per MI355X_MICROARCH.md; counters cannot be collected
    inside a timed run).  Same aggregation as `achieved` (all conv launches of one forward).  None when no PMC pass
    exists for this geometry."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")))
    if not cands:
        return None
    path = cands[-1]
    try:
        t = json.load(open(path))
    except OSError:
        return None
    if (t.get("B"), t.get("H"), t.get("W")) != (B, H, W):
        return None
    return {"bytes_per_forward": t["hbm_bytes_per_forward"], "bytes_per_conv_launch": t["hbm_bytes_per_conv_launch"],
            "algorithmic_bytes_per_forward": conv_algorithmic_bytes(B, H, W, 2 if suffix else 16), "source": t["source"]}


def conv_algorithmic_bytes(B, H, W, min_cin=16):
    """Input + output activation bytes of the 27 convolutions (4 B per value: f16 hi + f16 lo, or one fp32), weights excluded;
    min_cin: channel count the first layer's input tensor is stored with (16 in the half-split layout, 2 in the fp32 one)."""
    blocks = [(2, 32, 0), (32, 64, 1), (64, 128, 2), (128, 256, 3), (256, 512, 4), (768, 256, 3), (384, 128, 2),
              (192, 64, 1), (96, 32, 0)]
    tot = 0
    for cin, cout, lvl in blocks:
        px = (H >> lvl) * (W >> lvl) * B
        for j in range(3):
            ci = cin if j == 0 else cout
            tot += 4 * px * (max(ci, min_cin) + cout)
    return tot


def cpu_quota():
    """What this process may actually use of the host: scheduler affinity and the cgroup CPU bandwidth limit (a container that owns
    8 of 256 cores gets slower, not faster, with more threads)."""
    q = {"affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            q["cgroup:" + os.path.basename(path)] = open(path).read().strip()
        except OSError:
            pass
    return q


def cpu_baseline(params, solver, dev, args, gpu_value):
    """The CPU oracle (oracle/pnp_oracle.py, pinned to the reference by tests/golden) timed on the host cores on
    a bounded sample: cpu_batch items x cpu_iters inner iterations of the same workload.  Also the accuracy gate:
    the HIP path on the same sample must match to 1e-4 relative L2."""
    from oracle import pnp_oracle as O           # the checker / timed baseline, never the product path
    cores = os.cpu_count() or 1
    Bc, Tc = args.cpu_batch, args.cpu_iters
    H = W = args.size
    d = synth.make_csmri_batch(Bc, H, W, ratio=args.ratio, sigma_n=15.0, seed=4321)
    a = synth.make_actions(Bc, N_POLICY_STEPS, ACTION_PACK)[0]
    sig, mu = t(a["sigma_d"][:, :Tc]), t(a["mu"][:, :Tc])
    oden = O.Denoiser(params)
    v0 = O.admm_reset(t(d["x0"]))
    with torch.no_grad():
        # oneDNN convolutions do not scale to every core of a large host (and a container may own a fraction of them: the quota
        # is reported below): calibrate the thread count ON THE BATCH THE SAMPLE USES (all Bc items x 1 iteration per candidate;
        # r4 calibrated on 4 items, whose working set behaves differently from 24 items' -- VERDICT r4 weak #8) and time the
        # sample with the fastest one.  The sample's rate and the calibration's best rate are both in the JSON.
        best, threads, calib = None, cores, []
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
        for n in sorted({c for c in (8, 16, 32, 64, usable) if c <= usable}):
            torch.set_num_threads(n)
            O.csmri_admm(oden, v0[:1], t(d["y0"][:1]), t(d["mask"][:1]), sig[:1, :1], mu[:1, :1])   # warm up
            t0 = time.perf_counter()
            O.csmri_admm(oden, v0, t(d["y0"]), t(d["mask"]), sig[:, :1], mu[:, :1])
            dt = time.perf_counter() - t0
            calib.append({"threads": n, "s_per_iteration_of_the_sample_batch": dt, "image_iters_per_s": Bc / dt})
            if best is None or dt < best:
                best, threads = dt, n
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        ref = O.csmri_admm(oden, v0, t(d["y0"]), t(d["mask"]), sig, mu)
        dt = time.perf_counter() - t0
        # a single-thread figure (1 item x 1 iteration) ...
        torch.set_num_threads(1)
        t1 = time.perf_counter()
        O.csmri_admm(oden, v0[:1], t(d["y0"][:1]), t(d["mask"][:1]), sig[:1, :1], mu[:1, :1])
        dt_1thread = time.perf_counter() - t1
        # ... and BASELINE config #1 in full: B = 1, 128 x 128, 6 x 5 = 30 inner iterations
        torch.set_num_threads(threads)
        d1 = synth.make_csmri_batch(1, 128, 128, ratio=args.ratio, sigma_n=15.0, seed=99)
        acts1 = synth.make_actions(1, N_POLICY_STEPS, ACTION_PACK)
        v1 = O.admm_reset(t(d1["x0"]))
        t1 = time.perf_counter()
        for a1 in acts1:
            v1 = O.csmri_admm(oden, v1, t(d1["y0"]), t(d1["mask"]), t(a1["sigma_d"]), t(a1["mu"]))
        dt_cfg1 = time.perf_counter() - t1
    g1 = lambda a_: t(a_).to(dev)
    gv1 = solver.reset({"x0": g1(d1["x0"])})
    for a1 in acts1:                                               # warm-up (workspace for 128 x 128)
        gv1 = solver((gv1, (g1(d1["y0"]), g1(d1["mask"]))), (g1(a1["sigma_d"]), g1(a1["mu"])))
    gv1 = solver.reset({"x0": g1(d1["x0"])})
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for a1 in acts1:
        gv1 = solver((gv1, (g1(d1["y0"]), g1(d1["mask"]))), (g1(a1["sigma_d"]), g1(a1["mu"])))
    torch.cuda.synchronize()
    dt_cfg1_gpu = time.perf_counter() - t1
    rel_cfg1 = float((gv1.cpu() - v1).norm() / v1.norm())
    # the same configuration on a NON-EXPANSIVE weight set (PyTorch default-init scale, synth.make_unet_params_default): the
    # He-scaled synthetic network amplifies round-off x1.5-2 per call, a denoiser-like one does not -- this is the margin
    # to the 1e-4 bar that a trained checkpoint would see
    pdef = synth.make_unet_params_default(0)
    with torch.no_grad():
        odef = O.Denoiser(pdef)
        vd = O.admm_reset(t(d1["x0"]))
        for a1 in acts1:
            vd = O.csmri_admm(odef, vd, t(d1["y0"]), t(d1["mask"]), t(a1["sigma_d"]), t(a1["mu"]))
    sdef = ADMMSolver_CSMRI(UNetDenoiser2D(state_dict=pdef))
    gvd = sdef.reset({"x0": g1(d1["x0"])})
    for a1 in acts1:
        gvd = sdef((gvd, (g1(d1["y0"]), g1(d1["mask"]))), (g1(a1["sigma_d"]), g1(a1["mu"])))
    rel_cfg1_def = float((gvd.cpu() - vd).norm() / vd.norm())
    gv0 = solver.reset({"x0": t(d["x0"]).to(dev)})
    got = solver((gv0, (t(d["y0"]).to(dev), t(d["mask"]).to(dev))), (sig.to(dev), mu.to(dev))).cpu()
    rel = float((got - ref).norm() / ref.norm())
    image_iters_per_s = Bc * Tc / dt
    base = {
        "value": image_iters_per_s / args.batch,      # same unit as `value`: iterations over a 48-image batch / s
        "unit": "iters/s",
        "cores": threads,
        "host_cores": cores,
        "cpu_quota": cpu_quota(),
        "thread_calibration": calib,
        "calibration_best_image_iters_per_s": Bc / best,
        "kind": "port",
        "sample": f"{Bc} items x {Tc} inner iterations of CS-MRI ADMM {H}x{W} ({dt:.1f} s of CPU wall), "
                  f"scaled to env_batch={args.batch}",
        "image_iters_per_s": image_iters_per_s,
        "one_thread": {"image_iters_per_s": 1.0 / dt_1thread, "sample": f"1 item x 1 iteration {H}x{W}, 1 thread"},
        "config1_full": {"workload": "BASELINE configs[0]: CS-MRI ADMM 128x128, B=1, 6 x 5 = 30 iterations",
                         "cpu_s": dt_cfg1, "cpu_iters_per_s": 30.0 / dt_cfg1, "cpu_threads": threads,
                         "gpu_s": dt_cfg1_gpu, "gpu_iters_per_s": 30.0 / dt_cfg1_gpu, "rel_l2_gpu_vs_cpu": rel_cfg1,
                         "rel_l2_gpu_vs_cpu_default_init_scale_weights": rel_cfg1_def},
    }
    return base, rel


if __name__ == "__main__":
    main()
