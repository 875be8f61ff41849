"""The RCCL / side-stream path of the per-step exchange (tfpnp_amd/dist.py::StepExchange) on ONE GPU: a one-rank 'nccl'
process group with force_collective=True runs exactly the code the 2/4/8-GPU bench runs (pack on the compute stream,
all_gather_into_tensor on the side stream, pinned-host landing, deferred result) -- the multi-rank arithmetic itself is
covered by the world-size-2 gloo tests."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_step_exchange_rccl_side_stream_single_rank():
    from tfpnp_amd import dist as D
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        ex = D.StepExchange(5, dev, force_collective=True)
        assert ex.on and ex.stream is not None
        pend = None
        for step in range(6):                                   # pipelined exactly like bench.py::make_episode
            reward = (torch.arange(5, device=dev, dtype=torch.float32) * 10 + step).view(5, 1)
            done = torch.arange(5, device=dev) <= step
            # keep the compute stream busy so that the exchange really overlaps something
            junk = torch.rand(2048, 2048, device=dev) @ torch.rand(2048, 2048, device=dev)
            nxt = ex.post(reward, done, rank_finished=step >= 4)
            if pend is not None:
                r, d, f = pend.result()
                assert r.view(-1).tolist() == [10.0 * i + step - 1 for i in range(5)]
                assert d.tolist() == [i <= step - 1 for i in range(5)] and f == (step - 1 >= 4)
            pend = nxt
        r, d, f = pend.result()
        assert r.view(-1).tolist() == [10.0 * i + 5 for i in range(5)] and f is True
        assert ex.posted == 6 and junk.isfinite().all()
        # the blocking ShardedEnv contract on the same path
        assert D.max_over_ranks(1.25, dev) == 1.25 and D.all_true(True, dev)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_share_the_gpu(scaling):
    """`python bench.py --gpus 2` end to end with the REAL native solver on a one-GPU box: bench.py starts its two ranks itself,
    both use device 0 over gloo (PNPX_BENCH_SHARE_GPU; RCCL refuses two ranks on one device), shard the batch, run the episode
    with one exchange per env step and rank 0 prints the one JSON line.  Everything but the RCCL transport of the N > 1 path."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PNPX_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "6",
           "--size", "64", "--scaling", scaling, "--no-cpu-baseline", "--no-batch-table", "--no-fast-mode"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["collectives_per_env_step"] == 1.0
    assert d["config"]["global_batch"] == (6 if scaling == "strong" else 12)
    assert "NOT_A_MEASUREMENT" in d
    # the roofline object is self-consistent with the step clock: 30 denoiser forwards fit inside one step
    rf = d["roofline"]
    assert 0 < rf["conv_ms_per_forward"] <= rf["denoiser_ms_per_forward"]
    assert rf["denoiser_ms_per_forward"] * d["config"]["iters_per_step"] <= d["ms_per_step"]
    assert abs(sum(rf["ms_by_kernel"].values()) - rf["denoiser_ms_per_forward"]) < 1e-6 * rf["denoiser_ms_per_forward"] + 1e-9
