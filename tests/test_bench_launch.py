"""`python bench.py --gpus N` must start by itself: with WORLD_SIZE unset it re-executes under
torch.distributed.run, one rank per device, and rank 0 prints the single JSON line.  Driven here on CPU through
--selftest-cpu (gloo + a stub solver): same launch, sharding, per-step exchange and timing code as the GPU run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-cpu", "--steps", "2", "--warmup", "1",
                        *extra], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_self_launches_two_ranks(scaling):
    out = _run("--gpus", "2", "--scaling", scaling)
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2 and out["scaling"] == scaling
    assert out["config"]["global_batch"] == (96 if scaling == "weak" else 48)
    assert out["reward_rows_seen"] == out["config"]["global_batch"]      # every rank's rewards reached rank 0
    assert out["collectives_per_env_step"] == 1.0
    assert "SELFTEST" in out["metric"]                                    # can never be mistaken for a measurement


def test_bench_single_rank_needs_no_launcher():
    out = _run("--gpus", "1")
    assert out["n_gpus"] == 1 and out["reward_rows_seen"] == 48


def test_strong_and_weak_rewards_agree_with_single_process():
    """The strong split of ONE 48-item batch over two ranks sees the same rewards as one process running all 48."""
    one = _run("--gpus", "1")
    two = _run("--gpus", "2", "--scaling", "strong")
    assert abs(one["reward_sum"] - two["reward_sum"]) < 1e-4 * abs(one["reward_sum"])


def test_bench_without_gpu_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)


def test_bench_self_launches_eight_ranks_strong_uneven():
    """`bench.py --gpus 8 --scaling strong` on BASELINE config #3's env batch of 36: 5/5/5/5/4/4/4/4 items per rank, one
    exchange per env step, every rank's rewards on rank 0, same reward sum as one process running all 36."""
    eight = _run("--gpus", "8", "--scaling", "strong", "--batch", "36")
    assert eight["n_gpus"] == 8 and eight["world_size_seen"] == 8 and eight["scaling"] == "strong"
    assert eight["config"]["global_batch"] == 36 and eight["reward_rows_seen"] == 36
    assert eight["collectives_per_env_step"] == 1.0
    one = _run("--gpus", "1", "--batch", "36")
    assert abs(one["reward_sum"] - eight["reward_sum"]) < 1e-4 * abs(one["reward_sum"])
