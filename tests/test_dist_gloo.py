"""World-size-2 gloo test of the multi-GPU layout (tfpnp_amd/dist.py): shard the env batch, run the per-rank
episode bookkeeping with a stand-in reward, all_gather per-item rewards -> every rank sees the global vector."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tfpnp_amd import dist as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_bounds(n_items, world, rank)
    items = torch.arange(n_items, dtype=torch.float32)
    local_reward = (items[lo:hi] * 2 + 1).view(-1, 1)            # per-item "delta PSNR"
    local_done = (items[lo:hi] % 2 == 0).to(torch.float32).view(-1, 1)
    rewards = D.all_gather_rows(local_reward, n_items)
    done = D.all_gather_rows(local_done, n_items)
    D.barrier()
    tmax = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    q.put((rank, rewards.view(-1).tolist(), done.view(-1).tolist(), tmax))
    dist.destroy_process_group()


def test_all_gather_rows_world2_uneven():
    world, n_items = 2, 5          # uneven shards: 3 + 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_r = [2.0 * i + 1 for i in range(n_items)]
    exp_d = [1.0 if i % 2 == 0 else 0.0 for i in range(n_items)]
    for rank, rewards, done, tmax in res:
        assert rewards == exp_r and done == exp_d and tmax == 2.0


def test_single_process_passthrough():
    from tfpnp_amd import dist as D
    x = torch.arange(6.0).view(3, 2)
    assert torch.equal(D.all_gather_rows(x, 3), x)
    assert D.max_over_ranks(1.5, torch.device("cpu")) == 1.5
