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


# ------------------------------------------------------------------------------------------------------------------
# Episode loop over two ranks (ShardedEnv): uneven shards, per-item early stopping, one rank finishing first.
class _StubSolver(torch.nn.Module):
    """CPU stand-in with the PnPSolver contract: x <- x + mu * (gt_hint - x) per call (no GPU needed)."""

    def reset(self, data):
        return data['x0'].clone()

    def get_output(self, state):
        return state

    def filter_aux_inputs(self, state):
        return (state['gt'],)

    def filter_hyperparameter(self, action):
        return (action['mu'],)

    def forward(self, inputs, parameters):
        x, (gt,) = inputs
        (mu,) = parameters
        return x + mu.view(-1, 1, 1, 1) * (gt - x)


def _count_collectives():
    """Wrap every torch.distributed collective entry point with a counter (this process only)."""
    counts = {}
    for name in ("all_gather", "all_gather_into_tensor", "all_reduce", "broadcast", "reduce", "gather", "scatter",
                 "all_to_all", "all_gather_object", "reduce_scatter", "barrier"):
        orig = getattr(dist, name)

        def wrapped(*a, _orig=orig, _name=name, **k):
            counts[_name] = counts.get(_name, 0) + 1
            return _orig(*a, **k)
        setattr(dist, name, wrapped)
    return counts


def _episode_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tfpnp_amd import dist as D
    from tfpnp_amd.env.base import PnPEnv
    D.init_from_env(backend="gloo")
    counts = _count_collectives()

    class StubEnv(PnPEnv):
        ob_keys = ()

    B = 5
    rs = np.random.RandomState(0)
    gt = torch.from_numpy(rs.rand(B, 1, 4, 4).astype(np.float32))
    data = {'gt': gt, 'x0': torch.zeros_like(gt), 'output': torch.zeros_like(gt)}
    env = StubEnv(None, _StubSolver(), max_episode_step=4)
    env.metric_fn = lambda out, g: -((out - g) ** 2).reshape(out.shape[0], -1).mean(1, keepdim=True)   # CPU metric
    senv = D.ShardedEnv(env)
    ob = senv.reset(data)
    # item i stops after step stop_at[i]; rank 1's shard (items 3, 4) finishes after step 1, rank 0's after step 3
    stop_at = torch.tensor([3, 2, 3, 1, 1])
    lo, hi = D.shard_bounds(B, world, rank)
    log = []
    for step in range(1, 5):
        live = env.idx_left.clone() if not senv._local_done else torch.empty(0, dtype=torch.long)
        action = {'mu': torch.full((len(live),), 0.5), 'idx_stop': (stop_at[lo:hi][live] <= step).long()}
        before = dict(counts)
        ob, rewards, finished, info = senv.step(action)
        made = {k: v - before.get(k, 0) for k, v in counts.items() if v != before.get(k, 0)}
        assert made == {"all_gather_into_tensor": 1}, made       # ONE small collective per env step
        log.append((rewards.view(-1).tolist(), info['done'].tolist(), finished))
        if finished:
            break
    assert senv.exchange.posted == len(log)
    q.put((rank, log))
    dist.destroy_process_group()


def test_sharded_env_episode_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_episode_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                   # every rank sees the same global trajectory
    log = res[0]
    assert len(log) == 3 and [f for _, _, f in log] == [False, False, True]
    rew1, done1, _ = log[0]
    assert all(r > 0 for r in rew1)                           # every item improved in step 1
    assert done1 == [False, False, False, True, True]
    rew2, done2, _ = log[1]
    assert rew2[3] == 0.0 and rew2[4] == 0.0 and rew2[0] > 0  # rank 1 finished: zero rewards, still in the collectives
    assert done2 == [False, True, False, True, True]
    rew3, done3, _ = log[2]
    assert rew3[1] == 0.0 and rew3[0] > 0 and done3 == [True] * 5


def _pipelined_worker(rank, world, port, q):
    """StepExchange used the way bench.py uses it: post step k, resolve it after step k+1 has been posted."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tfpnp_amd import dist as D
    D.init_from_env(backend="gloo")
    counts = _count_collectives()
    n_global = 7                                         # uneven: 4 + 3
    ex = D.StepExchange(n_global, torch.device("cpu"))
    lo, hi = D.shard_bounds(n_global, world, rank)
    items = torch.arange(n_global, dtype=torch.float32)
    got, pending = [], None
    for step in range(5):
        reward = (items[lo:hi] * 10 + step).view(-1, 1)
        done = (items[lo:hi] <= step)
        nxt = ex.post(reward, done, rank_finished=bool(done.all()))
        if pending is not None:
            got.append(pending.result())
        pending = nxt
    got.append(pending.result())
    assert counts == {"all_gather_into_tensor": 5}, counts
    q.put((rank, [(r.view(-1).tolist(), d.tolist(), f) for r, d, f in got]))
    dist.destroy_process_group()


def test_step_exchange_pipelined_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    for step, (rew, done, fin) in enumerate(res[0]):
        assert rew == [10.0 * i + step for i in range(7)]
        assert done == [i <= step for i in range(7)]
        assert fin == (step >= 6)          # never within 5 steps: rank 1 holds item 6
    # single process: the exchange degenerates to a local view, no process group needed
    from tfpnp_amd import dist as D
    ex = D.StepExchange(3, torch.device("cpu"))
    r, d, f = ex.post(torch.tensor([[1.0], [2.0], [3.0]]), torch.tensor([1, 0, 1]), True).result()
    assert r.view(-1).tolist() == [1.0, 2.0, 3.0] and d.tolist() == [True, False, True] and f is True


# ------------------------------------------------------------------------------------------------------------------
# bench.py --scaling strong: ONE global env batch of 48 (the reference's DataParallel scatter, tasks/csmri/main.py:79-80)
# split contiguously over the ranks; per step every rank contributes the rewards of ITS items to one all_gather.
def _strong_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tfpnp_amd import dist as D
    D.init_from_env(backend="gloo")
    n_global, steps = 48, 6
    lo, hi = D.shard_bounds(n_global, world, rank)
    rs = np.random.RandomState(0)                        # every rank builds the same global batch, keeps its rows
    gt = torch.from_numpy(rs.rand(n_global, 1, 4, 4).astype(np.float32))
    x = torch.zeros_like(gt)[lo:hi]
    log = []
    for s in range(steps):
        before = -((x - gt[lo:hi]) ** 2).reshape(hi - lo, -1).mean(1, keepdim=True)
        x = x + 0.5 * (gt[lo:hi] - x)
        after = -((x - gt[lo:hi]) ** 2).reshape(hi - lo, -1).mean(1, keepdim=True)
        log.append(D.all_gather_rows(after - before, n_global).view(-1))
    q.put((rank, (lo, hi), torch.stack(log).numpy()))
    dist.destroy_process_group()


def test_strong_split_of_one_env_batch_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (b, log) for r, b, log in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == (0, 24) and res[1][0] == (24, 48)            # 48 / 2 contiguous items per rank
    assert np.array_equal(res[0][1], res[1][1])                      # both ranks hold the global reward table
    # ... and it equals the single-process run of the whole batch
    rs = np.random.RandomState(0)
    gt = torch.from_numpy(rs.rand(48, 1, 4, 4).astype(np.float32))
    x = torch.zeros_like(gt)
    for s in range(6):
        before = -((x - gt) ** 2).reshape(48, -1).mean(1)
        x = x + 0.5 * (gt - x)
        after = -((x - gt) ** 2).reshape(48, -1).mean(1)
        assert np.allclose(res[0][1][s], (after - before).numpy(), rtol=1e-6, atol=1e-7)
    from tfpnp_amd import dist as D
    for g in (1, 2, 4, 8, 5):                                        # shard sizes of the strong split, incl. uneven
        bounds = [D.shard_bounds(48, g, r) for r in range(g)]
        assert bounds[0][0] == 0 and bounds[-1][1] == 48 and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
        assert max(h - l for l, h in bounds) - min(h - l for l, h in bounds) <= 1


# ------------------------------------------------------------------------------------------------------------------
# World size 8 (the node the driver scales to): BASELINE config #3's env batch of 36 does not divide by 8 -- shards of
# 5/5/5/5/4/4/4/4 -- through ShardedEnv with per-item early stopping; ranks finish at different steps and keep joining the
# one collective per env step until the whole batch is done.
def _episode8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from tfpnp_amd import dist as D
    from tfpnp_amd.env.base import PnPEnv
    D.init_from_env(backend="gloo")
    counts = _count_collectives()

    class StubEnv(PnPEnv):
        ob_keys = ()

    n_global = 36
    lo, hi = D.shard_bounds(n_global, world, rank)
    rs = np.random.RandomState(0)
    gt_all = torch.from_numpy(rs.rand(n_global, 1, 4, 4).astype(np.float32))
    gt = gt_all[lo:hi].clone()
    data = {'gt': gt, 'x0': torch.zeros_like(gt), 'output': torch.zeros_like(gt)}
    env = StubEnv(None, _StubSolver(), max_episode_step=6)
    env.metric_fn = lambda out, g: -((out - g) ** 2).reshape(out.shape[0], -1).mean(1, keepdim=True)
    senv = D.ShardedEnv(env)
    senv.reset(data, pre_sharded=True, n_global=n_global)          # every rank built only its own rows
    stop_at = 1 + (torch.arange(n_global) * 7) % 5           # item i stops after step 1..5 (global, deterministic)
    log = []
    for step in range(1, 7):
        live = env.idx_left.clone() if not senv._local_done else torch.empty(0, dtype=torch.long)
        action = {'mu': torch.full((len(live),), 0.5), 'idx_stop': (stop_at[lo:hi][live] <= step).long()}
        before = dict(counts)
        _, rewards, finished, info = senv.step(action)
        made = {k: v - before.get(k, 0) for k, v in counts.items() if v != before.get(k, 0)}
        assert made == {"all_gather_into_tensor": 1}, made
        log.append((rewards.view(-1).tolist(), info['done'].tolist(), finished))
        if finished:
            break
    q.put((rank, (lo, hi), log))
    dist.destroy_process_group()


def test_sharded_env_episode_world8_uneven_36():
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_episode8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (b, log) for r, b, log in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [res[r][0][1] - res[r][0][0] for r in range(world)] == [5, 5, 5, 5, 4, 4, 4, 4]
    assert res[0][0] == (0, 5) and res[7][0] == (32, 36)
    for r in range(1, world):
        assert res[r][1] == res[0][1]                           # every rank holds the same global trajectory
    log = res[0][1]
    stop_at = [1 + (i * 7) % 5 for i in range(36)]
    assert len(log) == 5 and [f for _, _, f in log] == [False] * 4 + [True]
    for step, (rew, done, _) in enumerate(log, start=1):
        assert len(rew) == 36 and done == [s <= step for s in stop_at]
        for i in range(36):       # an item earns a reward exactly while it is live (stopped items: zero rows, still gathered)
            assert (rew[i] > 0) == (stop_at[i] >= step), (step, i)
