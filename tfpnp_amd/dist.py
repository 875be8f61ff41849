"""Multi-GPU layout of the hot path: one process per GPU, batch items sharded, no data-path collective.

The reference's only multi-GPU mechanism is torch.nn.DataParallel over the batch dimension of the whole solver
call (tasks/csmri/main.py:79-80; tfpnp/policy/sync_batchnorm/replicate.py:50-75): every env step it scatters the
state (1.5 MB per 256x256 item), replicates the module and gathers the result on GPU 0.  Batch items are fully
independent in every solver loop, so here rank r of G owns a fixed contiguous shard of the env batch for the
whole episode -- state, y0/mask and packed weights stay resident on its GPU -- and the ONLY per-step exchange is
an all_gather of the per-item rewards / done flags ([B_local] fp32 each) so that every rank sees the whole
batch's PSNR trajectory.  Backend 'nccl' is RCCL over xGMI on ROCm; 'gloo' is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous, balanced partition: the first (n % G) ranks get one extra item."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(data, world_size, rank):
    """dict of [B,...] tensors/arrays -> this rank's rows."""
    B = len(next(iter(data.values())))
    lo, hi = shard_bounds(B, world_size, rank)
    return {k: v[lo:hi] for k, v in data.items()}


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def all_gather_rows(local, n_total, group=None):
    """local: [B_local, ...] -> [n_total, ...] in global item order (uneven shards allowed).

    One small all_gather; shards are padded to the largest shard size so the collective is regular."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    """max of a python float over ranks (timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_true(flag, device):
    """logical AND of a python bool over ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


class StepExchange:
    """The ONE collective of an env step: every rank contributes a pre-allocated `[B_max, 3]` fp32 block --
    per local item (delta-PSNR reward, done flag) and, in column 2, "this rank has no live items left" -- to a single
    `all_gather_into_tensor`, issued OFF the compute stream and consumed later (typically at the next step's host read),
    so the exchange overlaps with the next step's kernels.  No allocation after construction: two send / receive / pinned
    host buffers alternate, so step k+1 may be posted before step k's result has been looked at.

        ex = StepExchange(n_global, device)
        pending = ex.post(reward_local, done_local, rank_finished)     # returns immediately
        ...                                                            # next step's kernels are launched here
        rewards, done, all_finished = pending.result()                 # [n_global,1], bool [n_global], python bool

    RCCL ('nccl' on ROCm): the collective runs on a side stream that waits for an event recorded on the compute stream
    after the block was packed; the gathered block is then copied to pinned host memory on that same side stream and
    `result()` waits for that event only -- the compute stream is never blocked by the collective.  gloo (CPU tests):
    `async_op=True` + `wait()` at consumption."""

    COLS = 3

    def __init__(self, n_global, device, group=None, force_collective=False):
        """force_collective: issue the collective even in a one-rank group (exercises the RCCL / side-stream path on a
        single GPU: tests/test_gpu_dist.py)."""
        self.group = group
        self.on = dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.n_global = int(n_global)
        self.device = torch.device(device)
        bounds = [shard_bounds(self.n_global, self.world, r) for r in range(self.world)]
        self.lo, self.hi = bounds[self.rank]
        self.mx = max(1, max(hi - lo for lo, hi in bounds))
        rows = [r * self.mx + i for r, (lo, hi) in enumerate(bounds) for i in range(hi - lo)]
        self.rows = torch.tensor(rows, dtype=torch.int64, device=self.device)    # global item -> row of the gathered block
        self.rank_rows = torch.arange(self.world, device=self.device) * self.mx   # one row per rank (its finished flag)
        cuda = self.device.type == 'cuda'
        mk = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.device)
        self.send = [mk(self.mx, self.COLS) for _ in range(2)]
        self.recv = [mk(self.world * self.mx, self.COLS) for _ in range(2)]
        self.host = [torch.zeros(self.world * self.mx, self.COLS, dtype=torch.float32, pin_memory=True)
                     for _ in range(2)] if cuda else None
        self.stream = torch.cuda.Stream(self.device) if cuda else None
        self.packed = [torch.cuda.Event() for _ in range(2)] if cuda else None
        self.landed = [torch.cuda.Event() for _ in range(2)] if cuda else None
        self.posted = 0            # collectives issued so far (tests assert exactly one per env step)
        self._pending = [None, None]

    def post(self, reward, done, rank_finished):
        """reward: [n_local,1] or [n_local]; done: [n_local] (bool / int / float) or None; rank_finished: python bool."""
        k = self.posted & 1
        if self._pending[k] is not None:          # the buffers of step k-2 are about to be reused: its result is final
            self._pending[k].result()
        n = self.hi - self.lo
        send = self.send[k]
        if n:
            send[:n, 0].copy_(reward.reshape(-1))
            if done is None:
                send[:n, 1].zero_()
            else:
                send[:n, 1].copy_(done.reshape(-1))
        send[:, 2].fill_(1.0 if rank_finished else 0.0)
        self.posted += 1
        p = _Pending(self, k)
        if not self.on:
            p.block = send
        elif self.stream is not None:
            self.packed[k].record()                                  # on the compute stream, after the pack
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(self.packed[k])
                work = dist.all_gather_into_tensor(self.recv[k], send, group=self.group, async_op=True)
                work.wait()                                           # side stream waits for RCCL's stream; host does not
                self.host[k].copy_(self.recv[k], non_blocking=True)
                self.landed[k].record()
            p.block = self.recv[k]
        else:
            p.work = dist.all_gather_into_tensor(self.recv[k], send, group=self.group, async_op=True)
            p.block = self.recv[k]
        self._pending[k] = p
        return p


class _Pending:
    def __init__(self, ex, k):
        self.ex, self.k, self.work, self.block, self._res = ex, k, None, None, None

    def result(self):
        """-> (rewards [n_global,1] on the exchange's device, done bool [n_global], every rank finished?)."""
        if self._res is not None:
            return self._res
        ex = self.ex
        if self.work is not None:
            self.work.wait()
        if ex.on and ex.stream is not None:
            ex.landed[self.k].synchronize()                          # host waits for the side stream's copy only
            torch.cuda.current_stream(ex.device).wait_event(ex.landed[self.k])   # device-side consumers of `block`
            flags = ex.host[self.k]
            finished = bool((flags[::ex.mx, 2] != 0).all())
        else:
            finished = bool((self.block[ex.rank_rows, 2] != 0).all())
        block = self.block.clone()            # the receive buffer is reused two posts later
        self._res = (block[ex.rows, 0:1], block[ex.rows, 1] != 0, finished)
        if ex._pending[self.k] is self:
            ex._pending[self.k] = None
        return self._res


class ShardedEnv:
    """A PnPEnv over this rank's contiguous shard of a global env batch -- what the reference does with
    DataParallelWithCallback(solver) (tasks/csmri/main.py:79-80), without moving images between GPUs.

        env = ShardedEnv(CSMRIEnv(None, solver, max_episode_step))
        ob = env.reset(global_batch_dict)                 # every rank passes the same dict (or pre-sharded=True)
        while True:
            action = policy(env.get_policy_ob(ob), ...)   # local live items only
            ob, reward, all_done, info = env.step(action) # reward: [B_global, 1] on every rank
            if all_done: break

    Per step the ranks exchange exactly ONE small collective (StepExchange: rewards, done flags and "rank finished" in
    one pre-allocated block, off the compute stream).  `step` returns that step's global view (the reference's contract:
    `all_done` is a Python bool); `step_async` returns the pending exchange instead, to be resolved after the next
    step's kernels have been launched (rollouts that do not branch on `all_done`, e.g. fixed-length episodes).  A rank
    whose items have all stopped keeps taking part in the collective with zero rewards until the last rank finishes.
    """

    def __init__(self, env, group=None):
        self.env = env
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_global = self.n_local = 0
        self.lo = self.hi = 0
        self._local_done = False
        self.exchange = None
        # device of the collective's buffers: fixed at construction so that a rank WITHOUT local items still joins the
        # gather with tensors the backend accepts (RCCL wants this rank's GPU; gloo takes the CPU)
        backend = dist.get_backend(group) if dist.is_initialized() else 'gloo'
        env_dev = getattr(env, 'device', None)
        if env_dev is not None and torch.device(env_dev).type == 'cuda':
            self.coll_device = torch.device(env_dev)
        elif backend == 'nccl':
            self.coll_device = torch.device('cuda', torch.cuda.current_device())
        else:
            self.coll_device = torch.device('cpu')

    def __getattr__(self, name):          # get_policy_ob, get_images, solver, max_episode_step, ...
        return getattr(self.env, name)

    def reset(self, data, pre_sharded=False, n_global=None):
        if pre_sharded:
            if n_global is None:
                raise ValueError('pre_sharded batches need n_global')
            self.n_global = int(n_global)
            local = dict(data)
        else:
            self.n_global = len(next(iter(data.values())))
            local = shard_batch(data, self.world, self.rank)
        self.lo, self.hi = shard_bounds(self.n_global, self.world, self.rank)
        self.n_local = self.hi - self.lo
        self._local_done = self.n_local == 0
        ob = self.env.reset(local) if self.n_local else None
        device = self.env.state['gt'].device if self.n_local else self.coll_device
        ex = self.exchange
        if ex is None or ex.n_global != self.n_global or ex.device != torch.device(device):
            self.exchange = StepExchange(self.n_global, device, self.group)
        self._done_full = torch.ones(self.n_local, dtype=torch.float32, device=device)
        self._zero_reward = torch.zeros(self.n_local, 1, dtype=torch.float32, device=device)
        return ob

    def step_async(self, action):
        """-> (local observation of the still-live items, pending exchange).  `pending.result()` gives
        (global reward [B_global,1], global done flags, all ranks finished?)."""
        ob = None
        self._done_full.fill_(1.0)
        if not self._local_done:
            live_before = self.env.idx_left.clone()
            _, ob, reward, local_all_done, info = self.env.step(action)
            self._done_full[live_before] = info['done'].to(torch.float32)
            self._local_done = bool(local_all_done)
        else:
            reward = self._zero_reward
        return ob, self.exchange.post(reward, self._done_full, self._local_done)

    def step(self, action):
        """-> (local observation of the still-live items, global reward [B_global,1], all ranks done?, info)"""
        ob, pending = self.step_async(action)
        rewards, done, finished = pending.result()
        return ob, rewards, finished, {'done': done, 'local_done': self._local_done}
