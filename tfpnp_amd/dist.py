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


class ShardedEnv:
    """A PnPEnv over this rank's contiguous shard of a global env batch -- what the reference does with
    DataParallelWithCallback(solver) (tasks/csmri/main.py:79-80), without moving images between GPUs.

        env = ShardedEnv(CSMRIEnv(None, solver, max_episode_step))
        ob = env.reset(global_batch_dict)                 # every rank passes the same dict (or pre-sharded=True)
        while True:
            action = policy(env.get_policy_ob(ob), ...)   # local live items only
            ob, reward, all_done, info = env.step(action) # reward: [B_global, 1] on every rank
            if all_done: break

    Per step the ranks exchange one all_gather of [B_local] rewards, one of [B_local] done flags and one 4-byte
    all-reduce ("is every rank finished?").  A rank whose items have all stopped keeps taking part in the collectives
    with zero rewards until the last rank finishes.
    """

    def __init__(self, env, group=None):
        self.env = env
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_global = self.n_local = 0
        self.lo = self.hi = 0
        self._local_done = False
        # device of the collectives' buffers: fixed at construction so that a rank WITHOUT local items still joins the
        # gathers with tensors the backend accepts (RCCL wants this rank's GPU; gloo takes the CPU)
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
        return self.env.reset(local) if self.n_local else None

    def step(self, action):
        """-> (local observation of the still-live items, global reward [B_global,1], all ranks done?, info)"""
        ref = self.env.state['gt'] if self.n_local else None
        device = ref.device if ref is not None else self.coll_device
        done_full = torch.ones(self.n_local, dtype=torch.float32, device=device)
        ob = None
        if not self._local_done:
            live_before = self.env.idx_left.clone()
            _, ob, reward, local_all_done, info = self.env.step(action)
            done_full[live_before] = info['done'].to(torch.float32)
            self._local_done = bool(local_all_done)
        else:
            reward = torch.zeros(self.n_local, 1, dtype=torch.float32, device=device)
        rewards = all_gather_rows(reward, self.n_global, self.group)
        done = all_gather_rows(done_full.view(-1, 1), self.n_global, self.group)
        finished = all_true(self._local_done, device)
        return ob, rewards, finished, {'done': done.view(-1) != 0, 'local_done': self._local_done}
