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
