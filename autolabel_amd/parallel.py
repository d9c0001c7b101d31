"""Data parallelism over the 8 GPUs of one node: one process per GPU, frames sharded across ranks, model replicated,
ONE collective per step -- the all-reduce of the flat gradient buffer (hash grid 14.2 M + MLPs ~0.06 M fp32 = 57 MB) over
RCCL/xGMI.  The reference has no multi-GPU path; this is the single data-parallel axis of 8(e).

The GradScaler overflow flag rides in the tail of the same buffer so every rank skips the same steps.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """torchrun-style environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world, local_rank)."""
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def frame_shard(n_frames, rank, world):
    """Contiguous frame range [lo, hi) of this rank; the shards partition range(n_frames)."""
    per, rem = divmod(n_frames, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def rank_seed(base, rank):
    """Distinct counter-RNG seed per rank (pixel picks, jitter, stratified noise)."""
    return (int(base) + 0x9E3779B1 * (rank + 1)) & 0x7FFFFFFF


def allreduce_gradients(grad, n_total, found_inf, group=None):
    """Average `grad[:n_total]` over the group in place with a single collective; `found_inf` (int32[1]) becomes the
    logical OR over ranks.  `grad` must have at least one spare element at index n_total."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    grad[n_total] = found_inf[0].to(grad.dtype)
    if dist.get_backend(group) == 'nccl':
        dist.all_reduce(grad, op=dist.ReduceOp.AVG, group=group)
    else:  # gloo has no AVG
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
        grad.mul_(1.0 / world)
    found_inf[0] = (grad[n_total] > 0).to(found_inf.dtype)
    grad[n_total] = 0


def broadcast_parameters(flat, group=None, src=0):
    """Replicate the master parameters of rank `src` (model replicas must start identical)."""
    if dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
