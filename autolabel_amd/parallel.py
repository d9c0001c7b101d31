"""Data parallelism over the 8 GPUs of one node: one process per GPU, frames sharded across ranks, model replicated,
ONE exchange per step -- the all-reduce of the flat gradient buffer (hash grid 14.2 M + MLPs ~0.06 M parameters) over RCCL/xGMI;
the hash-grid block crosses the wire as fp16 (28.5 MB per rank and step instead of 57 MB).  The reference has no multi-GPU path; this is the single data-parallel axis of 8(e).

The GradScaler overflow flag rides in the tail of the same buffer so every rank skips the same steps.

Sharded optimizer (TrainEngine(shard_optimizer=True)): the all-reduce of every hash-grid bucket is split into its two halves --
reduce-scatter of the gradient, Adam on the 1 / world slice a rank owns (moments allocated for that slice only), all-gather of the
updated fp16 table -- same wire bytes, 1 / world of the optimizer's HBM traffic and state (a trainer with an EMA reads the fp32
masters every step and gathers THOSE instead, shard_gather='master': twice the all-gather bytes).
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


def _pack_tail(grad, tail, found_inf, counts):
    grad[tail] = found_inf[0].to(grad.dtype)
    if counts is not None:   # "this rank saw a labelled ray": torch DDP gives every rank a gradient for a parameter as soon
        grad[tail + 1] = (counts[1] > 0).to(grad.dtype)   # as one rank has one, so the optimizer's skip rule must be global


def _unpack_tail(grad, tail, found_inf, counts):
    found_inf[0] = (grad[tail] > 0).to(found_inf.dtype)
    if counts is not None:
        counts[1] = torch.maximum(counts[1], (grad[tail + 1] > 0).to(counts.dtype))
    grad[tail:tail + 2] = 0


def _avg_inplace(view, group, world):
    if dist.get_backend(group) == 'nccl':
        dist.all_reduce(view, op=dist.ReduceOp.AVG, group=group)
    else:  # gloo has no AVG
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
        view.mul_(1.0 / world)


def _avg_f16(grad, a, b, scratch, found_inf, group, world, prepacked=False):
    """grad[a:b] averaged over the ranks with fp16 on the wire: pack (x 1/world) -> SUM all-reduce of the halves -> unpack.
    `prepacked`: scratch[:b - a] already holds fp16(gradient x 1/world) -- the hash-grid scatter wrote the payload itself
    (aln_encode_bwd_binned_wire; grad[a:b] holds nothing yet) -- so the packing pass is skipped, and the averaged halves STAY in
    `scratch` for the optimizer (aln_adam_step_wire): they are only watched for a non-finite element, grad[a:b] is not written.
    A non-finite element after the reduction (the same on every rank) raises `found_inf`: the step is skipped like any other
    fp16 overflow.  Device tensors only: the two conversions are HIP kernels (csrc/adam.hip)."""
    from . import hip as H
    if not grad.is_cuda:
        raise RuntimeError("payload='f16' needs device tensors (the conversions are HIP kernels); use payload='f32' on the CPU")
    n = b - a
    assert scratch is not None and scratch.dtype == torch.float16 and scratch.numel() >= n, 'fp16 staging buffer too small'
    assert a % 4 == 0, 'bucket start must keep the fp32 side 16-byte aligned'
    wire = scratch[:n]
    if not prepacked:
        H.call('aln_grad_pack_f16', H.ptr(grad[a:b]), n, 1.0 / world, H.ptr(wire), H.stream())
    dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=group)
    if prepacked:
        assert found_inf is not None, 'the reduced halves must be watched: pass the flag word'
        H.call('aln_grad_unpack_f16', H.ptr(wire), n, None, H.ptr(found_inf), H.stream())
        return
    H.call('aln_grad_unpack_f16', H.ptr(wire), n, H.ptr(grad[a:b]), H.ptr(found_inf) if found_inf is not None else None, H.stream())


def wire_bytes(n_grid, n_total, payload='f16'):
    """Bytes one rank contributes to the gradient exchange of a step (before the ring's 2 (P - 1) / P factor)."""
    return n_grid * (2 if payload == 'f16' else 4) + (n_total - n_grid + 2) * 4


def allreduce_gradients(grad, n_total, found_inf, group=None, counts=None, n_grid=0, payload='f32', scratch=None, force=False, prepacked=False):
    """Average `grad[:n_total]` over the group in place; `found_inf` (int32[1]) becomes the logical OR over ranks, and so does
    "some rank had labelled rays" (`counts[1] > 0`, which decides whether the semantic heads take an optimizer step).  `grad`
    must have at least two spare elements at index n_total.  payload='f32': ONE collective over the flat buffer.
    payload='f16': the hash-grid block grad[:n_grid] crosses the wire as fp16 (28.5 MB instead of 57 MB at the default model),
    the small MLP block + flags stay fp32 (two collectives).  `force`: issue the collectives for a group of one rank too (they move
    nothing; TrainEngine(exchange_at_world_1=True) runs the RCCL code this way on a single GPU)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return
    _pack_tail(grad, n_total, found_inf, counts)
    if payload == 'f16' and n_grid > 0:
        _avg_inplace(grad[n_grid:n_total + 2], group, world)
        _unpack_tail(grad, n_total, found_inf, counts)
        _avg_f16(grad, 0, n_grid, scratch, found_inf, group, world, prepacked)
        return
    _avg_inplace(grad[:n_total + 2], group, world)
    _unpack_tail(grad, n_total, found_inf, counts)


def allreduce_bucket(grad, a, b, group=None, found_inf=None, tail=None, counts=None, payload='f32', scratch=None, flag=None, force=False,
                     prepacked=False):
    """Average `grad[a:b]` over the group in place (one collective on the current stream).  With `found_inf` the bucket
    must end at `tail` (= n_total): the flags travel in `grad[tail:tail + 2]` and come back as the OR over ranks.
    payload='f16' (hash-grid buckets): fp16 on the wire through `scratch`; `flag` (int32[1]) is raised on a non-finite result."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return
    if found_inf is not None:
        assert b == tail, 'the overflow flag rides right behind the bucket'
        _pack_tail(grad, tail, found_inf, counts)
        b = tail + 2
    elif payload == 'f16':
        _avg_f16(grad, a, b, scratch, flag, group, world, prepacked)
        return
    _avg_inplace(grad[a:b], group, world)
    if found_inf is not None:
        _unpack_tail(grad, tail, found_inf, counts)


def shard_range(a, b, rank, world, align=8):
    """The slice [lo, hi) of the bucket [a, b) rank `rank` owns under the sharded optimizer, and the common shard length S
    (a multiple of `align` elements -- 16 bytes of fp16; the collectives want `world` equal shards, so the last ranks of a
    ragged bucket own a short or empty slice and the staging buffers are padded to world * S)."""
    n = b - a
    S = -(-n // (world * align)) * align
    lo = min(a + rank * S, b)
    return lo, min(lo + S, b), S


def reduce_scatter_bucket(grad, a, b, group=None, payload='f32', scratch=None, flag=None):
    """Average `grad[a:b]` over the group, every rank keeping only the slice it owns (shard_range): grad[lo:hi] holds the
    average afterwards, the rest of grad[a:b] is ZERO (the scatter of the next step adds into it; nobody reads it before).
    Half the wire bytes of the all-reduce; the other half is the all-gather of the updated table (allgather_bucket).
    payload='f16' (device tensors): fp16 on the wire through `scratch` (>= 2 * world * S halves); `flag` (int32[1]) is raised on
    a non-finite element of the OWNED slice -- the caller reduces the flag over the ranks.  Returns (lo, hi)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi, S = shard_range(a, b, rank, world)
    n = b - a
    if payload == 'f16':
        from . import hip as H
        if not grad.is_cuda:
            raise RuntimeError("payload='f16' needs device tensors (the conversions are HIP kernels); use payload='f32' on the CPU")
        assert scratch is not None and scratch.dtype == torch.float16 and scratch.numel() >= (world + 1) * S, 'fp16 staging buffer too small'
        assert a % 4 == 0 and lo % 4 == 0, 'bucket and shard starts must keep the fp32 side 16-byte aligned'
        wire, mine = scratch[:world * S], scratch[world * S:(world + 1) * S]
        H.call('aln_grad_pack_f16_clear', H.ptr(grad[a:b]), n, world * S, 1.0 / world, H.ptr(wire), H.stream())
        dist.reduce_scatter_tensor(mine, wire, op=dist.ReduceOp.SUM, group=group)
        if hi > lo:
            H.call('aln_grad_unpack_f16', H.ptr(mine), hi - lo, H.ptr(grad[lo:hi]), H.ptr(flag) if flag is not None else None, H.stream())
        return lo, hi
    wire = torch.zeros(world * S, dtype=grad.dtype, device=grad.device)
    wire[:n] = grad[a:b]
    mine = torch.empty(S, dtype=grad.dtype, device=grad.device)
    dist.reduce_scatter_tensor(mine, wire, op=dist.ReduceOp.SUM, group=group)
    grad[a:b] = 0
    grad[lo:hi] = mine[:hi - lo] * (1.0 / world)
    return lo, hi


def allgather_bucket(buf, a, b, group=None, scratch=None):
    """Every rank contributes the slice of `buf[a:b]` it owns (shard_range) and ends up with the whole bucket: the updated
    fp16 table after a sharded optimizer step (or the fp32 masters / Adam moments for a checkpoint).  In place when the bucket
    splits evenly (the default model's level groups do); a ragged bucket goes through `scratch` (world * S elements)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi, S = shard_range(a, b, rank, world)
    if b - a == world * S:
        dist.all_gather_into_tensor(buf[a:b], buf[lo:hi], group=group)
        return
    if scratch is None or scratch.dtype != buf.dtype or scratch.numel() < (world + 1) * S:
        scratch = torch.empty((world + 1) * S, dtype=buf.dtype, device=buf.device)
    mine, full = scratch[world * S:(world + 1) * S], scratch[:world * S]
    mine[:hi - lo] = buf[lo:hi]
    dist.all_gather_into_tensor(full, mine, group=group)
    buf[a:b] = full[:b - a]


def broadcast_parameters(flat, group=None, src=0):
    """Replicate the master parameters of rank `src` (model replicas must start identical)."""
    if dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def ray_shard(n_rays, rank, world):
    """Contiguous slice [lo, hi) of a frame's rays for this rank (rendering shards over rays with no collective in the
    data path, SURVEY 8e: "rendering a single frame can split rows across GPUs")."""
    per, rem = divmod(n_rays, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


@torch.no_grad()
def render_sharded(model, rays_o, rays_d, direction_norms, group=None, **render_kwargs):
    """Every rank renders its slice of the rays with `model.render(staged=True)`; the slices are concatenated on every
    rank with one all_gather per output at the end (outputs only: nothing is exchanged while rendering).  Shapes follow
    `model.render`: the ray prefix of `rays_o` is restored."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    prefix = rays_o.shape[:-1]
    ro, rd, dn = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), direction_norms.reshape(-1)
    lo, hi = ray_shard(ro.shape[0], rank, world)
    out = model.render(ro[lo:hi].contiguous(), rd[lo:hi].contiguous(), dn[lo:hi].contiguous(), staged=True, **render_kwargs)
    if world > 1:
        sizes = [ray_shard(ro.shape[0], r, world) for r in range(world)]
        full = {}
        for k, v in out.items():
            parts = [torch.empty((b - a,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device) for a, b in sizes]
            dist.all_gather(parts, v.contiguous(), group=group)
            full[k] = torch.cat(parts, 0)
        out = full
    return {k: (v.reshape(*prefix, *v.shape[1:]) if v.dim() > 1 else v.reshape(*prefix)) for k, v in out.items()}
