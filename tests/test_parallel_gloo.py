"""world_size-2 tests of the data-parallel path on CPU (gloo): gradient averaging with the overflow flag in the tail,
frame sharding, per-rank seeds, parameter broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from autolabel_amd import parallel
    r, w, _ = parallel.init_distributed('gloo')
    assert (r, w) == (rank, world)
    n = 1000
    grad = torch.zeros(n + 8)
    grad[:n] = torch.arange(n, dtype=torch.float32) * (rank + 1)
    flag = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
    parallel.allreduce_gradients(grad, n, flag)
    want = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1)) / world
    ok_avg = torch.allclose(grad[:n], want) and grad[n].item() == 0
    ok_flag = flag.item() == 1  # OR over ranks: every rank skips the same step
    flag2 = torch.zeros(1, dtype=torch.int32)
    g2 = torch.ones(n + 8) * (rank + 1)
    parallel.allreduce_gradients(g2, n, flag2)
    ok_noflag = flag2.item() == 0 and torch.allclose(g2[:n], torch.full((n,), (world + 1) / 2))
    flat = torch.full((10,), float(rank + 5))
    parallel.broadcast_parameters(flat)
    ok_bcast = bool((flat == 5.0).all())
    lo, hi = parallel.frame_shard(11, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, parallel.rank_seed(99, rank)))
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = (ok_avg, ok_flag, ok_noflag, ok_bcast, gathered)


def test_two_rank_gradient_allreduce_and_sharding():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        res = dict(ret)
    for rank in range(world):
        ok_avg, ok_flag, ok_noflag, ok_bcast, gathered = res[rank]
        assert ok_avg and ok_flag and ok_noflag and ok_bcast, (rank, res[rank][:4])
        shards = [(a, b) for a, b, _ in gathered]
        assert shards == [(0, 6), (6, 11)]
        assert len({s for _, _, s in gathered}) == world  # distinct per-rank seeds


def test_frame_shards_partition_all_frames():
    from autolabel_amd.parallel import frame_shard
    for n in [1, 7, 8, 200, 201]:
        for world in [1, 2, 3, 8]:
            covered = []
            for r in range(world):
                lo, hi = frame_shard(n, r, world)
                covered += list(range(lo, hi))
            assert covered == list(range(n))


def test_ray_shards_partition_a_frame():
    from autolabel_amd.parallel import ray_shard
    for n in [1, 307200, 76801]:
        for world in [1, 2, 3, 8]:
            covered = []
            for r in range(world):
                lo, hi = ray_shard(n, r, world)
                assert hi - lo in (n // world, n // world + 1)
                covered.append((lo, hi))
            assert covered[0][0] == 0 and covered[-1][1] == n and all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))


def _shard_worker(rank, world, port, ret):
    """Sharded optimizer on CPU tensors: reduce-scatter of two gradient buckets (one even, one ragged), the oracle's Adam on the
    owned slices with COMPACT moments, all-gather of the parameters -- against all-reduce + Adam on the whole table."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from autolabel_amd import parallel
    from oracle.nerf_oracle import adam_update
    parallel.init_distributed('gloo')
    buckets = [(0, 4096), (4096, 4096 + 1000)]          # the second one does not split into world x 8-element shards
    n_grid, n_mlp = buckets[-1][1], 64
    g = torch.Generator().manual_seed(100 + rank)
    p0 = torch.randn(n_grid + n_mlp, generator=torch.Generator().manual_seed(7))      # same parameters on every rank
    steps = [torch.randn(n_grid + n_mlp + 8, generator=g) for _ in range(3)]          # different gradients per rank
    # ---- replicated: all-reduce, Adam everywhere
    pa, ma, va = p0.clone(), torch.zeros(n_grid + n_mlp), torch.zeros(n_grid + n_mlp)
    for t, gr in enumerate(steps, 1):
        gr = gr.clone()
        parallel.allreduce_gradients(gr, n_grid + n_mlp, torch.zeros(1, dtype=torch.int32))
        adam_update(pa, gr[:n_grid + n_mlp], ma, va, t, 5e-3)
    # ---- sharded: reduce-scatter, Adam on the owned slices (compact moments), all-gather
    own = [parallel.shard_range(a, b, rank, world)[:2] for a, b in buckets]
    n_own = sum(hi - lo for lo, hi in own)
    pb, mb, vb = p0.clone(), torch.zeros(n_own + n_mlp), torch.zeros(n_own + n_mlp)
    zero_outside = True
    for t, gr in enumerate(steps, 1):
        gr = gr.clone()
        parallel.allreduce_bucket(gr, n_grid, n_grid + n_mlp, found_inf=torch.zeros(1, dtype=torch.int32), tail=n_grid + n_mlp)
        at = 0
        for (a, b), (lo, hi) in zip(buckets, own):
            got = parallel.reduce_scatter_bucket(gr, a, b)
            assert got == (lo, hi)
            zero_outside &= bool((gr[a:lo] == 0).all() and (gr[hi:b] == 0).all())
            adam_update(pb[lo:hi], gr[lo:hi], mb[at:at + hi - lo], vb[at:at + hi - lo], t, 5e-3)
            at += hi - lo
        adam_update(pb[n_grid:], gr[n_grid:n_grid + n_mlp], mb[at:], vb[at:], t, 5e-3)
        for a, b in buckets:
            parallel.allgather_bucket(pb, a, b)
    # the moments back in the replicated layout (what TrainEngine.state_dict does)
    mfull = torch.zeros(n_grid + n_mlp)
    at = 0
    for lo, hi in own:
        mfull[lo:hi] = mb[at:at + hi - lo]; at += hi - lo
    mfull[n_grid:] = mb[at:]
    for a, b in buckets:
        parallel.allgather_bucket(mfull, a, b)
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = dict(params_equal=bool(torch.equal(pa, pb)), moments_equal=bool(torch.equal(ma, mfull)), zero_outside=zero_outside, own=own,
                     moved=float((pa - p0).abs().max()))


def test_sharded_optimizer_is_bit_identical_to_the_replicated_one_on_two_ranks():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_shard_worker, args=(world, port, ret), nprocs=world, join=True)
        res = dict(ret)
    assert res[0]['own'] == [(0, 2048), (4096, 4096 + 504)] and res[1]['own'] == [(2048, 4096), (4096 + 504, 4096 + 1000)]
    for rank in range(world):
        r = res[rank]
        assert r['moved'] > 1e-3
        assert r['params_equal'] and r['moments_equal'] and r['zero_outside'], (rank, r)


def test_shard_ranges_partition_a_bucket():
    from autolabel_amd.parallel import shard_range
    for a, b in [(0, 4096), (4096, 5096), (8, 16), (0, 7)]:
        for world in [1, 2, 3, 8]:
            parts = [shard_range(a, b, r, world) for r in range(world)]
            S = parts[0][2]
            assert S % 8 == 0 and world * S >= b - a and all(p[2] == S for p in parts)
            assert parts[0][0] == a and parts[-1][1] == b
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert all(lo % 8 == a % 8 or lo == b for lo, _, _ in parts)
