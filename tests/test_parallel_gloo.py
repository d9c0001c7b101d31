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
