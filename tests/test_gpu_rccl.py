"""RCCL itself, on the one GPU this box has: a process group of ONE rank on the `nccl` backend (= RCCL on ROCm)
drives every data-parallel form of the training step -- `TrainEngine(exchange_at_world_1=True)` issues all collectives although they
move nothing: communicator set-up, `ReduceOp.AVG`, the fp16 wire kernels around a SUM all-reduce, in-place `reduce_scatter_tensor` /
`all_gather_into_tensor` of the sharded optimizer, the communication stream's event ordering, and hipGraph capture WITH the
collectives inside.  With one rank the average is the identity, so every form must reproduce the plain single-GPU step: bit for bit
with fp32 on the wire, and bit for bit among themselves with fp16 on the wire (the payload's rounding is the only difference).

The reference has no multi-GPU path (scripts/train.py:83 hard-codes cuda:0); SURVEY.md 8(e) is this build's own axis.  What this
does NOT show is a scaling curve: no second GPU is involved."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

STEPS = 4


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=dev)
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
    frames = DeviceFrames.from_scene(scene, dev)
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=3.0)
    out = {'backend': dist.get_backend(), 'world': dist.get_world_size()}

    def run(name, pg, payload='f16', overlap=True, shard=False, graph=False, fuse=False):
        P = Params(layout, dev); P.init_(seed=0)
        eng = TrainEngine(HipPipeline(layout, P), process_group=pg, num_steps=32, upsample_steps=32, overlap_comm=overlap, grad_payload=payload,
                          shard_optimizer=shard, fuse_grid_adam=fuse, exchange_at_world_1=pg is not None)
        batch = frames.alloc_batch(1024)
        try:
            if graph:    # the step captured WITH its collectives (torch's RCCL process group is capturable)
                g = eng.graphed(frames, batch, data_seed=5, seed=7, first_step=0, warmup=1)
                for _ in range(STEPS - g.steps):
                    g()
            else:
                for i in range(STEPS):
                    frames.next_train(batch, seed=5, step=i)
                    eng.step(batch, seed=7, step=i)
            sd = eng.state_dict()        # (a collective under the sharded optimizer)
            eng.sync_master()
            torch.cuda.synchronize()
            out[name] = dict(flat=P.flat.cpu(), table=P.table16.cpu(), m=sd['m'].cpu(), v=sd['v'].cpu(), steps=int(eng.state_i[0].item()),
                             dp=eng.dp, sharded=eng.shard is not None, fused=eng.fuse_grid_adam, finite=bool(torch.isfinite(P.flat).all()),
                             loss=float(eng.terms[4]))
        except Exception as e:   # reported, compared by the parent
            out[name] = dict(error=f'{type(e).__name__}: {e}'[:400])

    W = dist.group.WORLD
    run('single', None)                                   # gradient through P.grad, separate optimizer pass, no process group
    run('single_fused', None, fuse=True)                  # the benchmarked single-GPU step (table optimizer inside the scatter)
    run('simple_f32', W, 'f32', overlap=False)            # ONE AVG all-reduce of the flat buffer
    run('overlap_f32', W, 'f32', overlap=True)            # five buckets on the communication stream behind the scatter's level groups
    run('simple_f16', W, 'f16', overlap=False)
    run('overlap_f16', W, 'f16', overlap=True)
    run('sharded_f16', W, 'f16', overlap=True, shard=True)   # reduce_scatter_tensor + owned-slice Adam + all_gather_into_tensor
    run('sharded_simple_f16', W, 'f16', overlap=False, shard=True)
    run('graph_f16', W, 'f16', overlap=True, graph=True)
    run('sharded_graph_f16', W, 'f16', overlap=True, shard=True, graph=True)
    dist.barrier()
    torch.cuda.synchronize()
    ret[0] = out            # (a synchronous call into the manager process: delivered before anything below)
    # Captured graphs that hold RCCL kernels and a destroyed communicator do not always unwind in a safe order at interpreter exit
    # (bench.py's dp_world1 child aborted once in three in-bench runs AFTER its legs had finished): nothing is left to verify here, so
    # the process ends without running destructors.
    import gc
    gc.collect()
    try:
        dist.destroy_process_group()
    finally:
        os._exit(0)


def test_rccl_world_of_one_runs_every_data_parallel_form_of_the_step():
    res, errors = None, []
    for attempt in range(2):     # RCCL failing to come up (once in ~20 runs on the gpurun boxes: a c10 DistBackendError out of the worker before any
        port = _free_port()      # step has run) is retried once with a fresh rendezvous; what the worker COMPUTED is never retried
        with mp.Manager() as mgr:
            ret = mgr.dict()
            try:
                mp.spawn(_worker, args=(port, ret), nprocs=1, join=True)
            except Exception as e:   # noqa: BLE001 -- ProcessRaisedException / ProcessExitedException of the spawned rank
                errors.append(f'{type(e).__name__}: {str(e)[:1500]}')
            got = dict(ret)
        if 0 in got:
            res = got[0]
            break
    if errors:   # counted and reported, never silent: gpurun_out/rccl_setup_retries.json says how often the set-up retry fired and why
        import json
        import warnings
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        rec = os.path.join(root, 'gpurun_out', 'rccl_setup_retries.json')
        prev = json.load(open(rec)) if os.path.exists(rec) else []
        json.dump(prev + [dict(attempts=len(errors) + (res is not None), delivered=res is not None, errors=errors)], open(rec, 'w'))
        warnings.warn('the RCCL worker failed before delivering results (set-up only; computed results are never retried): ' + ' | '.join(errors))
    assert res is not None, errors
    assert res['backend'] == 'nccl' and res['world'] == 1
    errors = {k: v['error'] for k, v in res.items() if isinstance(v, dict) and 'error' in v}
    assert not errors, errors
    for name, r in res.items():
        if isinstance(r, dict):
            assert r['steps'] == STEPS and r['finite'], (name, r['steps'], r['finite'])
            assert r['dp'] == (not name.startswith('single')) and r['sharded'] == name.startswith('sharded'), (name, r['dp'], r['sharded'])
    same = lambda a, b: all(torch.equal(res[a][k], res[b][k]) for k in ('flat', 'table', 'm', 'v'))
    diff = lambda a, b: {k: int((res[a][k] != res[b][k]).sum()) for k in ('flat', 'table', 'm', 'v')}
    # fp32 on the wire: the average over one rank is the identity -- the plain single-GPU step, bit for bit
    assert res['single_fused']['fused'] and same('single', 'single_fused'), diff('single', 'single_fused')
    for name in ('simple_f32', 'overlap_f32'):
        assert same(name, 'single'), (name, diff(name, 'single'))
    # fp16 on the wire: every form moves the same numbers (eager, bucketed on the side stream, sharded optimizer, captured)
    for name in ('overlap_f16', 'sharded_f16', 'sharded_simple_f16', 'graph_f16', 'sharded_graph_f16'):
        assert same(name, 'simple_f16'), (name, diff(name, 'simple_f16'))
    # ... and the payload's rounding is all that separates them from the fp32 exchange (Adam normalises the step: compare the losses)
    assert abs(res['simple_f16']['loss'] - res['single']['loss']) <= 2e-2 * abs(res['single']['loss'])
