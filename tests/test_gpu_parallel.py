"""Data-parallel training step on the GPU with two ranks sharing one device (gloo carries the collectives: functional
check of the bucketed, stream-overlapped gradient all-reduce -- RCCL itself needs one GPU per rank and is exercised by
`bench.py --gpus N`)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _block_checksums(layout, *bufs):
    """(names, device int64 scalars): the wrapped sum of the bit patterns of every hash-grid level and every MLP block of each buffer
    (no host sync: the values are read after the last step)."""
    g, F = layout.enc.grid, 2
    nl = int(g.n_levels)
    spans = [(f'level{l:02d}', int(g.offset[l]) * F, int(g.offset[l + 1]) * F if l + 1 < nl else layout.n_grid) for l in range(nl)]
    spans += [(k, layout.offsets[k], layout.offsets[k] + layout.nets[k].n_params) for k in layout.nets]
    names, sums = [], []
    for which, buf in zip(('p', 'm', 'v'), bufs):
        bits = buf.view(torch.int32)
        for name, a, b in spans:
            names.append(f'{which}/{name}')
            sums.append(bits[a:b].to(torch.int64).sum())
    return names, sums


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo')
    torch.cuda.set_device(0)
    from autolabel_amd import parallel, synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    dev = torch.device('cuda', 0)
    scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
    frames = DeviceFrames.from_scene(scene, dev)
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=3.0)
    out = {}
    for overlap in (True, False):
        P = Params(layout, dev); P.init_(seed=0)
        parallel.broadcast_parameters(P.flat)
        P.refresh_shadows()
        eng = TrainEngine(HipPipeline(layout, P), process_group=dist.group.WORLD, num_steps=32, upsample_steps=32)
        eng.overlap_comm = overlap
        buckets = []
        if overlap:
            inner = eng._bucket_ready
            eng._bucket_ready = lambda kind, a, b: (buckets.append((kind, a, b)), inner(kind, a, b))[1]
        lo, hi = parallel.frame_shard(8, rank, world)
        batch = frames.alloc_batch(1024)
        trail = []     # per step: a bit-exact checksum of every parameter block (and of the Adam moments), so that a divergence between the
        for i in range(3):   # two exchange modes is reported with the step and the block it started in
            frames.next_train(batch, seed=parallel.rank_seed(5, rank), step=i, frame_range=(lo, hi))
            eng.step(batch, seed=parallel.rank_seed(7, rank), step=i)
            trail.append(_block_checksums(layout, P.flat, eng.m, eng.v))
        torch.cuda.synchronize()
        flat = P.flat.detach().cpu()
        both = [None] * world
        dist.all_gather_object(both, flat)
        if overlap:   # sharded rendering of one frame: every rank ends up with the full, identical image
            from autolabel_amd.models import ALNetwork
            from autolabel_amd.parallel import render_sharded
            torch.manual_seed(0)
            net = ALNetwork(encoding='hg+freq', num_layers=2, hidden_dim=128, geo_feat_dim=15, num_layers_color=2, hidden_dim_color=128,
                            hidden_dim_semantic=64, semantic_classes=scene['n_classes'], bound=3.0, cuda_ray=False, density_scale=1).cuda()
            t = frames.get_test(1)
            whole = net.render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False, num_steps=32, upsample_steps=16)
            shard = render_sharded(net, t['rays_o'], t['rays_d'], t['direction_norms'], perturb=False, num_steps=32, upsample_steps=16)
            out['render'] = {k: float((whole[k].float() - shard[k].float()).abs().max()) for k in whole}
            out['render_shape'] = tuple(shard['image'].shape) == tuple(whole['image'].shape)
        out[overlap] = dict(same=bool(torch.equal(both[0], both[1])), finite=bool(torch.isfinite(flat).all()), flat=flat,
                            trail=[{k: int(v) for k, v in zip(t[0], torch.stack(t[1]).cpu().tolist())} for t in trail],
                            steps=int(eng.state_i[0].item()), buckets=buckets, n_total=layout.n_total, n_grid=layout.n_grid)
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = out


def _lockstep_run():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        return dict(ret)


def _lockstep_divergence(res, world=2):
    """None when the overlapped and the single-collective exchange moved the same numbers on every rank; otherwise a description."""
    for rank in range(world):
        f1, f0 = res[rank][True]['flat'], res[rank][False]['flat']
        t1, t0 = res[rank][True]['trail'], res[rank][False]['trail']
        first = next(((i, sorted(k for k in a if a[k] != b[k])) for i, (a, b) in enumerate(zip(t1, t0)) if a != b), None)
        if not torch.equal(f1, f0) or first is not None:
            return dict(rank=rank, parameters_differing=int((f1 != f0).sum().item()), first_step=first[0] if first else None,
                        first_blocks=first[1] if first else None)
    return None


def test_two_ranks_on_one_gpu_stay_in_lockstep_with_bucketed_allreduce():
    res = _lockstep_run()
    world = 2
    for rank in range(world):
        # sharded render of a frame == the single-process render, bit for bit
        assert res[rank]['render_shape'] and max(res[rank]['render'].values()) == 0.0, res[rank]['render']
        for overlap in (True, False):
            r = res[rank][overlap]
            assert r['same'] and r['finite'] and r['steps'] == 3, (rank, overlap, r['same'], r['finite'], r['steps'])
        # every step: the MLP bucket first, then the level groups; together they tile [0, n_total) exactly once
        b = res[rank][True]['buckets']
        per_step = len(b) // 3
        step0 = b[:per_step]
        assert step0[0][0] == 'mlp' and all(k == 'grid' for k, _, _ in step0[1:])
        spans = sorted((a, e) for _, a, e in step0)
        assert spans[0][0] == 0 and spans[-1][1] == res[rank][True]['n_total']
        assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        # the small coarse levels go last (the only all-reduce nothing overlaps)
        assert step0[-1][1] == 0
    # overlapped buckets and the single exchange after the backward pass move the same numbers: every kernel of the step is
    # order-independent and the level groups reproduce the single scatter launch bit for bit.
    # This comparison failed in GPUTEST_r04 (3.4 M parameters apart after three steps, both ranks still in agreement) and then about once in
    # twenty runs.  Round 5 found the cause outside the exchange: packed fp32 instructions of the scatter returning +0 in lanes 48..63
    # while the OTHER rank's MFMA kernels ran on the same SIMD -- two processes on one GPU (scripts/dev/probe_pk_f32.hip).
    # The library is now built without them (tests/test_build_hygiene.py).  STRICT since round 6: the first divergence fails the test
    # (round 5's record-and-repeat would have hidden a regression of exactly that class); the step and parameter block it started in
    # are still written to gpurun_out/lockstep_divergence.json for whoever has to chase it.
    div = _lockstep_divergence(res)
    if div is not None:
        import json
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(root, 'gpurun_out', 'lockstep_divergence.json'), 'w') as f:
            json.dump(div, f)
    assert div is None, f'overlapped and single-collective steps diverged: {div}'


def _shard_worker(rank, world, port, ret):
    """TrainEngine(shard_optimizer=True) against the replicated optimizer: same batches, same seeds, 4 steps (one of them skipped
    by a forced overflow), overlapped and single-stream exchange."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo')
    torch.cuda.set_device(0)
    from autolabel_amd import parallel, synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    dev = torch.device('cuda', 0)
    scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
    frames = DeviceFrames.from_scene(scene, dev)
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=3.0)
    out = {}
    for shard in (False, True):
        for overlap in (True, False):
            P = Params(layout, dev); P.init_(seed=0)
            eng = TrainEngine(HipPipeline(layout, P), process_group=dist.group.WORLD, num_steps=32, upsample_steps=32, overlap_comm=overlap,
                              shard_optimizer=shard)
            lo, hi = parallel.frame_shard(8, rank, world)
            batch = frames.alloc_batch(1024)
            for i in range(4):
                frames.next_train(batch, seed=parallel.rank_seed(5, rank), step=i, frame_range=(lo, hi))
                inner = eng.pipe._k
                if i == 2 and rank == 1:       # a non-finite table gradient on ONE rank: BOTH ranks must skip the step
                    if shard:                  # ... as a VALUE in rank 0's shard (the sharded scatter adds into P.grad; the flag is raised on the owner)
                        P.grad[5] = float('inf')
                    else:                      # ... upstream of the scatter (replicated, fp16 on the wire: the scatter writes the payload itself and no
                        def poisoned(name, *a, **kw):    # fp32 table gradient exists to plant a value in): one infinite element of d_enc
                            if name.startswith('aln_encode_bwd_binned'):
                                eng.ws.bufs['d_enc'][1][777, 20] = float('inf')
                            return inner(name, *a, **kw)
                        eng.pipe._k = poisoned
                eng.step(batch, seed=parallel.rank_seed(7, rank), step=i)
                eng.pipe._k = inner
            sd = eng.state_dict()              # (a collective under the sharded optimizer)
            eng.sync_master()
            torch.cuda.synchronize()
            out[(shard, overlap)] = dict(flat=P.flat.cpu(), table=P.table16.cpu(), m=sd['m'].cpu(), v=sd['v'].cpu(), steps=int(eng.state_i[0].item()),
                                         scale=float(eng.state_f[0].item()), grad_clean=bool((P.grad[:layout.n_total] == 0).all().item()),
                                         m_elems=eng.m.numel(), n_total=layout.n_total, n_grid=layout.n_grid)
            if shard and overlap:   # a checkpoint written by the sharded engine restores into a replicated one and vice versa
                eng2 = TrainEngine(HipPipeline(layout, P), process_group=dist.group.WORLD, num_steps=32, upsample_steps=32, shard_optimizer=False)
                eng2.load_state_dict(sd)
                eng.load_state_dict(eng2.state_dict())
                out['roundtrip'] = bool(torch.equal(eng2.m.cpu(), sd['m'].cpu()) and torch.equal(eng.state_dict()['v'].cpu(), sd['v'].cpu()))
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = out


def test_sharded_table_optimizer_matches_the_replicated_one_bit_for_bit():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_shard_worker, args=(world, port, ret), nprocs=world, join=True)
        res = dict(ret)
    ref = res[0][(False, True)]
    assert ref['steps'] == 3 and ref['scale'] == 32768.0, 'the overflow injected on rank 1 must skip the step (and back the scale off) on both ranks'
    for rank in range(world):
        assert res[rank]['roundtrip']
        for key in [(False, True), (False, False), (True, True), (True, False)]:
            r = res[rank][key]
            assert r['steps'] == 3 and r['grad_clean'], (rank, key, r['steps'], r['grad_clean'])
            for name in ('flat', 'table', 'm', 'v'):
                assert torch.equal(r[name], ref[name]), f'rank {rank} {key}: {name} differs in {(r[name] != ref[name]).sum().item()} elements'
        sharded, full = res[rank][(True, True)], res[rank][(False, True)]
        # the optimizer state of the table is halved
        assert sharded['m_elems'] == full['n_grid'] // 2 + full['n_total'] - full['n_grid'] and full['m_elems'] == full['n_total']


def _trainer_worker(rank, world, port, ret):
    """scripts/train.py's data-parallel wiring driven through SimpleTrainer: DeviceLoader over this rank's frame shard,
    rank seeds, broadcast of rank 0's initialisation, process_group handed to the trainer."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    from argparse import Namespace
    from autolabel_amd import parallel, synthetic
    r, w, local = parallel.init_distributed('gloo')
    assert (r, w) == (rank, world)
    torch.cuda.set_device(0)
    from autolabel_amd.dataset import DeviceFrames, DeviceLoader
    from autolabel_amd.models import ALNetwork
    from autolabel_amd.trainer import SimpleTrainer
    scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device='cuda')
    opt = Namespace(rand_pose=-1, color_space='srgb', feature_loss=False, rgb_weight=1.0, depth_weight=0.1, semantic_weight=1.0,
                    feature_weight=0.5, num_steps=32, upsample_steps=32)
    optimizer = lambda m: torch.optim.Adam([{'name': 'encoding', 'params': list(m.encoder.parameters())},
                                            {'name': 'net', 'params': m.network_parameters(), 'weight_decay': 1e-6}],
                                           lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    out = {}
    import tempfile
    for shard, ema in [(False, None), (False, 0.95), (True, 0.95)]:
        torch.manual_seed(rank)   # different initialisations on purpose: the broadcast must make them equal
        model = ALNetwork(encoding='hg+freq', num_layers=2, hidden_dim=128, geo_feat_dim=15, num_layers_color=2, hidden_dim_color=128,
                          hidden_dim_semantic=64, semantic_classes=scene['n_classes'], bound=6.0, cuda_ray=False, density_scale=1)
        model.reset_parameters(seed=rank)
        wsdir = tempfile.mkdtemp() if shard else None
        tr = SimpleTrainer('ngp', opt, model, device='cuda:0', workspace=wsdir, optimizer=optimizer,
                           criterion=torch.nn.MSELoss(reduction='none'), fp16=True, ema_decay=ema,
                           lr_scheduler=lambda o: torch.optim.lr_scheduler.StepLR(o, gamma=0.5, step_size=10), metrics=[],
                           use_checkpoint='scratch', local_rank=rank, world_size=world, process_group=dist.group.WORLD, mute=True,
                           shard_optimizer=shard)
        parallel.broadcast_parameters(model._ensure_device().P.flat, dist.group.WORLD)
        model._shadow_version = None
        loader = DeviceLoader(DeviceFrames.from_scene(scene, 'cuda'), 1024, 1000, seed=parallel.rank_seed(0, rank),
                              frame_range=parallel.frame_shard(8, rank, world))
        tr.train_iterations(loader, 4)
        torch.cuda.synchronize()
        flat = model._P.flat.detach().cpu()
        both = [None] * world
        dist.all_gather_object(both, flat)
        r = dict(same=bool(torch.equal(both[0], both[1])), finite=bool(torch.isfinite(flat).all()), fused=tr.fused, flat=flat,
                 steps=int(tr.engine.state_i[0].item()), world=tr.engine.world, loss=float(tr.engine.terms[4]),
                 sharded=tr.engine.shard is not None, gather=tr.engine.shard_gather,
                 ema=[p.detach().cpu() for p in tr.ema.shadow] if tr.ema is not None else None)
        if shard:   # every rank calls, rank 0 writes; the file holds the full moments
            tr.save_checkpoint('dp')
            dist.barrier()
            path = f'{tr.ckpt_path}/dp.pth'
            r['ckpt_written_by_rank0_only'] = os.path.exists(path) == (rank == 0)
            if rank == 0:
                ck = torch.load(path, map_location='cpu', weights_only=False)
                r['ckpt_m_elems'] = ck['engine']['m'].numel()
        out[(shard, ema)] = r
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_simple_trainer_data_parallel_on_two_ranks():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_trainer_worker, args=(world, port, ret), nprocs=world, join=True)
        res = dict(ret)
    for rank in range(world):
        for key, r in res[rank].items():
            assert r['fused'] and r['world'] == 2 and r['steps'] == 4, (key, r['fused'], r['world'], r['steps'])
            assert r['same'] and r['finite'], 'replicas must stay bit-identical after the averaged-gradient steps'
        rep, shd = res[rank][(False, 0.95)], res[rank][(True, 0.95)]
        # the sharded optimizer under the trainer (EMA on: the fp32 masters are gathered every step) moves the same numbers
        assert shd['sharded'] and shd['gather'] == 'master' and not rep['sharded']
        assert torch.equal(rep['flat'], shd['flat'])
        assert all(torch.equal(a, b) for a, b in zip(rep['ema'], shd['ema']))
        assert shd['ckpt_written_by_rank0_only']
    assert res[0][(True, 0.95)]['ckpt_m_elems'] == res[0][(True, 0.95)]['flat'].numel()


def _equiv_worker(rank, world, port, ret):
    """ONE process on a batch of P x B rays == P ranks on B rays each, gradients averaged (fp32 on the wire: to
    1e-5 of the gradient norm; fp16 on the wire: to the fp16 rounding of the payload)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo')
    torch.cuda.set_device(0)
    from autolabel_amd import parallel
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    dev = torch.device('cuda', 0)
    B, S = 1024, 32
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, 5, bound=2.0)
    g = torch.Generator().manual_seed(3)      # the same 2 B rays on every rank
    N = world * B
    full = {'rays_o': (torch.rand(N, 3, generator=g) - 0.5) * 1.5, 'rays_d': torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1),
            'direction_norms': 1.0 + 0.2 * torch.rand(N, 1, generator=g), 'pixels': torch.rand(N, 3, generator=g),
            'depth': 0.5 + 2.0 * torch.rand(N, generator=g),                      # every ray has depth and a label: the per-rank
            'semantic': torch.randint(0, 5, (N,), generator=g).int(),            # means then average to the global mean exactly
            'features': torch.randn(N, 64, generator=g)}
    noise, u = torch.rand(N, S, generator=g), torch.rand(N, S, generator=g)
    full = {k: v.to(dev).contiguous() for k, v in full.items()}
    noise, u = noise.to(dev), u.to(dev)

    def grads(eng, batch, nz, uu):
        eng.P.grad.zero_()
        eng.state_i[2:4] = 0
        eng.forward_backward(batch, seed=1, step=0, noise=nz.contiguous(), u=uu.contiguous())
        eng.all_reduce_grads()
        torch.cuda.synchronize()
        return eng.averaged_gradient(), int(eng.state_i[2].item())    # (fp16 on the wire: the table's part is the exchange's payload itself)

    def engine(pg, payload='f16', overlap=True):
        P = Params(layout, dev); P.init_(seed=0)
        with torch.no_grad():
            P.flat[:layout.n_grid].mul_(3e3)       # a table that shapes the density (the initialisation is +-1e-4)
        P.refresh_shadows()
        return TrainEngine(HipPipeline(layout, P), process_group=pg, num_steps=S, upsample_steps=S, feature_loss=True,
                           grad_payload=payload, overlap_comm=overlap)

    ref, flag = grads(engine(None), full, noise, u)              # single process, 2 B rays
    assert flag == 0
    lo, hi = rank * B, (rank + 1) * B
    mine = {k: v[lo:hi].contiguous() for k, v in full.items()}
    out = {}
    for payload, overlap in (('f32', True), ('f32', False), ('f16', True), ('f16', False)):
        got, flag = grads(engine(dist.group.WORLD, payload, overlap), mine, noise[lo:hi], u[lo:hi])
        out[(payload, overlap)] = (float((got - ref).norm() / ref.norm()), float((got[layout.n_grid:] - ref[layout.n_grid:]).norm() /
                                   ref[layout.n_grid:].norm()), flag, parallel.wire_bytes(layout.n_grid, layout.n_total, payload))
    # an fp16 overflow on the wire must skip the step on EVERY rank: rank 1 plants a huge grid gradient
    eng = engine(dist.group.WORLD, 'f16', False)
    eng.P.grad.zero_(); eng.state_i[2:4] = 0
    eng.forward_backward(mine, seed=1, step=0, noise=noise[lo:hi].contiguous(), u=u[lo:hi].contiguous())
    if rank == 1:
        if eng._wire_direct():     # (the scatter wrote the payload itself: the planted gradient as the half it would have become)
            eng._wire_full()[100] = float('inf')
        else:
            eng.P.grad[100] = 1e9
    eng.all_reduce_grads()
    torch.cuda.synchronize()
    out['overflow_flag'] = int(eng.state_i[2].item())
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = out


def test_two_ranks_on_half_batches_equal_one_process_on_the_full_batch():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_equiv_worker, args=(world, port, ret), nprocs=world, join=True)
        res = dict(ret)
    for rank in range(world):
        r = res[rank]
        for overlap in (True, False):
            rel, rel_mlp, flag, nbytes = r[('f32', overlap)]
            assert flag == 0 and rel <= 1e-5 and rel_mlp <= 1e-5, (rank, 'f32', overlap, rel, rel_mlp)
            rel, rel_mlp, flag, nbytes16 = r[('f16', overlap)]
            assert flag == 0 and rel <= 1e-3 and rel_mlp <= 1e-5, (rank, 'f16', overlap, rel, rel_mlp)
            assert nbytes16 < 0.51 * nbytes      # the hash-grid block crosses the wire as halves
        assert r['overflow_flag'] == 1


def test_bench_contract_with_two_ranks_sharing_the_device():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), with gloo carrying the
    collectives so that both ranks can share this box's single GPU: the barrier / max-over-ranks timing, the frame shards, the
    data-parallel step (launch by launch, overlapped buckets, fp16 payload) and the ONE JSON line of rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALN_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
           '--batch', '512', '--no-cpu-baseline', '--no-pmc', '--no-march', '--no-lseg', '--quality-steps', '0', '--render-frames', '0',
           '--event-steps', '0']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly one JSON line'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['warmup'] == 2 and d['scaling'] == 'weak'
    assert d['n_ranks_seen'] == 2 and d['value'] > 0 and d['higher_is_better'] is True
    assert abs(d['value'] - 2 * 512 * 4 / (d['ms_per_step'] * 4e-3)) <= 1e-6 * d['value'], 'value = rays of ALL ranks / max-over-ranks time'
    assert d['config']['gradient_exchange_bytes_per_rank_and_step'] > 0 and d['config']['parallelism'] == 'dp2'


def test_bench_watchdog_prints_a_line_when_a_leg_never_finishes():
    """A collective that never returns must not cost the JSON line: with a leg timeout far below a leg's duration the watchdog fires
    in the first leg, rank 0 prints ONE line naming the watchdog, and every rank ends (no leg finished: exit code 1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALN_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
           '--batch', '512', '--no-cpu-baseline', '--no-pmc', '--no-march', '--no-lseg', '--quality-steps', '0', '--render-frames', '0',
           '--event-steps', '0', '--dp-leg-timeout', '0.05']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert 'watchdog' in d and d['value'] is None and d['n_gpus'] == 2 and 'dp_simple' in d['dp_legs']
    assert r.returncode != 0


def test_the_training_step_is_bit_stable_next_to_a_second_training_process():
    """The condition that turned the lock-step test above red in round 4, without the ranks: ONE process repeats the same forward +
    backward from the same state while a SECOND process trains on the same GPU, and every gradient block and output has to match the
    first repetition bit for bit.  Before the library was built without packed fp32 instructions (autolabel_amd/build.py,
    tests/test_build_hygiene.py) 5 - 14 % of the repetitions differed in a few hash-grid levels: v_pk_mul_f32 results zeroed in lanes
    48..63 while the neighbour's v_mfma_f32_32x32x16_f16 was in flight on the same SIMD (scripts/dev/probe_pk_f32.hip)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'scripts', 'dev', 'stress_determinism.py')
    r = subprocess.run([sys.executable, tool, '--disturb', '--seconds', '60', '--iters', '200'], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GRAFT_REPO_ROOT=root))
    tail = '\n'.join(r.stdout.strip().splitlines()[-12:])
    assert 'disturb role: finished' in r.stdout, 'the competing process did not run to the end:\n' + tail + '\n' + r.stderr[-2000:]
    assert r.returncode == 0 and 'all gradients and outputs bit-identical over 200 runs' in r.stdout, tail
