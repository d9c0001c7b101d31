"""Headline benchmark: NeRF training throughput (rays/s) on a synthetic 640x480 RGB-D scene, at stated held-out quality.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  `--gpus N` with N > 1 and no RANK in the environment re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (before any GPU call) and exits with the child's code; under torchrun
the flag must equal WORLD_SIZE.  Weak scaling by default (every rank trains on its own 4096-ray batches drawn from its shard
of the frames, gradients averaged by one RCCL all-reduce per step); `--global-batch B` fixes the global batch instead
(B / N rays per rank: strong scaling).

A "step" = device ray generation + render forward + loss + backward + Adam for one batch, frames resident in HBM.  On one
GPU the step is replayed from a hipGraph (engine.GraphedStep; `--no-graph` issues it launch by launch).  After the timed
region the same step runs launch by launch with HIP events around the timed kernels (roofline figures), then training
continues to `--quality-steps` total steps and the held-out PSNR / depth L1 / mIoU of that state are reported, so the
throughput travels with its quality.  Prints ONE JSON line on rank 0 (contract in the task description).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')   # see autolabel_amd/__init__.py: must be set before HIP initialises


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=200)
    p.add_argument('--warmup', type=int, default=50)
    p.add_argument('--batch', type=int, default=4096, help='rays per GPU per step')
    p.add_argument('--global-batch', type=int, default=0, help='fixed global batch (strong scaling); 0 = weak scaling')
    p.add_argument('--frames', type=int, default=200)
    p.add_argument('--feature-dim', type=int, default=64)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-graph', action='store_true')
    p.add_argument('--render-frames', type=int, default=6)
    p.add_argument('--event-steps', type=int, default=20, help='launch-by-launch steps with HIP events (roofline); 0 = skip')
    p.add_argument('--quality-steps', type=int, default=1500, help='total optimizer steps before the held-out metrics; 0 = skip')
    p.add_argument('--no-march', action='store_true', help='skip the occupancy-grid marching leg (second config)')
    p.add_argument('--march-samples', type=int, default=64)
    p.add_argument('--march-thresh', type=float, default=10.0)
    return p.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` typed by hand: start N ranks in child processes (never after the GPU has been touched)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def host_threads():
    """Cores this process may use (cgroup/affinity aware), capped: torch CPU ops on tiny tensors collapse when
    oversubscribed (256 threads on the GPU box ran 25x slower than 8)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count()
    return max(1, min(n, 32))


def build(args, device, rank, world):
    import torch
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.parallel import frame_shard, broadcast_parameters
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    from autolabel_amd.quality import split_heldout
    # S1 scene: 640x480, trained at factor 2 like scripts/train.py:15; every 20th frame is held out
    scene = synthetic.make_room_scene(n_frames=args.frames, seed=0, device=device, feat_dim=64, feat_hw=(60, 80))
    half = synthetic.subsample(scene, 2)
    train_ids, held = split_heldout(args.frames)
    pick = lambda sc, ids, sem: dict(sc, images=sc['images'][ids], depths=sc['depths'][ids], semantics=sc[sem][ids],
                                     features=sc['features'][ids] if sem == 'semantics' else None, T_CW=sc['T_CW'][ids])
    train = DeviceFrames.from_scene(pick(half, train_ids, 'semantics'), device)
    test = DeviceFrames.from_scene(pick(half, held, 'semantics_full'), device)
    full = DeviceFrames.from_scene(pick(scene, held[:max(args.render_frames, 1)], 'semantics_full'), device)
    lo, hi = scene['min_bounds'], scene['max_bounds']
    bound = float(((hi - lo) - (lo + hi) * 0.5).max())  # autolabel/model_utils.py:62-63
    layout = ModelLayout('hg+freq', 15, 128, 128, args.feature_dim, scene['n_classes'], bound=bound)
    P = Params(layout, device)
    P.init_(seed=0)
    if world > 1:
        broadcast_parameters(P.flat)
        P.refresh_shadows()
    pipe = HipPipeline(layout, P)
    pg = torch.distributed.group.WORLD if world > 1 else None
    eng = TrainEngine(pipe, feature_loss=True, process_group=pg)
    return scene, half, train, test, full, eng, frame_shard(len(train_ids), rank, world), bound


def cpu_baseline(half, train_ids, feature_dim, n_classes, bound):
    """the reference has no CPU path, so the CPU baseline is this repo's fp32 oracle ("port") on the host cores:
    (1) S1 -- the bench workload itself: rays drawn from the bench scene by the host mirror of `_next_train`, same model
    config, 128+128 samples, bounded to 256-ray batches (the full 4096-ray batch costs ~2 min per step); (2) S0 --
    BASELINE configs[0]: 32x32 cube, hash grid L=4, bounded to 2048 of its 8192 rays per step; (3) the reference-style host
    ray generation (`_next_train`, numpy) on the S1 scene, B=4096."""
    import numpy as np
    import torch
    from oracle import nerf_oracle as O
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import ArrayDataset
    torch.set_num_threads(host_threads())

    def train_steps(ds, cfg, B, steps, Cf):
        m = O.OracleModel(cfg, seed=0)
        st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in m.params.items()}
        g = torch.Generator().manual_seed(0)
        times = []
        for it in range(steps + 1):
            b = ds._next_train()
            t = lambda k, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(b[k])).to(dt)[:B]
            batch = {'pixels': t('pixels'), 'depth': t('depth'), 'semantic': t('semantic', torch.int64)}
            if Cf:
                batch['features'] = t('features')
            t0 = time.time()
            out = m.run(t('rays_o'), t('rays_d'), t('direction_norms'), 128, 128, perturb=True,
                        noise_coarse=torch.rand(B, 128, generator=g), u_fine=torch.rand(B, 128, generator=g))
            loss, _ = O.loss_fn(out, batch, feature_loss=bool(Cf))
            for p in m.params.values():
                p.grad = None
            loss.backward()
            with torch.no_grad():
                for k, p in m.params.items():
                    if p.grad is None:   # no labelled ray in the batch: torch skips the tensor (autolabel/trainer.py:61-62)
                        continue
                    O.adam_update(p, p.grad, st[k][0], st[k][1], it + 1, 5e-3, weight_decay=0.0 if k == 'grid' else 1e-6)
            if it > 0:
                times.append(time.time() - t0)
        return B / (sum(times) / len(times))

    cpu = lambda sc: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sc.items()}
    sub = cpu(half)
    ids = train_ids[:24]   # the host dataset keeps float images: a 24-frame subset of the bench scene is enough to draw rays from
    s1 = dict(sub, images=sub['images'][ids], depths=sub['depths'][ids], semantics=sub['semantics'][ids],
              features=sub['features'][ids], T_CW=sub['T_CW'][ids])
    ds1 = ArrayDataset(s1, batch_size=512)
    v1 = train_steps(ds1, O.ModelConfig(feature_dim=feature_dim, n_classes=n_classes, bound=bound), 256, 3, 64)
    ds1b = ArrayDataset(s1, batch_size=4096)
    ds1b._next_train()
    t0 = time.time()
    for _ in range(5):
        ds1b._next_train()
    raygen = 5 * 4096 / (time.time() - t0)
    cube = cpu(synthetic.make_cube_scene())
    lo, hi = cube['min_bounds'], cube['max_bounds']
    b0 = float(((hi - lo) - (lo + hi) * 0.5).max())
    ds0 = ArrayDataset(cube, batch_size=2048)
    v0 = train_steps(ds0, O.ModelConfig(encoding='hg+freq', feature_dim=64, n_classes=cube['n_classes'], bound=b0,
                                        grid=O.GridSpec(n_levels=4)), 2048, 2, 0)
    return {'value': v1, 'unit': 'rays/s', 'cores': host_threads(), 'kind': 'port',
            'sample': '3 timed oracle (fp32 PyTorch) train steps of 256 rays x (128+128) samples drawn from the bench scene (S1, '
                      'host _next_train), same model config; scale: the bench batch is 4096 rays = 16 such sub-batches',
            's0_value': v0, 's0_sample': '2 timed oracle train steps of 2048 rays (of configs[0]\'s 8192) on the 32x32 cube, hash grid L=4, 128+128 samples',
            'host_raygen_rays_per_s': raygen, 'host_raygen_sample': '5 batches of 4096 rays, numpy mirror of dataset._next_train on the S1 scene'}


def marching_leg(args, scene, train, test, bound, device, B, full=None):
    """Second configuration: the same scene, model and batch trained through occupancy-grid marching
    (cuda_ray=True: `--march-samples` rows per ray inside occupied cells instead of 128 + 128 along the whole ray), from scratch
    for --quality-steps steps; rays/s over the last --steps replays, then the same held-out metrics."""
    import torch
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    from autolabel_amd.quality import heldout_metrics, pipe_renderer
    layout = ModelLayout('hg+freq', 15, 128, 128, args.feature_dim, scene['n_classes'], bound=bound)
    P = Params(layout, device)
    P.init_(seed=0)
    pipe = HipPipeline(layout, P)
    occ = pipe.enable_marching(G=128, max_steps=1024, samples=args.march_samples, density_thresh=args.march_thresh)
    fx, fy, cx, cy = train.desc.fx, train.desc.fy, train.desc.cx, train.desc.cy
    pipe.mark_untrained_grid(train.world_to_camera(), (fx, fy, cx, cy), size=(train.w, train.h))
    eng = TrainEngine(pipe, feature_loss=True)
    batch = train.alloc_batch(B)
    dbg = lambda m: (torch.cuda.synchronize(), print('[march]', m, file=sys.stderr, flush=True)) if os.environ.get('ALN_BENCH_DEBUG') else None
    dbg('engine built')
    g = eng.graphed(train, batch, 1234, 99, warmup=3)
    dbg('captured')
    total = max(args.quality_steps, args.steps + args.warmup)
    n = g.steps
    while n < args.warmup:
        g(); n += 1
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):      # timed like the dense leg: the steps right after the warm-up, lr 5e-3
        g()
    torch.cuda.synchronize()
    dt = time.time() - t0
    n += args.steps
    dbg('timed')
    for frac, lr in ((0.6, 5e-3), (0.8, 2.5e-3), (1.0, 1.25e-3)):     # the dense leg's schedule
        eng.lr = lr
        dbg(f'lr set {lr} state_f {eng.state_f.tolist()} terms {eng.terms.tolist()}')
        while n < int(frac * total):
            g(); n += 1
            if os.environ.get('ALN_BENCH_DEBUG') and n % 16 in (0, 1, 2):
                dbg(f'n={n} terms {eng.terms.tolist()} scale {eng.state_f[0].item()}')
        dbg(f'lr {lr} done n={n}')
    torch.cuda.synchronize()
    def render(ro, rd, dn):   # marching render: the trained field is only meaningful where the grid lets samples fall
        parts = []
        ro, rd, dn = ro.reshape(-1, 3), rd.reshape(-1, 3), dn.reshape(-1)
        for a in range(0, ro.shape[0], 16384):
            out, _ = pipe.forward(ro[a:a + 16384].contiguous(), rd[a:a + 16384].contiguous(), dn[a:a + 16384].contiguous(),
                                  max(args.march_samples, 128), 0, False, train=False, march=True)
            parts.append({k: out[k].clone() for k in ('image', 'depth', 'semantic')})
        return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
    q = heldout_metrics(render, test, scene['n_classes'])
    q.update(steps=n)
    # render throughput through the occupancy grid: full 640x480 frames, the same 128 rows per ray as the quality render
    render_mrays = None
    if full is not None and args.render_frames > 0:
        fb = full.alloc_batch(full.w * full.h)
        def render_frame(f):
            full.get_test(f, fb)
            for a in range(0, full.w * full.h, 16384):
                pipe.forward(fb['rays_o'][a:a + 16384], fb['rays_d'][a:a + 16384], fb['direction_norms'][a:a + 16384].reshape(-1),
                             max(args.march_samples, 128), 0, False, train=False, march=True)
        render_frame(0)
        torch.cuda.synchronize()
        t1 = time.time()
        for f in range(args.render_frames):
            render_frame(f % full.n_frames)
        torch.cuda.synchronize()
        render_mrays = full.w * full.h * args.render_frames / (time.time() - t1) / 1e6
    return {'value': B * args.steps / dt, 'unit': 'rays/s', 'ms_per_step': 1000 * dt / args.steps, 'samples_per_ray': args.march_samples,
            'sample_rows_per_step': B * args.march_samples, 'grid': '128^3, one level', 'max_steps': 1024, 'density_thresh': args.march_thresh,
            'render_Mrays_per_s': render_mrays, 'render_rows_per_ray': max(args.march_samples, 128), 'occupied_fraction': occ.occupancy(), 'grid_updates': (n + occ.update_interval - 1) // occ.update_interval, 'quality': q,
            'note': 'cuda_ray=True path (dead in the reference: model_utils.py:72); timed over %d steps after %d warm-up steps like the dense '
                    'leg, grid refresh (every 16th step, its own captured graph) included; then trained on to the same step count and '
                    'learning-rate schedule as the dense leg' % (args.steps, args.warmup)}


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}',
              file=sys.stderr)
        sys.exit(2)
    import torch
    local = int(os.environ.get('LOCAL_RANK', 0))
    backend = os.environ.get('ALN_DIST_BACKEND', 'nccl')  # 'gloo' lets two ranks share one GPU (functional test only)
    if backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        kw = {'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}
        torch.distributed.init_process_group(backend, **kw)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    from autolabel_amd.parallel import rank_seed
    from autolabel_amd.quality import heldout_metrics, pipe_renderer, split_heldout
    scene, half, train, test, full, eng, frange, bound = build(args, device, rank, world)
    n_ranks_seen = torch.distributed.get_world_size() if world > 1 else 1
    B = args.global_batch // world if args.global_batch else args.batch
    assert B % 512 == 0 and B > 0, 'per-rank batch must be a multiple of the 512-ray chunk (autolabel/dataset.py:171)'
    batch = train.alloc_batch(B)
    dseed, mseed = rank_seed(1234, rank), rank_seed(99, rank)
    use_graph = not args.no_graph and world == 1   # data-parallel steps are issued launch by launch (collectives on a side stream)
    done = [0]

    def eager_step():
        train.next_train(batch, seed=dseed, step=done[0], frame_range=frange)
        eng.step(batch, seed=mseed, step=done[0])
        done[0] += 1

    if use_graph:
        graphed = eng.graphed(train, batch, dseed, mseed, frame_range=frange, warmup=min(3, max(args.warmup, 1)))
        done[0] = graphed.steps

        def step():
            graphed()
            done[0] += 1
    else:
        step = eager_step

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    while done[0] < args.warmup:
        step()
    sync()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    rays_per_s = B * world * args.steps / dt
    terms = eng.terms.tolist()

    # the same step launch by launch, HIP events (on the launch stream) around the timed kernels
    events = []
    if args.event_steps > 0:
        if use_graph:   # continue the step numbering of the replays
            graphed.counter.zero_()
        eng.pipe.kernel_events = events
        for _ in range(args.event_steps):
            eager_step()
        sync()
        eng.pipe.kernel_events = None
    live_rows = float(eng.ws.get('n_live', (1,), torch.int32).item())   # color-head rows of the last step (w > 1e-4)
    records_evt = eng.pipe.binned_record_count(B * (eng.S1 + eng.S2), ws=eng.ws)          # records of the last hash-grid backward (same state as the events)

    # render throughput: full 640x480 frames, 512 coarse steps, no upsampling (scripts/render.py:96-102)
    render_mrays, render_roof = None, None
    if rank == 0 and args.render_frames > 0:
        fb = full.alloc_batch(full.w * full.h)
        chunk = 16384
        def render_frame(f):
            full.get_test(f, fb)
            for a in range(0, full.w * full.h, chunk):
                eng.pipe.forward(fb['rays_o'][a:a + chunk], fb['rays_d'][a:a + chunk], fb['direction_norms'][a:a + chunk].reshape(-1),
                                 512, 0, False, train=False)
        render_frame(0)
        torch.cuda.synchronize()
        t1 = time.time()
        for f in range(args.render_frames):
            render_frame(f % full.n_frames)
        torch.cuda.synchronize()
        render_mrays = full.w * full.h * args.render_frames / (time.time() - t1) / 1e6
        # roofline of the render's dominant kernel pair (level-phased hash-grid gather): HIP events around it for one frame
        eng.pipe.kernel_events = rev = []
        render_frame(0)
        torch.cuda.synchronize()
        eng.pipe.kernel_events = None
        gd = [(e[0].elapsed_time(e[1]) * 1e-3, t) for e, n, t in rev if n == 'aln_encode_fwd_phased']
        if gd:
            rows = sum(t[1] for _, t in gd) / len(gd)
            avg = sum(d for d, _ in gd) / len(gd)
            nl, pad = eng.L.enc.grid.n_levels, eng.L.enc.enc_pad
            per_row = nl * 8 * 4 + 12 + pad * 2     # 16 x 8 fp16x2 corner reads + xyz in + the encoded row out (SURVEY 8d Bytes_fwd)
            render_roof = {'kernel': 'k_encode_grid_phased + k_encode_assemble (hash-grid gather, render)', 'bound': 'hbm',
                           'achieved': per_row * rows / avg / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                           'frac': per_row * rows / avg / 1e9 / 8000.0, 'avg_launch_us': avg * 1e6, 'launches': len(gd),
                           'rows_per_launch': rows, 'algorithmic_bytes_per_row': per_row}

    # quality of the trained state: continue to --quality-steps optimizer steps, then held-out metrics
    quality = None
    if args.quality_steps > 0:
        # StepLR of scripts/train.py:70-75 compressed to this run: lr halves at 60 % and 80 % of the steps (the learning rate is a
        # device word the Adam kernel reads, so the captured step follows it without re-capture)
        for frac, lr in ((0.6, 5e-3), (0.8, 2.5e-3), (1.0, 1.25e-3)):
            eng.lr = lr
            while done[0] < int(frac * args.quality_steps):
                step()
        sync()
        if rank == 0:
            quality = heldout_metrics(pipe_renderer(eng.pipe), test, scene['n_classes'])
            quality.update(steps=done[0], batch_per_gpu=B, lr='5e-3, halved at 60 % and 80 % of the steps', adam_steps_applied=int(eng.state_i[0].item()),
                           note='held-out frames (every 20th) of the bench scene at the training resolution, 256 samples/ray; labels on '
                                'every 10th training frame only; oracle parity of the same metrics: '
                                'tests/test_gpu_quality.py')

    if rank == 0:
        res = {
            'metric': 'train rays/sec (640x480 synthetic RGB-D scene, hg+freq, DINO-like features)', 'value': rays_per_s,
            'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1000 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'strong' if args.global_batch else 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': "S1 synthetic room standing in for the 'bench' scene: %d frames 640x480 trained at factor 2, " % args.frames +
                                   'DINO-like 64-d features, hg+freq L=16 T=2^19, 128+128 samples/ray',
                       'rays_per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'dp{world}', 'feature_dim': args.feature_dim,
                       'hip_graph': bool(use_graph)},
            'n_ranks_seen': n_ranks_seen,
            'render_Mrays_per_s': render_mrays,
            'loss_terms_last_timed_step': {'rgb': terms[0], 'depth': terms[1], 'feature': terms[2], 'semantic': terms[3], 'total': terms[4]},
            'quality': quality,
        }
        L = eng.L
        nl = L.enc.grid.n_levels
        # roofline of the dominant kernel(s): the hash-grid backward = k_encode_bwd_bin + k_encode_bwd_accum (one C-ABI call, one
        # pair of HIP events).  Algorithmic bytes per sample row: 16 levels x 8 corners x 2 features x 4 B fp32 RMW
        # counted once + the d_enc row (enc_pad x 2 B) + z (4 B).
        enc_ev = [(e, t) for e, n, t in events if n.startswith('aln_encode_bwd')]
        if enc_ev:
            durs = [e[0].elapsed_time(e[1]) * 1e-3 for e, _ in enc_ev]
            per_launch = [r * (lv * 8 * 2 * 4 + (L.enc.enc_pad * 2 + 4) * lv / nl) for _, (r, lv) in enc_ev]
            avg_s = sum(durs) / len(durs)
            achieved = (sum(per_launch) / len(per_launch)) / avg_s / 1e9
            pmc = {}
            try:
                with open(os.path.join(ROOT, 'profiles', 'r02_pmc_summary.json')) as f:
                    pmc = json.load(f)
            except (OSError, ValueError):
                pass
            res['roofline'] = {
                'kernel': 'k_encode_bwd_bin + k_encode_bwd_accum (hash-grid backward, one launch pair)',
                'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0,
                'traffic': pmc.get('encode_bwd_traffic_bytes_per_launch'), 'avg_launch_us': avg_s * 1e6, 'launches': len(durs),
                'algorithmic_bytes_per_launch': sum(per_launch) / len(per_launch),
                'records_per_launch': records_evt, 'record_bytes': 8,
                'per_level_split': 'profiles/r02_probe_encode_bwd_binned.txt (scripts/dev/probe_encode_bwd_binned.py: per level and per level group, binned vs atomic)',
                'atomic_requests_per_launch_round1_kernel': (pmc.get('atomic_kernel_requests_per_launch') or {}).get('TCC_ATOMIC_sum'),
                'note': 'HIP events around the launch pair over %d launch-by-launch steps right after the timed region (the timed '
                        'region itself replays a hipGraph, which cannot carry events); phase 1 sorts fp16x2 (index, value) records '
                        'by 64 KB table slice in LDS and streams them out (8 B/record), phase 2 streams them back and accumulates '
                        'in 64-bit fixed point in LDS: no global atomics; traffic = FETCH_SIZE + WRITE_SIZE of both kernels from '
                        'profiles/r02_pmc_summary.json' % args.event_steps}
            if res['roofline']['records_per_launch']:
                res['roofline']['records_per_s'] = res['roofline']['records_per_launch'] / avg_s
                res['roofline']['record_traffic_bytes_per_launch'] = 2 * 8 * res['roofline']['records_per_launch']   # written once, read once
        # second regime: the MLP heads against the dense fp16 MFMA peak.  Algorithmic FLOPs = 2 x MAC per
        # evaluated sample forward, 4 x MAC backward (data + weight gradients), unpadded widths; the forward recompute
        # inside the backward kernels is extra work, not counted.  Live rows of the color head are the device counter.
        def macs(k):
            m = L.nets[k]
            return m.n_in * m.hidden + (m.n_hidden - 1) * m.hidden * m.hidden + m.hidden * m.n_out
        mac = {'sigma': macs('sigma'), 'color': macs('color'), 'sem': macs('semf') + macs('semo')}
        flops = t_mlp = 0.0
        for e, n, t in events:
            if n.startswith('aln_wide'):   # one GEMM per launch: 2 M N K
                flops += 2.0 * t[1] * t[2]
                t_mlp += e[0].elapsed_time(e[1]) * 1e-3
                continue
            if not n.startswith(('aln_mlp', 'aln_sem_heads')):
                continue
            head, r = t
            if torch.is_tensor(r):
                r = live_rows   # the device counter is reused every step; the last step's value stands for all
            flops += (2.0 if n.endswith('_fwd') else 4.0) * mac[head] * r
            t_mlp += e[0].elapsed_time(e[1]) * 1e-3
        if t_mlp > 0:
            res['roofline_mlp'] = {'kernels': 'k_mlp_fwd + k_sem_fwd_fused + k_mlp_bwd_recomp8 + k_dw_reduce (all heads)' + (' + k_wide_nt / k_wide_tn' if L.sem_wide else ''), 'bound': 'mfma',
                                   'achieved': flops / t_mlp / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s',
                                   'frac': flops / t_mlp / 1e12 / 2500.0, 'us_per_step': t_mlp * 1e6 / args.event_steps,
                                   'algorithmic_gflop_per_step': flops / 1e9 / args.event_steps, 'live_color_rows': live_rows}
        if render_roof:
            res['roofline_render'] = render_roof
        if not args.no_march and world == 1:
            res['marching'] = marching_leg(args, scene, train, test, bound, device, B, full)
        if not args.no_cpu_baseline and world == 1:
            train_ids, _ = split_heldout(args.frames)
            res['cpu_baseline'] = cpu_baseline(half, train_ids, args.feature_dim, scene['n_classes'], bound)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
