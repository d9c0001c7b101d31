"""Headline benchmark: NeRF training throughput (rays/s) on a synthetic 640x480 RGB-D scene, at stated held-out quality.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  `--gpus N` with N > 1 and no RANK in the environment re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (before any GPU call) and exits with the child's code; under torchrun
the flag must equal WORLD_SIZE.  Weak scaling by default (every rank trains on its own 4096-ray batches drawn from its shard
of the frames, gradients averaged over RCCL once per step, hash-grid block as fp16); `--global-batch B` fixes the global batch
instead (B / N rays per rank: strong scaling).

A "step" = device ray generation + render forward + loss + backward + Adam for one batch, frames resident in HBM.  On one
GPU the step is replayed from a hipGraph (engine.GraphedStep; `--no-graph` issues it launch by launch).  After the timed
region the same step runs launch by launch with HIP events around the timed kernels (roofline figures), then training
continues to `--quality-steps` total steps and the held-out PSNR / depth L1 / mIoU of that state are reported -- and of
`--quality-seeds` - 1 further runs from other seeds (mean / min: the throughput travels with its quality).  Further legs on one
GPU: `marching` (cuda_ray=True, the same scene through the occupancy grid), `lseg` (512-d feature head), `roofline.traffic`
(HBM bytes of the dominant kernel pair from rocprofv3 PMC passes over a child run of this script), `cpu_baseline` (the oracle
on the host cores).  Prints ONE JSON line on rank 0 (contract in the task description).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')   # see autolabel_amd/__init__.py: must be set before HIP initialises

SCHEDULE = ((0.6, 5e-3), (0.8, 2.5e-3), (1.0, 1.25e-3))   # StepLR of scripts/train.py:70-75 compressed to the run: lr halves at 60 % / 80 %
SEEDS = [(0, 1234, 99), (1, 2234, 199), (2, 3234, 299)]    # (initialisation, data, sample noise) per quality run


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
    p.add_argument('--trained-steps', type=int, default=200, help='replays timed again AFTER the quality schedule (the `trained` sub-leg); 0 = skip')
    p.add_argument('--quality-seeds', type=int, default=3, help='runs (seed sets) the reported quality is the mean / min over')
    p.add_argument('--dp-leg-timeout', type=float, default=300.0, help='--gpus N: seconds a data-parallel leg may take before the watchdog prints the best finished leg and exits')
    p.add_argument('--dp-tail-timeout', type=float, default=1200.0, help='--gpus N: the same for everything after the legs')
    p.add_argument('--no-march', action='store_true', help='skip the occupancy-grid marching leg (second config)')
    p.add_argument('--march-samples', type=int, default=64)
    p.add_argument('--march-thresh', type=float, default=10.0)
    p.add_argument('--no-lseg', action='store_true', help='skip the LSeg-width (512-d feature head) leg')
    p.add_argument('--no-tiled-enc', action='store_true', help='A/B: hash-grid forward through plane buffers + the assembly pass instead of the tiled layout')
    p.add_argument('--no-planes-enc', action='store_true', help='A/B: the training step through row-major encoded rows + the assembly pass instead of the pair planes the 128-wide kernels read themselves')
    p.add_argument('--no-fold-dsigma', action='store_true', help="A/B: the density head's dL/dout rows through aln_assemble_grads instead of the backward kernel's own loader")
    p.add_argument('--no-fold-color-in', action='store_true', help="A/B: the colour head's input rows through aln_build_color_in instead of the compaction's second pass")
    p.add_argument('--tiled-enc-train', action='store_true', help='A/B: the tiled hash-grid output in the training step too (default: rendering only)')
    p.add_argument('--no-dp1', action='store_true', help='skip the leg that runs the data-parallel forms of the step through a one-rank RCCL group')
    p.add_argument('--no-dropin', action='store_true', help="skip the leg that times the reference's own route (scene directory -> SimpleTrainer)")
    p.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 PMC passes behind roofline.traffic')
    p.add_argument('--dp1-child', action='store_true', help=argparse.SUPPRESS)    # the child run behind `dp_world1`
    p.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)   # the run rocprofv3 wraps: a few eager steps, no JSON
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


class Workload:
    """The S1 scene on the device: 640x480, trained at factor 2 like scripts/train.py:15; every 20th frame held out."""

    def __init__(self, args, device):
        from autolabel_amd import synthetic
        from autolabel_amd.dataset import DeviceFrames
        from autolabel_amd.quality import split_heldout
        self.args, self.device = args, device
        scene = self.scene = synthetic.make_room_scene(n_frames=args.frames, seed=0, device=device, feat_dim=64, feat_hw=(60, 80))
        half = self.half = synthetic.subsample(scene, 2)
        self.train_ids, held = split_heldout(args.frames)
        pick = lambda sc, ids, sem: dict(sc, images=sc['images'][ids], depths=sc['depths'][ids], semantics=sc[sem][ids],
                                         features=sc['features'][ids] if sem == 'semantics' else None, T_CW=sc['T_CW'][ids])
        self.train = DeviceFrames.from_scene(pick(half, self.train_ids, 'semantics'), device)
        self.test = DeviceFrames.from_scene(pick(half, held, 'semantics_full'), device)
        self.full = DeviceFrames.from_scene(pick(scene, held[:max(args.render_frames, 1)], 'semantics_full'), device)
        lo, hi = scene['min_bounds'], scene['max_bounds']
        self.bound = float(((hi - lo) - (lo + hi) * 0.5).max())  # autolabel/model_utils.py:62-63
        self.n_classes = scene['n_classes']

    def engine(self, init_seed=0, march=False, feature_dim=None, world=1, pg=None, semantic_weight=1.0, overlap_comm=True, shard_optimizer=False,
               exchange_at_world_1=False, level_group=4):
        from autolabel_amd.engine import TrainEngine
        from autolabel_amd.parallel import broadcast_parameters
        from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
        a = self.args
        layout = ModelLayout('hg+freq', 15, 128, 128, feature_dim or a.feature_dim, self.n_classes, bound=self.bound)
        P = Params(layout, self.device)
        P.init_(seed=init_seed)
        if world > 1:
            broadcast_parameters(P.flat)
            P.refresh_shadows()
        pipe = HipPipeline(layout, P)
        pipe.tiled_enc_enabled = not getattr(a, 'no_tiled_enc', False)
        pipe.tiled_enc_train = bool(getattr(a, 'tiled_enc_train', False))
        pipe.planes_enc_train = not getattr(a, 'no_planes_enc', False)
        pipe.fold_dsigma = not getattr(a, 'no_fold_dsigma', False)
        pipe.fold_color_in = not getattr(a, 'no_fold_color_in', False)
        if march:
            t = self.train
            pipe.enable_marching(G=128, max_steps=1024, samples=a.march_samples, density_thresh=a.march_thresh)
            pipe.mark_untrained_grid(t.world_to_camera(), (t.desc.fx, t.desc.fy, t.desc.cx, t.desc.cy), size=(t.w, t.h))
        return TrainEngine(pipe, feature_loss=True, process_group=pg, semantic_weight=semantic_weight, overlap_comm=overlap_comm,
                           shard_optimizer=shard_optimizer, exchange_at_world_1=exchange_at_world_1, level_group=level_group,
                           direct_wire=os.environ.get('ALN_DIRECT_WIRE', '1') != '0')   # (A/B knob: 0 = fp32 table gradient + conversion passes)

    def renderer(self, eng, march):
        """render callable for quality.heldout_metrics: 256 rows per ray along the whole ray (dense) / 128 rows inside occupied cells."""
        import torch
        from autolabel_amd.quality import pipe_renderer
        if not march:
            return pipe_renderer(eng.pipe)
        rows = max(self.args.march_samples, 128)

        def render(ro, rd, dn):   # marching render: the trained field is only meaningful where the grid lets samples fall
            parts = []
            ro, rd, dn = ro.reshape(-1, 3), rd.reshape(-1, 3), dn.reshape(-1)
            for a in range(0, ro.shape[0], 16384):
                out, _ = eng.pipe.forward(ro[a:a + 16384].contiguous(), rd[a:a + 16384].contiguous(), dn[a:a + 16384].contiguous(), rows, 0, False,
                                          train=False, march=True)
                parts.append({k: out[k].clone() for k in ('image', 'depth', 'semantic')})
            return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
        return render

    def render_throughput(self, eng, march):
        """Full 640x480 frames: 512 coarse steps, no upsampling (scripts/render.py:96-102) / 128 rows per ray through the grid."""
        import torch
        full, n = self.full, self.args.render_frames
        fb = full.alloc_batch(full.w * full.h)
        rows = max(self.args.march_samples, 128) if march else 512

        def frame(f):
            full.get_test(f, fb)
            for a in range(0, full.w * full.h, 16384):
                eng.pipe.forward(fb['rays_o'][a:a + 16384], fb['rays_d'][a:a + 16384], fb['direction_norms'][a:a + 16384].reshape(-1), rows, 0,
                                 False, train=False, march=march)
        frame(0)
        torch.cuda.synchronize()
        t1 = time.time()
        for f in range(n):
            frame(f % full.n_frames)
        torch.cuda.synchronize()
        return full.w * full.h * n / (time.time() - t1) / 1e6, frame


def render_roofline(eng, frame, what):
    """HBM roofline of a render pass's dominant kernel pair, the level-phased hash-grid gather + row assembly, from HIP events
    around the launches of one frame.  Algorithmic bytes per sample row (SURVEY 8d Bytes_fwd): 16 levels x 8 fp16x2 corner reads
    + the position in (12 B) + the encoded row out (enc_pad x 2 B)."""
    import torch
    eng.pipe.kernel_events = rev = []
    frame(0)
    torch.cuda.synchronize()
    eng.pipe.kernel_events = None
    gd = [(e[0].elapsed_time(e[1]) * 1e-3, t) for e, n, t in rev if n == 'aln_encode_fwd_phased']
    if not gd:
        return None
    rows = sum(t[1] for _, t in gd) / len(gd)
    avg = sum(d for d, _ in gd) / len(gd)
    nl, pad = eng.L.enc.grid.n_levels, eng.L.enc.enc_pad
    per_row = nl * 8 * 4 + 12 + pad * 2
    return {'kernel': 'k_encode_grid_phased + k_encode_assemble (hash-grid gather, %s)' % what,
            'bound': 'hbm (limiter: L1 request rate of the 8-corner gathers, DESIGN.md 4.2)',
            'achieved': per_row * rows / avg / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': per_row * rows / avg / 1e9 / 8000.0,
            'avg_launch_us': avg * 1e6, 'launches': len(gd), 'rows_per_launch': rows, 'algorithmic_bytes_per_row': per_row}


def train_schedule(eng, step, done, total):
    """Continue to `total` optimizer steps under SCHEDULE (the learning rate is a device word: a captured step follows it)."""
    for frac, lr in SCHEDULE:
        eng.lr = lr
        while done[0] < int(frac * total):
            step()


def quality_runs(wl, args, first, march):
    """Held-out metrics of `first` (the timed run, already trained to --quality-steps) plus --quality-seeds - 1 fresh runs from
    other (initialisation, data, noise) seeds through the same schedule -> mean / min and the runs themselves."""
    import torch
    from autolabel_amd.quality import heldout_metrics
    runs = [dict(first, seeds=SEEDS[0])]
    B = args.batch
    for init_seed, data_seed, noise_seed in SEEDS[1:max(args.quality_seeds, 1)]:
        eng = wl.engine(init_seed, march=march)
        batch = wl.train.alloc_batch(B)
        g = eng.graphed(wl.train, batch, data_seed, noise_seed, warmup=3)
        done = [g.steps]

        def step():
            g(); done[0] += 1
        train_schedule(eng, step, done, args.quality_steps)
        torch.cuda.synchronize()
        q = heldout_metrics(wl.renderer(eng, march), wl.test, wl.n_classes)
        q.update(steps=done[0], adam_steps_applied=int(eng.state_i[0].item()), seeds=(init_seed, data_seed, noise_seed))
        runs.append(q)
        del eng, g, batch
        torch.cuda.empty_cache()
    agg = lambda k, f: float(f([r[k] for r in runs]))
    mean = lambda v: sum(v) / len(v)
    return {'vs_oracle': None if march else oracle_gate_record(), 'psnr_db_mean': agg('psnr_db', mean), 'psnr_db_min': agg('psnr_db', min), 'miou_mean': agg('miou', mean), 'miou_min': agg('miou', min),
            'depth_l1_m_mean': agg('depth_l1_m', mean), 'depth_l1_m_max': agg('depth_l1_m', max), 'n_runs': len(runs),
            'steps': args.quality_steps, 'batch_per_gpu': B, 'lr': '5e-3, halved at 60 % and 80 % of the steps', 'runs': runs,
            'note': 'held-out frames (every 20th) of the bench scene at the training resolution; labels on every 10th training frame only; '
                    'seeds = (initialisation, data, sample noise); training is bit-reproducible, so a seed set always gives these numbers; '
                    'oracle parity of the same metrics: tests/test_gpu_quality.py'}


def oracle_gate_record():
    """HIP-vs-oracle quality deltas of the matched-quality gate, read from the COMMITTED records of its last multi-seed runs
    (scripts/dev/quality_gate_run.py = tests/test_gpu_quality.run_gate): profiles/r05_quality_gate_5x5.json (5 + 5 seeds on the reduced
    gate model) and profiles/r06_quality_gate_benchmodel.json (3 + 3 seeds at the BENCH's own model: L = 16, T = 2^19, 128 + 128 samples).
    The oracle trainees are far too slow to run inside a bench; this only carries their numbers next to the bench's own quality."""
    def rec(name):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', name)
        try:
            r = json.load(open(path))
        except Exception:
            return None
        return {'source': 'profiles/%s (committed record, not measured in this run)' % name,
                'delta_psnr_db': r.get('delta_psnr_db'), 'delta_miou_pt': r.get('delta_miou_pt'), 'se_delta_psnr_db': r.get('se_delta_psnr_db'),
                'se_delta_miou_pt': r.get('se_delta_miou_pt'), 'n_hip_runs': len(r.get('hip', [])), 'n_oracle_runs': len(r.get('oracle_runs', [])),
                'two_sided_ok': r.get('two_sided_ok'), 'gate_model': r.get('model'), 'gate_scene': r.get('scene')}
    out = rec('r05_quality_gate_5x5.json')
    if out is not None:
        out['bench_model'] = rec('r06_quality_gate_benchmodel.json')
    return out


def scatter_roofline(eng, events, rows_per_step, n_event_steps):
    """Dominant kernel pair: k_encode_bwd_bin + k_encode_bwd_accum (one C-ABI call, one pair of HIP events).  Algorithmic bytes per
    sample row: 16 levels x 8 corners x 2 features x 4 B fp32 RMW counted once + the d_enc row (enc_pad x 2 B) + z (4 B)."""
    L = eng.L
    nl = L.enc.grid.n_levels
    enc_ev = [(e, t) for e, n, t in events if n.startswith('aln_encode_bwd')]
    if not enc_ev:
        return None
    durs = [e[0].elapsed_time(e[1]) * 1e-3 for e, _ in enc_ev]
    per_launch = [r * (lv * 8 * 2 * 4 + (L.enc.enc_pad * 2 + 4) * lv / nl) for _, (r, lv) in enc_ev]
    avg_s = sum(durs) / len(durs)
    alg_scatter = sum(per_launch) / len(per_launch)
    # one GPU: phase 2 also takes the Adam step for the table (12 B read + 14 B written per parameter: p, m, v; p, m, v, fp16 shadow)
    fused = bool(eng.fuse_grid_adam)
    alg_adam = 26.0 * L.n_grid if fused else 0.0
    alg = alg_scatter + alg_adam
    records = eng.pipe.binned_record_count(rows_per_step, ws=eng.ws)   # PAIR records of the last launch (same state as the events)
    rec_bytes = 2 * 12 * records if records else None                 # 12 bytes each, written once by phase 1, read once by phase 2
    return {'kernel': 'k_encode_bwd_bin + k_encode_bwd_accum (hash-grid backward, one launch pair' + ("; the table's Adam step inside phase 2)" if fused else ')'),
            'bound': "hbm (phase 2 with the table's optimizer inside runs at the HBM rate of its own traffic; phase 1 is latency- and issue-bound)",
            'achieved': alg / avg_s / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': alg / avg_s / 1e9 / 8000.0, 'traffic': None,
            'avg_launch_us': avg_s * 1e6, 'launches': len(durs), 'rows_per_launch': rows_per_step, 'algorithmic_bytes_per_launch': alg,
            'algorithmic_bytes_scatter': alg_scatter, 'algorithmic_bytes_optimizer': alg_adam, 'optimizer_fused_into_phase2': fused,
            'pair_without_optimizer_us': getattr(eng, 'pair_without_optimizer_us', None),
            'pair_records_per_launch': records, 'pair_record_bytes': 12, 'record_traffic_bytes_per_launch': rec_bytes,
            'pair_records_per_s': records / avg_s if records else None,
            'limiter': 'round-6 section clocks and stub builds (profiles/NOTES_experiments.md): phase 2 with the optimizer inside reads the 585 MB of '
                       'pair records and moves 436 MB of optimizer state in ~205 us = 4.9 TB/s -- taking the LDS atomics or the conversions out of it '
                       'changes nothing (without the optimizer: 232 -> 191 / 211 us), its own traffic is the roof; phase 1 (~295 us) is three times '
                       'its HBM time: per level corner arithmetic, ranking atomics, the sorted LDS stores and the copy-out share one critical '
                       'path between two barriers, 2.67 rounds of 768 resident blocks',
            'note': 'HIP events around the launch pair over %d launch-by-launch steps right after the timed region (the timed region itself '
                    'replays a hipGraph, which cannot carry events); phase 1 sorts PAIR records (two slots, two fp16x2 values: 12 B) by table slice in LDS and '
                    'streams them out, phase 2 streams them back and accumulates in 64-bit fixed point in LDS: no global '
                    'atomics, bit-reproducible; on one GPU phase 2 applies Adam to its slice of the table from those sums (no gradient round '
                    'trip through HBM, no separate optimizer pass over the 14.2 M table parameters); pair_without_optimizer_us = the same pair '
                    'with the optimizer left to aln_adam_step (4 extra launch-by-launch steps, bit-identical results)' % n_event_steps}


def mlp_roofline(eng, events, n_event_steps, live_rows):
    """Second regime: the MLP heads against the dense fp16 MFMA peak.  Algorithmic FLOPs = 2 x MAC per evaluated sample
    forward, 4 x MAC backward (data + weight gradients), unpadded widths; the forward recompute inside the backward kernels is
    extra work, not counted.  Live rows of the color head are the device counter."""
    import torch
    L = eng.L

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
        if not n.startswith(('aln_mlp', 'aln_sem_heads', 'aln_density')):
            continue
        if t is None:   # aln_mlp_dw_reduce_all: time of the MLP backward, no FLOPs of its own
            t_mlp += e[0].elapsed_time(e[1]) * 1e-3
            continue
        head, r = t
        if torch.is_tensor(r):
            r = live_rows   # the device counter is reused every step; the last step's value stands for all
        flops += (2.0 if n.endswith(('_fwd', '_fwd_sums')) else 4.0) * mac[head] * r
        t_mlp += e[0].elapsed_time(e[1]) * 1e-3
    if t_mlp <= 0:
        return None
    return {'kernels': 'k_mlp_fwd + k_sem_fwd_fused + ' + ('k_mlp_bwd_recomp8 + k_wide_nt / k_wide_tn' if L.sem_wide else 'k_mlp_bwd128 + k_sem_bwd_pair') +
                       ' + k_dw_reduce_all (all heads)',
            'bound': 'mfma (limiter: one wave per SIMD -- LDS operand reads and the pack / mask VALU work between MFMA groups do not overlap them)', 'achieved': flops / t_mlp / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': flops / t_mlp / 1e12 / 2500.0,
            'us_per_step': t_mlp * 1e6 / n_event_steps, 'algorithmic_gflop_per_step': flops / 1e9 / n_event_steps, 'live_color_rows': live_rows}


def event_steps(eng, eager_step, n, sync):
    events = []
    eng.pipe.kernel_events = events
    for _ in range(n):
        eager_step()
    sync()
    eng.pipe.kernel_events = None
    eng.pair_without_optimizer_us = None
    if eng.fuse_grid_adam:
        # for reference only: the same launch pair WITHOUT the optimizer inside its second phase (the gradient goes to P.grad and
        # aln_adam_step covers the table too) -- the results are bit-identical, so the trained state does not notice
        ref = []
        eng.fuse_grid_adam, eng.pipe.kernel_events = False, ref
        for _ in range(4):
            eager_step()
        sync()
        eng.fuse_grid_adam, eng.pipe.kernel_events = True, None
        d = [e[0].elapsed_time(e[1]) for e, n_, t in ref if n_.startswith('aln_encode_bwd')]
        eng.pair_without_optimizer_us = 1e3 * sum(d) / len(d) if d else None
    return events


def timed_leg(wl, args, eng, B, dseed, mseed, frange, use_graph, world, sync):
    """warm-up, the timed region (K steps, barrier + synchronize on both sides), then the launch-by-launch event steps."""
    import torch
    batch = wl.train.alloc_batch(B)
    done = [0]

    def eager_step():
        wl.train.next_train(batch, seed=dseed, step=done[0], frame_range=frange)
        eng.step(batch, seed=mseed, step=done[0])
        done[0] += 1
    graphed = None
    if use_graph:
        graphed = eng.graphed(wl.train, batch, dseed, mseed, frame_range=frange, warmup=min(3, max(args.warmup, 1)))
        done[0] = graphed.steps

        def step():
            graphed()
            done[0] += 1
    else:
        step = eager_step
    while done[0] < args.warmup:
        step()
    sync()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], device=wl.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    terms = eng.terms.tolist()
    events = []
    if args.event_steps > 0:
        # the eager steps continue the step numbering of the replays (absolute step numbers, the device counter is not used), and the
        # replays afterwards continue theirs: the trained state does not depend on how --steps / --warmup / --event-steps split the run
        eng._calls = done[0]     # (marching: the density-grid refresh of the eager steps follows the absolute step number too)
        events = event_steps(eng, eager_step, args.event_steps, sync)
        if graphed is not None:
            graphed.counter.fill_(done[0])
            graphed.steps = done[0]
    return dt, terms, events, step, done


def dp_world1_leg(wl, args, B, teardown=True):
    """The data-parallel forms of the step on ONE GPU: a one-rank `nccl` (= RCCL) process group, every collective really issued
    (TrainEngine(exchange_at_world_1=True)).  What this measures is what a rank's step costs once the gradient takes the data-parallel
    route -- table gradient through HBM instead of the optimizer inside the scatter, wire-format kernels, collectives, the communication
    stream, a separate optimizer pass over the table (replicated) or its owned slices (sharded) -- NOT scaling: nothing crosses xGMI."""
    import socket
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return {'skipped': 'a process group already exists'}
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=wl.device)
    out = {'backend': 'nccl (RCCL)', 'world': 1, 'note': 'one rank: the collectives run but move nothing; per-rank cost of the data-parallel '
           'route, not a scaling measurement'}
    a2 = argparse.Namespace(**vars(args))
    a2.event_steps = 0
    try:
        for name, overlap, graph, shard, lg in [('dp_simple', False, False, False, 4), ('dp_overlap', True, False, False, 4), ('dp_overlap_g8', True, False, False, 8),
                                                ('dp_sharded', True, False, True, 4), ('dp_graph', True, True, False, 4), ('dp_graph_g8', True, True, False, 8),
                                                ('dp_sharded_graph', True, True, True, 4)]:
            if os.environ.get('ALN_DP1_ONLY') and name not in os.environ['ALN_DP1_ONLY'].split(','):   # (profiling runs: scripts/dev/run_profiles.sh)
                continue
            try:
                e = wl.engine(SEEDS[0][0], pg=dist.group.WORLD, overlap_comm=overlap, shard_optimizer=shard, exchange_at_world_1=True, level_group=lg)
                dt = timed_leg(wl, a2, e, B, SEEDS[0][1], SEEDS[0][2], None, graph, 1, torch.cuda.synchronize)[0]
                out[name] = {'ms_per_step': 1000 * dt / args.steps, 'rays_per_s': B * args.steps / dt, 'hip_graph': graph}
                del e
            except Exception as ex:
                out[name] = {'error': f'{type(ex).__name__}: {ex}'[:300]}
            torch.cuda.empty_cache()
    finally:
        if teardown:
            import gc
            gc.collect()
            torch.cuda.synchronize()
            dist.destroy_process_group()
    return out


def dp_world1_in_child(args, B, timeout=240):
    """dp_world1_leg in a child process under a timeout: RCCL set-up is the one part of a default run that could hang on a box where it
    has never run, and the JSON line of this process must not depend on it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--dp1-child', '--steps', str(args.steps), '--warmup', str(args.warmup), '--batch', str(B),
           '--frames', str(args.frames), '--feature-dim', str(args.feature_dim)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {'error': f'did not finish within {timeout} s'}
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not lines:
        err = ' | '.join(l for l in r.stderr.splitlines() if l.strip() and not l.startswith('frame #'))
        return {'error': f'child exited with {r.returncode}: {err[-600:]}'}
    out = json.loads(lines[-1])
    if r.returncode != 0:
        out['child_exit_code'] = r.returncode   # (the legs had finished: the line is printed before the process group is torn down)
    return out


def dp1_child(args):
    import torch
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    wl = Workload(args, device)
    if os.environ.get('ALN_DP1_ONLY'):   # a profiled run of chosen eager legs: tear down and leave normally, so that rocprofv3 writes its trace
        print(json.dumps(dp_world1_leg(wl, args, args.batch, teardown=True)), flush=True)
        return 0
    print(json.dumps(dp_world1_leg(wl, args, args.batch, teardown=False)), flush=True)
    # captured graphs that hold RCCL kernels and a destroyed communicator do not always unwind in a safe order at interpreter exit (one
    # abort in three in-bench runs, after the line above): leave without running destructors
    sys.stdout.flush()
    os._exit(0)


def marching_leg(wl, args, B):
    """Second configuration: the same scene, model and batch trained through occupancy-grid marching
    (cuda_ray=True: `--march-samples` rows per ray inside occupied cells instead of 128 + 128 along the whole ray)."""
    import torch
    from autolabel_amd.quality import heldout_metrics
    eng = wl.engine(SEEDS[0][0], march=True)
    occ = eng.pipe.occ
    sync = torch.cuda.synchronize
    # every 16th step carries the density-grid refresh (2 M cells through the density head): the timed region of this leg is whole
    # refresh periods, at least four of them, whatever --steps says for the headline leg
    a2 = argparse.Namespace(**vars(args))
    a2.steps = max(64, (args.steps + 15) // 16 * 16)
    a2.warmup = max(args.warmup, 16)
    args = a2
    dt, terms, events, step, done = timed_leg(wl, args, eng, B, SEEDS[0][1], SEEDS[0][2], None, True, 1, sync)
    rows = B * args.march_samples
    roof = scatter_roofline(eng, events, rows, args.event_steps) if events else None
    live = float(eng.ws.get('n_live', (1,), torch.int32).item())
    roof_mlp = mlp_roofline(eng, events, args.event_steps, live) if events else None
    quality = None
    if args.quality_steps > 0:
        train_schedule(eng, step, done, args.quality_steps)
        sync()
        q = heldout_metrics(wl.renderer(eng, True), wl.test, wl.n_classes)
        q.update(steps=done[0], adam_steps_applied=int(eng.state_i[0].item()))
        quality = quality_runs(wl, args, q, march=True)
    render_mrays, render_roof = None, None
    if args.render_frames > 0:
        render_mrays, frame = wl.render_throughput(eng, True)
        render_roof = render_roofline(eng, frame, 'render through the occupancy grid: %d rows per ray inside occupied cells' % max(args.march_samples, 128))
    return {'value': B * args.steps / dt, 'unit': 'rays/s', 'ms_per_step': 1000 * dt / args.steps, 'steps': args.steps, 'warmup': args.warmup,
            'samples_per_ray': args.march_samples,
            'sample_rows_per_step': rows, 'grid': '128^3, one level', 'max_steps': 1024, 'density_thresh': args.march_thresh,
            'render_Mrays_per_s': render_mrays, 'render_rows_per_ray': max(args.march_samples, 128), 'occupied_fraction': occ.occupancy(),
            'roofline': roof, 'roofline_mlp': roof_mlp, 'roofline_render': render_roof, 'quality': quality,
            'note': 'cuda_ray=True path (dead in the reference: model_utils.py:72); timed over %d steps after %d warm-up steps like the dense '
                    'leg, grid refresh (every 16th step, its own captured graph) included; every ray gets %d rows, rows outside occupied cells '
                    'are zero-length padding (the record / byte counts of `roofline` are what the scatter actually moved)'
                    % (args.steps, args.warmup, args.march_samples)}


def lseg_leg(wl, args, B):
    """configs[4] shape on one GPU: 512-d feature head (LSeg, docs/vision-language.md:19), both semantic heads on wide.hip."""
    import torch
    a2 = argparse.Namespace(**vars(args))
    a2.steps, a2.warmup, a2.event_steps = min(args.steps, 30), min(args.warmup, 10), min(args.event_steps, 5)
    eng = wl.engine(SEEDS[0][0], feature_dim=512, semantic_weight=0.0)   # scripts/ros/node.py:166-176: feature_dim 512, semantic_weight 0.0
    sync = torch.cuda.synchronize
    dt, terms, events, step, done = timed_leg(wl, a2, eng, B, SEEDS[0][1], SEEDS[0][2], None, True, 1, sync)
    live = float(eng.ws.get('n_live', (1,), torch.int32).item())
    return {'value': B * a2.steps / dt, 'unit': 'rays/s', 'ms_per_step': 1000 * dt / a2.steps, 'steps': a2.steps, 'warmup': a2.warmup,
            'feature_dim': 512, 'semantic_weight': 0.0, 'linear_last_layer_per_ray': bool(eng.sem_linear), 'roofline_mlp': mlp_roofline(eng, events, a2.event_steps, live) if events else None,
            'loss_terms_last_timed_step': dict(zip(('rgb', 'depth', 'feature', 'semantic', 'total'), terms)),
            'note': 'the reference\'s LSeg training configuration (scripts/ros/node.py:166-176: feature_dim 512, semantic_weight 0.0): '
                    'semantic_features 16->512->512 per sample on the hand-written MFMA GEMMs (k_wide_nt / k_wide_tn), its linear last '
                    'layer once per RAY on the composited hidden activation, semantic_out skipped (no loss reaches it); the 64-d '
                    'DINO-like targets supervise the first 64 of the 512 feature channels'}


def dropin_leg(wl, args, B):
    """The reference's own route, timed: the training frames written as a scene DIRECTORY (README.md:107-135 layout), read back by
    `SceneDataset` at factor 2 with DINO-like feature maps, wrapped in `DataLoader(num_workers=1)` and trained by
    `SimpleTrainer.train_iterations` with the optimizer / schedule / EMA objects of scripts/train.py:50-93 -- what
    `python scripts/train.py <scene> --features dino` executes.  SimpleTrainer moves the frames into HBM when they fit (they do) and
    replays the step from the hipGraph; `host_loader` times the same trainer with device_data=False (the reference's data path:
    numpy batch assembly in a worker process + H2D copy per step)."""
    import math
    import shutil
    import tempfile
    from argparse import Namespace
    import torch
    from torch import optim
    from autolabel_amd import model_utils, utils
    from autolabel_amd.dataset import LenDataset, SceneDataset
    from autolabel_amd.trainer import SimpleTrainer
    tmp = tempfile.mkdtemp(prefix='aln_scene_')
    try:
        t0 = time.time()
        sc, ids = wl.scene, wl.train_ids
        cpu = lambda v: v.cpu() if torch.is_tensor(v) else v
        utils.write_scene(dict(sc, images=cpu(sc['images'][ids]), depths=cpu(sc['depths'][ids]), semantics=cpu(sc['semantics'][ids]),
                               features=cpu(sc['features'][ids]), T_CW=sc['T_CW'][ids]), tmp)
        t_write = time.time() - t0
        flags = model_utils.model_flag_parser().parse_args(['--features', 'dino', '--feature-dim', str(args.feature_dim)])
        t0 = time.time()
        dataset = SceneDataset('train', tmp, factor=2.0, batch_size=B, features='dino')
        t_load = time.time() - t0
        opt = Namespace(rand_pose=-1, color_space='srgb', feature_loss=True, rgb_weight=flags.rgb_weight, depth_weight=flags.depth_weight,
                        semantic_weight=flags.semantic_weight, feature_weight=flags.feature_weight)
        optimizer = lambda model: torch.optim.Adam([{'name': 'encoding', 'params': list(model.encoder.parameters())},
                                                    {'name': 'net', 'params': model.network_parameters(), 'weight_decay': 1e-6}],
                                                   lr=flags.lr, betas=(0.9, 0.99), eps=1e-15)
        scheduler = lambda o: optim.lr_scheduler.StepLR(o, gamma=0.5, step_size=max(10000 // math.log(1e-4 / flags.lr, 0.5) // 1000, 1))

        def run(device_data, warm, steps):
            torch.manual_seed(SEEDS[0][0])
            model = model_utils.create_model(dataset.min_bounds, dataset.max_bounds, dataset.n_classes, flags)
            tr = SimpleTrainer('ngp', opt, model, device=wl.device, workspace=None, optimizer=optimizer, criterion=torch.nn.MSELoss(reduction='none'),
                               fp16=True, ema_decay=0.95, lr_scheduler=scheduler, scheduler_update_every_step=False, metrics=[],
                               use_checkpoint='scratch', mute=True, device_data=device_data)
            loader = torch.utils.data.DataLoader(LenDataset(dataset, 1000), batch_size=None, num_workers=1)
            loader._data = dataset
            tr.train_iterations(loader, warm)
            torch.cuda.synchronize()
            t1 = time.time()
            tr.train_iterations(loader, steps)
            torch.cuda.synchronize()
            dt = time.time() - t1
            return B * steps / dt, 1000 * dt / steps, float(tr.engine.terms[4]), isinstance(tr.resident_loader(loader), type(loader))
        v, ms, loss, stayed_host = run('auto', max(args.warmup, 20), args.steps)
        res = {'value': v, 'unit': 'rays/s', 'ms_per_step': ms, 'steps': args.steps, 'frames_resident_in_hbm': not stayed_host,
               'loss_total_last_step': loss, 'scene_write_s': t_write, 'scene_load_s': t_load,
               'route': 'utils.write_scene -> SceneDataset(factor 2, features dino) -> DataLoader(num_workers=1) -> SimpleTrainer.train_iterations '
                        '(ema update + scheduler step per call included)'}
        try:
            hv, hms, _, _ = run(False, 10, min(args.steps, 40))
            res['host_loader'] = {'value': hv, 'unit': 'rays/s', 'ms_per_step': hms, 'steps': min(args.steps, 40),
                                  'note': "device_data=False: the reference's data path (numpy _next_train in a DataLoader worker, H2D per step)"}
        except Exception as e:   # a worker process that cannot be forked on this box must not cost the line
            res['host_loader'] = {'error': repr(e)[:200]}
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel pair from the PMC counters: rocprofv3 --pmc around a child run of this script
    (`--pmc-child`: a few launch-by-launch steps of the same step on a smaller frame set), FETCH_SIZE and WRITE_SIZE in separate
    passes as MI355X_MICROARCH.md prescribes (TCC slots); both counters are in KB.  gfx950 correction: FETCH_SIZE under-reports wide
    coalesced streaming reads by 2x -- that is the read pattern of phase 2 (12-byte record loads, 768 B per wave instruction) and of
    phase 1's d_enc rows, so the raw and the doubled-fetch figure are both reported; `traffic` is the raw sum (a lower bound)."""
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on PATH'
    res = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix='aln_pmc_', dir='/tmp')
        env = dict(os.environ, TMPDIR='/tmp', DEBUG_CLR_GRAPH_PACKET_CAPTURE='0')   # (the child steps launch by launch; exported anyway: the profiler's tool library initialises HIP before Python runs)
        cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__), '--pmc-child',
               '--batch', str(args.batch), '--frames', '40', '--feature-dim', str(args.feature_dim)]
        try:
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
        except (subprocess.SubprocessError, OSError) as e:
            shutil.rmtree(out, ignore_errors=True)
            return None, f'rocprofv3 --pmc {counter} failed: {type(e).__name__}'
        acc = {}
        for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == counter and r['Kernel_Name'].startswith('k_encode_bwd_'):
                    acc.setdefault(r['Kernel_Name'].split('(')[0], []).append(float(r['Counter_Value']))
        shutil.rmtree(out, ignore_errors=True)
        if not acc:
            return None, f'no {counter} rows for k_encode_bwd_* in the rocprofv3 output'
        res[counter] = {k: 1024.0 * sum(v) / len(v) for k, v in acc.items()}   # KB -> bytes, averaged over the launches
    fetch, write = sum(res['FETCH_SIZE'].values()), sum(res['WRITE_SIZE'].values())
    return {'traffic': fetch + write, 'fetch_bytes': fetch, 'write_bytes': write, 'traffic_fetch_doubled': 2 * fetch + write,
            'per_kernel': res, 'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around `bench.py --pmc-child` '
                                         '(4 launch-by-launch steps, same batch and model, 40 frames) in this run'}, None


def pmc_child(args):
    """The program rocprofv3 wraps for pmc_traffic: the same step, launch by launch, a handful of times."""
    import torch
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    a2 = argparse.Namespace(**vars(args))
    a2.render_frames = 1
    wl = Workload(a2, device)
    eng = wl.engine(SEEDS[0][0])
    batch = wl.train.alloc_batch(args.batch)
    for i in range(4):
        wl.train.next_train(batch, seed=SEEDS[0][1], step=i)
        eng.step(batch, seed=SEEDS[0][2], step=i)
    torch.cuda.synchronize()


def cpu_baseline(wl, args):
    """the reference has no CPU path, so the CPU baseline is this repo's fp32 oracle ("port") on the host cores:
    (1) S1 -- the bench workload itself: rays drawn from the bench scene by the host mirror of `_next_train`, same model config,
    128+128 samples, 1024-ray batches (a quarter of the 4096-ray bench batch: the full batch costs ~1 min per step), 5 timed steps;
    (2) S0 -- BASELINE configs[0]: 32x32 cube, hash grid L=4, its full 8192 rays per step; (3) the reference-style host ray
    generation (`_next_train`, numpy) on the S1 scene, B=4096."""
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
        return B / (sum(times) / len(times)), sum(times)

    cpu = lambda sc: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sc.items()}
    sub = cpu(wl.half)
    ids = wl.train_ids[:24]   # the host dataset keeps float images: a 24-frame subset of the bench scene is enough to draw rays from
    s1 = dict(sub, images=sub['images'][ids], depths=sub['depths'][ids], semantics=sub['semantics'][ids],
              features=sub['features'][ids], T_CW=sub['T_CW'][ids])
    v1, t1 = train_steps(ArrayDataset(s1, batch_size=1024), O.ModelConfig(feature_dim=args.feature_dim, n_classes=wl.n_classes, bound=wl.bound),
                         1024, 5, 64)
    ds1b = ArrayDataset(s1, batch_size=4096)
    ds1b._next_train()
    t0 = time.time()
    for _ in range(5):
        ds1b._next_train()
    raygen = 5 * 4096 / (time.time() - t0)
    cube = cpu(synthetic.make_cube_scene())
    lo, hi = cube['min_bounds'], cube['max_bounds']
    b0 = float(((hi - lo) - (lo + hi) * 0.5).max())
    v0, t0s = train_steps(ArrayDataset(cube, batch_size=8192), O.ModelConfig(encoding='hg+freq', feature_dim=64, n_classes=cube['n_classes'], bound=b0,
                                                                           grid=O.GridSpec(n_levels=4)), 8192, 2, 0)
    return {'value': v1, 'unit': 'rays/s', 'cores': host_threads(), 'kind': 'port',
            'sample': '5 timed oracle (fp32 PyTorch) train steps of 1024 rays x (128+128) samples drawn from the bench scene (S1, host '
                      '_next_train), same model config (%.0f s of CPU work); the bench batch is 4096 rays = 4 such sub-batches' % t1,
            's0_value': v0, 's0_sample': "2 timed oracle train steps of configs[0]'s full 8192-ray batch on the 32x32 cube, hash grid L=4, "
                                         '128+128 samples (%.0f s of CPU work)' % t0s,
            'host_raygen_rays_per_s': raygen, 'host_raygen_sample': '5 batches of 4096 rays, numpy mirror of dataset._next_train on the S1 scene'}


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if args.dp1_child:
        return dp1_child(args)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}',
              file=sys.stderr)
        sys.exit(2)
    # the PMC passes start other programs (rocprofv3 -> a child run of this script): done first, while this process has not
    # touched the GPU yet
    pmc, pmc_why = (None, 'skipped')
    if world == 1 and not args.no_pmc and args.event_steps > 0:
        pmc, pmc_why = pmc_traffic(args)
    import torch
    local = int(os.environ.get('LOCAL_RANK', 0))
    backend = os.environ.get('ALN_DIST_BACKEND', 'nccl')  # 'gloo' lets two ranks share one GPU (functional test of this script only)
    if backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        kw = {'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}
        torch.distributed.init_process_group(backend, **kw)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    from autolabel_amd.parallel import frame_shard, rank_seed, wire_bytes
    from autolabel_amd.quality import heldout_metrics
    wl = Workload(args, device)
    pg = torch.distributed.group.WORLD if world > 1 else None
    frange = frame_shard(len(wl.train_ids), rank, world)
    n_ranks_seen = torch.distributed.get_world_size() if world > 1 else 1
    B = args.global_batch // world if args.global_batch else args.batch
    assert B % 512 == 0 and B > 0, 'per-rank batch must be a multiple of the 512-ray chunk (autolabel/dataset.py:171)'
    args.batch = B
    dseed, mseed = rank_seed(SEEDS[0][1], rank), rank_seed(SEEDS[0][2], rank)
    use_graph = not args.no_graph and world == 1

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    dp_legs = None
    if world == 1:
        eng = wl.engine(SEEDS[0][0])
        dt, terms, events, step, done = timed_leg(wl, args, eng, B, dseed, mseed, frange, use_graph, world, sync)
    else:
        # Data parallel: the step is timed in up to three forms, each inside its own try -- the first RCCL run of this code may be the
        # driver's, and one exception must not cost the line.  dp_simple: ONE all-reduce of the gradient buffer after the backward
        # pass (the plainest use of the collective); dp_overlap: five buckets on a communication stream behind the scatter's level
        # groups (groups of four levels; dp_overlap_g8 / dp_graph_g8: groups of eight = three buckets -- 60 us less scatter time per step,
        # 11 MB instead of 1.4 MB of exchange with nothing to hide behind: which wins is the links' call); dp_graph: the overlapped step captured into a hipGraph, collectives included (launch-bound at small per-GPU
        # batches otherwise); dp_sharded: the overlapped step with the table's optimizer sharded over the ranks (reduce-scatter of the
        # gradient buckets, Adam on 1 / world of the table, all-gather of the fp16 table).  The headline is the fastest leg that
        # finished on EVERY rank.
        a2 = argparse.Namespace(**vars(args))
        a2.event_steps = 0
        dp_legs, best = {}, None
        can_graph = not args.no_graph and backend == 'nccl'   # (gloo collectives cannot be captured)
        # (the eager legs first: a capture that goes wrong can then only cost the graph legs)
        plan = ([('dp_simple', False, False, False, 4), ('dp_overlap', True, False, False, 4), ('dp_overlap_g8', True, False, False, 8),
                 ('dp_sharded', True, False, True, 4)] +
                ([('dp_graph', True, True, False, 4), ('dp_graph_g8', True, True, False, 8), ('dp_sharded_graph', True, True, True, 4)] if can_graph else []))

        # A collective that never returns (a rank lost, a communicator poisoned by a failed capture) would take the line with it: every
        # leg, and the rest of the run after the legs, runs under a watchdog that prints the line of the best leg finished so far
        # (rank 0) and ends the process on every rank.
        def fallback_line(why):
            line = {'metric': 'train rays/sec (640x480 synthetic RGB-D scene, hg+freq, DINO-like features)', 'value': None, 'unit': 'rays/s',
                    'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True,
                    'scaling': 'strong' if args.global_batch else 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
                    'config': {'workload': "S1 synthetic room standing in for the 'bench' scene: %d frames 640x480 trained at factor 2, " % args.frames +
                                           'DINO-like 64-d features, hg+freq L=16 T=2^19, 128+128 samples/ray',
                               'rays_per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'dp{world}', 'dp_leg': best[0] if best else None},
                    'n_ranks_seen': n_ranks_seen, 'dp_legs': dp_legs, 'watchdog': why}
            if best:
                line.update(value=B * world * args.steps / best[1][0], ms_per_step=1000 * best[1][0] / args.steps)
            else:
                line['error'] = 'no data-parallel leg finished'
            return line

        import threading
        line_lock = threading.Lock()     # ONE JSON line: the watchdog's fallback and the normal print exclude each other

        def watchdog(seconds, why):
            def fire():
                # a hang is a failure for the launcher even when a finished leg gives the line a value (exit code 3; 1 = no leg at all)
                line_lock.acquire()
                if rank == 0:
                    print(json.dumps(fallback_line(why)), flush=True)
                os._exit(3 if best else 1)
            t = threading.Timer(seconds, fire)
            t.daemon = True
            t.start()
            return t
        for name, overlap, graph, shard, lg in plan:
            ok, leg = 1, None
            info = {'overlap_comm': overlap, 'hip_graph': graph, 'shard_optimizer': shard, 'levels_per_bucket': lg}
            dp_legs[name] = dict(info, error='did not finish (watchdog)')
            guard = watchdog(args.dp_leg_timeout, f'leg {name} did not finish within {args.dp_leg_timeout} s')
            try:
                e = wl.engine(SEEDS[0][0], world=world, pg=pg, overlap_comm=overlap, shard_optimizer=shard, level_group=lg)
                leg = timed_leg(wl, a2, e, B, dseed, mseed, frange, graph, world, sync)
                dp_legs[name] = dict(info, value=B * world * args.steps / leg[0], unit='rays/s', ms_per_step=1000 * leg[0] / args.steps)
            except Exception as ex:   # recorded, not fatal
                ok = 0
                dp_legs[name] = dict(info, error=f'{type(ex).__name__}: {ex}'[:300])
            try:
                flag = torch.tensor([ok], device=device, dtype=torch.int32)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                ok = int(flag.item())
            except Exception as ex:
                dp_legs[name].setdefault('error', f'agreement all-reduce failed: {ex}'[:200])
                ok = 0
            guard.cancel()
            dp_legs[name]['finished_on_every_rank'] = bool(ok)
            if ok and leg is not None and (best is None or leg[0] < best[1][0]):
                best = (name, leg, e)
            elif leg is not None:
                del e
                torch.cuda.empty_cache()
        if best is None:
            if rank == 0:
                print(json.dumps({'metric': 'train rays/sec (640x480 synthetic RGB-D scene, hg+freq, DINO-like features)', 'value': None,
                                  'unit': 'rays/s', 'n_gpus': world, 'error': 'no data-parallel leg finished', 'dp_legs': dp_legs}), flush=True)
            sys.exit(1)
        (dp_best, (dt, terms, events, step, done), eng) = best
        use_graph = dp_legs[dp_best]['hip_graph']
        tail_guard = watchdog(args.dp_tail_timeout, f'the run after the legs (quality steps, held-out render) did not finish within {args.dp_tail_timeout} s')
    rays_per_s = B * world * args.steps / dt
    live_rows = float(eng.ws.get('n_live', (1,), torch.int32).item())   # color-head rows of the last step (w > 1e-4)
    roof = scatter_roofline(eng, events, B * (eng.S1 + eng.S2), args.event_steps) if events else None
    roof_mlp = mlp_roofline(eng, events, args.event_steps, live_rows) if events else None

    # render throughput of the dense path + roofline of its dominant kernel pair (level-phased hash-grid gather)
    render_dense, render_roof = None, None
    if rank == 0 and args.render_frames > 0:
        render_dense, frame = wl.render_throughput(eng, False)
        render_roof = render_roofline(eng, frame, 'dense render: 512 rows per ray')

    # quality of the trained state: continue to --quality-steps optimizer steps, held-out metrics, then the other seed sets
    quality = None
    trained = None
    if args.quality_steps > 0:
        train_schedule(eng, step, done, args.quality_steps)
        sync()
        if world == 1 and use_graph and args.trained_steps > 0:
            # the metric is "rays/s at fixed PSNR / mIoU": the same step timed again AT the trained state the quality numbers describe
            # (the density field has sharpened: fewer live colour rows, zero-weight records dropped by the scatter).  These replays train
            # on, like any others; the held-out metrics below are taken after them.
            t0 = time.time()
            for _ in range(args.trained_steps):
                step()
            sync()
            dt_tr = time.time() - t0
            live_tr = float(eng.ws.get('n_live', (1,), torch.int32).item())
            trained = {'value': B * args.trained_steps / dt_tr, 'unit': 'rays/s', 'ms_per_step': 1000 * dt_tr / args.trained_steps, 'steps': args.trained_steps,
                       'timed_state': 'trained: steps %d..%d (after the quality schedule)' % (done[0] - args.trained_steps, done[0]),
                       'live_row_fraction': live_tr / float(B * (eng.S1 + eng.S2))}
        if rank == 0:
            q = heldout_metrics(wl.renderer(eng, False), wl.test, wl.n_classes)
            q.update(steps=done[0], adam_steps_applied=int(eng.state_i[0].item()))
            quality = quality_runs(wl, args, q, march=False) if world == 1 else dict(q, note='one run (data-parallel launch)')

    if rank == 0:
        L = eng.L
        res = {
            'metric': 'train rays/sec (640x480 synthetic RGB-D scene, hg+freq, DINO-like features)', 'value': rays_per_s,
            'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1000 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'strong' if args.global_batch else 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            # the timed steps start from a fresh initialisation: the density field has not sharpened yet, so most sample rows are
            # still live for the colour head (a trained field prunes them: the colour head's work shrinks, nothing else changes)
            'timed_state': 'untrained: steps %d..%d after initialisation' % (args.warmup, args.warmup + args.steps),
            'live_row_fraction': live_rows / float(B * (eng.S1 + eng.S2)),
            'config': {'workload': "S1 synthetic room standing in for the 'bench' scene: %d frames 640x480 trained at factor 2, " % args.frames +
                                   'DINO-like 64-d features, hg+freq L=16 T=2^19, 128+128 samples/ray',
                       'rays_per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'dp{world}', 'feature_dim': args.feature_dim,
                       'hip_graph': bool(use_graph), 'dp_leg': dp_best if world > 1 else None,
                       'gradient_exchange_bytes_per_rank_and_step': wire_bytes(L.n_grid, L.n_total, eng.grad_payload) if world > 1 else 0},
            'n_ranks_seen': n_ranks_seen, 'dp_legs': dp_legs,
            'loss_terms_last_timed_step': dict(zip(('rgb', 'depth', 'feature', 'semantic', 'total'), terms)),
            'quality': quality,
            'trained': trained,
        }
        if roof:
            res['roofline'] = roof
        if roof_mlp:
            res['roofline_mlp'] = roof_mlp
        if render_roof:
            res['roofline_render'] = render_roof
        del eng
        torch.cuda.empty_cache()
        if world == 1:
            if not args.no_march:
                res['marching'] = marching_leg(wl, args, B)
            # the reported render rate is the faster, better-scoring path (through the occupancy grid); the dense 512-step render beside it
            res['render_Mrays_per_s'] = (res.get('marching') or {}).get('render_Mrays_per_s') or render_dense
            res['render_dense_Mrays_per_s'] = render_dense
            if not args.no_lseg and args.feature_dim != 512:
                res['lseg'] = lseg_leg(wl, args, B)
            if not args.no_dp1:
                res['dp_world1'] = dp_world1_in_child(args, B)
            if not args.no_dropin:
                try:
                    res['dropin'] = dropin_leg(wl, args, B)
                except Exception as e:
                    res['dropin'] = {'error': repr(e)[:300]}
            if roof:
                if pmc:
                    roof['traffic'] = pmc.pop('traffic')
                    roof['traffic_detail'] = pmc
                else:
                    roof['traffic_detail'] = {'unavailable': pmc_why}
            if not args.no_cpu_baseline:
                res['cpu_baseline'] = cpu_baseline(wl, args)
        else:
            res['render_Mrays_per_s'] = res['render_dense_Mrays_per_s'] = render_dense
        if world > 1:
            tail_guard.cancel()
            line_lock.acquire()    # (never released: a watchdog already past its cancel point must not print a second line)
        print(json.dumps(res), flush=True)
    if world > 1:
        tail_guard.cancel()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
