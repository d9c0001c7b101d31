"""Dev probe (round 3): where does the run-to-run quality spread of the dense path come from?

Trains the bench workload (S1 room, B = 4096, 128 + 128 samples, lr 5e-3 halved at 60 % / 80 %) several times and prints the
held-out metrics: (a) the same seeds twice -> must now be bit-identical; (b) other sample-noise seeds; (c) other data seeds;
(d) other initialisations.  Usage: python scripts/dev/variance.py [--steps 1500] [--march]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import torch

from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
from autolabel_amd.engine import TrainEngine
from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
from autolabel_amd.quality import heldout_metrics, pipe_renderer, split_heldout

p = argparse.ArgumentParser()
p.add_argument('--steps', type=int, default=1500)
p.add_argument('--frames', type=int, default=200)
p.add_argument('--batch', type=int, default=4096)
p.add_argument('--march', action='store_true')
p.add_argument('--eps', type=float, default=1e-15)
p.add_argument('--runs', type=str, default='base,base,noise1,noise2,data1,data2,init1,init2')
p.add_argument('--out', type=str, default='gpurun_out/variance.json')
args = p.parse_args()
dev = torch.device('cuda', 0)
scene = synthetic.make_room_scene(n_frames=args.frames, seed=0, device=dev, feat_dim=64, feat_hw=(60, 80))
half = synthetic.subsample(scene, 2)
train_ids, held = split_heldout(args.frames)
pick = lambda sc, ids, sem: dict(sc, images=sc['images'][ids], depths=sc['depths'][ids], semantics=sc[sem][ids],
                                 features=sc['features'][ids] if sem == 'semantics' else None, T_CW=sc['T_CW'][ids])
train = DeviceFrames.from_scene(pick(half, train_ids, 'semantics'), dev)
test = DeviceFrames.from_scene(pick(half, held, 'semantics_full'), dev)
lo, hi = scene['min_bounds'], scene['max_bounds']
bound = float(((hi - lo) - (lo + hi) * 0.5).max())


def run(init_seed, data_seed, noise_seed):
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=bound)
    P = Params(layout, dev)
    P.init_(seed=init_seed)
    pipe = HipPipeline(layout, P)
    if args.march:
        pipe.enable_marching(G=128, max_steps=1024, samples=64, density_thresh=10.0)
        pipe.mark_untrained_grid(train.world_to_camera(), (train.desc.fx, train.desc.fy, train.desc.cx, train.desc.cy), size=(train.w, train.h))
    eng = TrainEngine(pipe, feature_loss=True, eps=args.eps)
    batch = train.alloc_batch(args.batch)
    g = eng.graphed(train, batch, data_seed, noise_seed, warmup=3)
    n = g.steps
    t0 = time.time()
    for frac, lr in ((0.6, 5e-3), (0.8, 2.5e-3), (1.0, 1.25e-3)):
        eng.lr = lr
        while n < int(frac * args.steps):
            g(); n += 1
    torch.cuda.synchronize()
    dt = time.time() - t0
    if args.march:
        def render(ro, rd, dn):
            parts = []
            ro, rd, dn = ro.reshape(-1, 3), rd.reshape(-1, 3), dn.reshape(-1)
            for a in range(0, ro.shape[0], 16384):
                out, _ = pipe.forward(ro[a:a + 16384].contiguous(), rd[a:a + 16384].contiguous(), dn[a:a + 16384].contiguous(), 128, 0, False,
                                      train=False, march=True)
                parts.append({k: out[k].clone() for k in ('image', 'depth', 'semantic')})
            return {k: torch.cat([q[k] for q in parts]) for k in parts[0]}
    else:
        render = pipe_renderer(pipe)
    q = heldout_metrics(render, test, scene['n_classes'])
    q.update(steps=n, skipped=n - int(eng.state_i[0].item()), loss=eng.terms.tolist(), scale=float(eng.state_f[0].item()), train_s=dt,
             checksum=float(P.flat.double().sum().item()))
    return q


SEEDS = {'base': (0, 1234, 99), 'noise1': (0, 1234, 100), 'noise2': (0, 1234, 101), 'data1': (0, 1235, 99), 'data2': (0, 1236, 99),
         'init1': (1, 1234, 99), 'init2': (2, 1234, 99)}
res = []
for name in args.runs.split(','):
    q = run(*SEEDS[name])
    res.append(dict(q, run=name))
    print(name, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in q.items() if k != 'loss'}), flush=True)
os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
with open(os.path.join(ROOT, args.out), 'w') as f:
    json.dump({'args': vars(args), 'runs': res}, f, indent=1)
