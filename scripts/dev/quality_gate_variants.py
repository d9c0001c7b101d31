"""Where does the HIP-vs-fp32-oracle quality gap of tests/test_gpu_quality.py come from?  Same protocol, four trainees:
fp32 oracle, half_sim oracle (fp16 rounding points of the kernels simulated), HIP with the binned scatter, HIP with fp32 atomics."""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import nerf_oracle as O
from test_gpu_quality import _Frames, STEPS, B, S1, S2
from test_gpu_pipeline import build_pair
from autolabel_amd import synthetic
from autolabel_amd.dataset import ArrayDataset
from autolabel_amd.engine import TrainEngine
from autolabel_amd.quality import heldout_metrics, pipe_renderer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else STEPS
torch.set_num_threads(min(32, max(1, len(os.sched_getaffinity(0)))))
scene = synthetic.make_room_scene(n_frames=30, w=64, h=48, fx=32.0, fy=32.0, cx=31.5, cy=23.5, feat_dim=16, feat_hw=(6, 8), labelled_every=2)
held = [4, 11, 18, 25]; train_ids = [i for i in range(30) if i not in held]
cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in scene.items()}
tr = dict(cpu, images=cpu['images'][train_ids], depths=cpu['depths'][train_ids], semantics=cpu['semantics'][train_ids], features=cpu['features'][train_ids], T_CW=cpu['T_CW'][train_ids])
te = dict(cpu, images=cpu['images'][held], depths=cpu['depths'][held], semantics=cpu['semantics_full'][held], features=None, T_CW=cpu['T_CW'][held])
ds, ds_test = ArrayDataset(tr, batch_size=B), ArrayDataset(te, batch_size=B, split='test')
bound = float(((ds.max_bounds - ds.min_bounds) - (ds.min_bounds + ds.max_bounds) * 0.5).max())
C_ = scene['n_classes']
mk = lambda: build_pair(L=8, D=64, C_=C_, bound=bound, grid_scale=1.0, log2_T=14)
o32, p_bin, cfg = mk(); o16, p_atm, _ = mk()
o32.half_sim = False; o16.half_sim = True
p_atm.binned_bwd = False
engs = [TrainEngine(p, num_steps=S1, upsample_steps=S2, feature_loss=True) for p in (p_bin, p_atm)]
oracles = [o32, o16]
sts = [{k: [torch.zeros_like(v), torch.zeros_like(v), 0] for k, v in o.params.items()} for o in oracles]
np.random.seed(0); random.seed(0)
for it in range(steps):
    lr = 5e-3 * 0.5 ** (it // 250)
    b = ds._next_train()
    bt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
    dev = {k: v.cuda().float().contiguous() for k, v in bt.items() if k != 'semantic'}
    dev['semantic'] = bt['semantic'].int().cuda()
    for e in engs:
        e.lr = lr; e.step(dev, seed=7, step=it)
    noise = torch.from_numpy(O.rand_uniform(7, O.STREAM_PERTURB, it, np.arange(B * S1))).view(B, S1)
    u = torch.from_numpy(O.rand_uniform(7, O.STREAM_PDF, it, np.arange(B * S2))).view(B, S2)
    for o, st in zip(oracles, sts):
        out = o.run(bt['rays_o'], bt['rays_d'], bt['direction_norms'], S1, S2, perturb=True, noise_coarse=noise, u_fine=u)
        loss, _ = O.loss_fn(out, {'pixels': bt['pixels'], 'depth': bt['depth'], 'semantic': bt['semantic'], 'features': bt['features'].float()}, feature_loss=True)
        for p in o.params.values(): p.grad = None
        loss.backward()
        with torch.no_grad():
            for k, p in o.params.items():
                if p.grad is None: continue
                st[k][2] += 1
                O.adam_update(p, p.grad, st[k][0], st[k][1], st[k][2], lr, weight_decay=0.0 if k == 'grid' else 1e-6)
res = {}
fr = lambda dv: _Frames(ds_test, list(range(len(held))), dv)
for name, p in (('hip_binned', p_bin), ('hip_atomic', p_atm)):
    res[name] = heldout_metrics(pipe_renderer(p, num_steps=64, upsample_steps=0), fr('cuda'), C_)
for name, o in (('oracle_fp32', o32), ('oracle_half_sim', o16)):
    with torch.no_grad():
        res[name] = heldout_metrics(lambda ro, rd, dn, o=o: o.run(ro, rd, dn.reshape(-1, 1), 64, 0, perturb=False), fr('cpu'), C_)
    # the HIP renderer on the oracle's weights (separates training quality from render arithmetic)
res['steps'] = steps
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'quality_gate_variants.json'), 'w'), indent=1)
