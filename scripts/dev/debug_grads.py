"""GPU debug: per-segment gradient error of the HIP backward vs oracle autograd."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_pipeline import build_pair, make_rays, _batch, hip_loss, flat_grads, rel
from oracle import nerf_oracle as O

terms = sys.argv[1] if len(sys.argv) > 1 else 'all'
oracle, pipe, cfg = build_pair(L=16, D=64, C_=3)
N, S1, S2, C_, D, Cf = 64, int(sys.argv[3]) if len(sys.argv) > 3 else 64, int(sys.argv[4]) if len(sys.argv) > 4 else 64, 3, 64, 48
o, d, norms = make_rays(N, seed=2)
g = torch.Generator().manual_seed(11)
noise, u = torch.rand(N, S1, generator=g), torch.rand(N, max(S2, 1), generator=g)
batch = _batch(N, C_, Cf, seed=4)
W = {'all': (1.0, 0.1, 1.0, 0.5), 'rgb': (1.0, 0, 0, 0), 'depth': (0, 1.0, 0, 0), 'sem': (0, 0, 1.0, 0), 'feat': (0, 0, 0, 1.0)}[terms]
want = oracle.run(o, d, norms, num_steps=S1, upsample_steps=S2, perturb=True, noise_coarse=noise, u_fine=u)
loss, _ = O.loss_fn(want, batch, rgb_weight=W[0], depth_weight=W[1], semantic_weight=W[2], feature_weight=W[3], feature_loss=True)
loss.backward()
gw = flat_grads(oracle, cfg)
od, dd, nd, nz, ud = o.cuda(), d.cuda(), norms.cuda().reshape(-1), noise.cuda(), u.cuda()
out, ctx = pipe.forward(od, dd, nd, S1, S2, True, train=True, noise=nz, u=ud if S2 else None)
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1024.0
gi, gd, gs, gf, t = hip_loss(pipe, out, batch, N, C_, D, Cf, scale=scale, weights=W)
pipe.P.grad.zero_()
pipe.backward(ctx, gi, gd, gs, gf)
torch.cuda.synchronize()
got = pipe.P.grad[:pipe.L.n_total].cpu() / scale
L = pipe.L
print('terms', terms, 'loss', t.tolist(), loss.item(), 'found_inf', pipe.found_inf.item())
print('z max diff', (torch.sort(ctx['z'].cpu()[N*S1:].view(N,-1),1)[0] - 0).shape)
print('mask agreement', (ctx['w_row'].cpu() > 1e-4).sum().item(), want['_mask'].sum().item())
print('grid rel', rel(got[:L.n_grid], gw[:L.n_grid]))
lv = cfg.grid.levels()
for l in lv[::5]:
    a, b = 2 * l['offset'], 2 * (l['offset'] + l['size'])
    print('  level', lv.index(l), 'rel', round(rel(got[a:b], gw[a:b]), 5), 'norm', gw[a:b].norm().item())
for k in ['sigma', 'color', 'semf', 'semo']:
    a = L.offsets[k]
    for i, (o_, i_) in enumerate(L.nets[k].shapes):
        b = a + o_ * i_
        print(k, i, 'rel', round(rel(got[a:b], gw[a:b]), 5), 'norm', gw[a:b].norm().item())
        a = b
