import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_api import make_model, _trainer
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames, DeviceLoader
scene = synthetic.make_cube_scene(n_frames=8)
def run(graph, n, twice=False):
    os.environ['ALN_GRAPH'] = graph
    torch.manual_seed(0)
    model = make_model(D=64, C_=scene['n_classes'], bound=6.0, grid_scale=1.0)
    tr = _trainer(model, True)
    tr.opt.feature_loss = False
    loader = DeviceLoader(DeviceFrames.from_scene(scene, 'cuda'), 1024, 1000, seed=3)
    tr.train_iterations(loader, n)
    if twice:
        tr.train_iterations(loader, n)
    torch.cuda.synchronize()
    L = model._layout
    return model._P.flat.detach().cpu().clone(), L, tr.engine.terms.cpu().clone(), tr.engine.state_i.cpu().clone(), tr.engine.state_f.cpu().clone()
for n, twice in ((1, False), (2, False), (3, False), (6, False), (3, True)):
    a, L, ta, sa, fa = run('1', n, twice); b, _, tb, sb, fb = run('0', n, twice); c, _, tc, sc, fc = run('0', n, twice)
    print(f'--- n={n} twice={twice}  loss graph {ta[4]:.5f} eager {tb[4]:.5f} eager2 {tc[4]:.5f}  state_i {sa[:6].tolist()} {sb[:6].tolist()} scale {fa[0].item()} {fb[0].item()}')
    for k in ['sigma', 'color', 'semf', 'semo']:
        o = L.offsets[k]; e = o + L.nets[k].n_params
        print(f'   {k}: graph-eager max {(a[o:e]-b[o:e]).abs().max():.2e} frac>2e-3 {((a[o:e]-b[o:e]).abs()>2e-3).float().mean():.4f} | eager-eager max {(c[o:e]-b[o:e]).abs().max():.2e} frac {((c[o:e]-b[o:e]).abs()>2e-3).float().mean():.4f}')
    print(f'   grid: graph-eager frac>2e-3 {((a[:L.n_grid]-b[:L.n_grid]).abs()>2e-3).float().mean():.5f} | eager-eager {((c[:L.n_grid]-b[:L.n_grid]).abs()>2e-3).float().mean():.5f}')
