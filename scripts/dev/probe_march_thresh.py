import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from test_gpu_march import _model, _train
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
from autolabel_amd.engine import TrainEngine
scene = synthetic.make_cube_scene(n_frames=8)
frames = DeviceFrames.from_scene(scene, 'cuda')
for mode in ('dense', 'march'):
    torch.manual_seed(0)
    model = _model(True, scene['n_classes'], 6.0, grid_size=64, max_steps=256, march_samples=256, density_thresh=10.0)
    pipe = model._ensure_device()
    occ = pipe.occ
    if mode == 'dense':
        pipe.occ = None
        eng = _train(model, frames, 300)
        pipe.occ = occ
    else:
        eng = _train(model, frames, 300)
    occ.grid.zero_()
    occ.decay = 1.0
    for k in range(6):
        pipe.update_density_grid(step=1000 + k)
    g = occ.grid.cpu().numpy()
    print(mode, 'grid quantiles', np.quantile(g, [0.01, 0.1, 0.5, 0.9, 0.99, 0.999]).round(4), 'mean', g.mean(), 'loss', eng.terms.tolist())
    t = frames.get_test(2)
    with torch.inference_mode():
        occ.bits.fill_(-1)
        dense = model.eval().render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False)
        for th in (0.01, 0.1, 1.0, 10.0):
            occ.density_thresh = th
            pipe.refresh_bitfield()
            m = model.render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False)
            psnr = -10 * math.log10(((m['image'] - t['pixels'].view_as(m['image'])) ** 2).mean().item())
            print(f'   thresh {th}: occupancy {occ.occupancy():.4f} max|diff| {(m["image"]-dense["image"]).abs().max().item():.4f} mean {(m["image"]-dense["image"]).abs().mean().item():.5f} psnr {psnr:.2f}')
