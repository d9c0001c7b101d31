"""Spread of the 400-step marching fit of tests/test_gpu_march.py::test_training_through_marched_samples_converges (PSNR of training
view 0, occupancy) over repeated runs, eager and graph-replayed."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_march import _model, _train
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
scene = synthetic.make_cube_scene()
frames = DeviceFrames.from_scene(scene, 'cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for graph in (False, True):
    vals = []
    for trial in range(n):
        torch.manual_seed(0)
        model = _model(True, scene['n_classes'], 6.0, grid_size=64, max_steps=512, march_samples=96, density_thresh=10.0)
        eng = _train(model, frames, 400, graph=graph)
        t = frames.get_test(0)
        with torch.inference_mode():
            out = model.eval().render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False)
        psnr = -10 * math.log10(((out['image'] - t['pixels'].view_as(out['image'])) ** 2).mean().item())
        vals.append((round(psnr, 2), round(model._pipe.occ.occupancy(), 4), int(eng.state_i[0].item())))
    print('graph' if graph else 'eager', vals, flush=True)
