"""Spread of tests/test_gpu_api.py::test_training_reduces_loss_on_cube_scene (300 dense steps on the cube, PSNR of training view 0 and
loss ratios) over repeated runs."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_api import make_model
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
from autolabel_amd.engine import TrainEngine
scene = synthetic.make_cube_scene()
frames = DeviceFrames.from_scene(scene, 'cuda')
vals = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    model = make_model(D=64, C_=3, bound=6.0, grid_scale=1.0)
    eng = TrainEngine(model._ensure_device(), num_steps=64, upsample_steps=64)
    batch = frames.alloc_batch(2048)
    first = last = None
    for i in range(300):
        frames.next_train(batch, seed=1, step=i)
        eng.step(batch, seed=2, step=i)
        if i == 0: first = eng.terms.tolist()
    last = eng.terms.tolist()
    t = frames.get_test(0)
    with torch.inference_mode():
        out = model.eval().render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False, num_steps=128, upsample_steps=0)
    psnr = -10 * math.log10(((out['image'] - t['pixels']) ** 2).mean().item())
    vals.append((round(psnr, 2), round(last[0] / first[0], 3), round(last[1] / first[1], 3)))
print(vals)
