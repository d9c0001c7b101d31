"""Is a render bit-reproducible (a) twice through the same model, (b) through a second model loaded from the state dict?
(tests/test_gpu_march.py::test_training_through_marched_samples_converges compares (b) bit for bit and failed once in 5 runs.)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_march import _model, _train
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
scene = synthetic.make_cube_scene()
frames = DeviceFrames.from_scene(scene, 'cuda')
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    torch.manual_seed(0)
    model = _model(True, scene['n_classes'], 6.0, grid_size=64, max_steps=512, march_samples=96, density_thresh=10.0)
    _train(model, frames, 400)
    t = frames.get_test(0)
    def render(m):
        with torch.inference_mode():
            return {k: v.clone() for k, v in m.eval().render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False).items()}
    a, b = render(model), render(model)
    sd = model.state_dict()
    if os.environ.get('POISON'):   # hand the next allocations NaN-filled (or 0x7f-filled) memory: any read of an unwritten buffer shows
        xs = [torch.full((64 << 20,), float('nan'), device='cuda') for _ in range(24)]
        del xs
    m2 = _model(True, scene['n_classes'], 6.0, grid_size=64, max_steps=512, march_samples=96, density_thresh=10.0)
    m2.load_state_dict(sd)
    c, d = render(m2), render(m2)
    def diff(x, y):
        return {k: (float((x[k].float() - y[k].float()).abs().max()), int((x[k] != y[k]).sum())) for k in x if torch.is_tensor(x[k]) and x[k].shape == y[k].shape}
    print(trial, 'same model twice', {k: v for k, v in diff(a, b).items() if v[1]}, '| reloaded', {k: v for k, v in diff(a, c).items() if v[1]},
          '| reloaded twice', {k: v for k, v in diff(c, d).items() if v[1]}, flush=True)
