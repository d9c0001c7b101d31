import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
from autolabel_amd.engine import TrainEngine
from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
dev = torch.device('cuda', 0)
scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
frames = DeviceFrames.from_scene(scene, dev)
for hd, hc in [(64, 64), (128, 64), (64, 128)]:
    try:
        layout = ModelLayout('hg+freq', 15, hd, hc, 64, 3, bound=3.0)
        P = Params(layout, dev); P.init_(seed=0)
        eng = TrainEngine(HipPipeline(layout, P), num_steps=32, upsample_steps=32)
        batch = frames.alloc_batch(1024)
        for i in range(5):
            frames.next_train(batch, seed=5, step=i); eng.step(batch, seed=7, step=i)
        print(hd, hc, 'OK', eng.terms[4].item())
    except Exception as e:
        print(hd, hc, type(e).__name__, str(e)[:160])
