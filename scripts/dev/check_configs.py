import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
from autolabel_amd.engine import TrainEngine
from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
dev = torch.device('cuda', 0)
scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
frames = DeviceFrames.from_scene(scene, dev)
for (G, C_, D, enc) in [(15, 20, 64, 'hg+freq'), (15, 40, 64, 'hg'), (15, 2, 128, 'hg+freq'), (15, 3, 64, 'freq'), (15, 100, 64, 'hg+freq')]:
    layout = ModelLayout(enc, G, 128, 128, D, C_, bound=3.0)
    P = Params(layout, dev); P.init_(seed=0)
    eng = TrainEngine(HipPipeline(layout, P), num_steps=32, upsample_steps=32)
    batch = frames.alloc_batch(1024)
    batch['semantic'].clamp_(max=C_ - 1)
    losses = []
    for i in range(30):
        frames.next_train(batch, seed=5, step=i)
        batch['semantic'].clamp_(max=C_ - 1)
        eng.step(batch, seed=7, step=i)
        losses.append(eng.terms[4].item())
    print(G, C_, D, enc, 'loss %.3f -> %.3f' % (sum(losses[:3]) / 3, sum(losses[-3:]) / 3), 'finite', bool(torch.isfinite(P.flat).all()), 'steps', int(eng.state_i[0]))
