"""400 eager marching steps on the bench scene (for rocprofv3 --kernel-trace): the per-kernel split of the cuda_ray=True step."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import autolabel_amd, torch, bench
from autolabel_amd.engine import TrainEngine
from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
class A: pass
args = A(); args.frames = 200; args.feature_dim = 64; args.render_frames = 0
dev = torch.device('cuda', 0)
scene, half, train, test, full, eng0, frange, bound = bench.build(args, dev, 0, 1)
layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=bound)
P = Params(layout, dev); P.init_(seed=0)
pipe = HipPipeline(layout, P)
pipe.enable_marching(G=128, max_steps=1024, samples=64, density_thresh=10.0)
pipe.mark_untrained_grid(train.world_to_camera(), (train.desc.fx, train.desc.fy, train.desc.cx, train.desc.cy), size=(train.w, train.h))
eng = TrainEngine(pipe, feature_loss=True)
batch = train.alloc_batch(4096)
for i in range(400):
    train.next_train(batch, seed=1, step=i)
    eng.step(batch, seed=2, step=i)
torch.cuda.synchronize(); print('done', eng.terms.tolist())
