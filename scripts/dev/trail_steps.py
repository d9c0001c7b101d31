"""Per-step checksums of parameters / moments / loss terms over a training run -- to find the first step at which two builds of the
library part (both are deterministic; a single early step is bit-identical between them).
usage: trail_steps.py <out.pt> <steps> [path of libautolabel_hip.so] [dump_at_step]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import autolabel_amd  # noqa: F401
from autolabel_amd import hip as H
if len(sys.argv) > 3 and sys.argv[3] != '-':
    H.LIB = os.path.abspath(sys.argv[3])
dump_at = int(sys.argv[4]) if len(sys.argv) > 4 else -1
from stress_determinism import snapshot
from autolabel_amd import synthetic
from autolabel_amd.dataset import DeviceFrames
from autolabel_amd.engine import TrainEngine
from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
dev = torch.device('cuda', 0)
scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
frames = DeviceFrames.from_scene(scene, dev)
layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=3.0)
P = Params(layout, dev); P.init_(seed=0)
eng = TrainEngine(HipPipeline(layout, P), num_steps=32, upsample_steps=32)
batch = frames.alloc_batch(1024)
n = int(sys.argv[2])
trail = torch.zeros(n, 8, dtype=torch.int64)
ck = lambda t: int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())
extra = {}
for i in range(n):
    frames.next_train(batch, seed=5, step=i)
    if i == dump_at:
        extra['before/flat'] = P.flat.cpu().clone(); extra['before/m'] = eng.m.cpu().clone(); extra['before/v'] = eng.v.cpu().clone()
        extra['before/state_f'] = eng.state_f.cpu().clone(); extra['before/state_i'] = eng.state_i.cpu().clone()
    out = eng.step(batch, seed=7, step=i)
    torch.cuda.synchronize()
    L = layout
    trail[i, 0] = ck(P.flat[:L.n_grid]); trail[i, 1] = ck(P.flat[L.n_grid:]); trail[i, 2] = ck(eng.m); trail[i, 3] = ck(eng.v)
    trail[i, 4] = ck(eng.terms[:8]); trail[i, 5] = ck(eng.state_f[:4]); trail[i, 6] = ck(eng.state_i[:8]); trail[i, 7] = ck(eng.ws.bufs['n_live'][1])
    if i == dump_at:
        for k, v in snapshot(eng, P, layout, out).items():
            extra['at/' + k] = v.cpu().clone()
        for name in ('color_in', 'color_out', 'T_row', 'delta_row', 'd_color_in', 'sem_dots', 'h1', 'h2', 'ch1', 'ch2', 'logits', 'feat', 'cidx_row', 'd_h0', 'd_semf_in'):
            t = eng.ws.bufs.get(name)
            if t is not None:
                extra['at/ws/' + name] = t[1].cpu().clone()
        extra['after/flat'] = P.flat.cpu().clone(); extra['after/state_f'] = eng.state_f.cpu().clone(); extra['after/state_i'] = eng.state_i.cpu().clone()
torch.save({'trail': trail, 'extra': extra}, sys.argv[1])
print('trail of', n, 'steps saved; loss terms last', eng.terms[:5].tolist(), 'state_f', eng.state_f[:3].tolist(), 'state_i', eng.state_i[:6].tolist())
