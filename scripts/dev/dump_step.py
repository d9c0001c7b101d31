"""One training step's gradients, outputs and intermediates written to a file -- to compare two builds of the library bit for bit.
usage: dump_step.py <out.pt> [path of libautolabel_hip.so]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import autolabel_amd  # noqa: F401
from autolabel_amd import hip as H
if len(sys.argv) > 2:
    H.LIB = os.path.abspath(sys.argv[2])
from stress_determinism import build, snapshot
eng, P, layout, batch, frames = build(1024, 32)
P.grad.zero_()
out = eng.forward_backward(batch, seed=7, step=0)
torch.cuda.synchronize()
snap = {k: v.cpu() for k, v in snapshot(eng, P, layout, out).items()}
for name in ('color_in', 'color_out', 'T_row', 'delta_row', 'd_color_in', 'sem_dots'):
    t = eng.ws.bufs.get(name)
    if t is not None:
        snap['ws/' + name] = t[1].cpu()
# ... and the state after complete steps (optimizer included: the table's Adam inside the scatter, k_adam for the heads)
from autolabel_amd.engine import TrainEngine
eng2, P2, layout2, batch2, frames2 = build(1024, 32)
eng2.fuse_grid_adam = True
for i in range(3):
    frames2.next_train(batch2, seed=5, step=i)
    o2 = eng2.step(batch2, seed=7, step=i)
    torch.cuda.synchronize()
    snap[f'step{i}/flat'] = P2.flat.cpu().clone(); snap[f'step{i}/m'] = eng2.m.cpu().clone(); snap[f'step{i}/v'] = eng2.v.cpu().clone()
    snap[f'step{i}/table16'] = P2.table16.cpu().clone(); snap[f'step{i}/state_f'] = eng2.state_f.cpu().clone(); snap[f'step{i}/state_i'] = eng2.state_i.cpu().clone()
    snap[f'step{i}/terms'] = eng2.terms.cpu().clone()
    for k, v in batch2.items():
        if torch.is_tensor(v): snap[f'step{i}/batch/{k}'] = v.cpu().clone()
torch.save(snap, sys.argv[1])
print('saved', len(snap), 'tensors')
