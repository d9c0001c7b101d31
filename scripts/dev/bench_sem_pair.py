"""Dev only: backward of both semantic heads (aln_sem_heads_bwd, fold_geo = 1) alone, at the bench size: time per call.
  python scripts/dev/bench_sem_pair.py [--lib PATH]"""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument('--lib', default=None); ap.add_argument('--reps', type=int, default=20); ap.add_argument('--dots', action='store_true')
a = ap.parse_args()
import torch
from autolabel_amd import hip as H
if a.lib:
    H.LIB = a.lib
from autolabel_amd.pipeline import ModelLayout, Params
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=6.0)
P = Params(L, 'cuda'); P.init_(0)
N, S1, S2, D, G, Ccls = 4096, 128, 128, 64, 15, 7
rows = N * (S1 + S2)
sigma_out = (torch.randn(rows, 16, device='cuda') * 0.5).half()
feat = torch.zeros(rows, D, device='cuda', dtype=torch.float16)
logits = torch.zeros(rows, L.Cpad, device='cuda', dtype=torch.float16)
H.call('aln_sem_heads_fwd', C.byref(P.descs['semf']), C.byref(P.descs['semo']), H.ptr(sigma_out), rows, D, G, H.ptr(feat), H.ptr(logits), H.stream())
w_row = torch.rand(rows, device='cuda') * 0.05
g_sem, g_feat = torch.randn(N, Ccls, device='cuda') * 0.1, torch.randn(N, D, device='cuda') * 0.1
d_oin = torch.empty(rows, L.nets['semo'].in_pad, device='cuda', dtype=torch.float16)
d_fin = torch.empty(rows, 16, device='cuda', dtype=torch.float16)
dots = torch.zeros(rows, device='cuda'); grad = torch.zeros_like(P.grad); flag = torch.zeros(1, dtype=torch.int32, device='cuda')
gpf, gpo = (C.c_void_p(grad.data_ptr() + 4 * L.offsets[k]) for k in ('semf', 'semo'))
def run():
    H.call('aln_sem_heads_bwd', C.byref(P.descs['semf']), C.byref(P.descs['semo']), H.ptr(sigma_out), H.ptr(feat), H.ptr(w_row), H.ptr(g_sem),
           H.ptr(g_feat), N, S1, S2, Ccls, rows, D, G, H.ptr(d_oin), H.ptr(d_fin), gpf, gpo, 1, H.ptr(dots) if a.dots else None, H.ptr(flag), H.stream())
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(a.reps):
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
busy = 98 * 32 * (rows / 32) / 1024 / 2400.0
print(f'semantic pair backward, {rows} rows: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us  (MFMA pipe time {busy:.1f} us -> {100*busy/ts[len(ts)//2]:.1f} % busy at 2.4 GHz)  flag {flag.item()}  slabs {H.lib().aln_sem_heads_bwd_slabs(C.byref(P.descs["semf"]), C.byref(P.descs["semo"]), rows, D, G)}')

try:
    lib = H.lib(); lib.aln_debug_read_pair.argtypes = [C.c_void_p, C.c_int]
    lib.aln_debug_read_pair(None, 1)
    for _ in range(5): run()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 32)(); lib.aln_debug_read_pair(buf, 0)
    tiles = (rows // 64 + 255) // 256 * 5
    for w in range(4):
        v = [buf[w * 8 + i] / tiles for i in range(8)]
        print(f'wave {w}: work {v[0]:.0f}  wait {v[1]:.0f}  fb {v[2]:.0f} dW2 {v[3]:.0f} spin {v[4]:.0f}   total {sum(v):.0f} ticks per 64-row tile')
except AttributeError:
    pass
