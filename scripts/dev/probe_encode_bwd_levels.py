"""Per-level cost of the hash-grid scatter (aln_encode_bwd_levels one level at a time) on a bench-like batch."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autolabel_amd import hip as H
from autolabel_amd.pipeline import ModelLayout
N, S = 4096, 128
rows = N * S
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=3.0)
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
z = (torch.rand(N, S, generator=g).sort(dim=1)[0] * 5 + 0.2).cuda().reshape(-1).contiguous()
d_enc = (torch.randn(rows, 48, device='cuda') * 0.01).half()
grad = torch.zeros(L.n_grid + 8, device='cuda')
def run(lo, hi):
    H.call('aln_encode_bwd_levels', C.byref(L.enc), H.ptr(ro), H.ptr(rd), H.ptr(z), None, rows, S, H.ptr(d_enc), H.ptr(grad), lo, hi, H.stream())
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('all 16 levels: %.0f us' % timeit(lambda: run(0, 16)))
tot = 0
for l in range(16):
    t = timeit(lambda: run(l, l + 1)); tot += t
    print(f'level {l:2d} (res {int(L.enc.grid.res[l]):6d}, {"dense" if L.enc.grid.dense[l] else "hash "}): {t:6.0f} us')
print('sum of single-level launches: %.0f us' % tot)
for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16)):
    print(f'levels {lo}-{hi}: {timeit(lambda: run(lo, hi)):.0f} us')
