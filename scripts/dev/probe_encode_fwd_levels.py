"""Is the hash-grid forward bound by L2 misses?  Time the forward one level at a time (2 MB table: L2-resident) against all 16."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autolabel_amd import hip as H
from autolabel_amd.pipeline import ModelLayout, Params
N, S = 4096, 128
rows = N * S
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=3.0)
P = Params(L, 'cuda'); P.init_(0)
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
z = (torch.rand(N, S, generator=g).sort(dim=1)[0] * 5 + 0.2).cuda().reshape(-1).contiguous()
enc = torch.empty(rows, 48, device='cuda', dtype=torch.float16)
def run(lo, hi):
    H.call('aln_dev_encode_fwd_levels', C.byref(L.enc), H.ptr(P.table16), H.ptr(ro), H.ptr(rd), H.ptr(z), None, rows, S, H.ptr(enc), lo, hi,
           H.stream())
def timeit(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('no levels (positions, frequency part, tile write): %.0f us' % timeit(lambda: run(0, 0)))
print('all 16 levels: %.0f us' % timeit(lambda: run(0, 16)))
tot = 0
for l in (0, 2, 3, 6, 9, 12, 15):
    t = timeit(lambda: run(l, l + 1)); print(f'level {l:2d} alone: {t:.0f} us')
for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16), (8, 10), (8, 16)):
    print(f'levels {lo}-{hi}: {timeit(lambda: run(lo, hi)):.0f} us')
