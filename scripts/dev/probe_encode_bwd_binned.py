"""Binned (atomic-free) vs fp32-atomic hash-grid scatter on a bench-like batch: total, per level, per phase."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autolabel_amd import hip as H
from autolabel_amd.pipeline import ModelLayout
N, S1, S2 = 4096, 128, 128
M1, M = N * S1, N * (S1 + S2)
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=3.0)
e = L.enc
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
z = torch.cat([(torch.rand(N, S1, generator=g).sort(dim=1)[0] * 5 + 0.2).reshape(-1),
               (torch.rand(N, S2, generator=g).sort(dim=1)[0] * 5 + 0.2).reshape(-1)]).cuda().contiguous()
d_enc = (torch.randn(M, 48, device='cuda') * 0.01).half()
grad = torch.zeros(L.n_grid + 8, device='cuda')
ws = torch.empty(int(H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(e), M)), dtype=torch.uint8, device='cuda')
print(f'rows {M}, workspace {ws.numel() / 2**20:.0f} MiB')
def atomic(lo, hi):
    H.call('aln_encode_bwd_levels', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M1, S1, H.ptr(d_enc), H.ptr(grad), lo, hi, H.stream())
    H.call('aln_encode_bwd_levels', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z[M1:]), None, M - M1, S2, H.ptr(d_enc[M1:]), H.ptr(grad), lo, hi, H.stream())
def binned(lo, hi):
    H.call('aln_encode_bwd_binned', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, S2, None, H.ptr(d_enc), H.ptr(grad), H.ptr(ws),
           lo, hi, None, None, H.stream())
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('all 16 levels, both passes: atomic %.0f us   binned %.0f us' % (timeit(lambda: atomic(0, 16)), timeit(lambda: binned(0, 16))))
for l in range(16):
    print(f'level {l:2d} (res {int(e.grid.res[l]):6d}, {"dense" if e.grid.dense[l] else "hash "}): atomic {timeit(lambda: atomic(l, l + 1)):6.0f} us   '
          f'binned {timeit(lambda: binned(l, l + 1)):6.0f} us')
for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16)):
    print(f'levels {lo}-{hi}: atomic {timeit(lambda: atomic(lo, hi)):.0f} us   binned {timeit(lambda: binned(lo, hi)):.0f} us')
# records actually written (descriptor counts)
tile = int(H.lib().aln_encode_bwd_binned_tile_rows())
nt = (M + tile - 1) // tile
desc = ws[16 * nt * tile * 8 * 8:].view(torch.int32).view(16, 64, nt)
binned(0, 16); torch.cuda.synchronize()
cnt = ((desc >> 13) & 0x3FFF).sum(dim=(1, 2)).tolist()
print('records per level:', cnt, ' total %.1f M of %.1f M undeduped' % (sum(cnt) / 1e6, M * 8 * 16 / 1e6))
