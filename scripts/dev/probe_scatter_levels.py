"""Per-level time of the binned hash-grid scatter (one launch pair per level) on a bench-like batch, depth-order walk on."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autolabel_amd import hip as H
from autolabel_amd.pipeline import ModelLayout
N, S1, S2 = 4096, 128, 128
S, M1, M = S1 + S2, N * S1, N * (S1 + S2)
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=3.0)
e = L.enc
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
zc = torch.rand(N, S1, generator=g).sort(dim=1)[0] * 5 + 0.2
zf = (torch.rand(N, 1, generator=g) * 4 + 0.5 + torch.randn(N, S2, generator=g) * 0.15).sort(dim=1)[0]   # fine samples cluster at a surface
z = torch.cat([zc.reshape(-1), zf.reshape(-1)]).cuda().contiguous()
perm = torch.cat([zc, zf], 1).argsort(dim=1, stable=True).to(torch.int16).cuda().contiguous()
d_enc = (torch.randn(M, 48, device='cuda') * 0.01).half()
grad = torch.zeros(L.n_grid + 8, device='cuda')
ws = torch.empty(int(H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(e), M)), dtype=torch.uint8, device='cuda')
def binned(lo, hi):
    H.call('aln_encode_bwd_binned', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, S2, H.ptr(perm), H.ptr(d_enc), H.ptr(grad), H.ptr(ws),
           lo, hi, None, None, H.stream())
def timeit(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tile = int(H.lib().aln_encode_bwd_binned_tile_rows())
nt = (M + tile - 1) // tile
desc = ws[16 * nt * tile * 8 * 8:].view(torch.int32).view(16, 64, nt)
binned(0, 16); torch.cuda.synchronize()
cnt = ((desc >> 13) & 0x3FFF).sum(dim=(1, 2)).tolist()
print('all 16 levels: %.0f us;  records %.1f M of %.1f M undeduped' % (timeit(lambda: binned(0, 16)), sum(cnt) / 1e6, M * 8 * 16 / 1e6))
print(' '.join(f'L{l}:{timeit(lambda: binned(l, l + 1)):.0f}us/{cnt[l] / 1e6:.2f}M' for l in range(16)))
print(' '.join(f'L{lo}-{hi}:{timeit(lambda: binned(lo, hi)):.0f}us' for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16))))
