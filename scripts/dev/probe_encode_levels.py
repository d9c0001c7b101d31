"""Does hash-grid gather time per level drop when only ONE level's 2 MB table is touched (L2-resident)?"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autolabel_amd import hip as H
N, S = 4096, 128
rows = N * S
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
z = (torch.rand(N, S, generator=g).sort(dim=1)[0] * 6 + 0.2).cuda().reshape(-1).contiguous()
def run(desc, name):
    e = H.make_enc_desc('hg', 6.0, desc)
    table = torch.randn(e.grid.n_entries * 2, device='cuda').half()
    out = torch.empty(rows, e.enc_pad, dtype=torch.float16, device='cuda')
    for _ in range(3):
        H.call('aln_encode_fwd', C.byref(e), H.ptr(table), H.ptr(ro), H.ptr(rd), H.ptr(z), None, rows, S, H.ptr(out), H.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        H.call('aln_encode_fwd', C.byref(e), H.ptr(table), H.ptr(ro), H.ptr(rd), H.ptr(z), None, rows, S, H.ptr(out), H.stream())
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print(f'{name}: {t:.1f} us total, {t / e.grid.n_levels:.1f} us per level, table {e.grid.n_entries * 4 / 1e6:.1f} MB')
run(H.make_grid_desc(), '16 levels (28.5 MB table)')
for lvl in [15, 8, 3]:
    run(H.make_grid_desc(n_levels=1, base_resolution=16 * 2 ** lvl), f'1 level = level {lvl}')
run(H.make_grid_desc(n_levels=4, base_resolution=16 * 2 ** 12), '4 levels = levels 12-15')
