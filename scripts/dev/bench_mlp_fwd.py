"""Dev only: forward of a 128-wide head alone (aln_density_fwd / aln_mlp_fwd over plain rows): time per launch, checksum of the
outputs (compare two builds: --lib), error against an fp32 torch forward.
  python scripts/dev/bench_mlp_fwd.py [--lib PATH] [--head sigma|color] [--rows N]"""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default=None); ap.add_argument('--head', default='sigma'); ap.add_argument('--rows', type=int, default=1 << 19)
ap.add_argument('--reps', type=int, default=20)
a = ap.parse_args()
import torch
from autolabel_amd import hip as H
if a.lib:
    H.LIB = a.lib
from autolabel_amd.pipeline import ModelLayout, Params
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=6.0)
P = Params(L, 'cuda'); P.init_(0)
m = L.nets[a.head]
torch.manual_seed(3)


def run(x, out, sigma, rows, rows_dev=None):
    if a.head == 'sigma':
        H.call('aln_density_fwd', C.byref(P.descs['sigma']), H.ptr(x), rows, None, None, H.ptr(out), H.ptr(sigma), H.stream())
    else:
        H.call('aln_mlp_fwd', C.byref(P.descs[a.head]), H.ptr(x), rows, H.ptr(rows_dev), None, None, H.ptr(out), H.stream())


for rows in (1, 31, 33, 64, 65, 4096 + 17, 100000 - 37):
    x = (torch.randn(rows, m.in_pad, device='cuda') * 0.5).half()
    out = torch.full((rows + 64, m.out_pad), 7.0, device='cuda', dtype=torch.float16)
    sigma = torch.full((rows + 64,), 7.0, device='cuda')
    nd = torch.tensor([max(rows - 5, 1)], dtype=torch.int32, device='cuda') if a.head != 'sigma' else None
    run(x, out, sigma, rows, nd)
    torch.cuda.synchronize()
    live = rows if nd is None else int(nd.item())
    w = P.net_view(a.head)
    h, o = x[:live].float(), 0
    for li, (no, ni) in enumerate(m.shapes):
        W = w[o:o + no * ni].view(no, ni).half().float(); o += no * ni
        h = h @ W.t()
        if li < len(m.shapes) - 1:
            h = torch.relu(h).half().float()
    err = (out[:live].float() - h).abs().max().item() / (h.abs().max().item() + 1e-9)
    untouched = bool((out[live:] == 7.0).all()) and (a.head != 'sigma' or bool((sigma[live:] == 7.0).all()))
    sig_ok = a.head != 'sigma' or bool(torch.equal(sigma[:live], torch.exp(out[:live, 0].float())))
    cs = int(out[:live].view(torch.int16).to(torch.int64).sum().item())
    print(f'{a.head} rows {rows} (live {live}): rel err {err:.2e}  rows beyond untouched {untouched}  sigma = exp(h0) {sig_ok}  checksum {cs}')

rows = a.rows
x = (torch.randn(rows, m.in_pad, device='cuda') * 0.5).half()
out = torch.empty(rows, m.out_pad, device='cuda', dtype=torch.float16); sigma = torch.empty(rows, device='cuda')
nd = torch.tensor([rows], dtype=torch.int32, device='cuda') if a.head != 'sigma' else None
for _ in range(3): run(x, out, sigma, rows, nd)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(a.reps):
    e0.record(); run(x, out, sigma, rows, nd); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
busy = 48 * 32 * (rows / 32) / 1024 / 2400.0
print(f'{a.head}: {rows} rows: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us  (MFMA pipe time {busy:.1f} us -> {100*busy/ts[len(ts)//2]:.1f} % busy at 2.4 GHz)  checksum {int(out.view(torch.int16).to(torch.int64).sum().item())}')

try:
    lib = H.lib(); lib.aln_debug_read_fwd128.argtypes = [C.c_void_p, C.c_int]
    lib.aln_debug_read_fwd128(None, 1)
    for _ in range(5): run(x, out, sigma, rows, nd)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 8)(); lib.aln_debug_read_fwd128(buf, 0)
    pairs = ((rows + 63) // 64 + 1023) // 1024 * 5
    names = ['wait rows', 'S1 L0(A)|out(B)', 'S2 L0(B)|pack(A)', 'S3 L1(A)|pack(B)', 'S4 L1(B)|pack(A)', 'S5 L2(A)|pack(B)', 'S6 L2(B)|out(A)', 'loop edge']
    print('   per pair: ' + '  '.join(f'{n} {buf[i] / pairs:.0f}' for i, n in enumerate(names)) + f'   total {sum(buf) / pairs:.0f}')
except AttributeError:
    pass
