"""Dev only: the recompute backward of one fused head alone -- time per launch and errors against an fp32 torch backward
(fp16-rounded weights and activations at the kernel's rounding points).

  python scripts/dev/bench_mlp_bwd.py [--lib PATH] [--head sigma|color] [--rows N] [--save out.pt]
"""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default=None)
ap.add_argument('--head', default='sigma')
ap.add_argument('--rows', type=int, default=1 << 20)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--check-rows', type=int, default=1 << 16)
ap.add_argument('--save', default=None)
ap.add_argument('--phases', action='store_true')
a = ap.parse_args()
import torch
from autolabel_amd import hip as H
if a.lib:
    H.LIB = a.lib
from autolabel_amd.pipeline import ModelLayout, Params

L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=6.0)
P = Params(L, 'cuda'); P.init_(0)
m = L.nets[a.head]


def run(x, d_out, d_in, grad, flag, rows, rows_dev=None):
    gp = C.c_void_p(grad.data_ptr() + 4 * L.offsets[a.head])
    H.call('aln_mlp_bwd', C.byref(P.descs[a.head]), H.ptr(x), None, None, H.ptr(d_out), rows, H.ptr(rows_dev), None, None,
           H.ptr(d_in), gp, H.ptr(flag), H.stream())


def reduce(grad, rows):
    ds = (C.c_void_p * 1)(C.addressof(P.descs[a.head]))
    dws = (C.c_void_p * 1)(grad.data_ptr() + 4 * L.offsets[a.head])
    rr = (C.c_int32 * 1)(rows)
    H.call('aln_mlp_dw_reduce_all', 1, ds, dws, rr, H.stream())


def reference(x, d_out):
    w = P.net_view(a.head).clone()
    Ws, o = [], 0
    for (no, ni) in m.shapes:
        Ws.append(w[o:o + no * ni].view(no, ni).half().float().requires_grad_(True)); o += no * ni
    h = x.float()
    # straight-through fp16 rounding of activations (what the kernel holds in registers / LDS)
    def q(t):
        return t + (t.half().float() - t).detach()
    for W in Ws[:-1]:
        h = q(torch.relu(h @ W.t()))
    # gradients are rounded to fp16 between layers too; autograd keeps fp32 there -> tolerance, not equality
    xin = x.float().requires_grad_(True)
    h = xin
    for W in Ws[:-1]:
        h = q(torch.relu(h @ W.t()))
    out = h @ Ws[-1].t()
    out.backward(d_out.float())
    return xin.grad, torch.cat([W.grad.reshape(-1) for W in Ws])


def rel(x, y):
    return ((x.float() - y.float()).norm() / (y.float().norm() + 1e-30)).item()


torch.manual_seed(1)
# ---- correctness on a ragged row count
rows = a.check_rows - 37
x = (torch.randn(rows, m.in_pad, device='cuda') * 0.5).half()
x[:, m.n_in:] = 1.0
d_out = (torch.randn(rows, m.out_pad, device='cuda') * 0.02).half()
d_out[:, m.n_out:] = 0
d_in = torch.full((rows, m.in_pad), float('nan'), device='cuda', dtype=torch.float16)
grad = torch.zeros_like(P.grad)
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
run(x, d_out, d_in, grad, flag, rows); reduce(grad, rows)
torch.cuda.synchronize()
gx, gw = reference(x, d_out)
dw = grad[L.offsets[a.head]:L.offsets[a.head] + m.n_params]
print(f'{a.head}: rows {rows}: d_in rel err {rel(d_in, gx):.3e}  dW rel err {rel(dw, gw):.3e}  flag {flag.item()}  nan in d_in {torch.isnan(d_in.float()).sum().item()}')
o = 0
for li, (no, ni) in enumerate(m.shapes):
    A, B = dw[o:o + no * ni].view(no, ni), gw[o:o + no * ni].view(no, ni)
    rowerr = ((A - B).norm(dim=1) / (B.norm(dim=1) + 1e-20))
    print(f'   layer {li}: rel {rel(A, B):.3e}  worst row {rowerr.max().item():.3e}')
    o += no * ni
# with a device-side row count (the colour head's live rows)
nd = torch.tensor([rows - 1000], dtype=torch.int32, device='cuda')
d_in2 = torch.zeros_like(d_in); grad2 = torch.zeros_like(P.grad)
run(x, d_out, d_in2, grad2, flag, rows, nd); reduce(grad2, rows)
gx2, gw2 = reference(x[:rows - 1000], d_out[:rows - 1000])
dw2 = grad2[L.offsets[a.head]:L.offsets[a.head] + m.n_params]
print(f'   rows_dev: d_in rel {rel(d_in2[:rows - 1000], gx2):.3e}  dW rel {rel(dw2, gw2):.3e}  untouched tail {bool((d_in2[rows - 1000:] == 0).all())}')
# bit-reproducible
grad3 = torch.zeros_like(P.grad); d_in3 = torch.zeros_like(d_in)
run(x, d_out, d_in3, grad3, flag, rows); reduce(grad3, rows)
print('   repeat bit-identical:', torch.equal(grad3, grad), torch.equal(d_in3, d_in))
if a.save:
    torch.save({'d_in': d_in.cpu(), 'dw': dw.cpu()}, a.save)

# ---- timing
rows = a.rows
x = (torch.randn(rows, m.in_pad, device='cuda') * 0.5).half()
d_out = (torch.randn(rows, m.out_pad, device='cuda') * 0.02).half()
d_in = torch.empty(rows, m.in_pad, device='cuda', dtype=torch.float16)
for _ in range(3):
    run(x, d_out, d_in, grad, flag, rows)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(a.reps):
    e0.record(); run(x, d_out, d_in, grad, flag, rows); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
mf = {'sigma': 152, 'color': 132}.get(a.head, 0)
busy = mf * 32 * (rows / 32) / 1024 / 2400.0   # us of MFMA pipe time per SIMD at 2.4 GHz
print(f'{a.head}: {rows} rows: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us   (MFMA pipe time {busy:.1f} us -> {100 * busy / ts[len(ts) // 2]:.1f} % busy at 2.4 GHz)')

if a.phases:
    lib = H.lib()
    lib.aln_debug_read_phases128.argtypes = [C.c_void_p, C.c_int]
    lib.aln_debug_read_phases128(None, 1)
    n = 5
    for _ in range(n):
        run(x, d_out, d_in, grad, flag, rows)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 32)()
    lib.aln_debug_read_phases128(buf, 0)
    tiles = (rows // 128 + 255) // 256 * n
    names = ['D: whole phase + loop edge', 'B: whole phase', 'wait B_b', 'C: whole phase', 'wait B_c', 'D', 'wait end'] + ['sub %d' % i for i in range(7, 32)]
    tot = sum(buf[i] for i in range(32)) / tiles
    print(f'-- block 0 wave 0: {tot:.0f} ticks per tile')
    for i, nm in enumerate(names):
        if buf[i]:
            print(f'   {nm:42s} {buf[i] / tiles:8.0f}')
