import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autolabel_amd import hip as H
e = H.make_enc_desc('hg+freq', 1.0)
N, S = 300, 13
g = torch.Generator().manual_seed(5 + N)
ro = (torch.rand(N, 3, generator=g) - 0.5).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
z = (torch.rand(N, S, generator=g).sort(dim=1)[0] * 2).cuda().reshape(-1)
rows = N * S
d_enc = torch.zeros(rows, e.enc_pad, dtype=torch.float16)
d_enc[:, :44] = (torch.randn(rows, 44, generator=g) * 0.1).half()
d_enc = d_enc.cuda()
n = int(e.grid.n_entries) * 2
ref = torch.zeros(n, device='cuda'); got = torch.zeros(n, device='cuda')
H.call('aln_encode_bwd', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, rows, S, H.ptr(d_enc), H.ptr(ref), H.stream())
ws = torch.zeros(int(H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(e), rows)), dtype=torch.uint8, device='cuda')
H.call('aln_encode_bwd_binned', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, rows, rows, S, S, None, H.ptr(d_enc), H.ptr(got), H.ptr(ws), 0, 16, None, None, H.stream())
torch.cuda.synchronize()
d = (got - ref).abs()
print('max err', d.max().item(), 'ref max', ref.abs().max().item(), 'rel norm', (got - ref).norm().item() / ref.norm().item())
for l in range(16):
    a = int(e.grid.offset[l]) * 2; b = a + int(e.grid.size[l]) * 2
    lost = ((got[a:b] == 0) & (ref[a:b].abs() > 1e-6)).sum().item()
    extra = ((got[a:b] != 0) & (ref[a:b] == 0)).sum().item()
    print(f'level {l}: lost {lost} extra {extra} max err {d[a:b].max().item():.3e} nonzero ref {(ref[a:b] != 0).sum().item()} got {(got[a:b] != 0).sum().item()}')
