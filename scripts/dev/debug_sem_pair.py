import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from autolabel_amd import hip as H
from test_gpu_kernels import _mlp_setup
N, S1, S2, Ccls, out_pad, G = 37, 24, 20, 7, 16, 15
Wf, dF, keepF = _mlp_setup(H, 16, 64, 64, 2, seed=3)
Wo, dO, keepO = _mlp_setup(H, 80, 64, out_pad, 1, seed=4)
rows = N * (S1 + S2)
g = torch.Generator().manual_seed(9)
sigma_out = torch.randn(rows, 16, generator=g).half().cuda()
w_row = (torch.rand(rows, generator=g) * (torch.rand(rows, generator=g) > 0.3)).cuda()
g_sem = (torch.randn(N, Ccls, generator=g) * 0.1).cuda(); g_feat = (torch.randn(N, 64, generator=g) * 0.1).cuda()
feat = torch.zeros(rows, 64, dtype=torch.float16, device='cuda'); logits = torch.zeros(rows, out_pad, dtype=torch.float16, device='cuda')
H.call('aln_sem_heads_fwd', C.byref(dF), C.byref(dO), H.ptr(sigma_out), rows, 64, G, H.ptr(feat), H.ptr(logits), H.stream())
nf, no = sum(w.numel() for w in Wf), sum(w.numel() for w in Wo)
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
res = {}
for fold in (0, 1):
    d_oin = torch.zeros(rows, 80, dtype=torch.float16, device='cuda'); d_fin = torch.zeros(rows, 16, dtype=torch.float16, device='cuda')
    dWf, dWo = torch.zeros(nf, device='cuda'), torch.zeros(no, device='cuda')
    H.call('aln_sem_heads_bwd', C.byref(dF), C.byref(dO), H.ptr(sigma_out), H.ptr(feat), H.ptr(w_row), H.ptr(g_sem), H.ptr(g_feat),
           N, S1, S2, Ccls, rows, 64, G, H.ptr(d_oin), H.ptr(d_fin), H.ptr(dWf), H.ptr(dWo), fold, None, H.ptr(flag), H.stream())
    torch.cuda.synchronize()
    res[fold] = (dWf.cpu(), dWo.cpu(), d_fin.cpu(), d_oin.cpu())
a, b = res[0], res[1]
o = 0
for name, (no_, ni) in zip(['W0', 'W1', 'W2'], [(64, 16), (64, 64), (64, 64)]):
    A, B = a[0][o:o + no_ * ni].view(no_, ni), b[0][o:o + no_ * ni].view(no_, ni)
    d = (A - B).abs()
    print(name, 'max ref', A.abs().max().item(), 'max diff', d.max().item(), 'bad rows', (d.max(1)[0] > 1e-2 * A.abs().max()).nonzero().flatten().tolist()[:20],
          'bad cols', (d.max(0)[0] > 1e-2 * A.abs().max()).nonzero().flatten().tolist()[:20])
    o += no_ * ni
o = 0
for name, (no_, ni) in zip(['V0', 'V1'], [(64, 80), (out_pad, 64)]):
    A, B = a[1][o:o + no_ * ni].view(no_, ni), b[1][o:o + no_ * ni].view(no_, ni)
    d = (A - B).abs()
    print(name, 'max ref', A.abs().max().item(), 'max diff', d.max().item(), 'bad rows', (d.max(1)[0] > 1e-2 * A.abs().max()).nonzero().flatten().tolist()[:20],
          'bad cols', (d.max(0)[0] > 1e-2 * A.abs().max()).nonzero().flatten().tolist()[:20])
    o += no_ * ni
dg = a[2].float() + a[3][:, 64:80].float()
print('d_geo max diff', (dg - b[2].float()).abs().max().item(), 'of', dg.abs().max().item())
