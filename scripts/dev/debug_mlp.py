import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes as C, torch
from autolabel_amd import hip as H
from oracle import nerf_oracle as O
from test_gpu_kernels import _mlp_setup
def rel(a, b): return (a - b).norm().item() / max(b.norm().item(), 1e-20)
for (n_in, hid, n_out, nh) in [(48, 128, 16, 2), (32, 128, 16, 2), (16, 64, 64, 2), (80, 64, 16, 1)]:
    for gscale in [1.0, 1e-3]:
        rows = 4096
        Ws, desc, keep = _mlp_setup(H, n_in, hid, n_out, nh)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(rows, n_in, generator=g).half()
        xo = x.float().requires_grad_(True)
        Wo = [w.clone().requires_grad_(True) for w in Ws]
        want = O.mlp_forward(xo, Wo, half_sim=True)
        h1 = torch.zeros(rows, hid, dtype=torch.float16, device='cuda'); h2 = torch.zeros(rows, hid, dtype=torch.float16, device='cuda')
        out = torch.zeros(rows, n_out, dtype=torch.float16, device='cuda'); xd = x.cuda()
        H.call('aln_mlp_fwd', C.byref(desc), H.ptr(xd), rows, None, H.ptr(h1), H.ptr(h2), H.ptr(out), H.stream())
        d_out = (torch.randn(rows, n_out, generator=g) * 0.05 * gscale).half()
        (want * d_out.float()).sum().backward()
        dA1 = torch.zeros(rows, hid, dtype=torch.float16, device='cuda'); dA2 = torch.zeros(rows, hid, dtype=torch.float16, device='cuda')
        d_in = torch.zeros(rows, n_in, dtype=torch.float16, device='cuda'); dW = torch.zeros(sum(w.numel() for w in Ws), device='cuda')
        flag = torch.zeros(1, dtype=torch.int32, device='cuda'); dod = d_out.cuda()
        H.call('aln_mlp_bwd', C.byref(desc), H.ptr(xd), H.ptr(h1), H.ptr(h2), H.ptr(dod), rows, None, H.ptr(dA1), H.ptr(dA2), H.ptr(d_in), H.ptr(dW), H.ptr(flag), H.stream())
        torch.cuda.synchronize()
        gi = xo.grad
        cols = [round(rel(d_in.cpu().float()[:, j], gi[:, j]), 4) for j in range(0, n_in, max(1, n_in // 8))]
        print((n_in, hid, n_out, nh), 'gscale', gscale, 'out rel', round(rel(out.cpu().float(), want.detach()), 5), 'd_in rel', round(rel(d_in.cpu().float(), gi), 5), 'per-col', cols, 'absmax', gi.abs().max().item())
