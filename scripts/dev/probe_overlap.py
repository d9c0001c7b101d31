"""Can the atomic-bound hash-grid backward overlap with the MFMA/LDS-bound MLP backward on a second stream?"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autolabel_amd import hip as H
from autolabel_amd.pipeline import ModelLayout, Params
N, S = 2048, 128
rows = N * S
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=6.0)
P = Params(L, 'cuda'); P.init_(0)
for _d in P.descs.values():   # stand-alone launches: fold the slabs into the gradient inside the call (HipPipeline.backward defers it)
    _d.defer_dw_reduce = 0
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
z = (torch.rand(N, S, generator=g).sort(dim=1)[0] * 6 + 0.2).cuda().reshape(-1).contiguous()
d_enc = (torch.randn(rows, 48, device='cuda') * 0.01).half()
enc = torch.randn(rows, 48, device='cuda').half()
d_out = (torch.randn(rows, 16, device='cuda') * 0.01).half()
d_in = torch.empty(rows, 48, device='cuda', dtype=torch.float16)
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
gradA, gradB = torch.zeros_like(P.grad), torch.zeros_like(P.grad)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def enc_bwd(stream):
    H.call('aln_encode_bwd', C.byref(L.enc), H.ptr(ro), H.ptr(rd), H.ptr(z), None, rows, S, H.ptr(d_enc), H.ptr(gradA), C.c_void_p(stream.cuda_stream))
def mlp_bwd(stream, k='sigma'):
    gp = C.c_void_p(gradB.data_ptr() + 4 * L.offsets[k])
    H.call('aln_mlp_bwd', C.byref(P.descs[k]), H.ptr(enc), None, None, H.ptr(d_out), rows, None, None, None, H.ptr(d_in), gp, H.ptr(flag), C.c_void_p(stream.cuda_stream))
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cur = torch.cuda.current_stream()
t_enc = timeit(lambda: enc_bwd(cur)); t_mlp = timeit(lambda: [mlp_bwd(cur) for _ in range(4)])
def both():
    enc_bwd(s1)
    for _ in range(4): mlp_bwd(s2)
def both_rev():
    for _ in range(4): mlp_bwd(s2)
    enc_bwd(s1)
print(f'encode_bwd alone {t_enc:.0f} us ; 4x sigma mlp_bwd alone {t_mlp:.0f} us ; sum {t_enc + t_mlp:.0f}')
print(f'two streams (encode first) {timeit(both):.0f} us ; (mlp first) {timeit(both_rev):.0f} us')
