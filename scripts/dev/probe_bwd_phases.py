"""Where do the cycles of k_mlp_bwd_recomp8 go?  Builds mlp.hip with -DALN_PHASE_TIMING (clock64 stamps around every
barrier, block 0, chain wave 0 and dW wave 4) into scripts/dev/_build/ and prints cycles per phase and tile.

  python scripts/dev/probe_bwd_phases.py build     # here (hipcc cross-compiles)
  python scripts/dev/probe_bwd_phases.py [head]    # on the GPU box
"""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'scripts', 'dev', '_build', 'libaln_timing.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from autolabel_amd import build as B
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES if os.path.exists(os.path.join(B.CSRC, s))]
    subprocess.run(['hipcc'] + B.FLAGS + ['-shared', '-DALN_PHASE_TIMING'] + srcs + ['-o', OUT], check=True)
    sys.exit(0)
import torch
from autolabel_amd import hip as H
H.LIB = OUT          # load the instrumented build instead of the product library
from autolabel_amd.pipeline import ModelLayout, Params
head = sys.argv[1] if len(sys.argv) > 1 else 'sigma'
sem_step = head in ('semo_step', 'semf_step')      # the head inside aln_sem_heads_bwd, with its on-the-fly row sources
if sem_step:
    head = head[:4]
rows = 1 << 20
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=6.0)
P = Params(L, 'cuda'); P.init_(0)
for _d in P.descs.values():   # stand-alone launches: fold the slabs into the gradient inside the call (HipPipeline.backward defers it)
    _d.defer_dw_reduce = 0
m = L.nets[head]
x = torch.randn(rows, m.in_pad, device='cuda').half()
d_out = (torch.randn(rows, m.out_pad, device='cuda') * 0.01).half()
d_in = torch.empty(rows, m.in_pad, device='cuda', dtype=torch.float16)
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
grad = torch.zeros_like(P.grad)
gp = C.c_void_p(grad.data_ptr() + 4 * L.offsets[head])
if sem_step:
    N, S1, S2, D, G, Ccls = 4096, 128, 128, 64, 15, 7
    sigma_out = (torch.randn(rows, 16, device='cuda') * 0.5).half()
    feat = (torch.randn(rows, D, device='cuda') * 0.5).half()
    w_row = torch.rand(rows, device='cuda') * 0.05
    g_sem, g_feat = torch.randn(N, Ccls, device='cuda') * 0.1, torch.randn(N, D, device='cuda') * 0.1
    d_oin = torch.empty(rows, L.nets['semo'].in_pad, device='cuda', dtype=torch.float16)
    d_fin = torch.empty(rows, L.nets['semf'].in_pad, device='cuda', dtype=torch.float16)
    gpf, gpo = (C.c_void_p(grad.data_ptr() + 4 * L.offsets[k]) for k in ('semf', 'semo'))
def run():
    if sem_step:
        H.call('aln_sem_heads_bwd', C.byref(P.descs['semf']), C.byref(P.descs['semo']), H.ptr(sigma_out), H.ptr(feat), H.ptr(w_row), H.ptr(g_sem),
               H.ptr(g_feat), N, S1, S2, Ccls, rows, D, G, H.ptr(d_oin), H.ptr(d_fin), gpf, gpo, 1, None, H.ptr(flag), H.stream())
        return
    H.call('aln_mlp_bwd', C.byref(P.descs[head]), H.ptr(x), None, None, H.ptr(d_out), rows, None, None, None, H.ptr(d_in), gp,
           H.ptr(flag), H.stream())
lib = H.lib()
lib.aln_debug_read_phases.argtypes = [C.c_void_p, C.c_int]
for _ in range(3): run()
torch.cuda.synchronize()
lib.aln_debug_read_phases(None, int(m.in_pad) if sem_step else 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
buf = (C.c_longlong * 64)()
lib.aln_debug_read_phases(buf, 0)
us = e0.elapsed_time(e1) / reps * 1e3
ntiles = rows // 128
occ = 2 if (m.hidden == 64 and m.in_pad <= 32) else 1   # 64-wide heads: two blocks per CU
nblk = min(ntiles, occ * torch.cuda.get_device_properties(0).multi_processor_count)
iters = (ntiles + nblk - 1) // nblk * reps
print(f'{head}: {us:.0f} us per launch, {iters // reps} tiles per block')
names = ['tail(d_in/dW_first)', 'wait B0', 'load issue', 'wait B1', 'fwd recompute', 'wait B2', 'last layer', 'wait B3',
         'write dA2', 'wait B4', 'mid layer', 'wait B5', 'write dA1', 'wait B6']
for role, rn in [(0, 'chain wave 0'), (1, 'dW wave 4')]:
    v = [buf[role * 32 + i] / iters for i in range(32)]
    print(f'-- {rn}: {sum(v):.0f} clock ticks per tile')
    for i, n in enumerate(names):
        print(f'   {n:22s} {v[i]:8.0f}')
    sub = ['L0 mfma issue', 'relu_pack h1', 'write h1', 'L1 mfma issue', 'relu_pack h2', '(mid) mfma issue']
    if role == 0:
        print('   sub-stamps (the phase entry above = remainder after the last sub-stamp):')
        for i, n in enumerate(sub):
            print(f'     {n:20s} {v[16 + i]:8.0f}')
