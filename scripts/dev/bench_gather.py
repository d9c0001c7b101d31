"""Dev only: the level-phased hash-grid gather alone (pair-plane output) on 2^19 rows of a bench-shaped batch, for one or more libraries.
  python scripts/dev/bench_gather.py [--libs a.so,b.so] [--rounds 2]"""
import argparse, os, subprocess, sys, ctypes as C, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--libs', default='')
ap.add_argument('--rounds', type=int, default=2)
ap.add_argument('--child', default=None)
a = ap.parse_args()
if a.child is None:
    libs = [l for l in a.libs.split(',') if l] or ['']
    for rnd in range(a.rounds):
        for lib in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib or 'product'], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith('{')]
            print(f"{os.path.basename(lib) or 'product':32s} round {rnd}: " + (line[-1] if line else 'FAILED ' + out.stderr[-1500:]), flush=True)
    sys.exit(0)
import torch
from autolabel_amd import hip as H
if a.child != 'product':
    H.LIB = a.child if os.path.isabs(a.child) else os.path.join(ROOT, a.child)
from autolabel_amd.pipeline import ModelLayout, Params
N, S = 4096, 128
M = N * S
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=3.0)
P = Params(L, 'cuda'); P.init_(0)
e = L.enc
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
z = (torch.rand(N, S, generator=g).sort(dim=1)[0] * 5 + 0.2).reshape(-1).cuda().contiguous()
planes = torch.empty((24, M), dtype=torch.int32, device='cuda')
ws = torch.empty(16 * M * 4, dtype=torch.uint8, device='cuda')
enc = torch.empty((M, 48), dtype=torch.float16, device='cuda')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)
r = dict(planes=t(lambda: H.call('aln_encode_fwd_planes', C.byref(e), H.ptr(P.table16), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, S, H.ptr(planes), M, H.stream())),
         rows_assemble=t(lambda: H.call('aln_encode_fwd_phased', C.byref(e), H.ptr(P.table16), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, S, H.ptr(ws), H.ptr(enc), H.stream())))
print(json.dumps(r))
