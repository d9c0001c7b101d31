"""Dev only: the binned hash-grid scatter alone on a bench-shaped batch (4096 rays x (128 + 128) samples walked in depth order), per phase,
for one or more LIBRARIES (same-box A/B; every library runs in its own process, the processes alternate).

  python scripts/dev/bench_scatter.py [--libs a.so,b.so] [--rounds 3] [--check]      (no --libs: the in-tree library)
  --check: gradient table of every library against the first one (max |diff| / max |ref| per level) and against an fp64 scatter
"""
import argparse, os, subprocess, sys, ctypes as C, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--libs', default='')
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--child', default=None)
ap.add_argument('--check', action='store_true')
ap.add_argument('--save', default=None)
ap.add_argument('--rays', type=int, default=4096)
ap.add_argument('--p1-only', action='store_true')
a = ap.parse_args()

if a.child is None:
    libs = [l for l in a.libs.split(',') if l] or ['']
    for rnd in range(a.rounds):
        for i, lib in enumerate(libs):
            cmd = [sys.executable, os.path.abspath(__file__), '--child', lib or 'product', '--reps', str(a.reps), '--rays', str(a.rays)] + (['--p1-only'] if a.p1_only else [])
            if a.check and rnd == 0:
                cmd += ['--save', f'/tmp/scatter_{i}.pt']
            out = subprocess.run(cmd, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith('{')]
            if not line:
                print(lib, 'FAILED', out.stderr[-2000:]); continue
            for l in out.stderr.splitlines():
                if l.startswith('timing'):
                    print('   ', l)
            d = json.loads(line[-1])
            print(f"{os.path.basename(lib) or 'product':28s} round {rnd}: phase1 {d['p1']:7.1f}  phase2 {d['p2']:7.1f}  pair {d['pair']:7.1f}  fused-adam pair {d['fused']:7.1f} us (in 2 / 4 level groups {d.get('fg2', 0):.1f} / {d.get('fg4', 0):.1f})"
                  f"   records {d['records'] / 1e6:.2f} M", flush=True)
    if a.check:
        import torch
        ref = torch.load('/tmp/scatter_0.pt')
        for i in range(len(libs)):
            t = torch.load(f'/tmp/scatter_{i}.pt')
            off = ref['off']
            worst = []
            for l in range(16):
                sl = slice(2 * off[l], 2 * off[l + 1])
                worst.append(float((t['grad'][sl].double() - ref['exact'][sl]).abs().max() / ref['exact'][sl].abs().max()))
            same = bool(torch.equal(t['grad'], ref['grad']))
            print(f"{os.path.basename(libs[i]) or 'product':28s} vs fp64 scatter, worst level {max(worst):.2e} (per level: {' '.join('%.0e' % w for w in worst)})  bit-equal to first: {same}")
    sys.exit(0)

import torch
from autolabel_amd import hip as H
if a.child != 'product':
    H.LIB = a.child if os.path.isabs(a.child) else os.path.join(ROOT, a.child)
from autolabel_amd.pipeline import ModelLayout
N, S1, S2 = a.rays, 128, 128
M1, M = N * S1, N * (S1 + S2)
L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=3.0)
e = L.enc
g = torch.Generator().manual_seed(0)
ro = ((torch.rand(N, 3, generator=g) - 0.5) * 4).cuda()
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
zc = torch.rand(N, S1, generator=g).sort(dim=1)[0] * 5 + 0.2
# fine samples concentrate around a "surface" per ray, like the importance pass
zs = torch.rand(N, 1, generator=g) * 4 + 0.5
zf = (zs + torch.randn(N, S2, generator=g) * 0.15).clamp(0.2, 5.2).sort(dim=1)[0]
z = torch.cat([zc.reshape(-1), zf.reshape(-1)]).cuda().contiguous()
perm = torch.cat([zc, zf], dim=1).argsort(dim=1).to(torch.int16).cuda().contiguous()
d_enc = (torch.randn(M, 48, generator=torch.Generator().manual_seed(1)) * 0.01).half().cuda()
grad = torch.zeros(L.n_grid + 8, device='cuda')
ws = torch.empty(int(H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(e), M)), dtype=torch.uint8, device='cuda')
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
# optimizer state for the fused route
P, m_, v_ = torch.zeros(L.n_grid, device='cuda'), torch.zeros(L.n_grid, device='cuda'), torch.zeros(L.n_grid, device='cuda')
t16 = torch.zeros(L.n_grid, dtype=torch.float16, device='cuda')
si, sf = torch.zeros(16, dtype=torch.int32, device='cuda'), torch.ones(16, device='cuda')
sf[1] = 0.0
ad = H.AlnAdamFuse(P.data_ptr(), m_.data_ptr(), v_.data_ptr(), t16.data_ptr(), si.data_ptr(), sf.data_ptr(), 1e-2, 0.9, 0.99, 1e-15)


def phase(ph):
    H.call('aln_encode_bwd_binned_phase', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, S2, H.ptr(perm), H.ptr(d_enc), H.ptr(grad),
           H.ptr(ws), 0, 16, H.ptr(flag), None, 0.0, ph, H.stream())


def fused():
    H.call('aln_encode_bwd_binned', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, S2, H.ptr(perm), H.ptr(d_enc), None, H.ptr(ws),
           0, 16, H.ptr(flag), C.byref(ad), H.stream())


def fused_groups(ng):   # the same work as `fused`, level group by level group (is a group's pool still in the Infinity Cache when phase 2 reads it?)
    step = 16 // ng
    for lo in range(0, 16, step):
        H.call('aln_encode_bwd_binned', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, S2, H.ptr(perm), H.ptr(d_enc), None, H.ptr(ws),
               lo, lo + step, H.ptr(flag), C.byref(ad), H.stream())


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


lib = H.lib()
if hasattr(lib, 'aln_debug_read_bin_timing'):   # a -DBIN_TIMING build: ticks per section of phase 1 (block 300, waves 0 and 7), one launch
    lib.aln_debug_read_bin_timing.restype, lib.aln_debug_read_bin_timing.argtypes = C.c_int, [C.c_void_p, C.c_int]
    phase(1); torch.cuda.synchronize()
    lib.aln_debug_read_bin_timing(None, 1)
    phase(1); torch.cuda.synchronize()
    buf = (C.c_longlong * 24)()
    lib.aln_debug_read_bin_timing(buf, 0)
    names = ['top', 'gws', 'compute', 'atomics', 'B1', 'prefix', 'B2', 'stores', 'B3', 'copyout']
    for w in range(2):
        t = [buf[12 * w + i] for i in range(10)]
        print(f'timing wave {"0" if w == 0 else "7"}: total {sum(t)} ticks over 16 levels; per level: ' + '  '.join(f'{n} {v / 16:.0f}' for n, v in zip(names, t)), file=sys.stderr)
if hasattr(lib, 'aln_debug_read_acc_timing'):   # a -DACC_TIMING build: ticks per section of phase 2 (one block of a hashed level, waves 0 and 15), fused-optimizer launch
    lib.aln_debug_read_acc_timing.restype, lib.aln_debug_read_acc_timing.argtypes = C.c_int, [C.c_void_p, C.c_int]
    fused(); torch.cuda.synchronize()
    lib.aln_debug_read_acc_timing(None, 1)
    fused(); torch.cuda.synchronize()
    buf = (C.c_longlong * 24)()
    lib.aln_debug_read_acc_timing(buf, 0)
    names = ['zero', 'B0', 'bound', 'B1', 'desc', 'request', 'wait', 'consume', 'tail', 'B2', 'adam']
    for w in range(2):
        t = [buf[12 * w + i] for i in range(11)]
        print(f'timing acc wave {"0" if w == 0 else "15"}: total {sum(t)} ticks: ' + '  '.join(f'{n} {v}' for n, v in zip(names, t)), file=sys.stderr)
r = dict(p1=timeit(lambda: phase(1), a.reps), p2=0.0, pair=0.0, fused=0.0) if a.p1_only else dict(p1=timeit(lambda: phase(1), a.reps), p2=timeit(lambda: phase(2), a.reps), pair=timeit(lambda: phase(3), a.reps), fused=timeit(fused, a.reps), fg2=timeit(lambda: fused_groups(2), a.reps), fg4=timeit(lambda: fused_groups(4), a.reps))
tile = int(H.lib().aln_encode_bwd_binned_tile_rows())
nt = (M + tile - 1) // tile
desc = ws[ws.numel() - 16 * 64 * nt * 4:].view(torch.int32)[:16 * 64 * nt]     # (pair records since round 6)
r['records'] = int(((desc >> 13) & 0x3FFF).sum().item())
if a.save:
    grad.zero_(); phase(3); torch.cuda.synchronize()
    out = dict(grad=grad[:L.n_grid].cpu(), off=[int(e.grid.offset[l]) for l in range(16)] + [int(e.grid.n_entries)])
    if a.save.endswith('_0.pt'):   # exact scatter in fp64 through torch (index_add), on the same positions: the oracle's corner arithmetic
        from oracle import nerf_oracle as O
        levels = O.GridSpec().levels()
        exact = torch.zeros(L.n_grid, dtype=torch.float64, device='cuda')
        S = S1 + S2
        rows = torch.arange(M, device='cuda')
        ray = torch.where(rows < M1, rows // S1, (rows - M1) // S2)
        x = (ro[ray] + rd[ray] * z[:, None]).clamp(-3.0, 3.0)
        xn = ((x + 3.0) / torch.full_like(x, 6.0)).clamp(0, 1)
        for l in range(16):
            idx, w = O.grid_corner_indices(xn, levels[l])
            gl = d_enc[:, 12 + 2 * l:14 + 2 * l].double()
            for c in range(8):
                flat = (2 * (int(e.grid.offset[l]) + idx[:, c].long()))
                exact.index_add_(0, flat, w[:, c].double() * gl[:, 0])
                exact.index_add_(0, flat + 1, w[:, c].double() * gl[:, 1])
        out['exact'] = exact.cpu()
    torch.save(out, a.save)
print(json.dumps(r))
