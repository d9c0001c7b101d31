"""Which phase of the binned hash-grid scatter is not a pure function of its inputs when a second process shares the GPU?

Captures the arguments of ONE `aln_encode_bwd_binned` call of a real training step (fixed d_enc, z, depth order), then replays that
call alone, again and again, next to a competing process (stress_determinism.py --role disturb).  Every replay's gradient table is
compared with the first; on a mismatch the record pool of the differing level is
  (a) turned into a gradient on the host (float64 sums of value * 2^-shift per entry): equal to the GPU's table of THIS replay ->
      phase 2 added up what it was given, so phase 1 produced different records; unequal -> phase 2 is the culprit;
  (b) compared with the first replay's pool as a multiset per (tile, slice) run;
  (round 6: the -DBIN_DEBUG taps were removed from encode.hip with the kernel they instrumented; --map-lib / --times need the file as of commit 74cae74)
  (c) with --map-lib (a -DBIN_DEBUG=5 build of encode.hip, run once) traced to the (tile, wave, lane, corner) that emitted each wrong
      record, and with --dbg <a -DBIN_DEBUG=6 build> --times set against per-wave timestamps (was the wave switched out?).
Round 5: the differing records all came from lanes 48..63 of single waves, with products of packed fp32 multiplies zeroed; no timing
gap.  scripts/dev/probe_pk_f32.hip isolates the instruction-level cause; the shipped library is built without packed fp32 (build.py).
Building a tap library:  hipcc <build.py FLAGS> -DBIN_DEBUG=5 -c autolabel_amd/csrc/encode.hip -o /tmp/encode.o ;
                         hipcc --offload-arch=gfx950 -shared -fPIC /tmp/encode.o <the other objects of csrc/build> -o /tmp/dbg5/libautolabel_hip.so
"""
import argparse
import os
import subprocess
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch

from stress_determinism import build

CHUNK = 512 * 8


def level_tables(bins, nl, nt, l):
    pool = bins[:nl * nt * CHUNK * 8].view(torch.int64).reshape(nl, nt, CHUNK)[l].cpu().numpy()
    desc = bins[nl * nt * CHUNK * 8:nl * nt * CHUNK * 8 + nl * 64 * nt * 4].view(torch.int32).reshape(nl, 64, nt)[l].cpu().numpy().astype(np.uint32)
    return pool, desc


def host_grad(pool, desc, sl, size):
    """float64 gradient of one level from its records; also the canonical (sorted) record list per (tile, slice)."""
    g = np.zeros((size, 2), np.float64)
    canon = {}
    nt = desc.shape[1]
    for s in range(64):
        for t in range(nt):
            q = int(desc[s, t])
            st, n, sh = q & 0x1FFF, (q >> 13) & 0x3FFF, (q >> 27) - 8
            if not n:
                continue
            r = pool[t, st:st + n].view(np.uint64)
            slot = (r & np.uint64(0xFFFFFFFF)).astype(np.int64)
            val = (r >> np.uint64(32)).astype(np.uint32)
            h = val.view(np.uint32).astype(np.uint32)
            lo = (h & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float64)
            hi = (h >> 16).astype(np.uint16).view(np.float16).astype(np.float64)
            e = (s << sl) + slot
            np.add.at(g[:, 0], e, lo * 2.0 ** -sh)
            np.add.at(g[:, 1], e, hi * 2.0 ** -sh)
            canon[(s, t)] = (np.sort(r), sh)
    return g, canon


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2000)
    ap.add_argument('--seconds', type=float, default=70.0)
    ap.add_argument('--no-disturb', action='store_true')
    ap.add_argument('--max-report', type=int, default=3)
    ap.add_argument('--map-lib', default='', help='a -DBIN_DEBUG=5 library, run ONCE up front: maps the record values of every (level, row) so that '
                    'the records a failing replay got wrong can be traced to their lanes')
    ap.add_argument('--times', action='store_true', help='--dbg is a -DBIN_DEBUG=6 library: per (level, tile, wave) timestamps instead of value taps')
    ap.add_argument('--no-tap', action='store_true', help='with --dbg: load that library but leave its taps switched off')
    ap.add_argument('--dbg', default='', help='run THIS library instead of the shipped one (e.g. a build with other compiler flags); with --times '
                    'one built with -DBIN_DEBUG=6')
    a = ap.parse_args()
    import autolabel_amd  # noqa: F401
    from autolabel_amd import hip as H
    if a.dbg:
        assert H._lib is None
        H.LIB = os.path.abspath(a.dbg)
    eng, P, layout, batch, frames = build(1024, 32)
    pipe = eng.pipe
    cap = {}
    inner = pipe._k

    def spy(name, *args, **kw):
        if name == 'aln_encode_bwd_binned':
            cap['args'] = args
        return inner(name, *args, **kw)
    pipe._k = spy
    P.grad.zero_()
    out = eng.forward_backward(batch, seed=7, step=0)
    torch.cuda.synchronize()
    pipe._k = inner
    args = cap['args']
    g = layout.enc.grid
    nl = int(g.n_levels)
    M = int(args[5])
    nt = (M + 511) // 512
    bins = eng.ws.bufs['enc_bwd_bins'][1]
    print('captured scatter call: rows', M, 'tiles', nt, 'levels', args[13], args[14], flush=True)
    dbg = ref_dbg = None
    if a.dbg and not a.no_tap:
        import ctypes as C
        dbg = torch.zeros(nl, M, 16, dtype=torch.int32, device='cuda')
        fn = H.lib().aln_debug_set_bin_dbg
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
        assert fn(dbg.data_ptr()) == 0
    lane_map = None
    if a.map_lib:
        import ctypes as C
        ML = C.CDLL(os.path.abspath(a.map_lib))
        res, argt = H._SIGS['aln_encode_bwd_binned']
        ML.aln_encode_bwd_binned.restype, ML.aln_encode_bwd_binned.argtypes = res, argt
        ML.aln_debug_set_bin_dbg.restype, ML.aln_debug_set_bin_dbg.argtypes = C.c_int, [C.c_void_p]
        mt = torch.zeros(nl, M, 16, dtype=torch.int32, device='cuda')
        assert ML.aln_debug_set_bin_dbg(mt.data_ptr()) == 0
        assert ML.aln_encode_bwd_binned(*args) == 0
        torch.cuda.synchronize()
        assert ML.aln_debug_set_bin_dbg(None) == 0
        lane_map = mt.cpu().numpy().view(np.uint32)
        print('lane map taken', flush=True)
    child = None
    if not a.no_disturb:
        child = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'stress_determinism.py'), '--role', 'disturb',
                                  '--batch', '4096', '--S', '64', '--seconds', str(a.seconds)])
        time.sleep(20.0)
    n_grid = layout.n_grid
    ref = ref_bins = None
    reports = 0
    nbad = 0
    t0 = time.time()
    it = 0
    while it < a.iters and time.time() - t0 < a.seconds - 22:
        P.grad[:n_grid].zero_()
        H.call('aln_encode_bwd_binned', *args)
        torch.cuda.synchronize()
        got = P.grad[:n_grid].clone()
        if ref is None:
            ref, ref_bins = got, bins.clone()
            ref_dbg = None if dbg is None else dbg.clone()
            it += 1
            continue
        it += 1
        if torch.equal(ref.view(torch.int32), got.view(torch.int32)):
            continue
        nbad += 1
        if reports >= a.max_report:
            continue
        reports += 1
        now_bins = bins.clone()
        ts = None
        if dbg is not None and a.times:
            w = dbg.view(-1)[:nl * nt * 16].cpu().numpy().view(np.uint32).astype(np.int64).reshape(nl, nt, 8, 2)
            ts = w[..., 0] | (w[..., 1] << 32)          # 100 MHz ticks
            d = ts[1:] - ts[:-1]
            print(f'   level-to-level time of a wave (10 ns ticks): median {int(np.median(d))}, 99 % {int(np.percentile(d, 99))}, max {int(d.max())}; '
                  f'largest at (level, tile, wave) {[tuple(int(v) for v in np.unravel_index(i, d.shape)) for i in np.argsort(d, axis=None)[-6:]]}')
        diff = (ref.view(torch.int32) != got.view(torch.int32)).nonzero().flatten().cpu().numpy()
        print(f'--- replay {it}: {len(diff)} gradient words differ', flush=True)
        for l in range(nl):
            a0 = int(g.offset[l]) * 2
            b0 = int(g.offset[l + 1]) * 2 if l + 1 < nl else n_grid
            dl = diff[(diff >= a0) & (diff < b0)] - a0
            if not len(dl):
                continue
            size = (b0 - a0) // 2
            sl = max(0, min(13, int(np.ceil(np.log2(size))) - 6))
            ents = np.unique(dl // 2)
            print(f'level {l}: {len(dl)} words, {len(ents)} entries; entries {ents[:24].tolist()} slices {np.unique(ents >> sl).tolist()}')
            r_l, g_l = ref[a0:b0].cpu().numpy().reshape(-1, 2), got[a0:b0].cpu().numpy().reshape(-1, 2)
            for e in ents[:6]:
                print(f'   entry {e}: first run {r_l[e].tolist()} this run {g_l[e].tolist()} delta {(g_l[e].astype(np.float64) - r_l[e]).tolist()}')
            pool, desc = level_tables(now_bins, nl, nt, l)
            rpool, rdesc = level_tables(ref_bins, nl, nt, l)
            hg, canon = host_grad(pool, desc, sl, size)
            rhg, rcanon = host_grad(rpool, rdesc, sl, size)
            hg32, rhg32 = hg.astype(np.float32), rhg.astype(np.float32)
            print(f'   host sum of THIS replay\'s records vs this replay\'s table: {int((hg32 != g_l).sum())} words differ;'
                  f' vs the first replay\'s table: {int((hg32 != r_l).sum())};  first replay\'s records vs first table: {int((rhg32 != r_l).sum())}')
            ddesc = np.argwhere(desc != rdesc)
            print(f'   descriptors differing: {len(ddesc)} {ddesc[:6].tolist()}')
            nrun = 0
            shown = 0
            lanes = []
            for k in sorted(set(canon) | set(rcanon)):
                x, y = canon.get(k), rcanon.get(k)
                if x is None or y is None or x[1] != y[1] or len(x[0]) != len(y[0]) or not np.array_equal(x[0], y[0]):
                    nrun += 1
                    if x is None or y is None:
                        print(f'   run (slice {k[0]}, tile {k[1]}): present in one replay only')
                        continue
                    from collections import Counter
                    cx, cy = Counter(x[0].tolist()), Counter(y[0].tolist())
                    only_x, only_y = sorted((cx - cy).elements()), sorted((cy - cx).elements())
                    if lane_map is not None:
                        for r in only_y:
                            val = (r >> 32) & 0xFFFFFFFF
                            blk = lane_map[l, k[1] * 512:(k[1] + 1) * 512, 0:8]
                            hit = np.argwhere(blk == val)
                            for q, c in hit[:1]:
                                got_r = [x_ for x_ in only_x if (x_ & 0xFFFFFFFF) == (r & 0xFFFFFFFF)]
                                lanes.append((int(l), int(k[1]), int(q) // 64, int(q) % 64, int(c), f'{val:08x}', f'{(got_r[0] >> 32) & 0xFFFFFFFF:08x}' if got_r else '?'))
                    if shown < 6:
                        shown += 1
                        f = lambda r: f'slot {r & 0xFFFFFFFF:5d} h0 {(r >> 32) & 0xFFFF:04x} h1 {(r >> 48) & 0xFFFF:04x}'
                        print(f'   run (slice {k[0]:2d}, tile {k[1]:3d}) len {len(x[0])}/{len(y[0])} shift {x[1]}/{y[1]}: only in this replay [{"; ".join(f(r) for r in only_x[:4])}]'
                              f'  only in the first [{"; ".join(f(r) for r in only_y[:4])}]')
            print(f'   runs whose record multiset differs: {nrun}', flush=True)
            if lanes and ts is not None:
                for (l_, t_, w_) in sorted({(x_[0], x_[1], x_[2]) for x_ in lanes}):
                    g1 = int(ts[l_ + 1, t_, w_] - ts[l_, t_, w_]) if l_ + 1 < nl else None
                    g0 = int(ts[l_, t_, w_] - ts[l_ - 1, t_, w_]) if l_ > 0 else None
                    oth = [int(ts[l_ + 1, t_, k_] - ts[l_, t_, k_]) for k_ in range(8)] if l_ + 1 < nl else None
                    print(f'   failing wave (level {l_}, tile {t_}, wave {w_}): ticks from this level to the next {g1}, from the previous level {g0}; all 8 waves of the tile {oth}')
            if lanes:
                lanes.sort()
                lanes = lanes[:40]
                print('   wrong records traced to (level, tile, wave, lane, corner, right value, wrong value):')
                for t_ in lanes:
                    print('     ', t_)
    if child is not None:
        child.wait()
    print(f'stress_scatter: {nbad} of {it - 1} replays differed from the first', flush=True)
    return 1 if nbad else 0


if __name__ == '__main__':
    sys.exit(main())
