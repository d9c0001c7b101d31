"""profiles/<round>_pmc_summary.json from the per-pass summaries written by pmc_csv_summary.py.
usage: make_pmc_summary.py r02  (reads gpurun_out/pmc2_*.txt; r01 read gpurun_out/pmc_*.txt)"""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r02'
prefix = 'pmc_' if rnd == 'r01' else 'pmc%d_' % int(rnd[1:])
vals = {}
for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', prefix + '*.txt'))):
    k = None
    atomic_run = os.path.basename(f).startswith(prefix + 'atomic_')
    for line in open(f):
        if not line.startswith(' '):
            k = line.strip() + (' [ALN_ENC_BWD=atomic]' if atomic_run else '')
        else:
            m = re.match(r'\s+(\S+)\s+n=\s*(\d+) avg=(\S+)', line)
            vals.setdefault(k, {})[m.group(1)] = {'avg': float(m.group(3)), 'n': int(m.group(2))}
ours = {k: v for k, v in vals.items() if k.startswith(('_Z', 'k_')) and 'at::' not in k}
out = {'command': 'scripts/dev/run_pmc_%s.sh: one rocprofv3 --pmc pass per counter set around `python3 bench.py --steps 3 --warmup 1 '
                  '--no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0` (FETCH_SIZE | WRITE_SIZE | '
                  'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE | TCC_HIT_sum TCC_MISS_sum; with '
                  'ALN_ENC_BWD=atomic: TCC_ATOMIC_sum | TCC_EA0_ATOMIC_sum); kernel durations: %s_train_kernel_stats_rocprofv3.csv' % (rnd, rnd),
       'note': 'per-dispatch averages. FETCH_SIZE / WRITE_SIZE in KB as reported (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE '
               'under-reports wide coalesced reads by 2x, other widths uncalibrated). GRBM_GUI_ACTIVE is summed over the 8 XCDs; '
               'SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_32x32x16_f16; mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).',
       'kernels': {}}
for k, v in ours.items():
    e = {c: d['avg'] for c, d in v.items()}
    if e.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
        e['mfma_util'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] / 8 * 1024)
    if e.get('SQ_LDS_IDX_ACTIVE'):
        e['lds_conflict_frac'] = e.get('SQ_LDS_BANK_CONFLICT', 0.0) / e['SQ_LDS_IDX_ACTIVE']
    if 'TCC_HIT_sum' in e:
        e['l2_hit_rate'] = e['TCC_HIT_sum'] / max(e['TCC_HIT_sum'] + e['TCC_MISS_sum'], 1.0)
    out['kernels'][k] = e
kb = lambda name, c: next(v[c] for k, v in out['kernels'].items() if name in k and c in v) * 1024
if rnd == 'r01':
    out['k_encode_bwd_traffic_bytes_per_launch'] = kb('k_encode_bwd', 'FETCH_SIZE') + kb('k_encode_bwd', 'WRITE_SIZE')
else:
    parts = {n: {'FETCH_SIZE_bytes': kb(n, 'FETCH_SIZE'), 'WRITE_SIZE_bytes': kb(n, 'WRITE_SIZE')} for n in ('k_encode_bwd_bin', 'k_encode_bwd_accum')}
    out['encode_bwd_traffic_parts'] = parts
    out['encode_bwd_traffic_bytes_per_launch'] = sum(sum(p.values()) for p in parts.values())
    out['encode_bwd_traffic_note'] = ('raw counters (KB x 1024) of the launch pair; k_encode_bwd_accum reads exactly 8 B per record '
                                      '(bench.py roofline.records_per_launch x 8 B): the ratio of that to its FETCH_SIZE is this access '
                                      "pattern's calibration of the counter (8-byte-per-lane loads are tallied short on gfx950)")
    at = [v for k, v in out['kernels'].items() if 'ALN_ENC_BWD=atomic' in k]
    if at:
        out['atomic_kernel_requests_per_launch'] = {c: at[0].get(c) for c in ('TCC_ATOMIC_sum', 'TCC_EA0_ATOMIC_sum') if c in at[0] or any(c in a for a in at)}
        for a in at:
            out['atomic_kernel_requests_per_launch'].update({c: a[c] for c in a if c.startswith('TCC_')})
        out['atomic_kernel_note'] = ('round-1 fp32-atomic scatter (k_encode_bwd, one launch per renderer pass = 524288 rows): L2 atomic '
                                     'requests per launch; at 1.47 ms per launch (profiles/r01_kernel_stats_rocprofv3.csv) that is the '
                                     'request rate the atomic units deliver (profiles/r02_probe_atomics4.txt: 21 G/s)')
lseg = os.path.join(ROOT, 'gpurun_out', prefix + 'lseg512_mfma.txt')
if os.path.exists(lseg):   # bench.py --feature-dim 512 under --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (wide.hip kernels)
    wk, k = {}, None
    for line in open(lseg):
        if not line.startswith(' '):
            k = line.strip()
        else:
            m = re.match(r'\s+(\S+)\s+n=\s*(\d+) avg=(\S+)', line)
            wk.setdefault(k, {})[m.group(1)] = float(m.group(3))
    for k, e in wk.items():
        e['mfma_util'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] / 8 * 1024)
    out['lseg512_wide_kernels'] = wk
json.dump(out, open(os.path.join(ROOT, 'profiles', rnd + '_pmc_summary.json'), 'w'), indent=1)
for k, e in out['kernels'].items():
    if 'mfma_util' in e and e['mfma_util'] > 0:
        print(f"{k[:70]:70s} mfma_util {e['mfma_util']:.3f} lds_conflict {e.get('lds_conflict_frac', 0):.2f}")
print({k: v for k, v in out.items() if k.startswith(('encode_bwd_traffic_b', 'atomic_kernel_req', 'k_encode'))})
