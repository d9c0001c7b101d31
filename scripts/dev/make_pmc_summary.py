"""profiles/r01_pmc_summary.json from the per-pass summaries written by pmc_csv_summary.py (gpurun_out/pmc_*.txt)."""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vals = {}
for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_*.txt'))):
    k = None
    for line in open(f):
        if not line.startswith(' '):
            k = line.strip()
        else:
            m = re.match(r'\s+(\S+)\s+n=\s*(\d+) avg=(\S+)', line)
            vals.setdefault(k, {})[m.group(1)] = {'avg': float(m.group(3)), 'n': int(m.group(2))}
ours = {k: v for k, v in vals.items() if k.startswith(('_Z', 'k_')) and 'at::' not in k}
out = {'command': 'one rocprofv3 --pmc pass per counter set around `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline '
                  '--render-frames 0` (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_BANK_CONFLICT '
                  'SQ_LDS_IDX_ACTIVE); kernel durations: r01_kernel_stats_rocprofv3.csv (rocprofv3 --kernel-trace --stats, '
                  'bench.py --steps 50 --warmup 10)',
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
    out['kernels'][k] = e
enc = next(v for k, v in out['kernels'].items() if 'k_encode_bwd' in k)
out['k_encode_bwd_traffic_bytes_per_launch'] = (enc['FETCH_SIZE'] + enc['WRITE_SIZE']) * 1024
json.dump(out, open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json'), 'w'), indent=1)
for k, e in out['kernels'].items():
    if 'mfma_util' in e and e['mfma_util'] > 0:
        print(f"{k[:70]:70s} mfma_util {e['mfma_util']:.3f} lds_conflict {e.get('lds_conflict_frac', 0):.2f}")
print('encode_bwd traffic/launch', out['k_encode_bwd_traffic_bytes_per_launch'])
