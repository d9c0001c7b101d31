#!/bin/bash
# per-level durations of the two scatter kernels: rocprofv3 kernel trace of probe_scatter_levels.py, launches in program order
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/scripts/dev/probe_scatter_levels.py > /tmp/tr.log 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob('/tmp/tr/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_encode_bwd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(r['Kernel_Name'][:18], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000) for r in rows]
# program order: 1 launch pair (count), then 10 x all levels, then per level 10 pairs each, then 4 groups x 10
pairs = [(d[i][1], d[i + 1][1]) for i in range(0, len(d), 2)]
import statistics as st
def med(ps): return (st.median(p[0] for p in ps), st.median(p[1] for p in ps))
k = 1
print('all: bin %.0f accum %.0f' % med(pairs[k:k + 10])); k += 10
for l in range(16):
    print('L%d: bin %.0f accum %.0f' % ((l,) + med(pairs[k:k + 10]))); k += 10
for g in ('0-4', '4-8', '8-12', '12-16'):
    print('L%s: bin %.0f accum %.0f' % ((g,) + med(pairs[k:k + 10]))); k += 10
P
