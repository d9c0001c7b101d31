#!/bin/bash
# usage: prof_kernels.sh <name> <steps> <python script> [args...]
# rocprofv3 --kernel-trace --stats of a python program -> gpurun_out/<name>_kernel_stats.csv + a per-step table of this library's kernels
# (kernels whose call count is a multiple of <steps>).  The program itself follows `--` directly (no env / bash hop: gpurun rules).
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
NAME=$1; STEPS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
case "$1" in /*) PROG=$1;; *) PROG=$R/$1;; esac; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -- python3 $PROG "$@" > $O/${NAME}_prof.log 2>&1 < /dev/null
cp /tmp/prof_$NAME/*/*kernel_stats.csv $O/${NAME}_kernel_stats.csv
python3 - "$O/${NAME}_kernel_stats.csv" "$STEPS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = int(sys.argv[2]); tot = 0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    n, c, t = r['Name'], int(r['Calls']), float(r['TotalDurationNs'])
    if c % steps == 0 and c > 0 and not n.startswith(('at::', 'Cijk', 'void at', '__amd')):
        us = t / steps / 1e3; tot += us
        if us >= 1.0: print(f'{us:9.1f} us/step  x{c // steps:<3d} {n[:110]}')
print(f'{tot:9.1f} us/step  sum')
PY
