#!/bin/bash
# round 4: kernel-trace statistics of the training step, launch by launch -> gpurun_out/r04_train_kernel_stats.csv
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc --no-dropin --no-dp1 > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r04_train_kernel_stats.csv
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_train_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']; c=int(r['Calls']); t=float(r['TotalDurationNs'])
    if c%60==0 and c>0 and not n.startswith(('at::','Cijk','void at')):
        us=t/60/1e3; tot+=us
        print(f'{us:8.1f} us/step  x{c//60}  {n[:90]}')
print('sum', tot)
PY
