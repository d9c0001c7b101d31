#!/bin/bash
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/stl
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stl -- python3 $R/scripts/dev/lseg_steps.py > /tmp/stl.log 2>&1 < /dev/null
cp /tmp/stl/*/*kernel_stats.csv $O/r04_lseg_leg_kernel_stats.csv; tail -3 /tmp/stl.log
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_lseg_leg_kernel_stats.csv')))
rows=[r for r in rows if not r['Name'].startswith(('at::','Cijk','void at'))]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:22]: print(f"{float(r['TotalDurationNs'])/1e3:10.0f} us total  calls {r['Calls']:>5}  avg {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:70]}")
PY
