#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/scripts/dev/lseg_steps.py > /tmp/st.log 2>&1 < /dev/null
tail -2 /tmp/st.log | cut -c1-200
cp /tmp/st/*/*kernel_stats.csv $O/r03n_lseg_kernel_stats.csv
grep -v "at::native\|Cijk\|rocclr" $O/r03n_lseg_kernel_stats.csv | awk -F'",' '{split($2,a,","); printf "%-72s %6s %9.1f %10.0f\n", substr($1,2,72), a[1], a[3]/1000, a[2]/1000}' | head -40
