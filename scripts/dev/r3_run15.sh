#!/bin/bash
# kernel stats of the marching leg (graph replay) -- main leg shortened to nothing
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --render-frames 0 --event-steps 0 --quality-steps 300 --quality-seeds 1 --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
tail -2 /tmp/st.log | cut -c1-300
cp /tmp/st/*/*kernel_stats.csv $O/r03l_march_kernel_stats.csv
grep -v "at::native\|Cijk\|rocclr" $O/r03l_march_kernel_stats.csv | awk -F'",' '{split($2,a,","); printf "%-64s %6s %9.1f %10.0f\n", substr($1,2,64), a[1], a[3]/1000, a[2]/1000}' | head -50
