#!/bin/bash
# kernel-trace statistics of the training step (launch by launch and graph-replayed) -> gpurun_out/r02_train*_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r02_train_kernel_stats.csv; head -4 $O/r02_train_kernel_stats.csv | cut -c1-120
rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --event-steps 0 --quality-steps 0 --no-march > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r02_train_graph_kernel_stats.csv
