R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/stm
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stm -- python3 $R/scripts/dev/march_steps.py > /tmp/stm.log 2>&1
cp /tmp/stm/*/*kernel_stats.csv $R/gpurun_out/r02_march_kernel_stats.csv; tail -2 /tmp/stm.log
