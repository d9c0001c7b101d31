#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
stats() {  # tag, batch
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc --batch $2 > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03g_stats_$1_B$2.csv; echo "== $1 B=$2 $(grep 'k_adam\|composite_bwd\|raygen_train' $O/r03g_stats_$1_B$2.csv | awk -F, '{printf "%s=%.1f ", substr($1,2,18), $4/1000}')"
cd $R
}
cp autolabel_amd/csrc/libautolabel_hip.so /tmp/lib_product.so
stats product 4096
for v in a00 a01 a10; do cp scripts/dev/_build/lib_$v.so autolabel_amd/csrc/libautolabel_hip.so; stats $v 4096; done
cp /tmp/lib_product.so autolabel_amd/csrc/libautolabel_hip.so; stats product 4096
