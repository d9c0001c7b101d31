#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
stats() {
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
echo "== $1 $(grep "$2" /tmp/st/*/*kernel_stats.csv | awk -F, '{printf "%s=%.1f ", substr($1,2,24), $4/1000}')"
cd $R
}
PAT=$1; shift
cp autolabel_amd/csrc/libautolabel_hip.so /tmp/lib_product.so
stats product "$PAT"
for v in "$@"; do cp scripts/dev/_build/lib_$v.so autolabel_amd/csrc/libautolabel_hip.so; stats $v "$PAT"; done
cp /tmp/lib_product.so autolabel_amd/csrc/libautolabel_hip.so

