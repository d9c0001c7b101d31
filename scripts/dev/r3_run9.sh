#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality.py) > $O/r3_pytest9.log 2>&1; tail -6 $O/r3_pytest9.log | cut -c1-200
stats() {  # tag, batch
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc --batch $2 > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03f_stats_$1_B$2.csv; echo "== $1 B=$2"; grep "k_adam\|composite_bwd\|raygen_train" $O/r03f_stats_$1_B$2.csv | cut -d, -f1-4 | cut -c1-40,100-
cd $R
}
cp autolabel_amd/csrc/libautolabel_hip.so /tmp/lib_product.so
stats product 4096; stats product 1024
for v in cb24 cb22 cb444; do cp scripts/dev/_build/lib_$v.so autolabel_amd/csrc/libautolabel_hip.so; stats $v 4096; stats $v 1024; done
cp /tmp/lib_product.so autolabel_amd/csrc/libautolabel_hip.so
python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 > $O/r3_bench_h.json 2> $O/r3_bench_h.err; tail -3 $O/r3_bench_h.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_h.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
P
