#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_api.py -m gpu -q -x) > $O/r3_pytest11.log 2>&1; tail -6 $O/r3_pytest11.log | cut -c1-200
stats() {  # tag, batch
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc --batch $2 > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03h_stats_$1_B$2.csv; echo "== $1 B=$2 $(grep 'k_adam\|encode_bwd' $O/r03h_stats_$1_B$2.csv | awk -F, '{printf "%s=%.1f ", substr($1,2,18), $4/1000}')"
cd $R
}
cp autolabel_amd/csrc/libautolabel_hip.so /tmp/lib_product.so
stats product 4096
for v in dd10 dd12 dd16; do cp scripts/dev/_build/lib_$v.so autolabel_amd/csrc/libautolabel_hip.so; stats $v 4096; 
python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['ms_per_step'], d['roofline'].get('records_per_launch'), d['roofline']['avg_launch_us'])"
done
cp /tmp/lib_product.so autolabel_amd/csrc/libautolabel_hip.so
python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 > $O/r3_bench_i.json 2> $O/r3_bench_i.err; tail -3 $O/r3_bench_i.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_i.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline'], 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
P
