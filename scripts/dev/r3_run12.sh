#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_api.py -m gpu -q -x) > $O/r3_pytest12.log 2>&1; tail -6 $O/r3_pytest12.log | cut -c1-200
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03i_stats.csv; grep -v "at::native\|Cijk\|rocclr" $O/r03i_stats.csv | awk -F'",' '{split($2,a,","); printf "%-70s %s %.1f\n", substr($1,2,70), a[1], a[3]/1000}' | head -30
cd $R
python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 > $O/r3_bench_j.json 2> $O/r3_bench_j.err; tail -3 $O/r3_bench_j.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_j.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
P
