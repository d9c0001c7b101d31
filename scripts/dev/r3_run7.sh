#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality.py) > $O/r3_pytest7.log 2>&1; tail -6 $O/r3_pytest7.log | cut -c1-200
python bench.py > $O/r3_bench_f.json 2> $O/r3_bench_f.err; tail -3 $O/r3_bench_f.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_f.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), 'traffic', d['roofline'].get('traffic'), 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
m=d['marching']; print('march', m['value'], m['ms_per_step'], m['steps'], m['render_Mrays_per_s']); print('lseg', d['lseg']['ms_per_step']); print('q', d['quality']); print('cpu', d['cpu_baseline'])
P
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03d_train_kernel_stats.csv; grep -v "at::native\|Cijk\|rocclr" $O/r03d_train_kernel_stats.csv | cut -c1-120 | head -30
