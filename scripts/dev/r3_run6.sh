#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests -m gpu -q --durations=8) > $O/r3_pytest6.log 2>&1; tail -22 $O/r3_pytest6.log | cut -c1-200
python bench.py --no-cpu-baseline --no-pmc > $O/r3_bench_e.json 2> $O/r3_bench_e.err; tail -3 $O/r3_bench_e.err
for b in 1024 8192; do python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 --event-steps 0 --batch $b > $O/r3_bench_B$b.json 2>/dev/null; done
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_e.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
m=d['marching']; print('march', m['value'], m['ms_per_step'], m['steps'], m['render_Mrays_per_s']); print('lseg', d['lseg']['ms_per_step'])
for b in (1024, 8192):
    e=json.loads(open('gpurun_out/r3_bench_B%d.json'%b).read().strip().split('\n')[-1]); print('B',b,e['value'],e['ms_per_step'])
P
bash scripts/dev/run_pmc_r03.sh 2>&1 | tail -25
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03c_train_kernel_stats.csv; grep -v "at::native\|Cijk\|rocclr" $O/r03c_train_kernel_stats.csv | cut -c1-120 | head -30
