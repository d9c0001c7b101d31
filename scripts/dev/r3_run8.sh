#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests -m gpu -q -x -k "adam or step or train or reproduc or graph" --deselect tests/test_gpu_quality.py) > $O/r3_pytest8.log 2>&1; tail -6 $O/r3_pytest8.log | cut -c1-200
python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 > $O/r3_bench_g.json 2> $O/r3_bench_g.err; tail -3 $O/r3_bench_g.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_g.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
P
cd /tmp; export TMPDIR=/tmp
for b in 1024 4096; do
rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc --batch $b > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03e_train_kernel_stats_B$b.csv; echo "== B=$b"; grep -v "at::native\|Cijk\|rocclr" $O/r03e_train_kernel_stats_B$b.csv | cut -d, -f1-4 | cut -c1-60,200- | head -32
done
