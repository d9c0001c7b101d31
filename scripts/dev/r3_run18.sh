#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality.py) > $O/r3_pytest18.log 2>&1; tail -8 $O/r3_pytest18.log | cut -c1-300
python bench.py --no-cpu-baseline --no-pmc --no-lseg --quality-steps 0 --render-frames 0 > $O/r3_bench_m.json 2> $O/r3_bench_m.err; tail -3 $O/r3_bench_m.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_m.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['pair_without_optimizer_us'], 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
m=d['marching']; print('march', m['value'], m['ms_per_step'])
P
for b in 1024; do python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 --event-steps 0 --batch $b 2>/dev/null | python -c "
import json,sys; e=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('B',$b,e['value'],e['ms_per_step'])"; done
