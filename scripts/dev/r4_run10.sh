#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -x -q -m gpu -k "binned or fused or grid_adam or adam" > $OUT/r4_tests_f.txt 2>&1
tail -8 $OUT/r4_tests_f.txt
timeout 600 python3 bench.py --no-cpu-baseline --no-lseg --no-dropin --quality-steps 0 --no-pmc > $OUT/r4_bench_b.json 2> $OUT/r4_bench_b.err
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_bench_b.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['timed_state'], d['live_row_fraction'])
print(d['roofline']['bound'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
print(d['roofline_mlp']['kernels'], d['roofline_mlp']['us_per_step'], d['roofline_mlp']['frac'])
print(d['roofline_render'])
print(d['marching'].get('roofline_render'))
PY
