#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_api.py tests/test_gpu_march.py tests/test_gpu_parallel.py -x -q -m gpu > $OUT/r4_tests_n.txt 2>&1; tail -4 $OUT/r4_tests_n.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lseg --no-dropin --quality-steps 0 --no-pmc --render-frames 6 > $OUT/r4_bench_f.json 2> $OUT/r4_bench_f.err
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_bench_f.json').read().strip().split('\n')[-1])
print(d['value'], d['render_dense_Mrays_per_s'], d['render_Mrays_per_s'], d['roofline_render']['frac'])
PY
