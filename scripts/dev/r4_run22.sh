#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_api.py tests/test_gpu_march.py -x -q -m gpu > $OUT/r4_tests_k.txt 2>&1; tail -4 $OUT/r4_tests_k.txt
timeout 600 python3 bench.py --no-cpu-baseline --no-lseg --no-dropin --quality-steps 0 --no-pmc --no-march --render-frames 4 > $OUT/r4_bench_e.json 2> $OUT/r4_bench_e.err
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_bench_e.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['render_dense_Mrays_per_s'])
PY
