#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_api.py -x -q -m gpu > $OUT/r4_tests_h.txt 2>&1
tail -15 $OUT/r4_tests_h.txt
bash scripts/dev/r4_prof_train.sh 2>&1 | tail -30
timeout 600 python3 bench.py --no-cpu-baseline --no-lseg --no-dropin --quality-steps 0 --no-pmc --no-march --render-frames 0 > $OUT/r4_bench_c.json 2> $OUT/r4_bench_c.err
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_bench_c.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline_mlp']['us_per_step'], d['roofline_mlp']['frac'])
PY
