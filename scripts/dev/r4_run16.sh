#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_quality.py > $OUT/r4_tests_i.txt 2>&1
tail -8 $OUT/r4_tests_i.txt
timeout 600 python3 bench.py --no-cpu-baseline --no-lseg --no-dropin --quality-steps 0 --no-pmc --no-march --render-frames 4 > $OUT/r4_bench_d.json 2> $OUT/r4_bench_d.err
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_bench_d.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline_mlp']['us_per_step'], d['roofline_mlp']['frac'], d['render_dense_Mrays_per_s'])
PY
