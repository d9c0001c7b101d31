#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/r4_tests_final.txt 2>&1
tail -5 $OUT/r4_tests_final.txt
timeout 500 python3 __graft_entry__.py smoke 2>&1 | tail -1
timeout 1500 python3 bench.py > $OUT/r4_bench_final.json 2> $OUT/r4_bench_final.err
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_bench_final.json').read().strip().split('\n')[-1])
print('value',d['value'],d['ms_per_step'],'mlp',d['roofline_mlp']['us_per_step'],d['roofline_mlp']['frac'],'scatter',d['roofline']['avg_launch_us'],d['roofline']['frac'])
print('march',d['marching']['value'],'lseg',d['lseg']['ms_per_step'],'dropin',d['dropin']['value'])
PY
