#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_api.py -x -q -m gpu -k "subset or override or resident or scripts_train or loss" > $OUT/r4_tests_b.txt 2>&1
tail -15 $OUT/r4_tests_b.txt
timeout 600 python3 bench.py --steps 100 --warmup 20 --no-march --no-lseg --no-cpu-baseline --quality-steps 0 --render-frames 0 --no-pmc --event-steps 0 > $OUT/r4_bench_dropin.json 2> $OUT/r4_bench_dropin.err
python3 -c "
import json;d=json.load(open('$OUT/r4_bench_dropin.json'));print(d['value'], d['ms_per_step']); print(json.dumps(d.get('dropin'), indent=1))"
tail -3 $OUT/r4_bench_dropin.err
