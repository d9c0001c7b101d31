#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q -m gpu -k "binned or scatter or encode or adjoint or grid_adam or reproducible" > $OUT/r4_tests_d.txt 2>&1
tail -4 $OUT/r4_tests_d.txt
timeout 600 python3 bench.py --steps 60 --warmup 20 --no-march --no-lseg --no-cpu-baseline --quality-steps 0 --render-frames 0 --no-pmc --no-dropin > $OUT/r4_bench_b.json 2> $OUT/r4_bench_b.err
python3 -c "
import json;d=json.load(open('$OUT/r4_bench_b.json'));print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['avg_launch_us'], r['frac'], r.get('phase_us')); print(d['roofline_mlp']['us_per_step'])"
