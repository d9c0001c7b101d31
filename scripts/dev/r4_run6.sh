#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parallel.py tests/test_gpu_api.py -x -q -m gpu -k "parallel or ranks or bench or resident or loss or reproducible" > $OUT/r4_tests_c.txt 2>&1
tail -15 $OUT/r4_tests_c.txt
