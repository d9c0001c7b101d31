#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parallel.py -x -q -m gpu > $OUT/r4_tests_par.txt 2>&1
tail -25 $OUT/r4_tests_par.txt
