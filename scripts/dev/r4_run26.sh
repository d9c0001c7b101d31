#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parallel.py -x -q -m gpu -k "bench" > $OUT/r4_tests_o.txt 2>&1; tail -6 $OUT/r4_tests_o.txt
