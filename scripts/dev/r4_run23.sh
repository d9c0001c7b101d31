#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wide" > $OUT/r4_tests_l.txt 2>&1; tail -4 $OUT/r4_tests_l.txt
timeout 300 python3 scripts/dev/bench_wide.py 2>&1 | grep -E "nt |tn "
