#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "semantic" > $OUT/r4_tests_g.txt 2>&1
tail -15 $OUT/r4_tests_g.txt
timeout 300 python3 scripts/dev/bench_sem_pair.py 2>&1 | tail -1
timeout 300 python3 scripts/dev/bench_sem_pair.py --dots 2>&1 | tail -1
