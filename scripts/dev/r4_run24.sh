#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "semantic" > $OUT/r4_tests_m.txt 2>&1; tail -3 $OUT/r4_tests_m.txt
timeout 300 python3 scripts/dev/bench_sem_pair.py --dots 2>&1 | grep -E "median"
timeout 300 python3 scripts/dev/bench_sem_pair.py --dots --lib scripts/dev/_build/lib_trold.so 2>&1 | grep -E "median"
timeout 300 python3 scripts/dev/bench_sem_pair.py --dots --lib scripts/dev/_build/lib_ptpair.so 2>&1 | grep -E "median|wave"
