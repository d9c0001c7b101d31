#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "semantic_heads" > $OUT/r4_tests_e.txt 2>&1
tail -5 $OUT/r4_tests_e.txt
timeout 300 python3 scripts/dev/bench_sem_pair.py 2>&1 | tail -2
timeout 300 python3 scripts/dev/bench_sem_pair.py --lib scripts/dev/_build/lib_ptpair.so 2>&1 | tail -5
