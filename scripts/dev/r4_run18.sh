#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mlp or semantic" > $OUT/r4_tests_j.txt 2>&1; tail -4 $OUT/r4_tests_j.txt
timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head sigma 2>&1 | grep -E "median|rel err|bit-identical|rows_dev"
timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head color 2>&1 | grep -E "median|rel err|bit-identical|rows_dev"
timeout 300 python3 scripts/dev/bench_sem_pair.py --dots 2>&1 | grep -E "median"
