#!/bin/bash
# round 4: GPU suite without the long quality gate, then a default bench run
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_quality.py > $OUT/r4_tests.txt 2>&1
tail -5 $OUT/r4_tests.txt
timeout 900 python3 bench.py > $OUT/r4_bench_a.json 2> $OUT/r4_bench_a.err
tail -c 3000 $OUT/r4_bench_a.json
