#!/bin/bash
# full GPU suite + default bench (dev helper; results under gpurun_out/)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/r4_tests_full.txt 2>&1
tail -6 $OUT/r4_tests_full.txt
timeout 900 python3 bench.py > $OUT/r4_bench.json 2> $OUT/r4_bench.err
tail -c 3000 $OUT/r4_bench.json
