#!/bin/bash
# round 4, run 1: feature-sliced 128-wide backward against round 3's kernel (same box)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
for head in sigma color; do
  echo "== new $head"; timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head $head 2>&1 | tail -12
  echo "== old $head"; timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head $head --lib scripts/dev/_build/lib_old128.so 2>&1 | tail -12
done > $OUT/r4_run1.txt 2>&1
cat $OUT/r4_run1.txt
