#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{ echo "=== product"; timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head sigma 2>&1 | grep -E "median"
for lib in scripts/dev/_build/lib_ab_*.so; do
  echo "=== $lib"
  timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head sigma --lib $lib 2>&1 | grep -E "median"
done; } > $OUT/r4_run3.txt 2>&1
cat $OUT/r4_run3.txt
