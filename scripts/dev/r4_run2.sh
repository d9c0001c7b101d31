#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{ for head in sigma color; do
  timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head $head 2>&1 | grep -v amdgpu.ids
done
timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head sigma --lib scripts/dev/_build/lib_pt.so --phases 2>&1 | grep -E "ticks|   [A-Zl]"
} > $OUT/r4_run2.txt 2>&1
cat $OUT/r4_run2.txt
