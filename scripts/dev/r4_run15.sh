#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for h in sigma color; do
  timeout 300 python3 scripts/dev/bench_mlp_fwd.py --head $h --lib scripts/dev/_build/lib_ptf.so 2>&1 | grep -E "median|per pair"
  timeout 300 python3 scripts/dev/bench_mlp_fwd.py --head $h --rows 2097152 2>&1 | grep -E "median|per pair"
  timeout 300 python3 scripts/dev/bench_mlp_fwd.py --head $h --rows 131072 2>&1 | grep -E "median|per pair"
done
