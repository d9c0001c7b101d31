#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for h in sigma color; do
  timeout 300 python3 scripts/dev/bench_mlp_fwd.py --head $h 2>&1 | grep -v amdgpu.ids
  timeout 300 python3 scripts/dev/bench_mlp_fwd.py --head $h --lib scripts/dev/_build/lib_fwdold.so 2>&1 | grep -v amdgpu.ids
done
