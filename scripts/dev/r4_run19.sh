#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
for lib in "" "--lib scripts/dev/_build/lib_trold.so"; do
  echo "== $lib"
  timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head sigma $lib 2>&1 | grep -E "median"
  timeout 300 python3 scripts/dev/bench_mlp_bwd.py --head color $lib 2>&1 | grep -E "median"
  timeout 300 python3 scripts/dev/bench_sem_pair.py --dots $lib 2>&1 | grep -E "median"
done; done
