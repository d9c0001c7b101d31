#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python3 scripts/dev/bench_sem_pair.py --lib scripts/dev/_build/lib_ptpair.so 2>&1 | tail -5
timeout 300 python3 scripts/dev/bench_sem_pair.py --dots --lib scripts/dev/_build/lib_ptpair.so 2>&1 | tail -5
