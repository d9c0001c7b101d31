#!/bin/bash
# A/B: semantic-head backward at one vs two blocks per CU (ALN_MLP_OCC2), per-kernel times from rocprofv3
# usage: ab_occ2.sh "0 1"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/occ2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${1:-0 1}; do
  export ALN_MLP_OCC2=$v
  rm -rf /tmp/p$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$v -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march > $OUT/bench_$v.json 2> $OUT/bench_$v.err < /dev/null
  f=$(find /tmp/p$v -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/kernel_stats_$v.csv; grep -E "recomp8|k_sem_fwd|encode_bwd" $f | cut -d, -f1-4 | cut -c1-160; fi
done
