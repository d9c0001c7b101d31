#!/bin/bash
# A/B of the MLP backward switches, per-kernel times from rocprofv3.  usage: ab_mlp.sh "<tag>:<ENV=VAL,...> ..."
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/abmlp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  for kv in ${envs//,/ }; do [ -n "$kv" ] && export $kv; done
  rm -rf /tmp/p_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err < /dev/null
  f=$(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag ($envs): $(python3 -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print(d['ms_per_step'],'ms/step')" 2>&1 | tail -1)"
  if [ -n "$f" ]; then cp $f $OUT/kernel_stats_$tag.csv; grep -E "recomp8|k_dw_reduce|k_sem_fwd" $f | cut -d, -f1-4 | cut -c1-150; fi
  for kv in ${envs//,/ }; do [ -n "$kv" ] && unset ${kv%%=*}; done
done
