#!/bin/bash
# same-box A/B of environment switches on the graph-replayed training step.  usage: ab_bench.sh "<tag>:<ENV=VAL,...>" ...
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/abbench
for rep in 1 2; do
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  for kv in ${envs//,/ }; do [ -n "$kv" ] && export $kv; done
  timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --render-frames 0 --quality-steps 0 --no-march > gpurun_out/abbench/$tag.json 2> gpurun_out/abbench/$tag.err < /dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/abbench/$tag.json'))
print('$tag rep$rep', round(d['ms_per_step'],4), 'ms/step  scatter', round(d['roofline']['avg_launch_us'],1), ' mlp', round(d['roofline_mlp']['us_per_step'],1))"
  for kv in ${envs//,/ }; do [ -n "$kv" ] && unset ${kv%%=*}; done
done; done
