#!/bin/bash
# Same-box A/B of whole LIBRARIES (the product has no switches): ab_libs.sh <reps> <tag>=<lib.so> ...
# Every variant is copied over the in-tree library of THIS scratch copy in turn; per variant: the replayed training step (bench.py,
# hipGraph) <reps> times alternating with the others, then ONE launch-by-launch rocprofv3 kernel trace (per-kernel averages).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/ab; mkdir -p $O
LIB=$R/autolabel_amd/csrc/libautolabel_hip.so
cp $LIB /tmp/lib_product.so
reps=$1; shift
A="--steps 200 --warmup 30 --no-cpu-baseline --no-pmc --no-march --no-lseg --no-dropin --no-dp1 --quality-steps 0 --render-frames 0 --event-steps 0"
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    tag=${spec%%=*}; lib=${spec#*=}
    [ "$lib" = "product" ] && lib=/tmp/lib_product.so
    cp $lib $LIB
    timeout 300 python3 bench.py $A $AB_EXTRA 2>$O/$tag.err | tail -1 > $O/$tag.$rep.json
    python3 -c "
import json;d=json.load(open('$O/$tag.$rep.json'));print('$tag rep$rep', round(d['ms_per_step'],4),'ms/step', round(d['value']), 'rays/s')" 2>&1 | tail -1
  done
done
if [ -z "$AB_NO_TRACE" ]; then
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%=*}; lib=${spec#*=}
  [ "$lib" = "product" ] && lib=/tmp/lib_product.so
  cp $lib $LIB
  rm -rf /tmp/p_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-march --no-lseg --no-dropin --no-dp1 --quality-steps 0 --render-frames 0 --event-steps 0 --no-graph $AB_EXTRA > $O/trace_$tag.json 2> $O/trace_$tag.err < /dev/null
  f=$(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/kernel_stats_$tag.csv; echo "== $tag"; python3 $R/scripts/dev/prof_step.py $f; fi
done
fi
cp /tmp/lib_product.so $LIB
