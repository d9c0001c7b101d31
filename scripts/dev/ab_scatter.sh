#!/bin/bash
# per-kernel time of the binned scatter inside the training step (rocprofv3) + the binned-scatter tests
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/abscatter
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x -m gpu -k "binned or encode or scatter or adjoint" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
tag=${1:-cur}
rm -rf /tmp/p_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err < /dev/null
f=$(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag: $(python3 -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print(d['ms_per_step'],'ms/step')" 2>&1 | tail -1)"
if [ -n "$f" ]; then cp $f $OUT/kernel_stats_$tag.csv; grep -E "encode_bwd" $f | cut -d, -f1-4 | cut -c1-150; fi
