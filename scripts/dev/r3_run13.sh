#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp autolabel_amd/csrc/libautolabel_hip.so /tmp/lib_product.so
for v in product "$@"; do
  if [ $v = product ]; then cp /tmp/lib_product.so autolabel_amd/csrc/libautolabel_hip.so; else cp scripts/dev/_build/lib_$v.so autolabel_amd/csrc/libautolabel_hip.so; fi
  echo "== $v"; python scripts/dev/probe_scatter_levels.py 2>&1 | grep -v amdgpu.ids
  python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench $v', d['ms_per_step'], d['roofline'].get('records_per_launch'), d['roofline']['avg_launch_us'])"
done
cp /tmp/lib_product.so autolabel_amd/csrc/libautolabel_hip.so
