#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_march.py tests/test_gpu_api.py -q -x 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
echo "== $(grep "grid_phased\|encode_xn\|encode_assemble" /tmp/st/*/*kernel_stats.csv | awk -F, '{printf "%s=%.1f ", substr($1,2,24), $4/1000}')"
cd $R; python bench.py --no-cpu-baseline --no-pmc --no-lseg --quality-steps 0 --render-frames 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench', d['value'], d['ms_per_step'], 'march', d['marching']['value'], d['marching']['ms_per_step'], 'render', d['render_Mrays_per_s'], d.get('render_dense_Mrays_per_s'))"
