#!/bin/bash
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0   # graph mode under rocprofv3: the tool library brings HIP up before Python can set it
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality.py) > $O/r3_pytest16.log 2>&1; tail -12 $O/r3_pytest16.log | cut -c1-300
python bench.py --no-cpu-baseline --no-pmc --no-lseg --quality-steps 0 --render-frames 0 > $O/r3_bench_l.json 2> $O/r3_bench_l.err; tail -3 $O/r3_bench_l.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_l.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
m=d['marching']; print('march', m['value'], m['ms_per_step'])
P
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03m_stats.csv; grep "k_adam\|encode_bwd" $O/r03m_stats.csv | awk -F, '{printf "%s=%.1f ", substr($1,2,18), $4/1000}'; echo
