#!/bin/bash
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0   # graph mode under rocprofv3: the tool library brings HIP up before Python can set it
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(time python -m pytest tests -m gpu -q --durations=6) > $O/r3_pytest14.log 2>&1; tail -14 $O/r3_pytest14.log | cut -c1-200
python bench.py > $O/r3_bench_k.json 2> $O/r3_bench_k.err; tail -3 $O/r3_bench_k.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_k.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'traffic', d['roofline'].get('traffic'), 'mlp', d['roofline_mlp']['frac'], d['roofline_mlp']['us_per_step'])
m=d['marching']; print('march', m['value'], m['ms_per_step'], m['steps'], m['render_Mrays_per_s'], m.get('quality')); print('lseg', d['lseg']); print('q', d['quality']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])
P
for b in 1024 8192; do python bench.py --no-cpu-baseline --no-pmc --no-march --no-lseg --quality-steps 0 --render-frames 0 --event-steps 0 --batch $b > $O/r3_bench_k_B$b.json 2>/dev/null; python -c "
import json; e=json.loads(open('gpurun_out/r3_bench_k_B$b.json').read().strip().split('\n')[-1]); print('B',$b,e['value'],e['ms_per_step'])"; done
bash scripts/dev/run_pmc_r03.sh 2>&1 | tail -12
cd /tmp; export TMPDIR=/tmp
for b in 4096 1024; do rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc --batch $b > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03k_train_kernel_stats_B$b.csv; done
rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03k_train_graph_kernel_stats.csv
