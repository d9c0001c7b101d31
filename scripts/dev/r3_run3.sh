#!/bin/bash
# round 3, GPU call 3: tests touched by the fusions, full default bench (new legs), quality-gate regime probe
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -q --deselect tests/test_gpu_quality.py > $O/r3_pytest3.log 2>&1; tail -15 $O/r3_pytest3.log | cut -c1-300
(time python bench.py) > $O/r3_bench_b.json 2> $O/r3_bench_b.err; tail -5 $O/r3_bench_b.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench_b.json').read().strip().split('\n')[-1])
q=d.get('quality') or {}
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), 'traffic', d.get('roofline',{}).get('traffic'), d.get('roofline',{}).get('traffic_detail'))
print('quality', {k:v for k,v in q.items() if k!='runs' and k!='note'})
m=d.get('marching') or {}
print('march', m.get('value'), m.get('ms_per_step'), m.get('render_Mrays_per_s'), {k:v for k,v in (m.get('quality') or {}).items() if k not in('runs','note')}, (m.get('roofline') or {}).get('frac'), (m.get('roofline') or {}).get('avg_launch_us'), (m.get('roofline') or {}).get('records_per_launch'))
print('lseg', {k:v for k,v in (d.get('lseg') or {}).items() if k!='note'})
print('cpu', d.get('cpu_baseline'))
print('render', d.get('render_Mrays_per_s'), d.get('render_dense_Mrays_per_s'), 'mlp', d.get('roofline_mlp'))
P
python scripts/dev/quality_gate_probe.py steps=2500 > $O/r3_qgate_a.json 2> $O/r3_qgate_a.err; tail -c 2500 $O/r3_qgate_a.json; tail -3 $O/r3_qgate_a.err
