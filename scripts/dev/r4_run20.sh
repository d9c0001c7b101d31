#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1500 python3 bench.py > $OUT/r4_bench_full.json 2> $OUT/r4_bench_full.err
timeout 300 python3 bench.py --batch 1024 --no-cpu-baseline --no-lseg --no-dropin --quality-steps 0 --no-pmc --no-march --render-frames 0 > $OUT/r4_bench_b1024.json 2>/dev/null
timeout 300 python3 bench.py --batch 8192 --no-cpu-baseline --no-lseg --no-dropin --quality-steps 0 --no-pmc --no-march --render-frames 0 > $OUT/r4_bench_b8192.json 2>/dev/null
python3 - <<'PY'
import json,os
o=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'
d=json.loads(open(o+'r4_bench_full.json').read().strip().split('\n')[-1])
print('value',d['value'],d['ms_per_step'],'mlp',d['roofline_mlp']['us_per_step'],d['roofline_mlp']['frac'],'scatter',d['roofline']['avg_launch_us'],d['roofline']['frac'])
print('quality',{k:d['quality'][k] for k in d['quality'] if 'mean' in k or 'min' in k})
m=d['marching']; print('march',m['value'],m['ms_per_step'],m['render_Mrays_per_s'],{k:m['quality'][k] for k in m['quality'] if 'mean' in k})
print('lseg',d['lseg']['value'],d['lseg']['ms_per_step'],d['lseg']['roofline_mlp']['frac'])
print('dropin',d['dropin']['value'],'render dense',d['render_dense_Mrays_per_s'],'cpu',d['cpu_baseline']['value'])
for f in ('r4_bench_b1024.json','r4_bench_b8192.json'):
    e=json.loads(open(o+f).read().strip().split('\n')[-1]); print(f,e['value'],e['ms_per_step'])
PY
