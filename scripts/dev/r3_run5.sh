#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -q --deselect tests/test_gpu_quality.py > $O/r3_pytest5.log 2>&1; tail -15 $O/r3_pytest5.log | cut -c1-300
python bench.py --no-cpu-baseline --no-pmc > $O/r3_bench_c.json 2> $O/r3_bench_c.err; tail -3 $O/r3_bench_c.err
python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 --no-lseg > $O/r3_bench_d.json 2> $O/r3_bench_d.err; tail -3 $O/r3_bench_d.err
python - <<'P'
import json
for f in ('c','d'):
    d=json.loads(open('gpurun_out/r3_bench_%s.json'%f).read().strip().split('\n')[-1])
    q=d.get('quality') or {}
    print(f,'value', d['value'], 'ms', d['ms_per_step'], 'roof', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'))
    print('  quality', [ (round(r['psnr_db'],3), round(r['miou'],4)) for r in q.get('runs',[])])
    m=d.get('marching') or {}
    print('  march', m.get('value'), m.get('ms_per_step'), [ (round(r['psnr_db'],3), round(r['miou'],4)) for r in (m.get('quality') or {}).get('runs',[])], (m.get('roofline') or {}).get('avg_launch_us'))
    print('  lseg', {k:v for k,v in (d.get('lseg') or {}).items() if k in ('ms_per_step','linear_last_layer_per_ray')}, (d.get('lseg') or {}).get('roofline_mlp',{}).get('frac'))
    print('  mlp', d.get('roofline_mlp',{}).get('frac'), d.get('roofline_mlp',{}).get('us_per_step'))
P
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03b_train_kernel_stats.csv; grep -v "at::native\|Cijk\|rocclr" $O/r03b_train_kernel_stats.csv | cut -c1-150 | head -34
