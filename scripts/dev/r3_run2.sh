#!/bin/bash
# round 3, GPU call 2: full GPU suite, marching variance, backward phase probes, bench + kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -q --deselect tests/test_gpu_quality.py > $O/r3_pytest2.log 2>&1; tail -30 $O/r3_pytest2.log
python scripts/dev/variance.py --march --runs base,base,noise1,data1,init1 --out gpurun_out/variance_march.json > $O/variance_march.log 2>&1; tail -6 $O/variance_march.log
for h in sigma color semf semo; do python scripts/dev/probe_bwd_phases.py $h; done > $O/r3_bwd_phases.txt 2>&1; cat $O/r3_bwd_phases.txt
python bench.py --steps 200 --warmup 50 --quality-steps 0 --no-march --no-cpu-baseline > $O/r3_bench_a.json 2> $O/r3_bench_a.err; tail -c 1500 $O/r3_bench_a.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march > /tmp/st.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/r03a_train_kernel_stats.csv; grep -v "at::native\|Cijk\|rocclr" $O/r03a_train_kernel_stats.csv | cut -c1-150 | head -40
