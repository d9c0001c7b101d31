#!/bin/bash
# Profile collection of a round (usage: run_profiles.sh r05).  Kernel traces (rocprofv3 --kernel-trace --stats) of the training step (launch by launch, batch 4096
# and 1024), the dense render, the marching step + render and the LSeg step; then one rocprofv3 --pmc pass per counter set (no trace
# domains in those).  Outputs under gpurun_out/: <round>_*_kernel_stats.csv, pmc<N>_<set>.txt; summary: make_pmc_summary.py <round>
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
RND=${1:-r05}; PN=pmc$((10#${RND#r}))
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-graph --event-steps 0 --quality-steps 0 --no-pmc --no-dropin --no-dp1"
trace() {   # name, bench args...
  local name=$1; shift; rm -rf /tmp/st
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py "$@" > /tmp/st_$name.log 2>&1 < /dev/null
  cp /tmp/st/*/*kernel_stats.csv $O/${RND}_${name}_kernel_stats.csv && echo "== $name: $(wc -l < $O/${RND}_${name}_kernel_stats.csv) kernels"
}
trace train --steps 50 --warmup 10 $COMMON --render-frames 0 --no-march --no-lseg
trace train_B1024 --steps 50 --warmup 10 --batch 1024 $COMMON --render-frames 0 --no-march --no-lseg
trace render --steps 2 --warmup 1 $COMMON --render-frames 4 --no-march --no-lseg
trace march --steps 2 --warmup 1 $COMMON --render-frames 4 --no-lseg
trace lseg512 --steps 20 --warmup 5 $COMMON --render-frames 0 --no-march --no-lseg --feature-dim 512
# the lseg LEG's configuration (semantic_weight 0: linear last layer per ray): 30 launch-by-launch steps
rm -rf /tmp/st; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/scripts/dev/lseg_steps.py > /tmp/st_lseg_leg.log 2>&1 < /dev/null
cp /tmp/st/*/*kernel_stats.csv $O/${RND}_lseg_leg_kernel_stats.csv
BENCH="$R/bench.py --steps 3 --warmup 1 $COMMON --render-frames 0 --no-march --no-lseg"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '+'); rm -rf /tmp/pmc
  timeout 170 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python3 $BENCH > /tmp/pmc.log 2>&1
  python3 $R/scripts/dev/pmc_csv_summary.py /tmp/pmc > $O/${PN}_$name.txt; echo "== $name: $(grep -c n= $O/${PN}_$name.txt) rows"
done
rm -rf /tmp/pmc
timeout 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -- python3 $BENCH --feature-dim 512 > /tmp/pmc.log 2>&1
python3 $R/scripts/dev/pmc_csv_summary.py /tmp/pmc k_wide > $O/${PN}_lseg512_mfma.txt; echo "== lseg512: $(grep -c n= $O/${PN}_lseg512_mfma.txt) rows"
# (summary: run scripts/dev/make_pmc_summary.py r04 where profiles/ is tracked)
ls -la $O | grep -E "r04_|${PN}_" | head -30
