# Round-2 counter passes (one rocprofv3 --pmc run per counter set, kernel trace separately) around the bench's training step.
# Outputs: gpurun_out/pmc2_<set>.txt (per-kernel averages), gpurun_out/r02_train_kernel_stats.csv; summarised by make_pmc_summary.py r02
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
BENCH="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '+'); rm -rf /tmp/pmc
  timeout 170 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python3 $BENCH > /tmp/pmc.log 2>&1
  python3 $R/scripts/dev/pmc_csv_summary.py /tmp/pmc > $O/pmc2_$name.txt; echo "== $name: $(grep -c n= $O/pmc2_$name.txt) rows"
done
# the fp32-atomic scatter of round 1 (ALN_ENC_BWD=atomic): atomic requests reaching L2 per launch
export ALN_ENC_BWD=atomic
for set in "TCC_ATOMIC_sum" "TCC_EA0_ATOMIC_sum"; do
  rm -rf /tmp/pmc
  timeout 170 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python3 $BENCH > /tmp/pmc.log 2>&1
  python3 $R/scripts/dev/pmc_csv_summary.py /tmp/pmc k_encode_bwd > $O/pmc2_atomic_$set.txt; cat $O/pmc2_atomic_$set.txt
done
unset ALN_ENC_BWD
rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march > /tmp/st.log 2>&1
cp /tmp/st/*/*kernel_stats.csv $O/r02_train_kernel_stats.csv; head -5 $O/r02_train_kernel_stats.csv | cut -c1-150
rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --render-frames 0 --event-steps 0 --quality-steps 0 --no-march > /tmp/st.log 2>&1
cp /tmp/st/*/*kernel_stats.csv $O/r02_train_graph_kernel_stats.csv; head -3 $O/r02_train_graph_kernel_stats.csv | cut -c1-150
rm -rf /tmp/st
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --render-frames 6 --no-graph --event-steps 0 --quality-steps 0 --no-march > /tmp/st.log 2>&1
cp /tmp/st/*/*kernel_stats.csv $O/r02_render_kernel_stats.csv
mkdir -p $O/probes; timeout 200 python3 $R/scripts/dev/probe_encode_bwd_binned.py > $O/probes/probe_encode_bwd_binned.txt 2>&1
