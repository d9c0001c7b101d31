#!/bin/bash
# Round-3 counter passes (one rocprofv3 --pmc run per counter set, no trace domains) around bench.py's training step.
# Outputs: gpurun_out/pmc3_<set>.txt (per-kernel averages); summarised by scripts/dev/make_pmc_summary.py r03
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
BENCH="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '+'); rm -rf /tmp/pmc
  timeout 170 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python3 $BENCH > /tmp/pmc.log 2>&1
  python3 $R/scripts/dev/pmc_csv_summary.py /tmp/pmc > $O/pmc3_$name.txt; echo "== $name: $(grep -c n= $O/pmc3_$name.txt) rows"
done
rm -rf /tmp/pmc
timeout 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc --feature-dim 512 > /tmp/pmc.log 2>&1
python3 $R/scripts/dev/pmc_csv_summary.py /tmp/pmc k_wide > $O/pmc3_lseg512_mfma.txt; echo "== lseg512: $(grep -c n= $O/pmc3_lseg512_mfma.txt) rows"
python3 $R/scripts/dev/make_pmc_summary.py r03
