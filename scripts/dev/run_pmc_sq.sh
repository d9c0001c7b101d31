#!/bin/bash
# Shader-sequencer counters of the training step's kernels (one rocprofv3 --pmc pass per set, no trace domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
BENCH="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --render-frames 0 --no-graph --event-steps 0 --quality-steps 0 --no-march --no-lseg --no-pmc"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" \
           "SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc
  timeout 170 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python3 $BENCH > /tmp/pmc.log 2>&1
  python3 $R/scripts/dev/pmc_csv_summary.py /tmp/pmc > $O/pmc4_set$i.txt; echo "== set $i: $(grep -c n= $O/pmc4_set$i.txt) rows"
done
