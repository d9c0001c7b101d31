#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_configs.py -q > $O/r3_pytest4.log 2>&1; tail -25 $O/r3_pytest4.log | cut -c1-300
python scripts/dev/quality_gate_probe.py steps=3500 > $O/r3_qgate_b.json 2> $O/r3_qgate_b.err; tail -c 1200 $O/r3_qgate_b.json; tail -3 $O/r3_qgate_b.err
python scripts/dev/quality_gate_probe.py steps=2000 batch=2048 hip_seeds=7,8,9,10,11 > $O/r3_qgate_c.json 2> $O/r3_qgate_c.err; tail -c 1200 $O/r3_qgate_c.json; tail -3 $O/r3_qgate_c.err
