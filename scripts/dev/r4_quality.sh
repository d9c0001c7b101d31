#!/bin/bash
# full fits of the round's code: 3000 steps dense / marching, 12000-step soak of both (scripts/quality.py prints one JSON line each)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
: > $OUT/r04_quality_3000.jsonl
timeout 900 python3 scripts/quality.py --iters 3000 2>/dev/null | tail -1 >> $OUT/r04_quality_3000.jsonl
timeout 900 python3 scripts/quality.py --iters 3000 --cuda-ray 2>/dev/null | tail -1 >> $OUT/r04_quality_3000.jsonl
timeout 1500 python3 scripts/quality.py --iters 12000 2>/dev/null | tail -1 >> $OUT/r04_quality_3000.jsonl
timeout 1500 python3 scripts/quality.py --iters 12000 --cuda-ray 2>/dev/null | tail -1 >> $OUT/r04_quality_3000.jsonl
cut -c1-330 $OUT/r04_quality_3000.jsonl
