#!/bin/bash
# A/B: piece sizing of the sampled path's pass 2 (HJ_VAR_GUIDE), zipf config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/guide
timeout 600 python -m pytest tests/test_gpu_skew.py -m gpu -x -q 2>&1 | tail -2
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05})'
for g in ${GUIDES:-0 1 2 3}; do
HJ_VAR_GUIDE=$g timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "$summ" "guide=$g" | tee -a gpurun_out/guide/ab.txt
done
