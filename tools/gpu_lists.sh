#!/bin/bash
# list items (several probe ranges per table build): skew parity tests, then the zipf config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lists
timeout 900 python -m pytest tests/test_gpu_skew.py tests/test_gpu_join.py -m gpu -x -q > gpurun_out/lists/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/lists/tests.log
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05})'
for i in 1 2; do
timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "$summ" "zipf"
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "$summ" "2^30"
