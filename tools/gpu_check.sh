#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r2e/tests.log
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.1})'
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "2^30 fast "
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize --no-extras --exact-only 2>/dev/null | python3 -c "$summ" "2^30 exact"
timeout 600 python bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize --no-extras --exact-only 2>/dev/null | python3 -c "$summ" "2^27 exact"
timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 2>/dev/null | python3 -c "$summ" "zipf"
