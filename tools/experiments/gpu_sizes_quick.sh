#!/bin/bash
# the default step at a few sizes (kernel times from the instrumented steps), after the parity suites of the join kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sizes
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_skew.py -m gpu -x -q > gpurun_out/sizes/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/sizes/tests.log
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.02}, "mat", m.get("value"), m.get("ms_per_step"), m.get("k_join_materialize_ms"))'
for rep in 1 2; do
for l in 27 26 24 22 30; do
timeout 300 python bench.py --log2n $l --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "$summ" "2^$l" | tee -a gpurun_out/sizes/ab.txt
done
done
