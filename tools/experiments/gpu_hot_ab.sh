#!/bin/bash
# same-box A/B of the heavy-hitter bypass on config 4 (PK-FK 2^27 x 2^31, Zipf 1.0): HJ_HOT=0 (round 5's path) against the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hotab
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "first", d["first_call_ms"], {k:round(v["ms_per_step"]/v["launches_per_step"],3) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05}, "mat", m.get("value"), m.get("ms_per_step"), m.get("kernel_ms_of_one_step"), "hot", d["config"].get("heavy_hitter_bypass"))'
for rep in 1 2; do
for v in 0 1; do
HJ_HOT=$v timeout 900 python bench.py --workload zipf --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/hotab/err_$v.log | tee gpurun_out/hotab/line_${v}_$rep.json | python3 -c "$summ" "zipf HJ_HOT=$v" | tee -a gpurun_out/hotab/ab.txt
done
done
tail -3 gpurun_out/hotab/err_1.log
