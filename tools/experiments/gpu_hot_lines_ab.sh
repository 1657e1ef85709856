#!/bin/bash
# same-box A/B of the two forms of the WRITING bypass on config 4: HJ_HOT_LINES=0 (4-byte stores per hit, HOT 2) against whole lines through LDS (HOT 3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hotlines
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "first", d["first_call_ms"], {k:round(v["ms_per_step"]/v["launches_per_step"],3) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05}, "mat", m.get("value"), m.get("ms_per_step"), m.get("kernel_ms_of_one_step"), "hot", d["config"].get("heavy_hitter_bypass"))'
for rep in 1 2; do
for v in 0 1; do
HJ_HOT_LINES=$v timeout 900 python bench.py --workload zipf --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/hotlines/err_$v.log | tee gpurun_out/hotlines/line_${v}_$rep.json | python3 -c "$summ" "zipf HJ_HOT_LINES=$v" | tee -a gpurun_out/hotlines/ab.txt
done
done
tail -3 gpurun_out/hotlines/err_1.log
