#!/bin/bash
# 18 radix bits under skew: PK-FK 2^30 x 2^31 Zipf 1.0 (and 2^29 x 2^31: 17 bits) with the heavy-hitter bypass (sampled path) against HJ_HOT=0
# (at 18 bits: the exact passes, round 5's cliff)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/zipf18
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], d["config"]["radix_bits"], d["config"]["partition_layout_R_S"], {k:round(v["ms_per_step"]/v["launches_per_step"],3) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05}, "mat", m.get("value"), m.get("ms_per_step"), "hot", (d["config"].get("heavy_hitter_bypass") or {}).get("share"))'
for sz in "30 31" "29 31"; do
for v in 0 1; do
HJ_HOT=$v timeout 900 python bench.py --workload zipf --zipf-sizes $sz --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/zipf18/err_$v.log | python3 -c "$summ" "zipf $sz HJ_HOT=$v" | tee -a gpurun_out/zipf18/ab.txt
done
done
tail -3 gpurun_out/zipf18/err_1.log
