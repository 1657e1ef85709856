#!/bin/bash
# same-box A/B on config 4 of two BUILDS of the library: icde2019-gpu-join_amd/libhj.so (new) against libhj_prev.so (same exported symbols), processes alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/zipfab
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_new.so
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],3) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05}, "mat", m.get("value"), m.get("ms_per_step"), m.get("kernel_ms_of_one_step"))'
for rep in 1 2; do
for v in prev new; do
cp $P/libhj_$v.so $P/libhj.so; touch $P/libhj.so $P/bench
timeout 900 python bench.py --workload zipf --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/zipfab/err_$v.log | python3 -c "$summ" "zipf $v" | tee -a gpurun_out/zipfab/ab.txt
done
done
cp $P/libhj_new.so $P/libhj.so
tail -2 gpurun_out/zipfab/err_new.log
