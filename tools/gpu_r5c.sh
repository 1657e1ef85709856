#!/bin/bash
# round 5, call c: launch-structure A/B in ONE context (item 4) + general-item materialiser old/new library (item 5)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 900 python3 tools/launch_ab.py 5 22 24 26 27 28 30 2>&1 | grep "^{" | tee gpurun_out/r5c/launch_ab.txt
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_new.so
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "mat ms", m.get("ms_per_step"), "k_join_materialize_ms", m.get("k_join_materialize_ms"), "frac8", m.get("k_join_materialize_frac_of_8TBs"), {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05})'
for rep in 1 2; do
for v in new old; do
cp $P/libhj_$v.so $P/libhj.so
timeout 600 python bench.py --workload zipf --zipf-sizes 24 27 --build-side 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "$summ" "24x27 zipf-builds $v" | tee -a gpurun_out/r5c/gen_mat_ab.txt
timeout 600 python bench.py --workload zipf --zipf-sizes 26 29 --build-side 2 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "$summ" "26x29 zipf-builds $v" | tee -a gpurun_out/r5c/gen_mat_ab.txt
done
done
cp $P/libhj_new.so $P/libhj.so
timeout 600 python -m pytest tests/test_gpu_skew.py -m gpu -x -q 2>&1 | tail -3
