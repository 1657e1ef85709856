#!/bin/bash
# join-kernel occupancy sensitivity: same table capacity, more heads => more LDS per workgroup => 3 / 2 / 1 workgroups per CU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.1}, "mat", d.get("materialize"))'
for rep in 1 2; do
for h in 4096 8192 16384; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --lds 4608 $h 2>/dev/null | python3 -c "$summ" "[heads $h]"
done
done | tee gpurun_out/r3a/occ.txt
