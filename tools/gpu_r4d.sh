#!/bin/bash
# round-4 loop D: PK-FK 2^24 x 2^27 Zipf with the PK side building (probe-side skew) against the Zipf side building (general items)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4d
mkdir -p $OUT
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%-40s" % sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "first", d["first_call_ms"], d["config"]["partition_layout_R_S"], {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items()}, "mat", (d.get("materialize") or {}).get("ms_per_step"), (d.get("materialize") or {}).get("k_join_materialize_ms"))'
for rep in 1 2; do
for cfg in "--build-side 1" "--build-side 2" "--build-side 2 --exact-only"; do
  timeout 600 python bench.py --workload zipf --zipf-sizes 24 27 --steps 10 --warmup 3 --no-cpu-baseline $cfg 2>$OUT/err.txt | tee -a $OUT/zipf_24_27.json | python3 -c "$summ" "[24x27 $cfg]"
  tail -2 $OUT/err.txt | grep -i error
done
done
for cfg in "--build-side 1" "--build-side 2"; do
  timeout 600 python bench.py --workload zipf --zipf-sizes 26 29 --steps 6 --warmup 2 --no-cpu-baseline $cfg 2>$OUT/err.txt | tee -a $OUT/zipf_26_29.json | python3 -c "$summ" "[26x29 $cfg]"
done
