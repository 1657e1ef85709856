#!/bin/bash
# 2^27 x 2^27 (config 2) knob sweep on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%-34s" % sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], d["config"]["radix_bits"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05})'
for rep in 1 2; do
for cfg in "X=1|" "HJ_TARGET_SPANS=512|" "HJ_TARGET_SPANS=384|" "X=1|--bits 8 7" "X=1|--bits 9 7" "HJ_TARGET_SPANS=512|--bits 9 7" "X=1|--lds 4352 4096" ; do
  e=${cfg%%|*}; l=${cfg##*|}
  env $e timeout 600 python bench.py --steps 20 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize --no-extras $l 2>/dev/null | python3 -c "$summ" "[$cfg]"
done
done | tee gpurun_out/r3i/sweep27.txt
