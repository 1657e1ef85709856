#!/bin/bash
cd $GRAFT_REPO_ROOT
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%-14s" % sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05})'
for l in 27 26 28; do
for b in "9 5" "9 6" "9 7" "9 8" "8 8" "8 7"; do
  timeout 600 python bench.py --steps 20 --warmup 3 --log2n $l --bits $b --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "2^$l [$b]"
done
done
