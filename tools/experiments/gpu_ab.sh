#!/bin/bash
# A/B of experiment knobs on one box: tools/experiments/gpu_ab.sh "ENV1=.. ENV2=.." "ENV.." ...  (each arg = one environment, "-" = none)
cd $GRAFT_REPO_ROOT
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%-28s" % sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.1})'
for rep in 1 2; do
for e in "$@"; do
  if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
  env $ee timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize --no-extras $BARGS 2>/dev/null | python3 -c "$summ" "[$e]"
done
done
