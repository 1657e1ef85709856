#!/bin/bash
# step time over input sizes (launch-bound below ~2^24)
cd $GRAFT_REPO_ROOT
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%-28s" % sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], d["config"]["radix_bits"])'
for l in 20 22 24 26 27 28 30; do
  timeout 600 python bench.py --steps 20 --warmup 3 --log2n $l --no-cpu-baseline --no-materialize --no-extras $BARGS 2>/dev/null | python3 -c "$summ" "2^$l"
done
