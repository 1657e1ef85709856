#!/bin/bash
cd $GRAFT_REPO_ROOT
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%-24s" % sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.1})'
for a in "" "--lds 4608 2048" "--lds 4352 2048" "--lds 4352 1024" "--lds 4480 1024" "--lds 4608 4096 --probe-chunk 8192" ""; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize --no-extras $a 2>/dev/null | python3 -c "$summ" "[$a]"
done
timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 2>/dev/null | python3 -c "$summ" "zipf"
timeout 600 python bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize --no-extras --exact-only 2>/dev/null | python3 -c "$summ" "2^27 exact"
