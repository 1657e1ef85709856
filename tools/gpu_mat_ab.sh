#!/bin/bash
# materialising kernels A/B on one box: matches in registers (default) vs staged in LDS (HJ_STAGE_CAP=4608)
cd $GRAFT_REPO_ROOT
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}; print(sys.argv[1], "count-only ms", d["ms_per_step"], "mat step ms", m.get("ms_per_step"), "k_join_materialize ms", m.get("k_join_materialize_ms"))'
for rep in 1 2; do
for l in ${SIZES:-30 27}; do
for e in "X=1" "HJ_STAGE_CAP=4608"; do
  env $e timeout 600 python bench.py --steps 4 --warmup 1 --log2n $l --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "$summ" "[2^$l $e]"
done; done; done
