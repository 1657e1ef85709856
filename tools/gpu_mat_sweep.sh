#!/bin/bash
# one-probe materialising kernel: staging block size and table shape (same box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}; print(sys.argv[1], "count-only ms", d["ms_per_step"], "mat step ms", m.get("ms_per_step"), "k_join_materialize ms", m.get("k_join_materialize_ms"))'
for rep in 1 2; do
for cfg in "- -" "HJ_STAGE_CAP=2304 -" "HJ_STAGE_CAP=4608 --lds_4608_2048" "HJ_STAGE_CAP=4608 --lds_4608_8192"; do
  set -- $cfg
  e=$1; l=$(echo $2 | tr '_' ' ')
  [ "$e" = "-" ] && e="X=1"
  [ "$l" = "-" ] && l=""
  env $e timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras $l 2>/dev/null | python3 -c "$summ" "[$cfg]"
done
done | tee gpurun_out/r3d/mat_sweep.txt
