#!/bin/bash
# same-box A/B over sizes of a knob of the library ($KNOB, default HJ_TAGS_LEGACY): =1 (old) against =0 (new), processes alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/libab2
KNOB=${KNOB:-HJ_TAGS_LEGACY}
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.02}, "mat", m.get("value"), m.get("ms_per_step"), m.get("k_join_materialize_ms"))'
for rep in 1 2 3; do
for l in ${SIZES:-27 26 24}; do
for v in old new; do
if [ $v = old ]; then export $KNOB=1; else export $KNOB=0; fi
timeout 300 python bench.py --log2n $l --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "$summ" "2^$l $v" | tee -a gpurun_out/libab2/ab.txt
done
done
done
