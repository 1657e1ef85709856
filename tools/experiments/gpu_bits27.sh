#!/bin/bash
# 2^26 / 2^27: the default split (14 / 15 radix bits: full keys in the LDS table) against 16 bits (2048-tuple partitions, 16-bit tags)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bits27
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); m=d.get("materialize") or {}
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.02}, "mat", m.get("value"), m.get("ms_per_step"))'
for rep in 1 2; do
for l in 27 26; do
for b in "" "--bits 9 7" "--bits 8 8" "--bits 9 $((l-18))"; do
timeout 300 python bench.py --log2n $l --steps 20 --warmup 5 --no-cpu-baseline --no-extras $b 2>/dev/null | python3 -c "$summ" "2^$l [$b]" | tee -a gpurun_out/bits27/ab.txt
done
done
done
