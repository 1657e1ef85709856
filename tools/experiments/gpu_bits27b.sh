#!/bin/bash
# config 2 (2^27 x 2^27) with the round-6 kernels: the 15 radix bits split 9+6 (default) / 8+7 / 10+5 / 7+8, and the pass-1 span count
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bits27b
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line)
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.02})'
for rep in 1 2; do
for b in "" "--bits 8 7" "--bits 10 5" "--bits 7 8"; do
timeout 300 python bench.py --log2n 27 --steps 20 --warmup 5 --no-cpu-baseline --no-extras $b 2>/dev/null | python3 -c "$summ" "2^27 [$b]" | tee -a gpurun_out/bits27b/ab.txt
done
for ts in 128 384 512; do
HJ_TARGET_SPANS=$ts timeout 300 python bench.py --log2n 27 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "$summ" "2^27 [HJ_TARGET_SPANS=$ts]" | tee -a gpurun_out/bits27b/ab.txt
done
done
