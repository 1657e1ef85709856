#!/bin/bash
# A/B: k_join at 8 waves/SIMD (<= 64 VGPRs, 2 build loads in flight) vs the shipped 6 waves/SIMD, over LDS table shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/occ4
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_new.so
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05})'
for rep in 1 2; do
for v in new old; do
cp $P/libhj_$v.so $P/libhj.so
for lds in "" "--lds 4352 1024" "--lds 4352 2048" "--lds 4608 1024"; do
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras $lds 2>/dev/null | python3 -c "$summ" "2^30 $v [$lds]" | tee -a gpurun_out/occ4/ab.txt
done
timeout 600 python bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "2^27 $v" | tee -a gpurun_out/occ4/ab.txt
done
done
