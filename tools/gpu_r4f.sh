#!/bin/bash
# round-4 loop F: full GPU suite (line positions, > 2^32 tuples, headroom allocation), then same-box A/B of the library before
# (libhj_old.so = commit 9712a04) and after the 32-bit-line-number change
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4f
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; echo "tests rc=$?"
tail -8 $OUT/tests.txt
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_new.so
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05}, "first", d.get("first_call_ms"), d.get("first_call_split_ms"))'
for rep in 1 2 3; do
for v in new old; do
cp $P/libhj_$v.so $P/libhj.so
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "2^30 $v" | tee -a $OUT/ab.txt
timeout 600 python bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "2^27 $v" | tee -a $OUT/ab.txt
done
done
cp $P/libhj_new.so $P/libhj.so
for rep in 1 2; do
timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 --no-cpu-baseline --no-materialize 2>/dev/null | python3 -c "$summ" "zipf new" | tee -a $OUT/ab.txt
done
