#!/bin/bash
# same-box A/B of two builds of the library: icde2019-gpu-join_amd/libhj.so (new) against libhj_old.so (copied over it for the
# "old" runs on the box's scratch copy of the repo).  Parity tests of the new build first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/libab
P=icde2019-gpu-join_amd
timeout 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_skew.py -m gpu -x -q > gpurun_out/libab/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/libab/tests.log
cp $P/libhj.so $P/libhj_new.so
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05}, "copy", (d.get("roofline") or {}).get("stream_copy_ceiling"))'
for rep in 1 2; do
for v in new old; do
cp $P/libhj_$v.so $P/libhj.so
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize 2>/dev/null | python3 -c "$summ" "2^30 $v" | tee -a gpurun_out/libab/ab.txt
timeout 600 python bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "2^27 $v" | tee -a gpurun_out/libab/ab.txt
timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "$summ" "zipf $v" | tee -a gpurun_out/libab/ab.txt
done
done
