#!/bin/bash
# round-4 loop I: same-box A/B of library builds: streaming materialising probe old (count + second probe per segment, blocking
# read-back) vs new (one probe per segment, cursor read one segment later); k_join_mat_reg general variant at three (scratch) vs two
# (no scratch) workgroups per CU
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4i
mkdir -p $OUT
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_new.so
for rep in 1 2; do
for v in new old; do
cp $P/libhj_$v.so $P/libhj.so
timeout 900 python bench.py --workload stream --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('stream $v', d['ms_per_step'], d['h2d_GBs'], d['materialize'])" | tee -a $OUT/ab.txt
done
for v in new alt; do
cp $P/libhj_$v.so $P/libhj.so
timeout 600 python bench.py --workload zipf --zipf-sizes 24 27 --build-side 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('zipf-builds $v', d['ms_per_step'], d['materialize']['ms_per_step'], d['materialize']['k_join_materialize_ms'])" | tee -a $OUT/ab.txt
timeout 600 python bench.py --workload zipf --zipf-sizes 26 29 --build-side 2 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('zipf-builds 26x29 $v', d['ms_per_step'], d['materialize']['ms_per_step'], d['materialize']['k_join_materialize_ms'])" | tee -a $OUT/ab.txt
done
done
cp $P/libhj_new.so $P/libhj.so
