#!/bin/bash
# attribution builds of the writing bypass (make exp EXP="1 2 5 6": hj_part.hip under -DHJ_EXP=n): kernel time of pass 1 with one piece changed
# 1: without the cursor atomic   2: without hot_emit   5: the atomic on a word of the workgroup's own   6: the atomic on the cursor, result unused
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hotexp
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_shipped.so
for n in ${VARIANTS:-0 1 2 5 6}; do
  if [ $n = 0 ]; then cp $P/libhj_shipped.so $P/libhj.so; else cp $P/libhj_exp$n.so $P/libhj.so; fi
  touch $P/libhj.so $P/bench
  echo "== variant $n"
  HOT_PHASES_NOCHECK=1 timeout 600 python tools/experiments/hot_phases.py 27 31 2>gpurun_out/hotexp/err_$n.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['mode'], d['k_part1_var_ms'])
" | tee -a gpurun_out/hotexp/out.txt
done
cp $P/libhj_shipped.so $P/libhj.so
