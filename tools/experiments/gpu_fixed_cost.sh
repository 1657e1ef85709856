#!/bin/bash
# per-workgroup timelines of k_join_count and k_part2_fast: the -DHJ_STAMPS build in place of libhj.so on the box's
# scratch copy of the repo
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fixed
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_shipped.so
cp $P/libhj_stamps.so $P/libhj.so
touch $P/libhj.so $P/bench    # (make must not rebuild over it)
for l in 27 30 24; do
  timeout 600 python tools/experiments/fixed_cost.py $l 4 2>gpurun_out/fixed/err_$l.log | tee gpurun_out/fixed/timeline_$l.jsonl | cut -c1-900
done
cp $P/libhj_shipped.so $P/libhj.so
tail -2 gpurun_out/fixed/err_27.log
