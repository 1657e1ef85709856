#!/bin/bash
# same-box A/B of two BINARIES: the round-5 library (built from commit 0d0629e into tools/experiments/r5bin/, git-ignored) against the tree's,
# processes alternating, on the headline, config 2 and config 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/roundab; mkdir -p $O
sha256sum tools/experiments/r5bin/libhj.so icde2019-gpu-join_amd/libhj.so | tee $O/ab.txt
for rep in 1 2 3; do
for w in "uniform 30" "uniform 27" "zipf 27 31"; do
for v in r5 r6; do
if [ $v = r5 ]; then lib=tools/experiments/r5bin/libhj.so; else lib=icde2019-gpu-join_amd/libhj.so; fi
echo -n "$v  " | tee -a $O/ab.txt
timeout 300 python3 tools/experiments/round_ab.py $lib $w 20 2>&1 | tail -1 | tee -a $O/ab.txt
done; done; done
