#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe; mkdir -p $O
sha256sum icde2019-gpu-join_amd/libhj.so > $O/out.txt
timeout 900 python -m pytest tests/test_gpu_skew.py -m gpu -x -q -k "look_before" 2>&1 | tail -5 | tee -a $O/out.txt
for rep in 1 2 3; do for v in 0 26; do
HJ_SKEW_PROBE=$v ONLY_FIRST=1 timeout 600 python3 tools/experiments/first_call.py 27 31 2>&1 | grep "context 0" | sed "s/^/HJ_SKEW_PROBE=$v /" | tee -a $O/out.txt
done; done
