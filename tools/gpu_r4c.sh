#!/bin/bash
# round-4 loop C: general work items (skewed build side): skew + join parity suites, then PK-FK with the Zipf side as BUILD
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_skew.py tests/test_gpu_join.py -m gpu -x -q > $OUT/tests.txt 2>&1; echo "tests rc=$?"
tail -25 $OUT/tests.txt
bash tools/gpu_r4d.sh
