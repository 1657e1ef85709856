#!/bin/bash
# a soak of differential runs beyond gpu_fuzz_long.sh's seeds (half an hour of GPU time): more small seeds, more medium cases
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/soak; mkdir -p $OUT
{ sha256sum icde2019-gpu-join_amd/libhj.so
  echo "# tools/fuzz_more.py 6000 200000 1500 (as many seeds as fit 1500 s)"
  timeout 1800 python tools/fuzz_more.py 6000 200000 1500 2>&1 | tail -3
  echo "# tools/fuzz_medium.py 600 - 240 (medium cases, seeds 240..839)"
  timeout 1500 python tools/fuzz_medium.py 600 - 240 2>&1 | grep -v "^seed" | tail -3
  echo "# tools/fuzz_medium.py 200 hibits 72 (16-18 forced radix bits, seeds 72..271)"
  timeout 900 python tools/fuzz_medium.py 200 hibits 72 2>&1 | grep -v "^seed" | tail -3; } > $OUT/soak.txt 2>&1
cat $OUT/soak.txt
