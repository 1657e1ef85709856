#!/bin/bash
# a longer differential soak of the final round-6 binary (the evidence set has the short runs): medium fuzz 150 cases from seed 200, with the lowered probe bar
# 100 cases from seed 400, 16-18 forced radix bits 100 cases from seed 300, the small-case fuzz 400..3000, the multi-GPU join 2000 cases
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak; mkdir -p $O
sha256sum icde2019-gpu-join_amd/libhj.so > $O/soak.txt
{ echo "# tools/fuzz_medium.py 150 - 200"; timeout 1500 python tools/fuzz_medium.py 150 - 200 2>&1 | grep "FAIL\|medium fuzz"
  echo "# HJ_SKEW_PROBE=18 tools/fuzz_medium.py 100 - 400"; HJ_SKEW_PROBE=18 timeout 1500 python tools/fuzz_medium.py 100 - 400 2>&1 | grep "FAIL\|medium fuzz"
  echo "# tools/fuzz_medium.py 100 hibits 300"; timeout 1500 python tools/fuzz_medium.py 100 hibits 300 2>&1 | grep "FAIL\|medium fuzz"
  echo "# tools/fuzz_more.py 400 3000"; timeout 1500 python tools/fuzz_more.py 400 3000 2>&1 | tail -1
  echo "# tools/fuzz_dist.py 2000"; timeout 1500 python tools/fuzz_dist.py 2000 2>&1 | tail -1; } >> $O/soak.txt 2>&1
cat $O/soak.txt
