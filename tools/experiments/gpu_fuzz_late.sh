#!/bin/bash
# differential runs for what came late in round 6: the sampled path at 18 bits with the bypass, the look before the first attempt (bar lowered to 2^18 tuples
# so that the fuzz sizes are looked at: relations of >= 2^20 tuples)
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzzlate; mkdir -p $O
sha256sum icde2019-gpu-join_amd/libhj.so > $O/out.txt
{ echo "# HJ_SKEW_PROBE=18 tools/fuzz_medium.py 36"; HJ_SKEW_PROBE=18 timeout 900 python tools/fuzz_medium.py 36 2>&1 | grep -v amdgpu.ids
  echo "# tools/fuzz_medium.py 24 hibits"; timeout 900 python tools/fuzz_medium.py 24 hibits 2>&1 | grep -v amdgpu.ids
  echo "# HJ_SKEW_PROBE=18 tools/fuzz_medium.py 24 hibits 100"; HJ_SKEW_PROBE=18 timeout 900 python tools/fuzz_medium.py 24 hibits 100 2>&1 | grep -v amdgpu.ids; } >> $O/out.txt
grep -c "^seed" $O/out.txt; grep "FAIL\|fuzz" $O/out.txt
