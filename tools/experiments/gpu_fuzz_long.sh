#!/bin/bash
# extended differential runs against the oracle (beyond tools/gpu_final.sh's): more cases, more seeds, build-side variation
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/fuzzlong; mkdir -p $OUT
sha256sum icde2019-gpu-join_amd/libhj.so > $OUT/fuzz_extended.txt
{ echo "# tools/fuzz_medium.py 240 (2^18-2^23-tuple relations, six key distributions, build sides 0/1/2, default and exact_only, learned skew, materialisation digest, hipGraph)"
  timeout 2400 python tools/fuzz_medium.py 240 2>&1 | tail -4
  echo "# tools/fuzz_medium.py 72 hibits (the same cases with 16-18 forced radix bits: 9+7, 8+8, 9+8, 9+9)"
  timeout 2400 python tools/fuzz_medium.py 72 hibits 2>&1 | tail -4
  echo "# tools/fuzz_more.py 400 6000 (the suite's small-case fuzz over 5600 more seeds)"
  timeout 1800 python tools/fuzz_more.py 400 6000 2>&1 | tail -2; } >> $OUT/fuzz_extended.txt
cat $OUT/fuzz_extended.txt
