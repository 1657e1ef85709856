#!/bin/bash
# the full GPU suite, smoke(), then the extended differential runs — the parity evidence of a round
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "tests rc=$?"
tail -3 $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
{ echo "# tools/fuzz_medium.py 36 on one MI355X: differential runs against the oracle, 2^18-2^23-tuple relations, six key distributions,"
  echo "# default and exact_only, second join with the learned skew, one-probe materialisation digest, three joins with hj_config.graph"
  timeout 900 python tools/fuzz_medium.py 36 2>&1
  echo "# the same with the look before the first attempt taken from 2^18 tuples up (HJ_SKEW_PROBE=18: every relation of >= 2^20 tuples is looked at), then 24 cases"
  echo "# with 16-18 forced radix bits (the sampled path at 18 bits where the bypass applies), default bar and lowered bar"
  HJ_SKEW_PROBE=18 timeout 900 python tools/fuzz_medium.py 36 2>&1 | tail -1
  timeout 900 python tools/fuzz_medium.py 24 hibits 2>&1 | tail -1
  HJ_SKEW_PROBE=18 timeout 900 python tools/fuzz_medium.py 24 hibits 100 2>&1 | tail -1
  timeout 600 python tools/fuzz_more.py 40 400 2>&1 | tail -1
  echo "# tools/fuzz_dist.py 300: the multi-GPU join (in-process group on one GPU, world 2-8, count-only and materialising) against the oracle"
  timeout 900 python tools/fuzz_dist.py 300 2>&1 | tail -3; } > $OUT/fuzz.txt
tail -4 $OUT/fuzz.txt
sha256sum icde2019-gpu-join_amd/libhj.so | tee $OUT/libhj.sha256
