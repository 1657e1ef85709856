#!/bin/bash
# tools/experiments/rocprof_dist_bisect.sh — the compact reproducer of profiles/r5_rocprof_suite_crash.txt: tools/fuzz_dist.py (in-process
# hj_dist groups, one host thread per rank on cuda:0) under `rocprofv3 --kernel-trace`, 250 cases per configuration; a configuration is a
# name followed by environment assignments (FUZZ_WORLD, FUZZ_MAT, GPU_MAX_HW_QUEUES, ...).  Prints how many cases each survived.
cd /tmp && export TMPDIR=/tmp; export FUZZ_VERBOSE=1
run() { # name, env...
  name=$1; shift
  ( export "$@"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fd_$name -- python3 $GRAFT_REPO_ROOT/tools/fuzz_dist.py 250 7000 > /tmp/fz_$name.log 2>&1; echo "$name rc=$? cases $(grep -c '^case' /tmp/fz_$name.log) $(grep 'dist fuzz' /tmp/fz_$name.log | cut -c1-80)" )
  rm -rf /tmp/fd_$name
}
run w2_mat FUZZ_WORLD=2 FUZZ_MAT=1
run w3_mat FUZZ_WORLD=3 FUZZ_MAT=1
run w4_mat FUZZ_WORLD=4 FUZZ_MAT=1
run w5_mat FUZZ_WORLD=5 FUZZ_MAT=1
run w8_mat FUZZ_WORLD=8 FUZZ_MAT=1
run w8_count FUZZ_WORLD=8 FUZZ_MAT=0
run w8_mat_1_hw_queue FUZZ_WORLD=8 FUZZ_MAT=1 GPU_MAX_HW_QUEUES=1
run w8_mat_16_hw_queues FUZZ_WORLD=8 FUZZ_MAT=1 GPU_MAX_HW_QUEUES=16
# and the same fuzz WITHOUT the profiler
( export FUZZ_WORLD=8 FUZZ_MAT=1; timeout 600 python3 $GRAFT_REPO_ROOT/tools/fuzz_dist.py 250 7000 2>&1 | grep "dist fuzz" | sed 's/^/w8_mat_unprofiled: /' )
