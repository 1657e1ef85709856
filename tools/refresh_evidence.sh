#!/bin/bash
# The whole evidence refresh for the library in the tree, from the build container: four gpurun calls, copies, regenerated tables.
# Stops at the first step that fails (a busy pool returns "transient": run it again).
set -e
cd "$(dirname "$0")/.."
G=/usr/local/graft/bin/gpurun
ROUND=${ROUND:-r6}
export ROUND
tools/kernel_sizes.sh > profiles/${ROUND}_libhj_kernels.txt   # (the coverage call below reads it on the box)
ok() { grep -q '"status": *"ok"' gpurun_out/.last_call.json || { echo "gpurun did not run the command: $(cat gpurun_out/.last_call.json | head -c 300)"; exit 3; }; }
$G --timeout 3600 -- 'bash tools/gpu_final.sh' | tail -8; ok
bash tools/copy_evidence.sh final
$G --timeout 5400 -- 'bash tools/gpu_profiles.sh' | grep "rc=" | tr '\n' ' '; ok
bash tools/copy_evidence.sh profiles
$G --timeout 1800 -- 'bash tools/gpu_bench_lines.sh' | tail -4; ok
bash tools/copy_evidence.sh lines
$G --timeout 3600 -- 'bash tools/gpu_kernel_coverage.sh' | tail -3; ok
cp gpurun_out/coverage/kernel_coverage.txt profiles/${ROUND}_kernel_coverage.txt
[ "$SKIP_DOCS" = 1 ] || python3 tools/update_docs.py
