#!/bin/bash
# gpurun_out/${ROUND}prof (tools/gpu_profiles.sh), gpurun_out/final (tools/gpu_final.sh), gpurun_out/${ROUND}lines (tools/gpu_bench_lines.sh)
# -> profiles/${ROUND}_*: what the documents quote.  Run from the repository root, in this order: profiles, final, then (after the PMC files
# are in place and tools/gpu_bench_lines.sh has run) lines.
set -e
ROUND=${ROUND:-r6}
S=gpurun_out/${ROUND}prof; D=profiles
LIB=$(sha256sum icde2019-gpu-join_amd/libhj.so | cut -d" " -f1)
same_binary() { # the results must come from the library that is in the tree now
  grep -q "$LIB" "$1" || { echo "copy_evidence: $1 is not from the current libhj.so ($LIB): rerun the GPU script first"; exit 2; }
}
case "$1" in
profiles)
  same_binary $S/libhj.sha256
  for f in 2p30_exact zipf_exact zipf_24_27_pk_builds zipf_24_27_zipf_builds stream coprocess baselines forcedist forcedist_torch phantom2 phantom4 phantom8 phantom8_single_group phantom2_strong phantom4_strong phantom8_strong; do cp $S/bench_$f.json $D/${ROUND}_bench_$f.json; done
  for f in 2p30 2p27 zipf; do cp $S/bench_$f.json $D/${ROUND}_bench_${f}_evidence_call.json; done
  cp $S/stats30.kernel_stats.csv $D/${ROUND}_kernel_stats_2p30.csv; cp $S/stats27.kernel_stats.csv $D/${ROUND}_kernel_stats_2p27.csv; cp $S/statszipf.kernel_stats.csv $D/${ROUND}_kernel_stats_zipf.csv
  # the bench line each rocprofv3 process printed itself: CSV and line are ONE process (their per-kernel averages agree)
  cp $S/stats30.bench_line.json $D/${ROUND}_bench_2p30_under_rocprof.json; cp $S/stats27.bench_line.json $D/${ROUND}_bench_2p27_under_rocprof.json; cp $S/statszipf.bench_line.json $D/${ROUND}_bench_zipf_under_rocprof.json
  cp $S/pmc30/pmc.json $D/${ROUND}_pmc_2p30.json; cp $S/pmc30_mat/pmc.json $D/${ROUND}_pmc_2p30_materialize.json; cp $S/pmc27/pmc.json $D/${ROUND}_pmc_2p27.json
  cp $S/pmczipf/pmc.json $D/${ROUND}_pmc_zipf.json; cp $S/pmczipf_mat/pmc.json $D/${ROUND}_pmc_zipf_materialize.json
  cp $S/step_vs_size.txt $D/${ROUND}_step_vs_size.txt; cp $S/libhj.sha256 $D/${ROUND}_libhj.sha256 ;;
final)
  same_binary gpurun_out/final/libhj.sha256
  cp gpurun_out/final/gpu_tests.txt $D/${ROUND}_gpu_tests.txt; cp gpurun_out/final/fuzz.txt $D/${ROUND}_fuzz.txt ;;
lines)
  same_binary gpurun_out/${ROUND}lines/bench_2p30.json
  for f in 2p30 2p27 zipf; do cp gpurun_out/${ROUND}lines/bench_$f.json $D/${ROUND}_bench_$f.json; done ;;
*) echo "usage: copy_evidence.sh profiles|final|lines"; exit 1 ;;
esac
