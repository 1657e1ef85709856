#!/bin/bash
# the round's evidence (ROUND=r5 by default): bench lines + rocprofv3 kernel stats + PMC passes, all into gpurun_out/${ROUND}prof/ (copy what should be judged
# into profiles/).  --pmc runs are separate from --kernel-trace --stats runs and never combined with sys/runtime tracing.
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r6}
OUT=$ROOT/gpurun_out/${ROUND}prof
mkdir -p $OUT
cd $ROOT
sha256sum icde2019-gpu-join_amd/libhj.so > $OUT/libhj.sha256
python bench.py --steps 10 --warmup 3 > $OUT/bench_2p30.json 2> $OUT/bench_2p30.err; echo "bench30 rc=$?"
python bench.py --steps 10 --warmup 3 --log2n 27 > $OUT/bench_2p27.json 2>/dev/null; echo "bench27 rc=$?"
python bench.py --steps 10 --warmup 3 --exact-only --no-cpu-baseline > $OUT/bench_2p30_exact.json 2>/dev/null
python bench.py --workload zipf --steps 5 --warmup 2 > $OUT/bench_zipf.json 2>/dev/null; echo "zipf rc=$?"
python bench.py --workload zipf --steps 5 --warmup 2 --exact-only --no-cpu-baseline > $OUT/bench_zipf_exact.json 2>/dev/null; echo "zipf exact rc=$?"
python bench.py --workload zipf --zipf-sizes 24 27 --build-side 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_zipf_24_27_pk_builds.json 2>/dev/null
python bench.py --workload zipf --zipf-sizes 24 27 --build-side 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_zipf_24_27_zipf_builds.json 2>/dev/null; echo "zipf-builds rc=$?"
python bench.py --workload stream --steps 3 --warmup 1 > $OUT/bench_stream.json 2>/dev/null; echo "stream rc=$?"
python bench.py --workload coprocess --log2n 27 --steps 9 --warmup 2 > $OUT/bench_coprocess.json 2>/dev/null; echo "coprocess rc=$?"
python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline > $OUT/bench_forcedist.json 2>/dev/null; echo "forcedist rc=$?"
for g in 2 4 8; do python bench.py --steps 5 --warmup 2 --force-dist --phantom $g --no-cpu-baseline > $OUT/bench_phantom$g.json 2>/dev/null; done
python bench.py --steps 5 --warmup 2 --force-dist --phantom 8 --single-group --no-cpu-baseline > $OUT/bench_phantom8_single_group.json 2>/dev/null
# the STRONG shape (BASELINE's metric read as "2^30 x 2^30 ... 1/2/4/8 GPU": 2^30 tuples per relation in TOTAL): 2^(30 - log2 G) per GPU
python bench.py --steps 5 --warmup 2 --force-dist --phantom 2 --log2n 29 --no-cpu-baseline > $OUT/bench_phantom2_strong.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --force-dist --phantom 4 --log2n 28 --no-cpu-baseline > $OUT/bench_phantom4_strong.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --force-dist --phantom 8 --log2n 27 --no-cpu-baseline > $OUT/bench_phantom8_strong.json 2>/dev/null; echo "phantom strong rc=$?"
python bench.py --steps 5 --warmup 2 --force-dist --dist-impl torch --no-cpu-baseline > $OUT/bench_forcedist_torch.json 2>/dev/null
python tools/step_vs_size.py 2>/dev/null > $OUT/step_vs_size.txt
for l in 24 27 30; do python bench.py --workload baselines --log2n $l --steps 3 --warmup 1 2>/dev/null; done > $OUT/bench_baselines.json
cd /tmp
# per-kernel durations: the passes of R and S serialised (HJ_FORK_LOG2=0), as in bench.py's instrumented steps — with the two
# relations' kernels overlapped on two streams (the timed steps) a kernel's start-to-end time includes the other kernel's share
# of the chip.  The variable is exported here: no env/sh hop between rocprofv3 and python3.
export HJ_FORK_LOG2=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats30 -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize > $OUT/stats30.log 2>&1; echo "stats30 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats27 -- python3 $ROOT/bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize > $OUT/stats27.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statszipf -- python3 $ROOT/bench.py --workload zipf --steps 5 --warmup 2 --no-cpu-baseline --no-materialize > $OUT/statszipf.log 2>&1
for d in stats30 stats27 statszipf; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/$d.kernel_stats.csv; rm -rf $OUT/$d; grep '^{' $OUT/$d.log | tail -1 > $OUT/$d.bench_line.json; done
unset HJ_FORK_LOG2
cd $ROOT
tools/pmc_collect.sh ${ROUND}prof/pmc30
tools/pmc_collect.sh ${ROUND}prof/pmc30_mat --with-materialize
tools/pmc_collect.sh ${ROUND}prof/pmc27 --log2n 27
tools/pmc_collect.sh ${ROUND}prof/pmczipf --workload zipf --warmup 1
tools/pmc_collect.sh ${ROUND}prof/pmczipf_mat --with-materialize --workload zipf --warmup 1
rm -rf $OUT/pmc30/p? $OUT/pmc30_mat/p? $OUT/pmc27/p? $OUT/pmczipf/p? $OUT/pmczipf_mat/p?
ls -la $OUT
