#!/bin/bash
# first GPU call of round 2: tests, baseline numbers of the round-1 kernels, fresh PMC
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2a/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r2a/tests.log
grep "config 2" gpurun_out/r2a/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
cat gpurun_out/r2a/bench.json
timeout 300 python bench.py --steps 5 --warmup 2 --bits 6 9 --no-cpu-baseline --no-materialize > gpurun_out/r2a/bench_6_9.json 2>&1
cat gpurun_out/r2a/bench_6_9.json
timeout 300 python bench.py --steps 5 --warmup 2 --bits 8 8 --no-cpu-baseline --no-materialize > gpurun_out/r2a/bench_8_8.json 2>&1
cat gpurun_out/r2a/bench_8_8.json
tools/pmc_collect.sh r2a/pmc
