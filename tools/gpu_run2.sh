#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2b/tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r2b/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; echo "bench rc=$?"
cat gpurun_out/r2b/bench.json; tail -3 gpurun_out/r2b/bench.err
HJ_FAST_PATH=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize > gpurun_out/r2b/bench_exact.json 2>&1
cat gpurun_out/r2b/bench_exact.json
timeout 600 python bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline --no-materialize > gpurun_out/r2b/bench27.json 2>&1
cat gpurun_out/r2b/bench27.json
