#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2c/tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r2c/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; echo "bench rc=$?"
cat gpurun_out/r2c/bench.json; tail -3 gpurun_out/r2c/bench.err
timeout 600 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline > gpurun_out/r2c/bench_forcedist.json 2>&1; echo "force-dist rc=$?"
cat gpurun_out/r2c/bench_forcedist.json
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2c/bench_gpus2.json 2>&1; echo "gpus2 rc=$? (expected to fail on a 1-GPU box)"
tail -3 gpurun_out/r2c/bench_gpus2.json
