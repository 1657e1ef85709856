#!/bin/bash
# round 5, call d: co-processing rewrite (tests + bench), multi-GPU suites again, config-4 pass experiments
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_join.py tests/test_dist_c.py tests/test_dist.py -m gpu -x -q -k "coprocess or host or dist or materialis or cli" > gpurun_out/r5d/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5d/tests.log
for i in 1 2 3; do timeout 600 python bench.py --workload coprocess --log2n 27 --steps 5 --warmup 2 2>/dev/null | grep "^{" | tee -a gpurun_out/r5d/coprocess.txt; done
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line)
        print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05}, d["config"]["radix_bits"], d["config"]["partition_layout_R_S"])'
for rep in 1 2; do
timeout 600 python bench.py --workload zipf --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "zipf default" | tee -a gpurun_out/r5d/zipf_exp.txt
HJ_TARGET_SPANS=512 timeout 600 python bench.py --workload zipf --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "zipf spans512" | tee -a gpurun_out/r5d/zipf_exp.txt
HJ_VAR_GUIDE=4 timeout 600 python bench.py --workload zipf --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "zipf guide4" | tee -a gpurun_out/r5d/zipf_exp.txt
HJ_VAR_GUIDE=1 timeout 600 python bench.py --workload zipf --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "zipf guide1" | tee -a gpurun_out/r5d/zipf_exp.txt
timeout 600 python bench.py --workload zipf --zipf-theta 0 --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "uniform-FK fast path" | tee -a gpurun_out/r5d/zipf_exp.txt
HJ_FORCE_SAMPLED=2 timeout 600 python bench.py --workload zipf --zipf-theta 0 --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "uniform-FK forced sampled" | tee -a gpurun_out/r5d/zipf_exp.txt
HJ_FORCE_SAMPLED=2 timeout 600 python bench.py --workload zipf --zipf-theta 0 --bits 9 6 --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "uniform-FK forced sampled 9+6" | tee -a gpurun_out/r5d/zipf_exp.txt
timeout 600 python bench.py --workload zipf --zipf-theta 0.5 --steps 6 --warmup 2 --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "$summ" "zipf theta 0.5" | tee -a gpurun_out/r5d/zipf_exp.txt
done
