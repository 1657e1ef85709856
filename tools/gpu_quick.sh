#!/bin/bash
# quick A/B: the partition/join parity tests, then the headline bench at 2^30 and 2^27
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests/test_gpu_join.py -m gpu -x -q -k "fast_path or partition_parity or fuzz or golden or ragged or skew or tag16 or config1" > gpurun_out/quick/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/quick/tests.log
for l in 30 27; do
timeout 600 python bench.py --steps 10 --warmup 3 --log2n $l --no-cpu-baseline --no-materialize $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('log2n', $l, 'value', d['value'], 'ms', d['ms_per_step'], {k:round(v['ms_per_step']/v['launches_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']>0.1})
"
done
