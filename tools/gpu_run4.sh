#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2d/tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r2d/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize 2>/dev/null | cut -c1-1500
timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 2>/dev/null
HJ_FAST_PATH=0 timeout 600 python bench.py --workload zipf --steps 5 --warmup 2 2>/dev/null
for l in 20 22 24 26; do
 timeout 600 python bench.py --steps 20 --warmup 3 --log2n $l --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('fast  log2n', $l, 'value', d['value'], 'ms', d['ms_per_step'])"
 HJ_FAST_PATH=0 timeout 600 python bench.py --steps 20 --warmup 3 --log2n $l --no-cpu-baseline --no-materialize --no-extras 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('exact log2n', $l, 'value', d['value'], 'ms', d['ms_per_step'])"
done
