#!/bin/bash
# round-3 quick loop: join parity tests, then the headline bench (with the materialising leg) at 2^30 and 2^27
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r3b}
mkdir -p gpurun_out/$TAG
timeout 1200 python -m pytest tests/test_gpu_join.py tests/test_gpu_skew.py -m gpu -x -q ${TEST_K:+-k "$TEST_K"} > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/$TAG/tests.log
for l in ${SIZES:-30 27}; do
timeout 600 python bench.py --steps 6 --warmup 2 --log2n $l --no-cpu-baseline --no-extras $BENCH_ARGS 2>gpurun_out/$TAG/bench$l.err | tee gpurun_out/$TAG/bench$l.json | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('log2n', $l, 'value', d['value'], 'ms', d['ms_per_step'], {k:round(v['ms_per_step']/v['launches_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']>0.1}, 'mat', d.get('materialize'))
"
tail -3 gpurun_out/$TAG/bench$l.err
done
