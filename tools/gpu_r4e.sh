#!/bin/bash
# round-4 loop E: positions in lines (relations beyond 2^32 tuples): the full GPU suite, then the headline and config 2 / config 4
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4e
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; echo "tests rc=$?"
tail -15 $OUT/tests.txt
for l in 30 27; do
timeout 600 python bench.py --steps 10 --warmup 3 --log2n $l --no-cpu-baseline 2>/dev/null | tee $OUT/bench$l.json | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('log2n', $l, 'value', d['value'], 'ms', d['ms_per_step'], {k:round(v['ms_per_step']/v['launches_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']>0.1}, 'mat', (d.get('materialize') or {}).get('value'), 'roof', d['roofline']['frac'], 'probe', d['probe_phase']['frac_of_8TBs'])
"
done
timeout 900 python bench.py --workload zipf --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_zipf.json | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('zipf', d['value'], d['ms_per_step'], 'first', d['first_call_ms'], d['first_call_split_ms'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()}, 'mat', d['materialize']['ms_per_step'], d['materialize']['k_join_materialize_frac_of_8TBs'])
"
