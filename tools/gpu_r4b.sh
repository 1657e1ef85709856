#!/bin/bash
# round-4 loop B: full GPU suite after the pruning, config 4 with its materialising leg and the first-call breakdown (cold process,
# HJ_DEBUG allocation log), then the headline and config 2
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4b
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; echo "tests rc=$?"
tail -5 $OUT/tests.txt
HJ_DEBUG=1 timeout 900 python bench.py --workload zipf --steps 5 --warmup 2 > $OUT/bench_zipf.json 2> $OUT/bench_zipf.err; echo "zipf rc=$?"
grep -v "parent " $OUT/bench_zipf.err | tail -30
python3 - <<'PY'
import json
for l in open("gpurun_out/r4b/bench_zipf.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("zipf", d["value"], d["ms_per_step"], "first", d["first_call_ms"], d["first_call_split_ms"], d["kernels"], "mat", d.get("materialize"))
PY
for l in 30 27; do
timeout 600 python bench.py --steps 10 --warmup 3 --log2n $l --no-cpu-baseline 2>/dev/null | tee $OUT/bench$l.json | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('log2n', $l, 'value', d['value'], 'ms', d['ms_per_step'], {k:round(v['ms_per_step']/v['launches_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']>0.1}, 'mat', (d.get('materialize') or {}).get('value'), 'roof', d['roofline']['frac'], 'probe', d['probe_phase']['frac_of_8TBs'])
"
done
