#!/bin/bash
# round-4 loop A: hj_dist robustness tests, the skew suite, config 4 with its materialising leg, one phantom-8 line
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4a
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_dist_c.py tests/test_dist.py tests/test_gpu_skew.py -m gpu -x -q > $OUT/tests.txt 2>&1; echo "tests rc=$?"
tail -12 $OUT/tests.txt
timeout 900 python bench.py --workload zipf --steps 5 --warmup 2 > $OUT/bench_zipf.json 2> $OUT/bench_zipf.err; echo "zipf rc=$?"
tail -3 $OUT/bench_zipf.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r4a/bench_zipf.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("zipf", d["value"], d["ms_per_step"], "first", d["first_call_ms"], d["kernels"], "mat", d.get("materialize"))
PY
timeout 600 python bench.py --steps 5 --warmup 2 --force-dist --phantom 8 --no-cpu-baseline > $OUT/bench_phantom8.json 2> $OUT/phantom8.err; echo "phantom rc=$?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r4a/bench_phantom8.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("phantom8", d["ms_per_step"], d["roofline"], d["dist"].get("model"))
PY
