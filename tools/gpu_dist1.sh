#!/bin/bash
# one-GPU runs of the multi-GPU driver: the hj_dist tests, world 1 over RCCL, the phantom-world shapes (G = 2, 4, 8) with the timeline model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_dist_c.py -m gpu -x -q > gpurun_out/r3f/dist_c.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3f/dist_c.log
for args in ${RUNS:-"--force-dist" "--force-dist --phantom 2" "--force-dist --phantom 4" "--force-dist --phantom 8" "--force-dist --phantom 8 --single-group" "--force-dist --phantom 8 --slices 8" "--force-dist --dist-impl torch"}; do
  tag=$(echo $args | tr -d ' -')
  timeout 900 python bench.py --steps 5 --warmup 2 --log2n ${LOG2N:-30} --no-cpu-baseline $args > gpurun_out/r3f/bench_$tag.json 2> gpurun_out/r3f/bench_$tag.err; echo "$args rc=$?"
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r3f/bench_$tag.json") if l.startswith("{")][-1])
    print("  value", d["value"], "ms", d["ms_per_step"], "model", json.dumps(d["dist"].get("model")), "rank0", {k:(round(v,2) if isinstance(v,float) else v) for k,v in d["dist"].get("rank0",{}).items() if k.endswith("_ms") or k in ("path","slices","probe_groups")})
except Exception as e:
    print("  no line:", e); print(open("gpurun_out/r3f/bench_$tag.err").read()[-1500:])
PY
done
