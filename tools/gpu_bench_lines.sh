#!/bin/bash
# the three bench lines that carry roofline.traffic, re-run once the PMC files of THIS binary are in profiles/ (the traffic field is
# filled only when the lib_sha256 stored in profiles/r4_pmc_*.json equals the running library's)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4lines
mkdir -p $OUT
sha256sum icde2019-gpu-join_amd/libhj.so
python bench.py --steps 10 --warmup 3 > $OUT/bench_2p30.json 2>/dev/null; echo "rc=$?"
python bench.py --steps 10 --warmup 3 --log2n 27 > $OUT/bench_2p27.json 2>/dev/null; echo "rc=$?"
python bench.py --workload zipf --steps 5 --warmup 2 > $OUT/bench_zipf.json 2>/dev/null; echo "rc=$?"
python3 - <<'PY'
import json
for f in ("2p30","2p27","zipf"):
    d=[json.loads(l) for l in open("gpurun_out/r4lines/bench_%s.json"%f) if l.startswith("{")][0]
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], (d.get("materialize") or {}).get("value"))
PY
