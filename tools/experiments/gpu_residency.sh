#!/bin/bash
# how many count-kernel workgroups does a CU really hold?  census by per-workgroup stamps (-DHJ_STAMPS build) over LDS table sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fixed
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_shipped.so
cp $P/libhj_stamps.so $P/libhj.so
touch $P/libhj.so $P/bench
summ='
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l)
        if d.get("kernel")=="k_join_count": print(sys.argv[1], "cap", d["lds_capacity"], "heads", d["lds_heads"], "resident_per_cu", d["resident_per_cu"], "event_ms", d["event_ms"], "steady wg_us", d["steady_workgroup_us"], "first_end", d["first_end_us"]["median"], "tail_us", d["tail_us"], "GB/s", d["bytes_over_event_GBs"], "steady GB/s", d["steady_rate_GBs"])'
for cfg in "0 0" "4608 2048" "4352 4096" "4352 2048" "4352 1024"; do
  timeout 600 python tools/experiments/fixed_cost.py 30 2 $cfg 2>/dev/null | tee -a gpurun_out/fixed/residency_30.jsonl | python3 -c "$summ" "2^30" | tail -1
done
for cfg in "0 0" "4352 4096" "4352 1024" "4224 1024"; do
  timeout 600 python tools/experiments/fixed_cost.py 27 2 $cfg 2>/dev/null | tee -a gpurun_out/fixed/residency_27.jsonl | python3 -c "$summ" "2^27" | tail -1
done
cp $P/libhj_shipped.so $P/libhj.so
