#!/bin/bash
# kernel timeline (start/end of every launch, default two-stream steps) of a few steps: where a step's time goes between the kernels
# usage: gpu_timeline.sh TAG bench-args...    -> gpurun_out/TAG/timeline.csv (columns trimmed), bench line in timeline.log
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $ROOT/bench.py "$@" > $OUT/timeline.log 2>&1; echo "rc=$?"
f=$(find $OUT/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" $OUT/timeline.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    o.write("start_us,dur_us,queue,kernel\n")
    for r in rows:
        o.write("%.1f,%.1f,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", ""), r["Kernel_Name"][:60].replace(",", ";")))
print(len(rows), "launches")
PY
rm -rf $OUT/tl
grep "^{" $OUT/timeline.log | head -1 | cut -c1-300
