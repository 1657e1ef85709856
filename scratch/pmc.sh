#!/bin/bash
# usage: scratch/pmc.sh <tag> <counter list...>   (one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-materialize > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== $grp -> $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    if any(x in k for x in ("k_scatter", "k_hist", "k_join")):
        print("%-42s %-28s launches=%d per_launch=%.4g" % (k, c, n, v / n))
PY
done
