#!/bin/bash
# the dispatch parameters rocprofv3 records for every kernel of a 2^24 step (LDS block size, VGPRs, SGPRs, scratch, workgroup size)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ktrace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktrace -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --log2n 24 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-materialize > $GRAFT_REPO_ROOT/gpurun_out/ktrace/bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/ktrace -name "*kernel_trace.csv" | head -1)
echo $f
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
seen={}
for r in rows:
    k=r["Kernel_Name"][:60]
    if k not in seen:
        seen[k]=1
        print(k, {c:r[c] for c in r if c in ("LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size","Grid_Size","Workgroup_Size_X","Grid_Size_X")})
PY
