#!/bin/bash
# tools/experiments/gpu_lds_exp.sh — LDS-side analysis of wc_fast, one box: the experiment builds of `make exp`
# (libhj_exp<N>.so, csrc/hj_part.hip HJ_EXP) copied over libhj.so one after the other; per build the pass kernels' times (two
# alternating rounds) and one rocprofv3 --pmc pass of the LDS counters.  Output: gpurun_out/lds/.
cd $GRAFT_REPO_ROOT
P=icde2019-gpu-join_amd
OUT=$GRAFT_REPO_ROOT/gpurun_out/lds
mkdir -p $OUT
cp $P/libhj.so $P/libhj_ship.so
VARIANTS=${VARIANTS:-"0 1 2 4 8 15 16 32 64 48 112"}
lib() { if [ "$1" = "0" ]; then cp $P/libhj_ship.so $P/libhj.so; else cp $P/libhj_exp$1.so $P/libhj.so; fi; }
for rep in 1 2; do
  for v in $VARIANTS; do
    lib $v
    HJ_EXP_TAG=$v timeout 300 python3 tools/experiments/lds_exp.py 30 10 2>/dev/null | tee -a $OUT/times.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for v in $VARIANTS; do
  ( cd $GRAFT_REPO_ROOT && lib $v )
  HJ_EXP_TAG=$v timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT \
     --kernel-trace --output-format csv -d $OUT/pmc$v -- python3 $GRAFT_REPO_ROOT/tools/experiments/lds_exp.py 30 2 > $OUT/pmc$v.log 2>&1
  echo "pmc $v rc=$?"
done
cd $GRAFT_REPO_ROOT
lib 0
python3 - <<'PY' | tee $OUT/pmc_summary.txt
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "lds")
for d in sorted(glob.glob(out + "/pmc*/"), key=lambda x: int(x.rstrip("/").split("pmc")[-1])):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if "k_part" in name:
                vals[name.split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(vals.items()):
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        extra = ""
        if e.get("SQ_LDS_IDX_ACTIVE"):
            extra = " conflict/active %.3f" % (e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"])
        print(d.rstrip("/").split("/")[-1], k, {c: "%.4g" % v for c, v in sorted(e.items())}, extra)
PY
