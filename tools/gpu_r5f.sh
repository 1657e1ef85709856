#!/bin/bash
# round 5, call f: the all-valid-wave specialisation of the round body (variant 128): same-context A/B + dynamic instruction mix
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
for s in 30 27; do timeout 500 python3 tools/lds_ab.py $s 4 8 0 128 2>&1 | grep "^{" | tee -a gpurun_out/r5f/ab.txt; done
HJ_WCV=128 timeout 600 python -m pytest tests/test_gpu_join.py -m gpu -x -q -k "parity or ragged or golden or fuzz or large_unique" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for v in 0 128; do
  export HJ_WCV=$v
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES \
     --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5f/pmc$v -- python3 $GRAFT_REPO_ROOT/tools/lds_exp.py 30 2 > $GRAFT_REPO_ROOT/gpurun_out/r5f/pmc$v.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
     --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5f/pmcb$v -- python3 $GRAFT_REPO_ROOT/tools/lds_exp.py 30 2 > $GRAFT_REPO_ROOT/gpurun_out/r5f/pmcb$v.log 2>&1
  echo "pmc $v rc=$?"
done
unset HJ_WCV
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee gpurun_out/r5f/pmc_summary.txt
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5f")
for d in sorted(glob.glob(out + "/pmc*/")):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if "k_part" in name:
                vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(vals.items()):
        print(d.rstrip("/").split("/")[-1], k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
