#!/bin/bash
# tools/pmc_collect.sh <tag> [bench.py args...] — on the GPU box: one rocprofv3 --pmc pass per counter group
# (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ has 8 slots) over a 1-step bench run,
# then tools/pmc_summary.py folds the per-dispatch CSVs into gpurun_out/<tag>/pmc.json.  Copy that file to
# profiles/ to have it judged.  --pmc is never combined with sys/runtime tracing here.
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
MAT=--no-materialize
if [ "$1" = "--with-materialize" ]; then MAT=""; shift; fi
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
GROUPS_=("FETCH_SIZE" "WRITE_SIZE"
  "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM")
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline $MAT --no-extras "$@" > $OUT/p$i.log 2>&1
  echo "== pass $i ($grp): rc=$?"
done
python3 $ROOT/tools/pmc_summary.py $OUT "$@" > $OUT/pmc.json && echo "wrote $OUT/pmc.json"
