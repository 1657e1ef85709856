#!/bin/bash
# tools/multigpu_session.sh [N] — ONE command for the first session on a multi-GPU node.
#   N = GPUs to use (default: all visible).  Every step runs under its own timeout; every JSON line is copied to
#   profiles/${ROUND}_multigpu_*.json; the summary (what ran, what was skipped and why, return codes, seconds) goes to
#   profiles/${ROUND}_multigpu_session.txt.  Budget on 8 GPUs: <= 25 min (5 tests + 3 x 6 bench + 2 driver).
#   1. the multi-GPU tests that every one-GPU box skips (tests/test_dist.py, tests/test_dist_c.py: RCCL at world 2)
#   2. python bench.py --gpus G for G = 2, 4, 8 <= N: ONE invocation each = the weak-scaling headline over RCCL + the sharded materialising
#      join + the strong-scaling point + both transports (RCCL / copy engines) in a one-process leg
#   3. the C++ driver: icde2019-gpu-join_amd/bench --gpus N (the reference's CLI, both transports)
# With fewer GPUs than a step needs the step is SKIPPED and the summary says so; on a ONE-GPU box the N > 1 plumbing of bench.py and of
# the C++ driver still runs (HJ_BENCH_BACKEND=gloo / HJ_BENCH_SHARE_GPU=1: every rank on GPU 0, labelled "not a measurement").
cd "$(dirname "$0")/.."
ROUND=${ROUND:-r6}
OUT=gpurun_out/multigpu
mkdir -p $OUT profiles
SUM=$OUT/session.txt
NDEV=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
N=${1:-$NDEV}
[ "$N" -gt "$NDEV" ] && N=$NDEV
T_TEST=${T_TEST:-300}; T_BENCH=${T_BENCH:-360}; T_DRV=${T_DRV:-120}
{ echo "multi-GPU session $(date -u +%Y-%m-%dT%H:%M:%SZ): $NDEV GPU(s) visible, using $N; library $(sha256sum icde2019-gpu-join_amd/libhj.so | cut -c1-16)"; } > $SUM
step() { # name, timeout seconds, command...
  local name=$1 t=$2; shift 2
  local t0=$(date +%s)
  timeout $t "$@" > $OUT/$name.out 2> $OUT/$name.err; local rc=$?
  echo "$name: rc=$rc $(( $(date +%s) - t0 )) s$([ $rc -eq 124 ] && echo ' (TIMEOUT)')" | tee -a $SUM
  return $rc
}
skip() { echo "$1: SKIPPED — $2" | tee -a $SUM; }
keep_lines() { # copy the JSON line(s) of a step to profiles/
  grep '^{' $OUT/$1.out > profiles/${ROUND}_multigpu_$1.json 2>/dev/null || rm -f profiles/${ROUND}_multigpu_$1.json
}
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.out 2>&1 || { echo "build failed" | tee -a $SUM; exit 1; }

# 1. the tests (they skip themselves below two GPUs; -rs lists which)
step tests $T_TEST python -m pytest tests/test_dist.py tests/test_dist_c.py -m gpu -q -rs -p no:cacheprovider
grep -E "passed|failed|SKIPPED" $OUT/tests.out | sed 's/^/    /' >> $SUM

# 2. bench.py --gpus G
for G in 2 4 8; do
  if [ "$G" -le "$N" ]; then
    step bench_gpus$G $T_BENCH python bench.py --gpus $G --steps 5 --warmup 2
    keep_lines bench_gpus$G
  elif [ "$NDEV" -eq 1 ] && [ "$G" -eq 2 ]; then
    # one GPU: the N > 1 code path of bench.py itself (rank launch, barriers, max over ranks, the line) over gloo — not a measurement
    HJ_BENCH_BACKEND=gloo step bench_gpus2_plumbing $T_BENCH python bench.py --gpus 2 --log2n 22 --steps 3 --warmup 1 --no-cpu-baseline
    keep_lines bench_gpus2_plumbing
    echo "    (one GPU: plumbing run over gloo, every rank on GPU 0 — is_measurement false)" >> $SUM
  else
    skip bench_gpus$G "needs $G GPUs, $N available"
  fi
done

# 3. the C++ driver with the reference's CLI
if [ "$N" -ge 2 ]; then
  for tr in rccl copy; do
    step driver_gpus${N}_$tr $T_DRV icde2019-gpu-join_amd/bench -b 7 -a HJC -R $((1 << 27)) -S $((1 << 27)) --gpus $N --transport $tr --json
    keep_lines driver_gpus${N}_$tr
  done
elif [ "$NDEV" -eq 1 ]; then
  HJ_BENCH_SHARE_GPU=1 step driver_gpus2_plumbing $T_DRV icde2019-gpu-join_amd/bench -b 7 -a HJC -R $((1 << 22)) -S $((1 << 22)) --gpus 2 --json
  keep_lines driver_gpus2_plumbing
  echo "    (one GPU: HJ_BENCH_SHARE_GPU=1, both ranks on GPU 0 over the device-copy transport — TEST MODE)" >> $SUM
else
  skip driver "no GPU"
fi
cp $SUM profiles/${ROUND}_multigpu_session.txt
cat $SUM
