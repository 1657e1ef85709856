#!/bin/bash
# tools/gpu_dist_soak.sh [N] [TESTS] — GPU tests (default: the in-process multi-rank tests, one host thread per rank, all ranks on cuda:0)
# N times over, plain and under rocprofv3's kernel trace, with a native backtrace on a fault (tools/segv_bt.c): looks for races between
# the rank threads.  MODE=plain|rocprof|both (default both).
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
N=${1:-6}
TESTS=${2:-tests/test_dist_c.py}
MODE=${MODE:-both}
OUT=$ROOT/gpurun_out/distsoak
rm -rf $OUT && mkdir -p $OUT
gcc -shared -fPIC -O1 -o /tmp/segv_bt.so $ROOT/tools/segv_bt.c || exit 1
export HJ_TEST_SEGV_BT=/tmp/segv_bt.so
cd $ROOT
ulimit -c unlimited
if [ $MODE != rocprof ]; then
for i in $(seq 1 $N); do
  timeout 900 python3 -m pytest $TESTS -m gpu -q -x -p no:cacheprovider -p no:faulthandler > $OUT/plain.$i.log 2>&1; rc=$?
  echo "plain $i rc=$rc $(grep -E 'passed|failed|error' $OUT/plain.$i.log | tail -1)"
done
fi
if [ $MODE != plain ]; then
for i in $(seq 1 $N); do
  timeout 1200 rocprofv3 --disable-signal-handlers true --kernel-trace --stats --output-format csv -d $OUT/trace$i -- python3 -m pytest $TESTS -m gpu -q -x -p no:cacheprovider -p no:faulthandler > $OUT/rocprof.$i.log 2>&1; rc=$?
  echo "rocprof $i rc=$rc $(grep -E 'passed|failed|error' $OUT/rocprof.$i.log | tail -1)"
  if [ $rc -ge 128 ]; then   # a fault: keep the log under its own name and, when the kernel left a core file, every thread's backtrace
    cp $OUT/rocprof.$i.log $OUT/crash.$(date +%H%M%S).log
    cat /proc/sys/kernel/core_pattern
    core=$(ls -t core* /tmp/core* 2>/dev/null | head -1)
    if [ -n "$core" ]; then
      timeout 300 /opt/rocm/bin/rocgdb -batch -ex "info threads" -ex "thread apply all bt 25" $(which python3) $core > $OUT/crash_backtraces.$(date +%H%M%S).txt 2>&1
      ls -la $core; rm -f $core
    else echo "no core file"; fi
  fi
  rm -rf $OUT/trace$i
done
fi
grep -l "native backtrace" $OUT/*.log | head
