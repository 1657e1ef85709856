cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3prof
mkdir -p $OUT
sha256sum $ROOT/icde2019-gpu-join_amd/libhj.so
export HJ_FORK_LOG2=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats30 -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats30.log 2>&1; echo "stats30 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats27 -- python3 $ROOT/bench.py --steps 10 --warmup 3 --log2n 27 --no-cpu-baseline > $OUT/stats27.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statszipf -- python3 $ROOT/bench.py --workload zipf --steps 5 --warmup 2 --no-cpu-baseline > $OUT/statszipf.log 2>&1
for d in stats30 stats27 statszipf; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/$d.kernel_stats.csv; rm -rf $OUT/$d; done
grep "^{" $OUT/stats30.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench under rocprof:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['probe_phase']['avg_launch_ms'], d['materialize']['k_join_materialize_ms'])"
