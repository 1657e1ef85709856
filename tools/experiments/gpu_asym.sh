#!/bin/bash
# round 4: relations of very different sizes — the larger one's passes on a high-priority second stream (partition_both).
# Same-box check of the shipped rule against the serial order (HJ_FORK_LOG2=0), at config 4 and at 2^24 x 2^27; skew + join parity.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4asym; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_skew.py tests/test_gpu_join.py -m gpu -q -x > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
run() { name=$1; args=$2; shift; shift
  env "$@" timeout 600 python bench.py $args --no-cpu-baseline --no-extras > $OUT/$name.json 2> $OUT/$name.err
  python3 - <<PY
import json
d=[json.loads(l) for l in open('$OUT/$name.json') if l.startswith('{')]
print('$name', d[0]['value'], d[0]['ms_per_step'], 'mat', d[0]['materialize']['ms_per_step']) if d else print('$name FAILED')
PY
}
for i in 1 2 3; do
  run z2731_shipped_$i "--workload zipf --steps 5 --warmup 2" HJ_X=0
  run z2731_serial_$i "--workload zipf --steps 5 --warmup 2" HJ_FORK_LOG2=0
  run z2427_shipped_$i "--workload zipf --zipf-sizes 24 27 --steps 20 --warmup 3" HJ_X=0
  run z2427_serial_$i "--workload zipf --zipf-sizes 24 27 --steps 20 --warmup 3" HJ_FORK_LOG2=0
done
