#!/bin/bash
# Round-4 gate: the histogram-free passes as 512- / 256-thread workgroups (256 / 128 LDS lines, two / four per
# CU) against the 1024-thread default, same box, alternating.  First the parity test of the new geometries, then the sweep.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4gate
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_join.py -q -x -k "workgroup_geometries or fast_path_layout or partition_parity" > $OUT/tests.txt 2>&1; echo "tests rc=$?"
tail -3 $OUT/tests.txt
summ='
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%-44s" % sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], d["config"]["radix_bits"], {k:round(v["ms_per_step"]/v["launches_per_step"],4) for k,v in d["kernels"].items() if v["ms_per_step"]>0.05})'
run() { # log2n, env, extra args
  env $2 timeout 600 python bench.py --steps 20 --warmup 3 --log2n $1 --no-cpu-baseline --no-materialize --no-extras $3 2>/dev/null | python3 -c "$summ" "[2^$1 $2 $3]"
}
for rep in 1 2; do
  for cfg in "X=1|" "HJ_WG2=512|" "HJ_WG2=256|" "HJ_WG2=512 HJ_TARGET_SPANS=512|" \
             "HJ_WG1=512 HJ_WG2=512|--bits 8 7" "HJ_WG1=512 HJ_WG2=256|--bits 8 7" "HJ_WG1=512 HJ_WG2=1024|--bits 8 7" "X=1|--bits 8 7" \
             "HJ_WG1=512 HJ_WG2=512|--bits 8 8" "X=1|--bits 8 8"; do
    run 27 "${cfg%%|*}" "${cfg##*|}"
  done
done | tee $OUT/sweep27.txt
for rep in 1 2; do
  for cfg in "X=1|" "HJ_WG2=512|" "HJ_WG2=256|"; do run 26 "${cfg%%|*}" "${cfg##*|}"; run 28 "${cfg%%|*}" "${cfg##*|}"; done
  for cfg in "X=1|" "HJ_WG2=512|--bits 9 8"  "X=1|--bits 9 8"; do run 29 "${cfg%%|*}" "${cfg##*|}"; done
  run 30 "X=1" ""
done | tee $OUT/sweep_sizes.txt
# the 8-way shard split of the multi-GPU path in the shape of an 8-GPU job (one slice: as many spans as the geometry wants)
for rep in 1 2; do
  for e in "X=1" "HJ_WG0=512" "HJ_WG0=256"; do
    env $e timeout 600 python bench.py --steps 5 --warmup 2 --log2n 28 --force-dist --phantom 8 --slices 1 --no-cpu-baseline 2>/dev/null | python3 -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); s=(d.get("dist") or {}).get("rank0",{}); print(sys.argv[1], "ms", d["ms_per_step"], {k:s.get(k) for k in ("split_ms","pass1_ms","spans_per_slice","slices","path")})' "[$e]"
  done
done | tee $OUT/split8.txt
