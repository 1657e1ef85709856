#!/bin/bash
# process-to-process spread on ONE box: the headline bench eight times back to back (value, ms per step, per-launch kernel ms)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/spread; mkdir -p $OUT
sha256sum icde2019-gpu-join_amd/libhj.so | cut -c1-16
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-materialize > $OUT/b$i.json 2>/dev/null
  python3 - <<PY
import json
d=[json.loads(l) for l in open('$OUT/b$i.json') if l.startswith('{')][0]
k=d['kernels']; r=d['roofline']
print('process $i: %.1f Gtuples/s %.2f ms/step  k_part1_fast %.2f  k_part2_fast %.2f  k_join %.2f ms per launch  copy ceiling %.0f GB/s  line-scatter ceiling %.0f' % (d['value'], d['ms_per_step'], k['k_part1_fast']['ms_per_step']/2, k['k_part2_fast']['ms_per_step']/2, k['k_join_count']['ms_per_step'], r['stream_copy_ceiling'], r['line_scatter_ceiling']))
PY
done
