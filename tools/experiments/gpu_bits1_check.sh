#!/bin/bash
# the 7-bit first pass for relations of similar size (choose_bits): the suite and the differential runs with the rule in place, then the bench line of
# config 2 with the rule off (HJ_BITS1_SIMILAR=9) and on, processes alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/bits1; mkdir -p $O
sha256sum icde2019-gpu-join_amd/libhj.so > $O/out.txt
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $O/out.txt
timeout 900 python tools/fuzz_medium.py 36 2>&1 | tail -1 | tee -a $O/out.txt
timeout 900 python tools/fuzz_more.py 40 600 2>&1 | tail -1 | tee -a $O/out.txt
timeout 900 python tools/fuzz_dist.py 300 2>&1 | tail -1 | tee -a $O/out.txt
fi
for rep in 1 2 3; do for l in ${SIZES:-27}; do for v in 9 7; do
HJ_BITS1_SIMILAR=$v timeout 300 python bench.py --log2n $l --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']; print('2^$l HJ_BITS1_SIMILAR=$v', d['value'], d['ms_per_step'], {k:round(v['ms_per_step']/v['launches_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']>0.02}, 'materialising step', (d.get('materialize') or {}).get('ms_per_step'), 'roofline', r['kernel'], r['achieved'], r['frac'])" | tee -a $O/out.txt
done; done; done
