#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
timeout 600 python bench.py --steps 6 --warmup 2 --log2n 27 --no-cpu-baseline > gpurun_out/r4q/b27.json 2> gpurun_out/r4q/b27.err; echo "rc=$?"; tail -3 gpurun_out/r4q/b27.err
python3 -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r4q/b27.json') if l.startswith('{')][0]
print(d['value'], d['ms_per_step'], d['config']['radix_bits'], d['config2_as_stated'], d['materialize'])"
timeout 900 python bench.py --workload zipf --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r4q/bz.json 2> gpurun_out/r4q/bz.err; echo "rc=$?"; tail -3 gpurun_out/r4q/bz.err
python3 -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r4q/bz.json') if l.startswith('{')][0]
print(d['value'], d['ms_per_step'], d['first_call_ms'], d['materialize'])"
