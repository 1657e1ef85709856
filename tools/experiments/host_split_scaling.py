#!/usr/bin/env python3
"""Thread scaling of the two host splits on this box (no GPU work): Mtuples/s of hj_host_split_blocks (one pass) and hj_host_split
(two passes) at 2^log2n keys for a list of thread counts.
    python3 tools/experiments/host_split_scaling.py [log2n] [threads,threads,...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g

p = g.load_package()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
n = 1 << log2n
keys = np.random.default_rng(1).permutation(n).astype(np.int32)
out = {"log2n": log2n, "cpus": os.cpu_count(), "one_pass_Mtuples_s": {}, "two_pass_Mtuples_s": {}}
for t in threads:
    for name, fn in (("one_pass_Mtuples_s", lambda: p.host_split_blocks(keys, None, 16, t)[5]), ("two_pass_Mtuples_s", lambda: p.host_split(keys, None, 16, t)[3])):
        best = 0.0
        for _ in range(4):
            best = max(best, fn())          # GB/s of 8 bytes per tuple
        out[name][t] = round(best / 8 * 1e3, 1)
print(json.dumps(out))
