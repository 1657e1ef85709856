#!/usr/bin/env python3
"""Same-context A/B of the co-processing join's host split: one pass into blocks with the uploads running beside it (the default), the same
with the uploads after the split (HJ_COPROCESS_SPLIT=3), and the two-pass split of round 4
(HJ_COPROCESS_SPLIT=2, read per call), alternating calls on the same context and the same host arrays.
    python3 tools/experiments/coprocess_ab.py [log2n] [rounds]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << log2n
dev = torch.device("cuda:0")
hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
k = torch.empty(n, dtype=torch.int32, device=dev)
hj.gen_unique(k, n, 0, n, 1); hj.sync(); R = k.cpu().numpy()
hj.gen_unique(k, n, 0, n, 2); hj.sync(); S = k.cpu().numpy()
del k
VARIANTS = {"one_pass_streamed": "1", "one_pass_upload_after": "3", "two_pass": "2"}
res = {k_: [] for k_ in VARIANTS}
for i in range(len(VARIANTS) * (rounds + 1)):
    name = list(VARIANTS)[i % len(VARIANTS)]
    os.environ["HJ_COPROCESS_SPLIT"] = VARIANTS[name]
    t0 = time.perf_counter()
    m, _ = hj.join_coprocess(R, None, S, None)
    dt = time.perf_counter() - t0
    assert m == n
    if i >= len(VARIANTS):
        res[name].append((round(dt * 1e3, 2), round(hj.host_split_throughput(), 2)))
out = {"log2n": log2n, "cpus": os.cpu_count()}
for name, v in res.items():
    ms = sorted(x[0] for x in v)
    out[name] = {"ms": [x[0] for x in v], "median_ms": ms[len(ms) // 2], "split_GBs": [x[1] for x in v]}
print(json.dumps(out))
