#!/usr/bin/env python3
"""The co-processing join at 2^log2n x 2^log2n under smaller and smaller residency-group budgets (HJ_COPROCESS_GROUP_TUPLES, read per
call): one group (uploads beside the split) against 2, 4, 8, 16 groups (both relations split first, then group by group: upload(g+1)
beside join(g)).  Same context, calls alternating.   python3 tools/experiments/coprocess_groups.py [log2n] [rounds]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << log2n
dev = torch.device("cuda:0")
hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
k = torch.empty(n, dtype=torch.int32, device=dev)
hj.gen_unique(k, n, 0, n, 1); hj.sync(); R = k.cpu().numpy()
hj.gen_unique(k, n, 0, n, 2); hj.sync(); S = k.cpu().numpy()
del k
budgets = {"card": None, "2n": 2 * n, "n": n, "n/2": n // 2, "n/4": n // 4, "n/8": n // 8}
res = {b: [] for b in budgets}
groups = {}
for i in range(len(budgets) * (rounds + 1)):
    name = list(budgets)[i % len(budgets)]
    if budgets[name] is None:
        os.environ.pop("HJ_COPROCESS_GROUP_TUPLES", None)
    else:
        os.environ["HJ_COPROCESS_GROUP_TUPLES"] = str(budgets[name])
    t0 = time.perf_counter()
    m, _ = hj.join_coprocess(R, None, S, None)
    dt = time.perf_counter() - t0
    assert m == n
    groups[name] = hj.coprocess_groups()
    if i >= len(budgets):
        res[name].append(round(dt * 1e3, 2))
print(json.dumps({"log2n": log2n, "budget_tuples -> groups, median ms, calls": {b: [groups[b], sorted(v)[len(v) // 2], v] for b, v in res.items()}}))
