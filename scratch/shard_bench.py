# time hj_shard_split (level-0 split of the multi-GPU path) on one GPU, 2^30 tuples
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
P = g.load_package()
n = 1 << 30
dev = torch.device("cuda:0")
hj = P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
k = torch.empty(n, dtype=torch.int32, device=dev); p = torch.empty(n, dtype=torch.int32, device=dev)
ok = torch.empty_like(k); op = torch.empty_like(p)
hj.gen_unique(k, n, 0, n, 1); hj.fill_payload(p, n, "ones"); hj.sync()
for variant in (None, "4"):
    if variant: os.environ["HJ_SCATTER_VARIANT"] = variant
    h2 = P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
    for w in (2, 4, 8):
        h2.shard_split(k, p, n, w, ok, op)
        h2.timings_reset()
        t0 = time.perf_counter()
        for _ in range(3):
            c = h2.shard_split(k, p, n, w, ok, op)
        dt = (time.perf_counter() - t0) / 3
        t = h2.timings()
        print("variant", variant, "shards", w, "ms", round(dt * 1e3, 2), {kk: round(v["total_ms"] / 3, 2) for kk, v in t.items() if v["total_ms"] > 0.3}, "balance", round(max(c) / (n / w), 4))
