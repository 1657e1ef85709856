"""hj_shard_split throughput on one GPU: 2^30 tuples, G = 2, 4, 8 shards (the pre-exchange step of the multi-GPU path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
hj.enable_timings(1)
k, p, ok, op = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
hj.gen_unique(k, n, 0, n, 1); hj.fill_payload(p, n, "rowid"); hj.sync()
for G in (2, 3, 4, 8, 16, 64):
    hj.shard_split(k, p, n, G, ok, op)
    hj.timings_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        counts = hj.shard_split(k, p, n, G, ok, op)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    kt = {a: round(b["total_ms"] / max(b["launches"], 1), 3) for a, b in hj.timings().items() if b["launches"]}
    print("G=%d  %.2f ms  %.2f TB/s (16 B/tuple)  balance %.3f  %s" % (G, dt * 1e3, 16 * n / dt / 1e12, max(counts) / (n / G), kt))
