import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from tests.hjtest import pkg
P = pkg()
dev = torch.device("cuda:0")
for L in (16, 20, 24):
    n = 1 << L
    hj = P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
    Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    hj.gen_unique(Rk, n, 0, n, 1); hj.gen_unique(Sk, n, 0, n, 2)
    hj.fill_payload(Rp, n, "ones"); hj.fill_payload(Sp, n, "ones"); hj.sync()
    hj.bind_device(P.REL_R, Rk, Rp); hj.bind_device(P.REL_S, Sk, Sp)
    for _ in range(20): hj.join()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): hj.join()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200 * 1e3
    hj.enable_timings(2); hj.timings_reset()
    for _ in range(10): hj.join()
    kt = hj.timings()
    print("2^%d: %.4f ms/step" % (L, dt), {k: (v["launches"] / 10, round(v["total_ms"] / 10 * 1e3, 1)) for k, v in kt.items() if v["launches"]})
    hj.close() if hasattr(hj, "close") else None
