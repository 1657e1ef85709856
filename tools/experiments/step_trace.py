"""Per-step times of the headline step over the first N steps of a fresh process (is there a settling phase?), with the device
addresses of the input columns (does the level of a process follow where its buffers landed?).
usage: python tools/experiments/step_trace.py [N=60] [log2n=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.hjtest import pkg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
L = int(sys.argv[2]) if len(sys.argv) > 2 else 30
P = pkg()
n = 1 << L
dev = torch.device("cuda:0")
hj = P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
hj.gen_unique(Rk, n, 0, n, 1); hj.gen_unique(Sk, n, 0, n, 2)
hj.fill_payload(Rp, n, "ones"); hj.fill_payload(Sp, n, "ones")
hj.sync()
hj.bind_device(P.REL_R, Rk, Rp); hj.bind_device(P.REL_S, Sk, Sp)
ts = []
for i in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = hj.join()[0]
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
assert m == n
hj.enable_timings(1); hj.timings_reset()
for i in range(5): hj.join()
kt = hj.timings()
print("median %.3f ms; first steps:" % sorted(ts)[len(ts) // 2], " ".join("%.2f" % t for t in ts[:6]),
      "| kernels", {k: round(v["total_ms"] / v["launches"], 3) for k, v in kt.items() if v["launches"] and v["total_ms"] > 0.5},
      "| inputs at", " ".join("%x" % t.data_ptr() for t in (Rk, Rp, Sk, Sp)))
