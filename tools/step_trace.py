"""Per-step times of the headline step over the first N steps of a fresh process (is there a settling phase?).
usage: python tools/step_trace.py [N=60] [log2n=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
L = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tests.hjtest import pkg
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
print("steps ms:", " ".join("%.2f" % t for t in ts))
