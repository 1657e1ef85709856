import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as o
print("affinity", len(os.sched_getaffinity(0)), "omp max", o.max_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
n = 1 << 26
rng = np.random.default_rng(1)
R = rng.permutation(n).astype(np.int32); S = rng.permutation(n).astype(np.int32)
for t in (8, 16, 32, 64, 128, 256):
    t0 = time.perf_counter(); m, _ = o.radix_join_omp(R, None, S, None, 7, 7, t); dt = time.perf_counter() - t0
    print(t, m == n, round(2 * n / dt / 1e9, 3), "Gtuples/s", flush=True)
