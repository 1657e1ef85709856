"""Extended run of tests/test_gpu_join.py::test_fuzz_against_oracle over many more seeds (GPU box, a few minutes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_join as T
from hjtest import pkg
P = pkg()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # seconds; 0 = no limit (a soak stops by itself and still prints its summary)
t0 = time.time(); bad = 0
for seed in range(lo, hi):
    if budget and time.time() - t0 > budget:
        hi = seed
        break
    try:
        T.test_fuzz_against_oracle(P, seed)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL seed", seed, repr(e)[:300], flush=True)
        if bad > 5: break
print("fuzz seeds %d..%d: %d failures, %.0f s" % (lo, hi, bad, time.time() - t0))
