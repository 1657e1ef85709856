"""Config 4's FIRST call on a fresh binding, taken apart: (1) hj_last_call_breakdown of an untimed first call on a fresh context (the process
has already had its device memory mapped once: a throw-away context ran the same join before); (2) the same with hj_enable_timings(2): the
kernels inside the attempt / the sample-and-plan part, so that what is left is host time."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hjtest import pkg
P = pkg()
lr, ls = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (27, 31)
nR, nS = 1 << lr, 1 << ls
dev = torch.device("cuda:0")
Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for n in (nR, nR, nS, nS))
with P.HashJoin(0) as g:
    g.gen_unique(Rk, nR, 0, nR, 3); g.gen_zipf(Sk, nS, 0, nR, 1.0, 4); g.fill_payload(Rp, nR, "ones"); g.fill_payload(Sp, nS, "ones"); g.sync()
for rep, lvl in enumerate((0,) if os.environ.get('ONLY_FIRST') else (0, 0, 0, 2, 0)):
    with P.HashJoin(0) as hj:
        hj.enable_timings(lvl)
        hj.bind_device(P.REL_R, Rk, Rp, nR); hj.bind_device(P.REL_S, Sk, Sp, nS)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); r = hj.join(); first = (time.perf_counter() - t0) * 1e3
        bd = hj.last_call_breakdown()
        tm = {k: round(v["total_ms"], 3) for k, v in hj.timings().items()} if lvl else None
        t0 = time.perf_counter(); hj.join(); second = (time.perf_counter() - t0) * 1e3
        for _ in range(3): hj.join()
        t0 = time.perf_counter()
        for _ in range(10): hj.join()
        steady = (time.perf_counter() - t0) * 100
        print("context %d timings level %d: first call %.2f ms, second %.2f, steady %.2f   breakdown %s   library work of the first call %.2f ms"
              % (rep, lvl, first, second, steady, json.dumps(bd), bd["total_ms"] - bd["allocation_ms"]), flush=True)
        if tm: print("   kernels of the first call (ms):", json.dumps(tm), flush=True)
