#!/usr/bin/env python3
"""Step time (partition R + partition S + build/probe + count read-back) over input sizes, eager launches vs the step replayed
from a captured hipGraph (hj_config.graph).  The context runs on its own stream (HIP's legacy default stream cannot be captured)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
print("%-6s %12s %12s %10s  %s" % ("size", "eager ms", "graph ms", "graph/eager", "radix bits"))
for lg in (16, 18, 20, 22, 24, 26, 27, 28, 30):
    n = 1 << lg
    res = {}
    for mode in ("eager", "graph"):
        hj = pkg.HashJoin(0)
        hj.configure(graph=(mode == "graph"))
        Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4))
        hj.gen_unique(Rk, n, 0, n, 1)
        hj.gen_unique(Sk, n, 0, n, 2)
        hj.fill_payload(Rp, n, "ones")
        hj.fill_payload(Sp, n, "ones")
        hj.sync()
        hj.bind_device(pkg.REL_R, Rk, Rp)
        hj.bind_device(pkg.REL_S, Sk, Sp)
        for _ in range(4):
            assert hj.join()[0] == n
        steps = 200 if lg <= 24 else 20
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m = hj.join()[0]
        dt = (time.perf_counter() - t0) / steps
        assert m == n
        res[mode] = dt * 1e3
        bits = (hj.config()["bits1"], hj.config()["bits2"])
        hj.close()
        del Rk, Rp, Sk, Sp
    print("2^%-4d %12.4f %12.4f %10.2f  %s" % (lg, res["eager"], res["graph"], res["graph"] / res["eager"], bits))
