#!/usr/bin/env python3
"""tools/experiments/launch_ab.py [rounds] [sizes...] — same-process A/B of the launch structure of a step: contexts that share
the SAME input columns and differ in $HJ_MERGE_LOG2 (both relations' passes in one launch per pass, no k_set_root, no event fork) and
$HJ_PLAN_ATOMIC (plan + expand in one launch), interleaved round-robin; ms per hj_join step (wall, synchronised)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

VARIANTS = [("r4: two streams, plan+scan+expand", {"HJ_MERGE_LOG2": "0", "HJ_PLAN_ATOMIC": "0", "HJ_FORK_LOG2": "40"}),
            ("atomic plan only", {"HJ_MERGE_LOG2": "0", "HJ_PLAN_ATOMIC": "1", "HJ_FORK_LOG2": "40"}),
            ("merged passes only", {"HJ_MERGE_LOG2": "40", "HJ_PLAN_ATOMIC": "0", "HJ_FORK_LOG2": "40"}),
            ("merged passes + atomic plan", {"HJ_MERGE_LOG2": "40", "HJ_PLAN_ATOMIC": "1", "HJ_FORK_LOG2": "40"}),
            ("one stream, no merge, r4 plan", {"HJ_MERGE_LOG2": "0", "HJ_PLAN_ATOMIC": "0", "HJ_FORK_LOG2": "0"})]


def main():
    import torch
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    sizes = [int(x) for x in sys.argv[2:]] or [24, 26, 27, 28]
    pkg = graft.load_package()
    dev = torch.device("cuda", 0)
    for log2n in sizes:
        n = 1 << log2n
        # ONE context, the same partition buffers for every variant: the knobs are read per call
        g = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
        ctxs = [(name, g) for name, _ in VARIANTS]
        envs = dict(VARIANTS)
        Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
        g.gen_unique(Rk, n, 0, n, 1)
        g.gen_unique(Sk, n, 0, n, 2)
        g.fill_payload(Rp, n, "ones")
        g.fill_payload(Sp, n, "ones")
        g.sync()
        reps = max(10, min(200, (1 << 31) // n // 4))
        acc = {name: [] for name, _ in VARIANTS}
        for name, hj in ctxs:
            os.environ.update(envs[name])
            hj.reload_knobs()                        # (the library reads its knobs once, in hj_create)
            hj.bind_device(pkg.REL_R, Rk, Rp)
            hj.bind_device(pkg.REL_S, Sk, Sp)
            for _ in range(3):
                assert hj.join()[0] == n
        for r in range(rounds):
            for name, hj in ctxs:
                os.environ.update(envs[name])
                hj.reload_knobs()
                hj.join()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    hj.bind_device(pkg.REL_R, Rk, Rp)
                    hj.bind_device(pkg.REL_S, Sk, Sp)
                    m = hj.join()[0]
                torch.cuda.synchronize()
                acc[name].append((time.perf_counter() - t0) / reps * 1e3)
                assert m == n
        base = sorted(acc[VARIANTS[0][0]])[rounds // 2]
        for name, _ in VARIANTS:
            med = sorted(acc[name])[rounds // 2]
            print(json.dumps({"log2n": log2n, "variant": name, "median_ms": round(med, 4), "vs_r4": round(med / base, 4), "Gtuples_s": round(2 * n / med / 1e6, 2),
                              "rounds_ms": [round(x, 4) for x in acc[name]]}))
        g.close()
        del Rk, Rp, Sk, Sp


if __name__ == "__main__":
    main()
