#!/usr/bin/env python3
"""tools/experiments/lds_ab.py [log2n] [rounds] [reps] [variants...] — same-process A/B of the LDS-side variants of the write-combining rounds
(csrc/hj_part.hip: template parameter WV of wc_fast, selected per launch by $HJ_WCV): the SAME buffers, the variants interleaved
round-robin, per-kernel times from HIP events with the passes serialised; every variant's join count is checked first."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    variants = [int(x) for x in sys.argv[4:]] or [0, 16, 32, 48, 64, 112]
    pkg = graft.load_package()
    n = 1 << log2n
    dev = torch.device("cuda", 0)
    hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
    Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    hj.gen_unique(Rk, n, 0, n, 1)
    hj.gen_unique(Sk, n, 0, n, 2)
    hj.fill_payload(Rp, n, "ones")
    hj.fill_payload(Sp, n, "ones")
    hj.sync()
    hj.bind_device(pkg.REL_R, Rk, Rp)
    hj.bind_device(pkg.REL_S, Sk, Sp)
    for v in variants:
        os.environ["HJ_WCV"] = str(v)
        assert hj.join()[0] == n, v
    hj.enable_timings(1)
    acc = {v: {} for v in variants}
    for r in range(rounds):
        for v in variants:
            os.environ["HJ_WCV"] = str(v)
            hj.timings_reset()
            for _ in range(reps):
                hj.partition_both()
                hj.sync()
            for k, t in hj.timings().items():
                if t["launches"] and k.startswith("k_part"):
                    acc[v].setdefault(k, []).append(t["total_ms"] / t["launches"])
    base = acc[variants[0]]
    for v in variants:
        row = {k: [round(x, 4) for x in xs] for k, xs in acc[v].items()}
        med = {k: sorted(xs)[len(xs) // 2] for k, xs in acc[v].items()}
        rel = {k: round(med[k] / sorted(base[k])[len(base[k]) // 2], 4) for k in med}
        print(json.dumps({"log2n": log2n, "wv": v, "median_ms": {k: round(x, 4) for k, x in med.items()}, "vs_first": rel, "rounds": row}))


if __name__ == "__main__":
    main()
