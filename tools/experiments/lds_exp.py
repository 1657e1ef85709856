#!/usr/bin/env python3
"""tools/experiments/lds_exp.py [log2n] [reps] — the two histogram-free passes of both relations, timed per kernel (HIP events, passes serialised),
nothing checked: the attribution builds of the LDS-side analysis (csrc/hj_part.hip HJ_EXP) write garbage on purpose.  Only
k_part1_fast is meaningful under those builds (pass 2 reads what pass 1 wrote).  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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
    hj.partition_both()
    hj.sync()
    hj.enable_timings(1)
    hj.timings_reset()
    for _ in range(reps):
        hj.partition_both()
        hj.sync()
    kt = hj.timings()
    print(json.dumps({"log2n": log2n, "reps": reps, "exp": os.environ.get("HJ_EXP_TAG", ""),
                      "avg_ms": {k: round(v["total_ms"] / v["launches"], 4) for k, v in kt.items() if v["launches"] and k.startswith("k_part")}}))


if __name__ == "__main__":
    main()
