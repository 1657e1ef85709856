#!/usr/bin/env python3
"""tools/experiments/hot_phases.py [log2R] [log2S] — where a round of the bypassing pass 1 spends its time (config 4: 27 31): phase A (kept
stores, table lookups, ranks), B (placement), C (flush of the ordinary lines; HOT 3: + the wait for the reservation), and the hot flush, as
s_memrealtime ticks summed per workgroup by a -DHJ_STAMPS build (gpu_hot_phases.sh).  One JSON line per mode: count (HOT 1), write (HOT 2), and
write with capacity 0 (every kernel runs, no output tuple is stored)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    lr = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    ls = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    pkg = graft.load_package()
    nR, nS = 1 << lr, 1 << ls
    dev = torch.device("cuda", 0)
    hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
    Rk, Rp = (torch.empty(nR, dtype=torch.int32, device=dev) for _ in range(2))
    Sk, Sp = (torch.empty(nS, dtype=torch.int32, device=dev) for _ in range(2))
    hj.gen_unique(Rk, nR, 0, nR, 3)
    hj.gen_zipf(Sk, nS, 0, nR, 1.0, 4)
    hj.fill_payload(Rp, nR, "ones")
    hj.fill_payload(Sp, nS, "ones")
    hj.sync()
    expect = nS - int((Sk == nR).sum().item())
    hj.bind_device(pkg.REL_R, Rk, Rp)
    hj.bind_device(pkg.REL_S, Sk, Sp)
    assert hj.join()[0] == expect
    st = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
    sj = torch.zeros(8, dtype=torch.int64, device=dev)
    out = [torch.empty(expect, dtype=torch.int32, device=dev) for _ in range(3)]
    hj.enable_timings(1)

    def report(name):
        a = st.cpu().numpy().reshape(-1, 8)
        a = a[a[:, 4] > 0]
        r = a[:, 4].astype(np.float64)
        us = lambda col: round(float(np.median(a[:, col] / r)) * 0.01, 3)   # ticks of 10 ns per round -> us
        kt = hj.timings()
        k = kt.get("k_part1_var", {"launches": 1, "total_ms": 0})
        print(json.dumps({"mode": name, "workgroups": int(len(a)), "rounds_per_workgroup": int(np.median(r)) if len(a) else None,
                          "us_per_round": {"A": us(0), "B": us(1), "C": us(2), "wait_at_end_of_C": us(3), "sum": round(us(0) + us(1) + us(2) + us(3), 3)} if len(a) else None,
                          "k_part1_var_ms": round(k["total_ms"] / max(1, k["launches"]), 3), "hot": hj.hot_stats()}))

    def nostore():   # capacity 0: every kernel runs, no output tuple is stored (the call reports HJ_ECAPACITY with the true size)
        try:
            hj.join_and_materialize_into(out[0], out[1], out[2], 0)
        except pkg.HJError as e:
            assert e.code == pkg.ECAPACITY, e
        return expect

    for name, env, fn in (("count (HOT 1)", None, lambda: hj.join()[0]),
                          ("write, NO output stores (HOT 2, capacity 0)", None, nostore),
                          ("write (HOT 2)", None, lambda: hj.join_and_materialize_into(out[0], out[1], out[2], expect))):
        got = fn()
        assert got == expect or os.environ.get("HOT_PHASES_NOCHECK"), (got, expect)
        st.zero_()
        hj.debug_set_stamps(sj, st)
        hj.timings_reset()
        got = fn()
        assert got == expect or os.environ.get("HOT_PHASES_NOCHECK"), (got, expect)
        hj.sync()
        report(name)
        hj.debug_set_stamps(None, None)


if __name__ == "__main__":
    main()
