#!/usr/bin/env python3
"""tools/experiments/zipf_ab.py [rounds] — same-CONTEXT A/B of how the big relation of config 4 (PK-FK 2^27 x 2^31) is partitioned: one context, the
same columns and partition buffers, variants switched by hj_configure (radix bits of the two passes) and per-call experiment knobs
(HJ_FORCE_SAMPLED, HJ_TARGET_SPANS, HJ_VAR_GUIDE, HJ_REPLAN), interleaved; per-kernel ms from HIP events (passes serialised)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

KNOBS = ("HJ_FORCE_SAMPLED", "HJ_TARGET_SPANS", "HJ_VAR_GUIDE")


def main():
    import torch
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    lr, ls = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (27, 31)
    pkg = graft.load_package()
    dev = torch.device("cuda", 0)
    nR, nS = 1 << lr, 1 << ls
    hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
    Rk, Rp = (torch.empty(nR, dtype=torch.int32, device=dev) for _ in range(2))
    Sp = torch.empty(nS, dtype=torch.int32, device=dev)
    data = {}
    for name, theta in (("zipf1.0", 1.0), ("uniformFK", 0.0)):
        k = torch.empty(nS, dtype=torch.int32, device=dev)
        hj.gen_zipf(k, nS, 0, nR, theta, 4)
        data[name] = k
    hj.gen_unique(Rk, nR, 0, nR, 3)
    hj.fill_payload(Rp, nR, "ones")
    hj.fill_payload(Sp, nS, "ones")
    hj.sync()
    expect = {name: nS - int((k == nR).sum().item()) for name, k in data.items()}
    b = lr - 12
    if lr <= 27:
        variants = [  # (label, data, hj_configure arguments, env)
            ("zipf sampled 8+7 (default)", "zipf1.0", {}, {}),
            ("zipf sampled 9+6", "zipf1.0", dict(bits1=9, bits2=b - 9), {}),
            ("zipf sampled 7+8", "zipf1.0", dict(bits1=7, bits2=b - 7), {}),
            ("zipf exact passes", "zipf1.0", dict(exact_only=True), {}),
            ("uniform FK plain 9+6", "uniformFK", {}, {"HJ_FORCE_SAMPLED": "8"}),
            ("uniform FK plain 8+7", "uniformFK", dict(bits1=8, bits2=b - 8), {"HJ_FORCE_SAMPLED": "8"}),
            ("uniform FK sampled 9+6", "uniformFK", dict(bits1=9, bits2=b - 9), {"HJ_FORCE_SAMPLED": "2"}),
            ("uniform FK sampled 8+7", "uniformFK", dict(bits1=8, bits2=b - 8), {"HJ_FORCE_SAMPLED": "2"}),
        ]
    else:  # 16 - 18 radix bits: the sampled path (512-way passes under skew) against the exact passes
        variants = [
            ("zipf sampled (default bits)", "zipf1.0", {}, {}),
            ("zipf exact passes", "zipf1.0", dict(exact_only=True), {}),
            ("uniform FK plain", "uniformFK", {}, {"HJ_FORCE_SAMPLED": "8"}),
            ("uniform FK sampled", "uniformFK", {}, {"HJ_FORCE_SAMPLED": "2"}),
        ]
    os.environ["HJ_REPLAN"] = "1"
    hj.reload_knobs()
    acc = {v[0]: [] for v in variants}
    for r in range(rounds):
        for label, dname, cfg, env in variants:
            for kn in KNOBS:
                os.environ.pop(kn, None)
            os.environ.update(env)
            hj.reload_knobs()                        # (the library reads its knobs once, in hj_create)
            hj.configure(**cfg)
            hj.bind_device(pkg.REL_R, Rk, Rp)
            hj.bind_device(pkg.REL_S, data[dname], Sp)
            for _ in range(2):                       # first contact (overflow -> sample -> tables) and one steady step
                assert hj.join()[0] == expect[dname], label
            layout = hj.partition_layout(pkg.REL_S)
            hj.enable_timings(1)
            hj.timings_reset()
            for _ in range(3):
                assert hj.join()[0] == expect[dname]
            kt = hj.timings()
            hj.enable_timings(0)
            # S's launches: the longer of the two per kernel name when R and S share one (plain passes)
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                assert hj.join()[0] == expect[dname]
            torch.cuda.synchronize()
            step_ms = (time.perf_counter() - t0) / 3 * 1e3
            row = {k: round(v["total_ms"] / 3, 4) for k, v in kt.items() if v["launches"] and (k.startswith("k_part") or k.startswith("k_join") or k.startswith("k_hist") or k.startswith("k_scatter"))}
            acc[label].append({"layout_S": layout, "step_ms": round(step_ms, 3), "ms_per_step_by_kernel": row})
    for label, _, cfg, env in variants:
        print(json.dumps({"sizes": [lr, ls], "variant": label, "configure": cfg, "env": env, "rounds": acc[label]}))


if __name__ == "__main__":
    main()
