#!/usr/bin/env python3
"""tools/fuzz_dist.py N [first_seed] — differential runs of the multi-GPU join (hj_dist, in-process group: one host thread per rank, all
ranks on cuda:0, device-copy transport) against the oracle under random shapes: world size 2...8, relation sizes, key distribution
(unique PK-FK / duplicates on both sides / one key holding 40 % of S), uneven and empty local cuts, 1-5 slices, one or two probe-side
groups, radix bits, count-only and MATERIALISING (the union of the ranks' shares == the oracle's sorted (key, payR, payS) multiset).
Every case joins twice on the same group (buffers and learned state reused).  Uses the helpers of tests/test_dist_c.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_dist_c as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
fails = 0
paths = {}
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(1000 + seed)
    world = int(rng.choice([2, 3, 4, 5, 8]))
    kind = ["unique", "dups", "skew"][int(rng.integers(0, 3))]
    mat = bool(rng.integers(0, 2))
    if os.environ.get("FUZZ_WORLD"):      # pin the world size / the kind of join (experiments)
        world = int(os.environ["FUZZ_WORLD"])
    if os.environ.get("FUZZ_MAT"):
        mat = os.environ["FUZZ_MAT"] == "1"
    big = 150_000 if mat else 400_000
    nR = int(rng.integers(2_000, big))
    nS = int(rng.integers(2_000, big * 2))
    if kind == "dups" and mat:                # ~ nR * nS / 10^4 matches: keep the oracle's sorted multiset small
        nR, nS = min(nR, 40_000), min(nS, 60_000)
    R, S = T._inputs(nR, nS, seed, kind)
    cuts = np.sort(rng.random(world - 1)).tolist() + [1.0] if rng.random() < 0.6 else None
    if cuts and rng.random() < 0.3:
        cuts[0] = 0.0                          # rank 0 holds nothing
    bits = [None, dict(bits1=5, bits2=4), dict(bits1=4, bits2=3), dict(bits1=6, bits2=5)][int(rng.integers(0, 4))]
    dist_cfg = dict(slices=int(rng.integers(1, 6)))
    if rng.random() < 0.3:
        dist_cfg["single_group"] = True
    if os.environ.get("FUZZ_VERBOSE"):
        print("case seed %d world %d kind %s mat %s nR %d nS %d cuts %s bits %s cfg %s" % (seed, world, kind, mat, nR, nS, cuts, bits, dist_cfg), flush=True)
    try:
        if mat:
            stats = T._run_materialize([0] * world, R, S, cuts=cuts, dist_cfg=dist_cfg, ctx_cfg=bits)
        else:
            stats, _ = T._run([0] * world, R, S, cuts=cuts, dist_cfg=dist_cfg, ctx_cfg=bits)
        p = stats[0]["path"]
        paths[p] = paths.get(p, 0) + 1
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("FAIL seed %d world %d kind %s mat %s nR %d nS %d cuts %s bits %s cfg %s: %r" % (seed, world, kind, mat, nR, nS, cuts, bits, dist_cfg, e), flush=True)
print("dist fuzz: %d cases (seeds %d..%d), %d failures, paths %s, %d s" % (n_cases, first, first + n_cases - 1, fails, paths, time.time() - t0))
sys.exit(1 if fails else 0)
