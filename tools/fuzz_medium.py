"""Medium-size randomized differential run (GPU box): 2^18..2^23-tuple relations, default configuration (histogram-free
passes with fallback) and exact_only, several key distributions, the build side left to the library / forced to R / forced to S
(round 4: a skewed or larger designated build side runs on general work items), count + aggregate + partition digests against
the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hjtest import pkg
from oracle import pyoracle as o
P = pkg()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
# "hibits": every case with 16-18 forced radix bits (9+7, 8+8, 9+8, 9+9 in turn): tag tables, the sliced sampling pass of the sampled path
# at 16 / 17 bits, exact passes for a skewed relation at 18 — partitions of a few tuples, most of them empty
HIBITS = len(sys.argv) > 2 and sys.argv[2] == "hibits"
HB = [dict(bits1=9, bits2=7), dict(bits1=8, bits2=8), dict(bits1=9, bits2=8), dict(bits1=9, bits2=9)]
t0 = time.time(); bad = 0
FIRST = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # first seed (a soak run continues where an earlier one stopped)
for seed in range(FIRST, FIRST + n_cases):
    rng = np.random.default_rng(7000 + seed)
    nR = int(rng.integers(1 << 18, 1 << 22)); nS = int(rng.integers(1 << 18, 1 << 23))
    kind = seed % 6
    bside = (seed // 6) % 3   # 0 = the smaller relation builds, 1 = R, 2 = S
    if kind == 0:
        R = rng.permutation(nR); S = rng.integers(0, nR + 100, nS)
    elif kind == 1:   # zipf-ish probe side
        R = rng.permutation(nR); S = np.minimum((rng.pareto(1.0, nS) * 5).astype(np.int64), nR)
    elif kind == 2:   # a few heavy keys on both sides
        R = rng.integers(0, 1 << 20, nR); S = np.where(rng.random(nS) < 0.4, rng.integers(0, 4, nS) * 977, rng.integers(0, 1 << 20, nS))
        R[: nR // 1000] = 977
    elif kind == 3:   # keys with constant low bits (pass-2 digit degenerate)
        R = rng.permutation(nR) * 1024 % (1 << 31); S = rng.integers(0, nR, nS) * 1024 % (1 << 31)
    elif kind == 4:   # full int32 range incl. negatives
        R = rng.integers(-2**31, 2**31 - 1, nR); S = np.concatenate([rng.choice(R, nS // 2), rng.integers(-2**31, 2**31 - 1, nS - nS // 2)])
    else:             # sorted inputs
        R = np.sort(rng.integers(0, 1 << 22, nR)); S = np.sort(rng.integers(0, 1 << 22, nS))
    R, S = R.astype(np.int32), S.astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, len(R)).astype(np.int32); Ps = rng.integers(-2**31, 2**31 - 1, len(S)).astype(np.int32)
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=True)
    lays = []
    for exact in (False, True):
        with P.HashJoin(0) as hj:
            fb = HB[seed % 4] if HIBITS else {}
            hj.configure(exact_only=exact, build_side=bside, **fb)
            hj.load_host(P.REL_R, R, Pr); hj.load_host(P.REL_S, S, Ps)
            got = hj.join()
            c = hj.config(); bits = c["bits1"] + c["bits2"]
            okc = got == (em, eagg)
            lay = (hj.partition_layout(P.REL_R), hj.partition_layout(P.REL_S))
            for rel, (kk, pp) in ((P.REL_R, (R, Pr)), (P.REL_S, (S, Ps))):
                badp, dg = hj.verify_partitions(rel, with_digests=True)
                a, b, off = o.radix_partition(kk, pp, 0, bits)
                okc = okc and badp == 0 and np.array_equal(dg, o.partition_digest(a, b, off))
            got2 = hj.join()   # second run: what the first learned (skew: sampled capacities / exact passes) is in effect
            lay2 = (hj.partition_layout(P.REL_R), hj.partition_layout(P.REL_S))
            lays.append((lay, lay2))
            okc = okc and got2 == (em, eagg)
            if em <= 60_000_000:   # materialise in one probe on fresh partitions, order-independent digest against the oracle's
                hj.partition(P.REL_R); hj.partition(P.REL_S)
                k, pr, ps = hj.join_materialize(cap=em)
                okc = okc and len(k) == em and o.triples_checksum(k, pr, ps) == echk
                # round 6: partition both + one probe in ONE call (a skewed probe side's hot keys are written by pass 1), then a count on what it left
                k, pr, ps = hj.join_and_materialize(cap=em)
                okc = okc and len(k) == em and o.triples_checksum(k, pr, ps) == echk and hj.join_count() == (em, eagg)
            hj.configure(exact_only=exact, build_side=bside, graph=True, **fb)
            for _ in range(3):
                okc = okc and hj.join() == (em, eagg)
        if not okc:
            bad += 1; print("FAIL seed", seed, "kind", kind, "exact", exact, got, (em, eagg), lay, flush=True)
    print("seed %d kind %d build_side %d nR %d nS %d matches %d layouts (first join -> second join) default %s -> %s, exact_only %s -> %s"
          % (seed, kind, bside, len(R), len(S), em, lays[0][0], lays[0][1], lays[1][0], lays[1][1]), flush=True)
print("medium fuzz%s: %d cases, %d failures, %.0f s" % (" (16-18 forced radix bits)" if HIBITS else "", n_cases, bad, time.time() - t0))
