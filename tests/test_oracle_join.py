"""Oracle join semantics: closed-form counts from generator construction, two
independent restatements agreeing (sort-merge vs joinCpu chained hash vs OpenMP radix join), and
brute force on tiny inputs."""
import os

import numpy as np
import pytest

from oracle import pyoracle as o


def _load(golden_dir, name):
    return np.fromfile(os.path.join(golden_dir, name), dtype=np.int32)


def _brute(R, Pr, S, Ps):
    out = []
    for i, r in enumerate(R):
        for j, s in enumerate(S):
            if r == s:
                out.append((int(np.uint32(r)), int(np.uint32(Pr[i])), int(np.uint32(Ps[j]))))
    return sorted(out)


def test_config1_shape_unique_self_join(golden_dir):
    # bench -R N -S N: S is re-read from R's cache file → S ≡ R → matches = N (main.cu:135,143)
    R = _load(golden_dir, "unique_4096.bin")
    m, agg, _ = o.join_count(R, None, R, None)
    assert m == 4096 and agg == 4096
    assert o.joinCpu(R, R)[0] == 4096
    assert o.radix_join_omp(R, None, R, None, 3, 2, 2)[0] == 4096


def test_unique_fk_closed_form(golden_dir):
    # M > N: S = 0,1..N,1..N,... → matches = M - #{S == N} = M - floor((M-1)/N)
    R = _load(golden_dir, "unique_4096.bin")
    S = _load(golden_dir, "unique_fk10000_max4096.bin")
    expect = 10000 - (10000 - 1) // 4096
    assert int((S == 4096).sum()) == (10000 - 1) // 4096
    assert o.join_count(R, None, S, None)[0] == expect
    assert o.joinCpu(R, S)[0] == expect
    assert o.radix_join_omp(R, None, S, None, 4, 3, 3)[0] == expect


def test_zipf_closed_form(golden_dir):
    R = _load(golden_dir, "unique_4096.bin")
    S = _load(golden_dir, "zipf_S20000_a4096_t1.0_seed42.bin")
    expect = len(S) - int((S == 4096).sum())
    assert o.join_count(R, None, S, None)[0] == expect == o.joinCpu(R, S)[0]


def test_nonunique_histogram_formula(golden_dir):
    R = _load(golden_dir, "nonuniq_R6000_seed7.bin")
    S = _load(golden_dir, "nonuniq_S9000_seed8.bin")
    cr = np.bincount(R, minlength=3000).astype(np.int64)
    cs = np.bincount(S, minlength=3000).astype(np.int64)
    expect = int((cr * cs).sum())
    m, agg, chk = o.join_count(R, None, S, None)
    assert m == expect == agg
    assert o.joinCpu(R, S)[0] == expect
    assert o.radix_join_omp(R, None, S, None, 2, 2, 4) == (expect, expect)
    k, pr, ps = o.join_materialize(R, None, S, None)
    assert len(k) == expect and o.triples_checksum(k, pr, ps) == chk


def test_full_range_fk(golden_dir):
    R = _load(golden_dir, "pk_R3000_seed11.bin")
    S = _load(golden_dir, "fk_S7000_pk_R3000_seed11.bin")
    # FK = PK repeated → every FK tuple matches >= 1 PK tuple
    m = o.join_count(R, None, S, None)[0]
    assert m >= 7000 and m == o.joinCpu(R, S)[0]


def test_materialize_vs_bruteforce():
    rng = np.random.default_rng(3)
    R = rng.integers(-20, 20, 150).astype(np.int32)
    S = rng.integers(-20, 20, 170).astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, 150).astype(np.int32)
    Ps = np.arange(170, dtype=np.int32)
    k, pr, ps = o.join_materialize(R, Pr, S, Ps)
    got = list(zip(k.view(np.uint32).tolist(), pr.view(np.uint32).tolist(), ps.view(np.uint32).tolist()))
    assert got == _brute(R, Pr, S, Ps)
    m, agg, chk = o.join_count(R, Pr, S, Ps)
    assert m == len(got)
    expect_agg = sum(int(np.int32(a)) * int(np.int32(b)) for _, a, b in
                     [(x, np.uint32(y).astype(np.int32), np.uint32(z).astype(np.int32)) for x, y, z in got]) % 2**64
    assert agg == expect_agg
    assert chk == o.triples_checksum(k, pr, ps)
    # order independence of the checksum
    perm = rng.permutation(len(k))
    assert o.triples_checksum(k[perm], pr[perm], ps[perm]) == chk


def test_edge_cases():
    e = np.empty(0, np.int32)
    one = np.array([5], np.int32)
    assert o.join_count(e, None, e, None)[0] == 0
    assert o.join_count(one, None, e, None)[0] == 0
    assert o.join_count(e, None, one, None)[0] == 0
    assert o.join_count(one, None, one, None)[0] == 1
    dup = np.full(100, -7, np.int32)  # all-duplicate keys: 100*100 pairs
    assert o.join_count(dup, None, dup, None)[0] == 10000
    assert o.radix_join_omp(dup, None, dup, None, 3, 3, 2)[0] == 10000
    ext = np.array([-2**31, 2**31 - 1, 0, -1], np.int32)
    assert o.join_count(ext, None, ext[::-1].copy(), None)[0] == 4


def test_partition_function():
    rng = np.random.default_rng(5)
    k = rng.integers(-2**31, 2**31 - 1, 10000).astype(np.int32)
    p = np.arange(10000, dtype=np.int32)
    for shift, bits in [(0, 4), (5, 8), (23, 9), (0, 1)]:
        ok, op, off = o.radix_partition(k, p, shift, bits)
        assert off[0] == 0 and off[-1] == 10000
        d = (ok.view(np.uint32) >> shift) & ((1 << bits) - 1)
        assert np.all(np.diff(d.astype(np.int64)) >= 0)            # partition ids ascend
        assert np.array_equal(k[op], ok)                            # payload travels with its key
        for q in (0, (1 << bits) - 1):
            assert np.all(d[int(off[q]):int(off[q + 1])] == q)
        # stable: payload (= original index) ascends inside each partition
        assert all(np.all(np.diff(op[int(off[q]):int(off[q + 1])]) > 0) for q in range(1 << bits))


def test_config1_full_size():
    """BASELINE config 1: 2^20 ⋈ 2^20 unique keys, host only."""
    n = 1 << 20
    R = o.random_unique_gen(n, n, 12345)
    assert np.array_equal(np.sort(R), np.arange(n, dtype=np.int32))
    assert o.join_count(R, None, R, None, checksum=False)[0] == n
    assert o.joinCpu(R, R, threads=o.max_threads())[0] == n
    assert o.radix_join_omp(R, None, R, None, 5, 5, o.max_threads())[0] == n


# ---- the join half of the oracle, pinned to the REFERENCE's own CPU join ---------------------------
def test_oracle_matches_reference_joinCpu_answers(join_answers, golden_dir):
    """tests/golden/join_answers.json holds what the reference's joinCpu + h_hashMurmur
    (hash_join_clustered_probe.cu:2013-2059, compiled unmodified from where it lies: oracle/_ref/refjoin,
    script tests/golden/make_golden.py) printed for these golden pairs.  Every oracle restatement of the
    join must reproduce its match count s and its key sum g."""
    assert len(join_answers) >= 15
    for a in join_answers:
        R, S = _load(golden_dir, a["R"]), _load(golden_dir, a["S"])
        assert a["build"] == len(R) and a["c"] == len(R) + len(S)
        m, agg, chk = o.join_count(R, None, S, None)
        assert m == a["s"], (a, m)                                     # sort-merge restatement
        assert o.joinCpu(R, S) == (a["s"], a["g"]), a                  # restated joinCpu: same s AND g
        assert o.joinCpu(R, S, threads=3) == (a["s"], a["g"]), a
        assert o.radix_join_omp(R, None, S, None, 4, 3, 2)[0] == a["s"], a
        # g through the payload path: payR = 1, payS = key → sum payR*payS = sum of matching S keys
        assert o.join_count(R, None, S, S)[1] % 2**32 == a["g"], a
        k, pr, ps = o.join_materialize(R, None, S, None)
        assert len(k) == a["s"] and int(k.astype(np.int64).sum()) % 2**32 == a["g"], a


REFJOIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "refjoin")


@pytest.mark.skipif(not os.path.exists(REFJOIN), reason="oracle/_ref/refjoin is only built where /root/reference exists")
def test_oracle_vs_reference_joinCpu_live(tmp_path):
    """In the build container: the reference's joinCpu run live on fresh seeded inputs (duplicates on both
    sides, negative keys, empty sides) against the oracle."""
    import subprocess
    rng = np.random.default_rng(2024)
    cases = [(rng.integers(-500, 500, 3000), rng.integers(-500, 500, 5000)),
             (rng.integers(0, 2**31 - 1, 4000), rng.integers(0, 2**31 - 1, 4000)),
             (rng.permutation(20000), rng.integers(0, 25000, 50000)),
             (np.full(300, 7), np.full(200, 7)),
             (np.arange(10), np.empty(0, np.int64)),
             (np.array([-2**31, 2**31 - 1, 0, -1]), np.array([-1, -2**31, 5, 2**31 - 1, -1]))]
    for i, (R, S) in enumerate(cases):
        R, S = R.astype(np.int32), S.astype(np.int32)
        fr, fs = tmp_path / ("r%d.bin" % i), tmp_path / ("s%d.bin" % i)
        R.tofile(fr)
        S.tofile(fs)
        out = subprocess.run([REFJOIN, str(fr), str(fs)], stdout=subprocess.PIPE, check=True,
                             env=dict(os.environ, OMP_NUM_THREADS="1")).stdout.decode().splitlines()
        s, c, g = (int(x) for x in [l for l in out if l.startswith("===")][1][3:].split())
        assert c == len(R) + len(S)
        assert o.join_count(R, None, S, None)[0] == s
        assert o.joinCpu(R, S) == (s, g)
        assert o.join_count(R, None, S, S)[1] % 2**32 == g
