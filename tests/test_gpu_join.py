"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs.
Bit-exact bar: match count, sum(payR*payS) mod 2^64, the multiset of (key,payR,payS) output tuples,
and the (key,payload) multiset of every radix partition."""
import os

import numpy as np
import pytest

from hjtest import pkg, sorted_triples
from oracle import pyoracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return pkg()


def _load(golden_dir, name):
    return np.fromfile(os.path.join(golden_dir, name), dtype=np.int32)


def _check_join(P, R, Pr, S, Ps, cfg=None, materialize=True):
    """Every configuration is checked twice: with the histogram-free passes enabled (the default; they fall back to
    the exact passes on the device when a slot overflows) and with the exact passes only."""
    out = None
    for exact_only in (False, True):
        out = _check_join_1(P, R, Pr, S, Ps, dict(cfg or {}, exact_only=exact_only), materialize)
    return out


def _check_join_1(P, R, Pr, S, Ps, cfg=None, materialize=True):
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=materialize)
    with P.HashJoin(0) as hj:
        if cfg:
            hj.configure(**cfg)
        hj.load_host(P.REL_R, R, Pr)
        hj.load_host(P.REL_S, S, Ps)
        m, agg = hj.join()
        assert (m, agg) == (em, eagg), ("count/agg", (m, agg), (em, eagg), hj.config())
        if materialize:
            exp = sorted_triples(*o.join_materialize(R, Pr, S, Ps)) if em <= 2_000_000 else None

            def check(k, pr, ps, what):
                assert len(k) == em, what
                assert o.triples_checksum(k, pr, ps) == echk, what
                if exp is not None:
                    for a, b in zip(sorted_triples(k, pr, ps), exp):
                        assert np.array_equal(a, b), what

            # materialisation in ONE probe (the default): behind a count (item list reused) ...
            check(*hj.join_materialize(), "one probe, after a count")
            # ... on fresh partitions with no count before it (the timed shape: partition both, materialise) ...
            hj.partition_both()
            check(*hj.join_materialize(cap=em), "one probe, no count")
        # the step replayed from a captured hipGraph (hj_config.graph): eager call, capturing call, replays
        hj.configure(**dict(cfg or {}, graph=True))
        for _ in range(4):
            assert hj.join() == (em, eagg), ("graph", hj.config())
        return hj.config()


def _check_partitions(P, keys, pays, cfg):
    for exact_only in (False, True):
        _check_partitions_1(P, keys, pays, dict(cfg, exact_only=exact_only))


def _check_partitions_1(P, keys, pays, cfg):
    with P.HashJoin(0) as hj:
        hj.configure(**cfg)
        hj.load_host(P.REL_R, keys, pays)
        hj.load_host(P.REL_S, keys[:1], pays[:1])
        hj.partition(P.REL_R)
        c = hj.config()
        bits = c["bits1"] + c["bits2"]
        gk, gp, goff = hj.partitions(P.REL_R, len(keys))
        bad, dg = hj.verify_partitions(P.REL_R, with_digests=True)
    assert bad == 0
    ok, op, ooff = o.radix_partition(keys, pays, 0, bits)
    assert np.array_equal(goff, ooff)                                   # identical partition boundaries
    assert np.array_equal(o.partition_digest(gk, gp, goff), o.partition_digest(ok, op, ooff))
    assert np.array_equal(dg, o.partition_digest(ok, op, ooff))         # device-side digest kernel agrees
    d = gk.view(np.uint32) & ((1 << bits) - 1)
    assert np.all(np.diff(d.astype(np.int64)) >= 0)                     # every tuple sits in its partition


# ---- golden fixtures (outputs of the reference generator) --------------------------------------------
@pytest.mark.parametrize("cfg", [None, dict(bits1=3), dict(bits1=4, bits2=3), dict(force_bits=True)])
def test_golden_unique_self_join(P, golden_dir, cfg):
    R = _load(golden_dir, "unique_4096.bin")        # bench -R 4096 -S 4096: S ≡ R → 4096 matches
    Pr = np.arange(len(R), dtype=np.int32)
    _check_join(P, R, Pr, R, Pr, cfg)


@pytest.mark.parametrize("cfg", [None, dict(bits1=5, bits2=4)])
def test_golden_unique_fk(P, golden_dir, cfg):
    R = _load(golden_dir, "unique_4096.bin")
    S = _load(golden_dir, "unique_fk10000_max4096.bin")
    _check_join(P, R, np.arange(len(R), dtype=np.int32), S, np.arange(len(S), dtype=np.int32), cfg)
    with P.HashJoin(0) as hj:
        hj.load_host(P.REL_R, R)
        hj.load_host(P.REL_S, S)
        assert hj.join() == (10000 - (10000 - 1) // 4096,) * 2   # closed form, payloads = 1


@pytest.mark.parametrize("cfg", [None, dict(bits1=2, bits2=2), dict(bits1=6, bits2=5, lds_capacity=64, lds_heads=16, probe_chunk=100)])
def test_golden_zipf_nonunique_fullrange(P, golden_dir, cfg):
    R = _load(golden_dir, "unique_4096.bin")
    Z = _load(golden_dir, "zipf_S20000_a4096_t1.0_seed42.bin")
    _check_join(P, R, np.arange(len(R), dtype=np.int32), Z, np.arange(len(Z), dtype=np.int32), cfg)
    A = _load(golden_dir, "nonuniq_R6000_seed7.bin")
    B = _load(golden_dir, "nonuniq_S9000_seed8.bin")
    _check_join(P, A, np.arange(len(A), dtype=np.int32), B, -np.arange(len(B), dtype=np.int32), cfg)
    K = _load(golden_dir, "pk_R3000_seed11.bin")
    F = _load(golden_dir, "fk_S7000_pk_R3000_seed11.bin")
    _check_join(P, K, np.arange(len(K), dtype=np.int32), F, np.arange(len(F), dtype=np.int32), cfg)


@pytest.mark.parametrize("cfg", [None, dict(bits1=4, bits2=3), dict(bits1=9, bits2=7), dict(bits1=6), dict(force_bits=True)])
def test_reference_joinCpu_answers(P, join_answers, golden_dir, cfg):
    """The HIP path against the REFERENCE's own CPU join: tests/golden/join_answers.json holds what joinCpu +
    h_hashMurmur (hash_join_clustered_probe.cu:2013-2059), compiled unmodified from where they lie, printed for
    these golden pairs — s = matching pairs, g = sum of matching S keys mod 2^32."""
    for a in join_answers:
        R, S = _load(golden_dir, a["R"]), _load(golden_dir, a["S"])
        with P.HashJoin(0) as hj:
            if cfg:
                hj.configure(**cfg)
            hj.load_host(P.REL_R, R)                       # payloads = 1 (hjcp.cu:1994-1999)
            hj.load_host(P.REL_S, S, S)                    # payS = key → sum payR*payS = the reference's g
            m, agg = hj.join()
            assert m == a["s"] and agg % 2**32 == a["g"], (a, m, agg, cfg)
            k, pr, ps = hj.join_materialize()
            assert len(k) == a["s"] and int(k.astype(np.int64).sum()) % 2**32 == a["g"], (a, cfg)
            assert np.array_equal(k, ps) and np.all(pr == 1)
            if cfg is None:
                assert hj.join_nonpartitioned(1)[0] == a["s"]
    R, S = _load(golden_dir, "unique_4096.bin"), _load(golden_dir, "unique_fk10000_max4096.bin")
    r = P.hashJoinClusteredProbe(R, S)                     # the reference's entry point, same inputs
    assert r["matches"] == [a for a in join_answers if a["S"] == "unique_fk10000_max4096.bin"][0]["s"]


def test_graph_replay_follows_the_data(P):
    """hj_config.graph: the captured step is tied to the binding, not to the data under it.  New data in the same columns is
    joined correctly by a replay; data that turns skewed raises the overflow flag inside the replay, the graph is dropped, the
    relation redone with the exact passes, and the next calls capture again."""
    rng = np.random.default_rng(12)
    n = 1 << 18
    with P.HashJoin(0) as hj:          # its own stream: HIP's legacy default stream cannot be captured
        hj.configure(bits1=5, bits2=4, graph=True)
        R = rng.permutation(n).astype(np.int32)
        S = rng.integers(0, n, 3 * n).astype(np.int32)
        hj.load_host(P.REL_R, R)
        hj.load_host(P.REL_S, S)
        for _ in range(3):
            assert hj.join() == (3 * n, 3 * n)
        S2 = rng.integers(0, n // 2, 3 * n).astype(np.int32)        # same size, other values: same device columns
        hj.load_host(P.REL_S, S2)
        for _ in range(3):
            assert hj.join()[0] == 3 * n
        S3 = S2.copy()
        S3[: 2 * n] = 7                                               # now one key holds two thirds of S
        hj.load_host(P.REL_S, S3)
        for _ in range(4):
            assert hj.join()[0] == 3 * n
        assert hj.partition_layout(P.REL_S) in ("exact", "sampled")
        # timings on: the graph is dropped, the eager path answers
        hj.enable_timings(1)
        assert hj.join()[0] == 3 * n and hj.timings()["k_join_count"]["launches"] == 1


# ---- edge cases ----------------------------------------------------------------------------------------
def test_empty_and_tiny(P):
    e = np.empty(0, np.int32)
    one = np.array([7], np.int32)
    for R, S in [(e, e), (one, e), (e, one), (one, one), (one, np.array([8], np.int32))]:
        for cfg in (None, dict(bits1=3, bits2=2)):
            _check_join(P, R, np.arange(len(R), dtype=np.int32), S, np.arange(len(S), dtype=np.int32), cfg)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 63, 64, 65, 511, 512, 513, 4095, 4097, 8191, 8192, 8193, 20001])
def test_ragged_sizes(P, n):
    rng = np.random.default_rng(n)
    R = rng.permutation(n).astype(np.int32)
    S = rng.integers(0, n + 3, 2 * n + 1).astype(np.int32)
    _check_join(P, R, np.arange(n, dtype=np.int32), S, np.arange(len(S), dtype=np.int32), dict(bits1=3, bits2=2))


def test_extreme_keys_and_payload_overflow(P):
    rng = np.random.default_rng(9)
    keys = np.array([-2**31, 2**31 - 1, 0, -1, 1, 0x7FFF0000, -0x7FFF0000], np.int32)
    R = np.concatenate([keys, rng.integers(-2**31, 2**31 - 1, 5000).astype(np.int32)])
    S = np.concatenate([keys[::-1], R[::3], rng.integers(-2**31, 2**31 - 1, 5000).astype(np.int32)])
    Pr = rng.integers(-2**31, 2**31 - 1, len(R)).astype(np.int32)   # payR*payS overflows int32: agg is mod 2^64
    Ps = rng.integers(-2**31, 2**31 - 1, len(S)).astype(np.int32)
    for cfg in (None, dict(bits1=9, bits2=9), dict(bits1=8, bits2=8), dict(bits1=5)):
        _check_join(P, R, Pr, S, Ps, cfg)


def test_all_duplicates_overflow_lds_table(P):
    """One key everywhere: a single partition far larger than the LDS table (the reference's
    'S partition doesn't fit' branch, jp.cu:929-1003) and a quadratic output."""
    R = np.full(3000, 42, np.int32)
    S = np.full(1500, 42, np.int32)
    cfg = dict(bits1=4, bits2=4, lds_capacity=256, lds_heads=64, probe_chunk=512)
    _check_join(P, R, np.arange(3000, dtype=np.int32), S, np.arange(1500, dtype=np.int32), cfg)
    with P.HashJoin(0) as hj:
        hj.configure(**cfg)
        hj.load_host(P.REL_R, R)
        hj.load_host(P.REL_S, S)
        assert hj.join() == (4_500_000, 4_500_000)


def test_skew_heavy_hitters(P):
    rng = np.random.default_rng(4)
    n = 1 << 14
    R = rng.permutation(n).astype(np.int32)
    hot = rng.integers(0, n, 8)
    S = np.where(rng.random(1 << 18) < 0.6, hot[rng.integers(0, 8, 1 << 18)], rng.integers(0, n, 1 << 18)).astype(np.int32)
    for cfg in (None, dict(bits1=5, bits2=5, probe_chunk=1000), dict(build_side=2, lds_capacity=2048, lds_heads=512)):
        _check_join(P, R, np.arange(n, dtype=np.int32), S, np.arange(len(S), dtype=np.int32), cfg)


def test_tag16_vs_full_key_paths(P):
    # 16+ radix bits → 16-bit tags (the reference's compression, jp.cu:1029, exact only then: D2); fewer → full keys
    rng = np.random.default_rng(11)
    R = rng.integers(-2**31, 2**31 - 1, 1 << 16).astype(np.int32)
    S = np.concatenate([R[: 1 << 15], rng.integers(-2**31, 2**31 - 1, 1 << 15).astype(np.int32)])
    # keys that agree in the low bits and in the hash slot bits but differ above must not match
    R[:4] = [0x00010000, 0x10010000, 0x20010000, 0x30010000]
    S[:2] = [0x40010000, 0x10010000]
    # keys that differ ONLY in the one or two top bits a 16-bit tag could not hold at 15 / 14 radix bits: same partition, same
    # low 16 remaining bits — must not match (full keys are compared there)
    R[4:8] = [0x00012345, -0x7FFEDCBB, 0x40012345, -0x3FFEDCBB]       # 0x00012345 | top bits 00, 10, 01, 11
    S[2:5] = [-0x7FFEDCBB, 0x40012345, 0x00012345 | 0x20000000]
    R[8:10] = [-2**31, 2**31 - 1]
    S[5:7] = [2**31 - 1, -2**31]
    for cfg in (dict(bits1=8, bits2=8), dict(bits1=9, bits2=9), dict(bits1=8, bits2=7), dict(bits1=7, bits2=7), dict(bits1=9, bits2=6),
                dict(bits1=8, bits2=7, lds_heads=16, lds_capacity=256), dict(bits1=7, bits2=7, lds_heads=4, lds_capacity=100),
                dict(bits1=7, bits2=6), dict(bits1=2)):
        _check_join(P, R, np.arange(len(R), dtype=np.int32), S, np.arange(len(S), dtype=np.int32), cfg)


def test_long_streams_through_one_table_with_and_without_repeated_keys(P):
    """One table, many sub-chunks of the streamed side (65536-tuple work items against a <= 4608-entry table): the materialising kernel
    asks whether the table holds a key twice and, if not, ends every sub-chunk after its first round.  Unique build keys (the shortcut),
    one repeated key, every key repeated, and a build side of several table chunks — all against the oracle's multiset."""
    rng = np.random.default_rng(77)
    nS = 300_001
    for nR, dup in ((3000, 0), (3000, 1), (3000, 2), (4608, 0), (12000, 1)):
        R = rng.permutation(50_000)[:nR].astype(np.int32)
        if dup == 1:
            R[17] = R[1234]                                   # exactly one key twice
        elif dup == 2:
            R[nR // 2:] = R[: nR - nR // 2]                    # every key twice
        S = R[rng.integers(0, nR, nS)].astype(np.int32)
        S[::7] = -5                                            # tuples without a partner
        for cfg in (dict(force_bits=True), dict(bits1=1), dict(bits1=2, bits2=1, probe_chunk=20_000)):
            _check_join_1(P, R, np.arange(nR, dtype=np.int32), S, np.arange(nS, dtype=np.int32) * 3 + 1, cfg)


# ---- histogram-free passes: taken on uniform keys, abandoned (on the device) under skew -------------------------
def test_fast_path_layout_and_fallback(P):
    rng = np.random.default_rng(55)
    n = 1 << 18
    uni = rng.permutation(n).astype(np.int32)
    skew = np.where(rng.random(n) < 0.5, 12345, rng.integers(0, n, n)).astype(np.int32)      # one key holds half
    stride = (rng.permutation(n).astype(np.int64) * 512 % (1 << 31)).astype(np.int32)        # pass-2 digit always 0
    pay = np.arange(n, dtype=np.int32)
    for keys, expect in ((uni, "slotted"), (skew, "exact"), (stride, "exact")):
        for cfg in (dict(bits1=9, bits2=9), dict(bits1=9, bits2=7), dict(bits1=5, bits2=4), dict(bits1=7, bits2=2)):
            with P.HashJoin(0) as hj:
                hj.configure(**cfg)
                hj.load_host(P.REL_R, keys, pay)
                hj.load_host(P.REL_S, uni, pay)
                hj.partition(P.REL_R)
                hj.partition(P.REL_S)
                assert hj.partition_layout(P.REL_S) == "slotted"
                lay = hj.partition_layout(P.REL_R)
                if keys is uni or cfg["bits2"] >= 7:
                    assert lay == expect, (lay, expect, cfg)
                gk, gp, goff = hj.partitions(P.REL_R, n)
                ok, op, ooff = o.radix_partition(keys, pay, 0, cfg["bits1"] + cfg["bits2"])
                assert np.array_equal(goff, ooff)
                assert np.array_equal(o.partition_digest(gk, gp, goff), o.partition_digest(ok, op, ooff))
                assert hj.join_count() == o.join_count(keys, pay, uni, pay, checksum=False)[:2]
                hj.configure(exact_only=True, **cfg)
                hj.partition(P.REL_R)
                assert hj.partition_layout(P.REL_R) == "exact"
    # a single pass has no histogram-free form
    with P.HashJoin(0) as hj:
        hj.configure(bits1=6)
        hj.load_host(P.REL_R, uni, pay)
        hj.load_host(P.REL_S, uni, pay)
        hj.partition(P.REL_R)
        assert hj.partition_layout(P.REL_R) == "exact"


# ---- partition parity --------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [dict(bits1=1), dict(bits1=9), dict(bits1=4, bits2=4), dict(bits1=9, bits2=9), dict(bits1=7, bits2=2)])
def test_partition_parity(P, cfg):
    rng = np.random.default_rng(21)
    n = 300_001
    keys = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    keys[: n // 4] = rng.integers(0, 50, n // 4)            # a skewed quarter
    _check_partitions(P, keys, np.arange(n, dtype=np.int32), cfg)


def test_partition_repeatable_and_input_untouched(P):
    rng = np.random.default_rng(2)
    keys = rng.integers(0, 1 << 20, 100_000).astype(np.int32)
    pays = np.arange(100_000, dtype=np.int32)
    k0, p0 = keys.copy(), pays.copy()
    with P.HashJoin(0) as hj:
        hj.configure(bits1=6, bits2=5)
        hj.load_host(P.REL_R, keys, pays)
        hj.load_host(P.REL_S, keys, pays)
        first = hj.join()
        for _ in range(3):                                   # atomics decide in-partition order: results must not move
            assert hj.join() == first
    assert np.array_equal(keys, k0) and np.array_equal(pays, p0)


# ---- medium sizes: oracle still finishes in seconds ------------------------------------------------------
@pytest.mark.parametrize("logn", [20, 22])
def test_config1_and_up_unique(P, logn):
    n = 1 << logn
    P.generator.seed_generator(12345)
    R = P.generator.create_relation_unique(None, n, n)
    assert np.array_equal(R, o.random_unique_gen(n, n, 12345))
    rng = np.random.default_rng(logn)
    S = R[rng.permutation(n)]
    Pr = np.arange(n, dtype=np.int32)
    c = _check_join(P, R, Pr, S, Pr, None, materialize=(logn <= 20))
    assert c["bits1"] + c["bits2"] == logn - 12
    with P.HashJoin(0) as hj:
        hj.load_host(P.REL_R, R)
        hj.load_host(P.REL_S, S)
        assert hj.join() == (n, n)


def test_reference_entry_point(P, golden_dir, capfd):
    """hashJoinClusteredProbe(args*, timingInfo*) as main.cu calls it (src/main.cu:291)."""
    R = _load(golden_dir, "unique_4096.bin")
    S = _load(golden_dir, "unique_fk10000_max4096.bin")
    r = P.hashJoinClusteredProbe(R, S)
    out = capfd.readouterr().out
    assert r["status"] == 0 and r["return"] == 0
    assert r["matches"] == r["agg"] == r["materialized"] == 9998
    for line in ("With materialization", "Partition Throughput ", "Joins Throughput ", "Total Throughput  ",
                 "9998 results", "Without materialization", "Total Throughput "):
        assert line in out, line


def test_bench_cli_join(P, tmp_path):
    import subprocess
    r = subprocess.run([P._lib.BENCH_PATH, "-b", "7", "-a", "HJC", "-R", "65536", "-S", "200000", "--seed", "3"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "HJC : shareMemory = 30720\t#threads = 32" in r.stdout
    assert "%d results" % (200000 - (200000 - 1) // 65536) in r.stdout


def test_bench_cli_json_cpu_baseline_gpus(P, tmp_path):
    """The flags SURVEY §8(b) adds beside main.cu:445-457: --json, --cpu-baseline, --gpus N."""
    import json
    import subprocess
    import torch
    expect = 200000 - (200000 - 1) // 65536
    base = [P._lib.BENCH_PATH, "-b", "7", "-a", "HJC", "-R", "65536", "-S", "200000", "--seed", "3"]
    r = subprocess.run(base + ["--json", "--cpu-baseline"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "%d results" % expect in r.stdout and "CPU baseline (chained hash join" in r.stdout
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["status"] == 0 and line["matches"] == line["materialized"] == expect and line["gpus"] == 1
    assert line["cpu_baseline"]["matches"] == expect
    r = subprocess.run(base + ["--json", "--gpus", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["matches"] == expect and line["gpus"] == 2 and "Total Throughput (2 GPUs)" in r.stdout
    else:
        assert r.returncode != 0 and "GPU Error" in r.stderr     # one device only: fails loudly, no fallback
    # the transcript of the N > 1 mode on a one-GPU box: both ranks on GPU 0 over the device-copy transport (test mode, says so).
    # The materialising run comes first, as in the reference (hjcp.cu:937-940, 986-991); every GPU keeps its share of the output
    r = subprocess.run(base + ["--json", "--gpus", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HJ_BENCH_SHARE_GPU="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    out = r.stdout
    assert "TEST MODE" in out and out.index("With materialization") < out.index("Without materialization")
    assert out.count("Total Throughput (2 GPUs)") == 2 and out.count("%d results" % expect) == 2
    shares = [int(x) for x in out.split("Output (sharded, one share per GPU):")[1].split("tuples")[0].split()]
    assert len(shares) == 2 and sum(shares) == expect and min(shares) > 0
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["matches"] == line["materialized"] == expect and line["gpus"] == 2


def test_two_contexts_on_two_host_threads(P):
    """One context per host thread (SURVEY §8(b) threading): two threads, two contexts on the same GPU, different LDS
    table shapes and layouts, joins and the reference entry point running concurrently."""
    import threading
    rng = np.random.default_rng(101)
    data = []
    for t in range(2):
        R = rng.integers(0, 40_000, 50_000 + 7 * t).astype(np.int32)
        S = rng.integers(0, 40_000, 150_000 + 11 * t).astype(np.int32)
        data.append((R, S, o.join_count(R, None, S, R if False else None, checksum=False)[:2]))
    errs = []

    def work(t):
        try:
            R, S, exp = data[t]
            cfg = [dict(bits1=5, bits2=4, lds_capacity=512, lds_heads=128), dict(lds_capacity=6000, lds_heads=4096, exact_only=True)][t]
            with P.HashJoin(0) as hj:
                hj.configure(**cfg)
                hj.load_host(P.REL_R, R)
                hj.load_host(P.REL_S, S)
                for _ in range(5):
                    assert hj.join() == exp
                    k, pr, ps = hj.join_materialize()
                    assert len(k) == exp[0]
            r = P.hashJoinClusteredProbe(R, S)
            assert r["status"] == 0 and r["matches"] == exp[0]
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs


# ---- BASELINE sizes through size-independent properties --------------------------------------------------
def test_config2_single_pass_as_stated(P):
    """BASELINE config 2 as stated: 2^27 x 2^27, ONE radix pass of 9 bits (2^18-tuple partitions, far beyond the
    LDS table: every partition goes through the chunked-build path).  Count must hold; the time is recorded
    (DESIGN.md explains why 9+7 two-pass is the default at this size)."""
    import time
    import torch
    n = 1 << 27
    dev = torch.device("cuda:0")
    Rk, Sk, Rp, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    with P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream) as hj:
        hj.gen_unique(Rk, n, 0, n, 1)
        hj.gen_unique(Sk, n, 0, n, 2)
        hj.fill_payload(Rp, n, "ones")
        hj.fill_payload(Sp, n, "ones")
        hj.bind_device(P.REL_R, Rk, Rp)
        hj.bind_device(P.REL_S, Sk, Sp)
        res = {}
        for name, cfg in (("9+0 single pass", dict(bits1=9, bits2=0)), ("default", dict())):
            hj.configure(**cfg)
            assert hj.join() == (n, n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert hj.join() == (n, n)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) * 1e3
        print("config 2 (2^27 x 2^27): " + ", ".join("%s %.2f ms" % kv for kv in res.items()))


@pytest.mark.parametrize("logn", [27, 30])
def test_large_unique_properties(P, logn):
    """2^27 ⋈ 2^27 unique keys generated on the device (two independent pseudo-random permutations):
    matches = N; partitioning preserves the (key,payload) multiset and places every tuple; the
    materialised output has N tuples whose digest equals the digest of (R[i], i, pos_S(R[i]))."""
    import torch
    n = 1 << logn
    dev = torch.device("cuda:0")
    Rk, Sk = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    Rp, Sp = torch.empty_like(Rk), torch.empty_like(Sk)
    # run on torch's current stream so that torch ops and libhj kernels are ordered with each other
    with P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream) as hj:
        hj.gen_unique(Rk, n, 0, n, 1)
        hj.gen_unique(Sk, n, 0, n, 2)
        hj.fill_payload(Rp, n, "rowid")
        hj.fill_payload(Sp, n, "rowid")
        hj.sync()
        assert int(torch.unique(Rk).numel()) == n and int(Rk.min()) == 0 and int(Rk.max()) == n - 1
        hj.bind_device(P.REL_R, Rk, Rp)
        hj.bind_device(P.REL_S, Sk, Sp)
        before = hj.digest_pairs(Rk, Rp, n)
        m, agg = hj.join()
        assert m == n
        # sum_i rowidR(i) * rowidS(match) mod 2^64, from torch (independent of the join path)
        inv = torch.empty(n, dtype=torch.int64, device=dev)
        inv[Sk.long()] = torch.arange(n, device=dev)
        exp_agg = int((torch.arange(n, device=dev) * inv[Rk.long()]).sum().item()) % (1 << 64)
        assert agg == exp_agg
        bad, _ = hj.verify_partitions(P.REL_R)
        assert bad == 0
        k, p, off, nparts = hj.partition_pointers(P.REL_R)
        assert hj.digest_pairs(k, p, n) == before
        ok, opr, ops = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
        assert hj.join_materialize_into(ok, opr, ops, n) == n
        exp_digest = hj.digest_triples(Rk, Rp, inv[Rk.long()].int(), n)
        assert hj.digest_triples(ok, opr, ops, n) == exp_digest
        assert torch.equal(Rk[opr.long()], ok) and torch.equal(Sk[ops.long()], ok)   # key == R[payR] == S[payS]


def test_largest_two_pass_size(P):
    """2^31 x 2^31 unique keys: 18 radix bits = 8192-tuple partitions (two LDS-table chunks per partition).  Count,
    multiset-preserving partitioning, placement."""
    import torch
    n = 1 << 31
    dev = torch.device("cuda:0")
    Rk, Sk, Rp, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    with P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream) as hj:
        hj.gen_unique(Rk, n, 0, n, 11)
        hj.gen_unique(Sk, n, 0, n, 12)
        hj.fill_payload(Rp, n, "rowid")
        hj.fill_payload(Sp, n, "ones")
        hj.sync()
        hj.bind_device(P.REL_R, Rk, Rp)
        hj.bind_device(P.REL_S, Sk, Sp)
        before = hj.digest_pairs(Rk, Rp, n)
        m, agg = hj.join()
        assert m == n
        assert agg == (n * (n - 1) // 2) % (1 << 64)           # sum of R's row ids (payS = 1), every R tuple matched once
        assert hj.config()["bits1"] + hj.config()["bits2"] == 18
        bad, _ = hj.verify_partitions(P.REL_R)
        assert bad == 0
        k, p, off, nparts = hj.partition_pointers(P.REL_R)
        assert nparts == 1 << 18 and hj.digest_pairs(k, p, n) == before


def test_beyond_2p32_tuples(P):
    """288 GB of HBM, not 32-bit positions, is what bounds a relation.  PK-FK 2^32 x 2^31: R holds every int32
    value exactly once (a pseudo-random permutation of the whole key domain), S 2^31 distinct keys — positions inside the
    partition kernels are 32-bit LINE numbers, byte addresses 64-bit.  Size-independent properties: count = |S| (64-bit, beyond
    INT32_MAX), multiset-preserving partitioning of the 2^32-tuple relation (order-independent digest), every tuple in its
    partition; a relation that cannot fit the card is refused with a memory message before anything is allocated."""
    import torch
    nR, nS = 1 << 32, 1 << 31
    dev = torch.device("cuda:0")
    Rk, Rp = (torch.empty(nR, dtype=torch.int32, device=dev) for _ in range(2))
    Sk, Sp = (torch.empty(nS, dtype=torch.int32, device=dev) for _ in range(2))
    with P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream) as hj:
        hj.gen_unique(Rk, nR, 0, nR, 21)
        hj.gen_unique(Sk, nS, 0, nR, 22)
        hj.fill_payload(Rp, nR, "ones")
        hj.fill_payload(Sp, nS, "ones")
        hj.sync()
        hj.bind_device(P.REL_R, Rk, Rp)
        hj.bind_device(P.REL_S, Sk, Sp)
        before = hj.digest_pairs(Rk, Rp, nR)
        m, agg = hj.join()
        assert m == agg == nS and m > 2**31 - 1
        assert hj.config()["bits1"] + hj.config()["bits2"] == 18 and hj.config()["build_side"] == 2
        assert hj.partition_layout(P.REL_R) == "slotted" and hj.partition_layout(P.REL_S) == "slotted"
        bad, _ = hj.verify_partitions(P.REL_R)
        assert bad == 0
        del Sk, Sp                                   # room for the gap-free copy of R's partitions
        hj.bind_device(P.REL_S, Rk[:16], Rp[:16])
        k, p, off, nparts = hj.partition_pointers(P.REL_R)
        assert nparts == 1 << 18 and hj.digest_pairs(k, p, nR) == before
        # the exact passes address the same sizes (histogram + scan + scatter, 64-bit offsets)
        hj.configure(exact_only=True)
        hj.partition(P.REL_R)
        bad, _ = hj.verify_partitions(P.REL_R)
        assert bad == 0
        k, p, off, nparts = hj.partition_pointers(P.REL_R)
        assert hj.digest_pairs(k, p, nR) == before
        # 2^34 tuples of buffers cannot fit 288 GB: refused by the memory check, nothing allocated, the context stays usable
        with pytest.raises(P.HJError) as ei:
            hj.bind_device(P.REL_R, Rk, Rp, n=1 << 34)
            hj.partition(P.REL_R)
        assert "GiB" in str(ei.value)


# ---- streaming probe side (outOfGPU_Join3_payload, hjcp.cu:1684-1984) ----------------
@pytest.mark.parametrize("seg", [0, 1000, 4096, 50_000, 10**9])
def test_stream_probe_segments(P, seg):
    rng = np.random.default_rng(77)
    nR, nS = 40_000, 333_333
    R = rng.permutation(nR).astype(np.int32)
    S = rng.integers(0, nR + 50, nS).astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
    Ps = np.arange(nS, dtype=np.int32)
    em, eagg, _ = o.join_count(R, Pr, S, Ps, checksum=False)
    with P.HashJoin(0) as hj:
        hj.load_host(P.REL_R, R, Pr)
        assert hj.join_stream_probe(S, Ps, segment_tuples=seg) == (em, eagg)          # given payloads
        assert hj.join_stream_probe(S, None, "rowid", segment_tuples=seg) == (em, eagg)  # global row ids
        ones = o.join_count(R, Pr, S, None, checksum=False)
        assert hj.join_stream_probe(S, None, "ones", segment_tuples=seg) == ones[:2]
        # the resident path still works afterwards and agrees
        hj.load_host(P.REL_S, S, Ps)
        assert hj.join() == (em, eagg)


@pytest.mark.parametrize("seg", [0, 7000, 60_000])
def test_stream_probe_materialize(P, seg):
    """Join3 with its materialisation: per segment join_partitioned_results + copy of the output to the host on a
    third stream (hjcp.cu:1917-1961).  The multiset of (key,payR,payS) over all segments must equal the oracle's."""
    rng = np.random.default_rng(78)
    nR, nS = 30_000, 200_003
    R = rng.integers(0, 25_000, nR).astype(np.int32)                 # duplicates on the build side too
    S = rng.integers(-100, 26_000, nS).astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
    Ps = np.arange(nS, dtype=np.int32)
    ek, epr, eps = o.join_materialize(R, Pr, S, Ps)
    em, eagg, _ = o.join_count(R, Pr, S, Ps, checksum=False)
    with P.HashJoin(0) as hj:
        hj.load_host(P.REL_R, R, Pr)
        (k, pr, ps), agg = hj.join_stream_probe_materialize(S, Ps, segment_tuples=seg)
        assert len(k) == em and agg == eagg
        for a, b in zip(sorted_triples(k, pr, ps), sorted_triples(ek, epr, eps)):
            assert np.array_equal(a, b)
        (k2, pr2, ps2), _ = hj.join_stream_probe_materialize(S, None, "rowid", segment_tuples=seg)   # global row ids
        for a, b in zip(sorted_triples(k2, pr2, ps2), sorted_triples(ek, epr, eps)):
            assert np.array_equal(a, b)
        # capacity: nothing beyond cap is written, HJ_ECAPACITY reports the true size
        out = [np.full(em, -7, np.int32) for _ in range(3)]
        with pytest.raises(P.HJError) as ei:
            hj.join_stream_probe_materialize(S, Ps, segment_tuples=seg, cap=em // 2, out=out)
        assert ei.value.code == -4 and all(np.all(x[em // 2:] == -7) for x in out)


def test_stream_probe_materialize_segments_that_do_not_fit_are_redone(P):
    """The materialising streaming loop writes every segment's output in ONE probe into device columns sized for one match per
    probe tuple and looks at a segment's cursor one segment later (no blocking read per segment).  A segment that produces more
    (every R key three times here), or whose slots overflowed (half of a segment one key), left nothing usable: it is redone at
    the end with exact sizes.  The multiset over all segments must still equal the oracle's, the aggregate too."""
    rng = np.random.default_rng(79)
    nR, nS, seg = 3 * 20_000, 1 << 19, 1 << 17
    base = rng.permutation(20_000).astype(np.int32)
    R = np.concatenate([base, base, base])                          # 3 matches per probe tuple: every segment exceeds its columns
    S = base[rng.integers(0, 20_000, nS)].astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
    Ps = np.arange(nS, dtype=np.int32)
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=True)
    assert em == 3 * nS
    with P.HashJoin(0) as hj:
        hj.configure(bits1=4, bits2=3)
        hj.load_host(P.REL_R, R, Pr)
        (k, pr, ps), agg = hj.join_stream_probe_materialize(S, Ps, segment_tuples=seg)
        assert len(k) == em and agg == eagg and o.triples_checksum(k, pr, ps) == echk
    # unique R, two of four segments skewed (slot overflow -> redo), the others take the one-probe road
    R = rng.permutation(1 << 16).astype(np.int32)
    S = R[rng.integers(0, 1 << 16, nS)].astype(np.int32)
    S[: seg // 2] = R[5]
    S[2 * seg: 2 * seg + seg // 2] = R[9]
    Pr = np.arange(len(R), dtype=np.int32)
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=True)
    with P.HashJoin(0) as hj:
        hj.configure(bits1=5, bits2=4)
        hj.load_host(P.REL_R, R, Pr)
        for _ in range(2):
            (k, pr, ps), agg = hj.join_stream_probe_materialize(S, Ps, segment_tuples=seg)
            assert len(k) == em and agg == eagg and o.triples_checksum(k, pr, ps) == echk
            assert np.array_equal(R[pr], k) and np.array_equal(S[ps], k)


def test_stream_probe_edge_cases(P):
    R = np.arange(100, dtype=np.int32)
    with P.HashJoin(0) as hj:
        with pytest.raises(P.HJError):
            hj.join_stream_probe(R)                       # R not loaded
        hj.load_host(P.REL_R, R)
        assert hj.join_stream_probe(np.empty(0, np.int32)) == (0, 0)
        assert hj.join_stream_probe(np.array([5, 5, 200], np.int32), segment_tuples=1) == (2, 2)


# ---- API contract: errors, config, output capacity, CLI multipliers / --file -------------------------------
def test_api_errors_and_config(P):
    import ctypes as C
    R = np.arange(1000, dtype=np.int32)
    with P.HashJoin(0) as hj:
        with pytest.raises(P.HJError):
            hj.partition(P.REL_R)                                   # nothing loaded
        hj.load_host(P.REL_R, R, R)
        with pytest.raises(P.HJError):
            hj.join_count()                                         # not partitioned
        with pytest.raises(P.HJError):
            hj.configure(bits1=10)                                  # at most 9 bits per pass
        with pytest.raises(P.HJError):
            hj.configure(lds_heads=1000)                            # power of two
        hj.load_host(P.REL_S, np.repeat(R, 3), None, "rowid")
        hj.configure(bits1=3, bits2=2, build_side=2, lds_capacity=512, lds_heads=128, probe_chunk=64)
        c = hj.config()
        assert (c["bits1"], c["bits2"], c["build_side"], c["lds_capacity"], c["lds_heads"], c["probe_chunk"]) == (3, 2, 2, 512, 128, 64)
        assert hj.join() == (3000, o.join_count(R, R, np.repeat(R, 3), np.arange(3000, dtype=np.int32), checksum=False)[1])
        # output capacity: nothing beyond cap is written, the call reports HJ_ECAPACITY and the true size
        bufs = [hj.device_malloc(4 * 4000) for _ in range(3)]
        guard = np.full(1000, -7, np.int32)
        for b in bufs:
            hj._ck(hj._L.hj_memcpy_h2d(hj._h, C.c_void_p(b + 4 * 2000), guard.ctypes.data_as(C.c_void_p), 4000))
        n = C.c_uint64()
        rc = hj._L.hj_join_materialize(hj._h, C.c_void_p(bufs[0]), C.c_void_p(bufs[1]), C.c_void_p(bufs[2]), 2000, C.byref(n))
        assert rc == -4 and n.value == 3000                          # HJ_ECAPACITY
        for b in bufs:
            assert np.all(hj.to_host(b + 4 * 2000, 1000, np.int32) == -7)
            hj.device_free(b)
        # partitioning one relation with other bits than its partner is refused, not mis-joined
        hj.configure(bits1=4)
        hj.partition(P.REL_R)
        hj.configure(bits1=5)
        hj.partition(P.REL_S)
        hj.configure(bits1=4)   # invalidates both
        with pytest.raises(P.HJError):
            hj.join_count()
        # maximum size: what the card's memory holds (positions inside the partition kernels are 32-bit line numbers; the memory check
        # is exercised with real columns in test_beyond_2p32_tuples); beyond 2^34 tuples the structural limit answers before any
        # buffer is allocated or any byte of the (here fictitious) columns is read
        hj.configure()
        rc = hj._L.hj_bind_device(hj._h, P.REL_R, C.c_void_p(1 << 20), C.c_void_p(1 << 21), (1 << 34) + 16)
        assert rc == 0
        assert hj._L.hj_partition(hj._h, P.REL_R) == -1 and b"too large" in hj._L.hj_error(hj._h)
        hj.load_host(P.REL_R, R, R)


def test_bench_cli_file_and_multipliers(P, tmp_path):
    """--file -k/-l (main.cu:186-189) and -x/-y multipliers (create_relation_n, main.cu:103-109,208-248)."""
    import subprocess
    g = P.generator
    g.seed_generator(4)
    R = g.create_relation_unique(None, 5000, 5000)
    S = g.create_relation_unique(None, 12000, 5000)
    g.writeToFile(str(tmp_path / "r.bin"), R)
    g.writeToFile(str(tmp_path / "s.bin"), S)
    expect = o.join_count(R, None, S, None, checksum=False)[0]
    r = subprocess.run([P._lib.BENCH_PATH, "-b", "7", "-a", "HJC", "-R", "5000", "-S", "12000", "--file", "-k", "r.bin", "-l", "s.bin"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Reading from files" in r.stdout and "%d results" % expect in r.stdout
    # short file → error, like D12 says it should be
    r = subprocess.run([P._lib.BENCH_PATH, "-b", "7", "-a", "HJC", "-R", "5001", "-S", "12000", "--file", "-k", "r.bin", "-l", "s.bin"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
    # -y 3 -x 2: R = 3 copies of a 1000-key permutation, S = 2 copies → 1000 * 3 * 2 matches
    r = subprocess.run([P._lib.BENCH_PATH, "-b", "7", "-a", "HJC", "-R", "1000", "-S", "1000", "-y", "3", "-x", "2", "--seed", "6"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "6000 results" in r.stdout


# ---- CPU-GPU co-processing (outOfGPU_Join2_payload, hjcp.cu:1000-1680) ---------------
@pytest.mark.parametrize("parts,threads", [(0, 0), (1, 1), (16, 3), (7, 2), (64, 8)])
def test_coprocess(P, parts, threads):
    rng = np.random.default_rng(parts * 100 + threads)
    nR, nS = 60_001, 250_000
    R = rng.integers(-5000, 50_000, nR).astype(np.int32)
    S = rng.integers(-5000, 50_000, nS).astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
    Ps = np.arange(nS, dtype=np.int32)
    em, eagg, _ = o.join_count(R, Pr, S, Ps, checksum=False)
    with P.HashJoin(0) as hj:
        assert hj.join_coprocess(R, Pr, S, Ps, parts, threads) == (em, eagg)
        assert hj.join_coprocess(R, None, S, None, parts, threads) == o.join_count(R, None, S, None, checksum=False)[:2]
        assert hj.host_split_throughput() > 0                       # the host split reports its GB/s (pp.cu:218)
        assert hj.coprocess_groups() == 1                           # everything fits the card: one residency group, one join
        e = np.empty(0, np.int32)
        assert hj.join_coprocess(e, None, S, None, parts, threads) == (0, 0)
        # the resident path is usable afterwards
        hj.load_host(P.REL_R, R, Pr)
        hj.load_host(P.REL_S, S, Ps)
        assert hj.join() == (em, eagg)


def test_coprocess_residency_groups(P, monkeypatch):
    """Level-0 pairs are uploaded and joined in residency groups — runs of consecutive pairs whose tuples fit a device-memory budget
    (the reference's groupOptimal2, pp.cu:307-468).  A small budget forces many groups, a partition above the budget is a group of its
    own; the result does not depend on the grouping."""
    rng = np.random.default_rng(404)
    nR, nS = 200_000, 900_000
    R = rng.integers(0, 150_000, nR).astype(np.int32)
    S = rng.integers(0, 150_000, nS).astype(np.int32)
    S[: nS // 3] = 77                                               # one level-0 partition far above any small budget
    Pr, Ps = np.arange(nR, dtype=np.int32), np.arange(nS, dtype=np.int32)
    expect = o.join_count(R, Pr, S, Ps, checksum=False)[:2]
    for split in ("1", "2"):                                        # one pass into blocks (default) / the two-pass split of round 4
        monkeypatch.setenv("HJ_COPROCESS_SPLIT", split)
        seen = []
        for budget in ("0", "1", "60000", "150000", "400000", "100000000"):
            monkeypatch.setenv("HJ_COPROCESS_GROUP_TUPLES", budget)
            with P.HashJoin(0) as hj:
                assert hj.join_coprocess(R, Pr, S, Ps, 16, 4) == expect, (split, budget)
                seen.append(hj.coprocess_groups())
        assert seen[0] == 1 and seen[1] == 16 and seen[-1] == 1        # no override: the card's budget; 1 tuple: every pair alone
        assert seen[1] >= seen[2] >= seen[3] >= seen[4] >= seen[5]


def test_coprocess_many_blocks_per_partition(P):
    """The one-pass split hands a partition over as a list of blocks from every worker's arena; at 2^22 tuples a (worker, partition)
    share is several blocks, merged into one upload where they are neighbours.  Same result as the resident join."""
    n = 1 << 22
    rng = np.random.default_rng(5)
    R = rng.permutation(n).astype(np.int32)
    S = rng.integers(0, n + n // 8, n).astype(np.int32)
    Pr = np.arange(n, dtype=np.int32)
    Ps = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    with P.HashJoin(0) as hj:
        hj.load_host(P.REL_R, R, Pr)
        hj.load_host(P.REL_S, S, Ps)
        expect = hj.join()
        assert expect[0] == int((S < n).sum())
        for threads in (1, 5, 8):
            assert hj.join_coprocess(R, Pr, S, Ps, 16, threads) == expect, threads
        assert hj.join_coprocess(R, None, S, None, 16, 8)[0] == expect[0]


def test_coprocess_2p24_with_payload_columns(P):
    """2^24 x 2^24 from host memory WITH payload columns (four columns through the one-pass split and the uploads that run beside it),
    against closed forms: unique keys on both sides -> 2^24 matches, aggregate = sum over the keys of payR * payS (torch, int64 wrap)."""
    import torch
    n = 1 << 24
    g = torch.Generator().manual_seed(9)
    R = torch.randperm(n, generator=g, dtype=torch.int32)
    S = torch.randperm(n, generator=g, dtype=torch.int32)
    Pr = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, generator=g)
    Ps = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, generator=g)
    pr_by_key = torch.empty(n, dtype=torch.int64)
    ps_by_key = torch.empty(n, dtype=torch.int64)
    pr_by_key[R.long()] = Pr.long()
    ps_by_key[S.long()] = Ps.long()
    expect = int((pr_by_key * ps_by_key).sum().item()) % 2**64          # int64 arithmetic wraps mod 2^64 like the device's
    with P.HashJoin(0) as hj:
        for threads in (0, 3):
            m, agg = hj.join_coprocess(R.numpy(), Pr.numpy(), S.numpy(), Ps.numpy(), 16, threads)
            assert (m, agg) == (n, expect), threads
        assert hj.coprocess_groups() == 1


def test_reference_entry_dispatch(P, golden_dir, capfd, monkeypatch):
    """hj_ClusteredProbe's three-way dispatch (hjcp.cu:2001-2008): resident / streamed S / co-processing."""
    R = _load(golden_dir, "unique_4096.bin")
    S = _load(golden_dir, "zipf_S20000_a4096_t1.0_seed42.bin")
    expect = len(S) - int((S == 4096).sum())
    for path, line in (("stream", "Total Throughput (Streaming) "), ("coprocess", "Total Throughput (Co-processing) "),
                       ("resident", "With materialization")):
        monkeypatch.setenv("HJ_FORCE_PATH", path)
        r = P.hashJoinClusteredProbe(R, S)
        out = capfd.readouterr().out
        assert r["status"] == 0 and r["matches"] == r["agg"] == expect, (path, r)
        assert line in out and "%d results" % expect in out


# ---- §8(f) rank 3: late materialisation (jp.cu:1420-1557) / rank 4: non-partitioned baselines (jp.cu:628-742) ----
# (bits 8+8: 16 radix bits -> the kernels that compare 16-bit tags, k_join<true, 2>)
@pytest.mark.parametrize("cfg", [None, dict(bits1=5, bits2=4), dict(build_side=2), dict(bits1=8, bits2=8)])
def test_late_materialize(P, cfg):
    rng = np.random.default_rng(31)
    nR, nS, c1, c2 = 30_000, 100_000, 3, 2
    R = rng.permutation(nR).astype(np.int32)
    S = rng.integers(0, nR + 100, nS).astype(np.int32)
    Dr = rng.integers(-2**31, 2**31 - 1, (c1, nR)).astype(np.int32)     # column-major: column z at z*stride
    Ds = rng.integers(-2**31, 2**31 - 1, (c2, nS)).astype(np.int32)
    rid_r, rid_s = np.arange(nR, dtype=np.int32), np.arange(nS, dtype=np.int32)
    _, pr, ps = o.join_materialize(R, rid_r, S, rid_s)                 # matching (rowR, rowS) pairs from the oracle
    expect = (int(Dr[:, pr].astype(np.int64).sum()) + int(Ds[:, ps].astype(np.int64).sum())) % 2**64
    with P.HashJoin(0) as hj:
        if cfg:
            hj.configure(**cfg)
        hj.load_host(P.REL_R, R, None, "rowid")
        hj.load_host(P.REL_S, S, None, "rowid")
        hj.partition(P.REL_R)
        hj.partition(P.REL_S)
        dDr, dDs = hj.to_device(Dr), hj.to_device(Ds)
        try:
            assert hj.join_late_materialize(dDr, c1, nR, dDs, c2, nS) == (len(pr), expect)
            assert hj.join_late_materialize(dDr, 1, nR, None, 0, 0)[1] == int(Dr[0, pr].astype(np.int64).sum()) % 2**64
            assert hj.join_count()[0] == len(pr)                       # the ordinary count still works on these partitions
        finally:
            hj.device_free(dDr)
            hj.device_free(dDs)


def test_hbm_ceiling_microbenchmarks_move_what_they_say(P):
    """hj_ubench (the same-run HBM ceilings bench.py prices the pass kernels against): the copy really copies both columns, the line
    scatter writes every 128-byte line of its power-of-two prefix exactly once, the write-only stream writes, the read-only stream
    leaves the output alone — and each reports a rate."""
    import torch
    n = (1 << 20) + 4096
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, generator=g).to(dev)
    b = torch.arange(n, dtype=torch.int32, device=dev)
    with P.HashJoin(0) as hj:
        c, d = torch.zeros_like(a), torch.zeros_like(b)
        assert hj.ubench("copy", a, b, c, d, n, reps=2) > 0
        assert torch.equal(c, a) and torch.equal(d, b)
        c.fill_(-1)
        d.fill_(-1)
        assert hj.ubench("line_scatter", a, b, c, d, n, reps=2) > 0
        m = 1 << 20                                                  # the largest power-of-two number of 128-B lines: 2^20 tuples
        lines_in = b[:m].view(-1, 32)                                # payload = row id: a line is 32 consecutive ids
        lines_out = d[:m].view(-1, 32)
        assert torch.equal(lines_out[:, 1:] - lines_out[:, :1], lines_in[:, 1:] - lines_in[:, :1])   # whole lines, order kept inside
        assert torch.equal(torch.sort(lines_out[:, 0]).values, lines_in[:, 0])                          # every line exactly once
        assert torch.equal(a[d[:m].long()], c[:m])                   # the key column went with it
        assert bool((d[m:] == -1).all())
        c.fill_(-1)
        assert hj.ubench("read", a, b, c, d, n, reps=2) > 0 and bool((c == -1).all())
        assert hj.ubench("write", a, b, c, d, n, reps=2) > 0 and not bool((c == -1).any())


def test_layout_gate_microbenchmarks_move_what_they_say(P):
    """hj_ubench kinds 4-7 (the layout gate of round 6): one array per side.  The one-array copy copies; the line-pair copy copies;
    the two scattering kinds write every 256-byte line pair of their power-of-two prefix exactly once, keys in front of their payloads;
    columns that are not the halves of one allocation are refused."""
    import torch
    n = (1 << 18) + 2048
    dev = torch.device("cuda:0")
    src = torch.arange(2 * n, dtype=torch.int32, device=dev)
    dst = torch.full((2 * n,), -1, dtype=torch.int32, device=dev)
    a, b, c, d = src[:n], src[n:], dst[:n], dst[n:]
    with P.HashJoin(0) as hj:
        for kind in ("copy1", "pairs"):
            dst.fill_(-1)
            assert hj.ubench(kind, a, b, c, d, n, reps=2) > 0
            assert torch.equal(dst, src), kind
        m = 1 << 18                                                  # line pairs covered: the largest power of two, 2^13 of 32 tuples
        dst.fill_(-1)
        assert hj.ubench("pairs_scatter", a, b, c, d, n, reps=2) > 0
        pin, pout = src[:2 * m].view(-1, 64), dst[:2 * m].view(-1, 64)
        assert torch.equal(pout[:, 1:] - pout[:, :1], pin[:, 1:] - pin[:, :1])                  # whole 256-byte pairs, order kept inside
        assert torch.equal(torch.sort(pout[:, 0]).values, pin[:, 0])                                # every pair exactly once
        assert bool((dst[2 * m:] == -1).all())
        dst.fill_(-1)
        assert hj.ubench("soa_to_pairs_scatter", a, b, c, d, n, reps=2) > 0                         # columns in, line pairs out
        pout = dst[:2 * m].view(-1, 64)
        assert torch.equal(pout[:, 32:], pout[:, :32] + n)                                          # a line's payloads behind its keys
        assert torch.equal(torch.sort(pout[:, 0]).values, torch.arange(0, m, 32, dtype=torch.int32, device=dev))
        assert torch.equal(pout[:, 1:32] - pout[:, :1], torch.arange(1, 32, dtype=torch.int32, device=dev).expand(m // 32, 31))
        with pytest.raises(Exception):
            hj.ubench("copy1", a, torch.empty_like(a), c, d, n, reps=1)


def test_nonpartitioned_baselines(P, golden_dir):
    rng = np.random.default_rng(32)
    R = _load(golden_dir, "unique_4096.bin")
    Z = _load(golden_dir, "zipf_S20000_a4096_t1.0_seed42.bin")
    A = _load(golden_dir, "nonuniq_R6000_seed7.bin")
    B = _load(golden_dir, "nonuniq_S9000_seed8.bin")
    Pr = rng.integers(-2**31, 2**31 - 1, len(R)).astype(np.int32)
    Pz = rng.integers(-2**31, 2**31 - 1, len(Z)).astype(np.int32)
    with P.HashJoin(0) as hj:
        hj.load_host(P.REL_R, R, Pr)
        hj.load_host(P.REL_S, Z, Pz)
        exp = o.join_count(R, Pr, Z, Pz, checksum=False)[:2]
        assert hj.join_nonpartitioned(0) == exp                        # perfect array: unique build keys
        assert hj.join_nonpartitioned(1) == exp                        # global chained table
        assert hj.join() == exp                                        # and the partitioned path agrees
        hj.load_host(P.REL_R, A)
        hj.load_host(P.REL_S, B)
        assert hj.join_nonpartitioned(1) == o.join_count(A, None, B, None, checksum=False)[:2]   # duplicates on both sides
        neg = np.array([-5, 3, 7], np.int32)
        hj.load_host(P.REL_R, neg)
        hj.load_host(P.REL_S, np.array([3, -5, -5, 9], np.int32))
        assert hj.join_nonpartitioned(1) == (3, 3)
        with pytest.raises(P.HJError):
            hj.join_nonpartitioned(0)                                  # negative build keys: no direct addressing


# ---- randomised differential test: many small configurations against the oracle ----------------------------
@pytest.mark.parametrize("seed", range(40))
def test_fuzz_against_oracle(P, seed):
    rng = np.random.default_rng(1000 + seed)
    nR = int(rng.choice([0, 1, 7, 100, 4097, int(rng.integers(1, 60_000))]))
    nS = int(rng.choice([0, 1, 9, 333, 8193, int(rng.integers(1, 120_000))]))
    kind = seed % 5
    if kind == 0:      # few distinct keys: heavy duplicates, quadratic output
        dom = int(rng.integers(1, 40)); nR, nS = min(nR, 3000), min(nS, 3000)
        R = rng.integers(0, dom, nR); S = rng.integers(0, dom, nS)
    elif kind == 1:    # full int32 range incl. negatives; S partly drawn from R
        R = rng.integers(-2**31, 2**31 - 1, nR)
        S = np.concatenate([rng.choice(R, nS // 2) if nR else np.empty(0, np.int64), rng.integers(-2**31, 2**31 - 1, nS - nS // 2)])
    elif kind == 2:    # dense unique PK, uniform FK
        R = rng.permutation(nR); S = rng.integers(0, max(nR, 1) + 10, nS)
    elif kind == 3:    # keys that differ only in high bits (tag / full-key paths, one hot partition)
        R = (rng.integers(0, 1 << 12, nR) << 20) | 5; S = (rng.integers(0, 1 << 12, nS) << 20) | 5
        nR, nS = len(R), len(S)
    else:              # zipf-like
        R = rng.permutation(nR); S = np.minimum((rng.pareto(1.0, nS) * 3).astype(np.int64), max(nR, 1))
    R, S = R.astype(np.int32), S.astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, len(R)).astype(np.int32)
    Ps = rng.integers(-2**31, 2**31 - 1, len(S)).astype(np.int32)
    cfg = dict(bits1=int(rng.integers(0, 10)), bits2=int(rng.integers(0, 10)), force_bits=True,
               build_side=int(rng.integers(0, 3)), lds_capacity=int(rng.choice([0, 64, 300, 4608])),
               lds_heads=int(rng.choice([0, 1, 16, 1024])), probe_chunk=int(rng.choice([0, 50, 1000, 65536])),
               exact_only=bool(seed % 3 == 0))
    if seed % 7 == 0:
        cfg = None     # library defaults
    em, eagg, echk = o.join_count(R, Pr, S, Ps)
    with P.HashJoin(0) as hj:
        if cfg:
            hj.configure(**cfg)
        hj.load_host(P.REL_R, R, Pr)
        hj.load_host(P.REL_S, S, Ps)
        assert hj.join() == (em, eagg), cfg
        k, pr, ps = hj.join_materialize()
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk, cfg
        for rel, (kk, pp) in ((P.REL_R, (R, Pr)), (P.REL_S, (S, Ps))):
            bad, dg = hj.verify_partitions(rel, with_digests=True)
            c = hj.config()
            ok, op, ooff = o.radix_partition(kk, pp, 0, c["bits1"] + c["bits2"])
            assert bad == 0 and np.array_equal(dg, o.partition_digest(ok, op, ooff)), cfg
