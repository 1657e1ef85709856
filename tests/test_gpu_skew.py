"""BASELINE config 4: PK-FK join with Zipf-skewed foreign keys (the skew-handling probe path).
Reference semantics: R = perm(0..N-1), S in {1..N} by gen_zipf (src/generator_ETHZ.cu:299-348) →
matches = M - #{S == N}; the count exceeds INT32_MAX at full size (D5)."""
import math

import numpy as np
import pytest

from hjtest import pkg, sorted_triples
from oracle import pyoracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return pkg()


def test_device_zipf_generator_distribution(P):
    import torch
    n, alphabet, theta = 1 << 22, 1 << 16, 1.0
    k = torch.empty(n, dtype=torch.int32, device="cuda:0")
    with P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream) as hj:
        hj.gen_zipf(k, n, 0, alphabet, theta, 5)
        hj.sync()
    v = k.cpu().numpy()
    assert v.min() >= 1 and v.max() <= alphabet                 # alphabet 1..N, no zeros (gen.cu:245)
    cnt = np.sort(np.bincount(v, minlength=alphabet + 1))[::-1].astype(np.float64)
    HN = sum(1.0 / i for i in range(1, alphabet + 1))
    for rank in range(1, 9):                                    # head of the distribution: p_k = 1/(k H_N)
        p = 1.0 / (rank * HN)
        assert abs(cnt[rank - 1] - n * p) < 6 * math.sqrt(n * p), (rank, cnt[rank - 1], n * p)
    assert np.count_nonzero(cnt) > 0.6 * alphabet               # the tail is populated too


@pytest.mark.parametrize("cfg", [None, dict(bits1=6, bits2=5, probe_chunk=2048), dict(build_side=2)])
def test_config4_shape_reference_stream(P, cfg):
    """R unique, S from the reference's own Zipf stream (product generator = reference generator)."""
    g = P.generator
    nR, nS = 1 << 18, 1 << 22
    g.seed_generator(12345)
    R = g.create_relation_unique(None, nR, nR)
    S = g.create_relation_zipf(None, nS, nR, 1.0)
    expect = nS - int((S == nR).sum())
    em, eagg, echk = o.join_count(R, None, S, None, checksum=False)
    assert em == expect
    Pr, Ps = np.arange(nR, dtype=np.int32), np.arange(nS, dtype=np.int32)
    with P.HashJoin(0) as hj:
        if cfg:
            hj.configure(**cfg)
        hj.load_host(P.REL_R, R, Pr)
        hj.load_host(P.REL_S, S, Ps)
        m, agg = hj.join()
        assert m == expect
        assert agg == o.join_count(R, Pr, S, Ps, checksum=False)[1]
        bad, dg = hj.verify_partitions(P.REL_S, with_digests=True)
        assert bad == 0
        c = hj.config()
        ok, op, ooff = o.radix_partition(S, Ps, 0, c["bits1"] + c["bits2"])
        assert np.array_equal(dg, o.partition_digest(ok, op, ooff))    # skewed partitions hold the right multisets
        k, pr, ps = hj.join_materialize()
    assert len(k) == expect
    assert np.array_equal(R[pr], k) and np.array_equal(S[ps], k)
    assert len(np.unique(ps)) == expect                                  # every matching S tuple exactly once


def test_config4_full_size_properties(P):
    """2^27 ⋈ 2^31, Zipf theta = 1.0 generated on the device: 64-bit count, closed form, partition checks."""
    import torch
    nR, nS = 1 << 27, 1 << 31
    dev = torch.device("cuda:0")
    Rk = torch.empty(nR, dtype=torch.int32, device=dev)
    Rp = torch.empty(nR, dtype=torch.int32, device=dev)
    Sk = torch.empty(nS, dtype=torch.int32, device=dev)
    Sp = torch.empty(nS, dtype=torch.int32, device=dev)
    with P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream) as hj:
        hj.gen_unique(Rk, nR, 0, nR, 3)
        hj.gen_zipf(Sk, nS, 0, nR, 1.0, 4)
        hj.fill_payload(Rp, nR, "ones")
        hj.fill_payload(Sp, nS, "ones")
        hj.sync()
        absent = int((Sk == nR).sum().item())          # the one alphabet value R does not hold
        top = int(torch.bincount(Sk[: 1 << 26].long(), minlength=nR + 1).max().item())
        assert top > 0.04 * (1 << 26)                  # heaviest key ~ 1/H_N = 5.2 % of S
        hj.bind_device(P.REL_R, Rk, Rp)
        hj.bind_device(P.REL_S, Sk, Sp)
        before = hj.digest_pairs(Sk, Sp, nS)
        m, agg = hj.join()
        assert m == agg == nS - absent and m > 2**31 - 2**20 > 2**31 - 1 - 2**20
        assert hj.config()["build_side"] == 1          # the PK side builds; the skewed side is split into chunks
        bad, _ = hj.verify_partitions(P.REL_S)
        assert bad == 0
        k, p, off, nparts = hj.partition_pointers(P.REL_S)
        assert hj.digest_pairs(k, p, nS) == before
        # the first join found S skewed (its histogram-free attempt overflowed); the second goes straight to the exact
        # passes, splits the radix bits evenly and deals the LDS lines by need — same partitions, same result
        assert hj.partition_layout(P.REL_S) == "exact" and hj.partition_layout(P.REL_R) == "slotted"   # (introspection redid S exact)
        assert hj.join() == (m, agg)
        assert hj.partition_layout(P.REL_S) == "sampled"   # the skewed probe side: histogram-free passes, capacities from a sample
        bad, _ = hj.verify_partitions(P.REL_S)
        assert bad == 0
        k, p, off, nparts = hj.partition_pointers(P.REL_S)
        assert hj.digest_pairs(k, p, nS) == before


@pytest.mark.parametrize("cfg", [dict(bits1=8, bits2=8, exact_only=True), dict(bits1=6, bits2=6, exact_only=True),
                                 dict(bits1=9, bits2=7, exact_only=True), dict(bits1=4, bits2=9, exact_only=True), None])
def test_skewed_exact_passes_deal_lines_by_need(P, cfg):
    """Exact passes under skew: the 512 LDS lines are dealt to the digits by what the span's histogram says they will
    receive (one digit wanting more than all of them: scaled), heavy spans rank with the wave-aggregated atomic.
    Partition boundaries and contents must equal the oracle's, whatever the line allocation."""
    rng = np.random.default_rng(91)
    n = 1 << 20
    u = rng.random(n)
    keys = np.where(u < 0.55, 0x00ABCDEF, np.where(u < 0.75, 0x00ABCD11, np.where(u < 0.80, 77, rng.integers(0, 1 << 24, n)))).astype(np.int32)
    pays = np.arange(n, dtype=np.int32)
    R = rng.permutation(1 << 16).astype(np.int32)
    Rp = np.arange(len(R), dtype=np.int32)
    em, eagg, _ = o.join_count(R, Rp, keys, pays, checksum=False)
    with P.HashJoin(0) as hj:
        if cfg:
            hj.configure(**cfg)
        hj.load_host(P.REL_R, R, Rp)
        hj.load_host(P.REL_S, keys, pays)
        for _ in range(2):   # the second round of the default configuration runs with what the first learned
            assert hj.join() == (em, eagg)
            c = hj.config()
            bits = c["bits1"] + c["bits2"]
            gk, gp, goff = hj.partitions(P.REL_S, n)
            ok, op, ooff = o.radix_partition(keys, pays, 0, bits)
            assert np.array_equal(goff, ooff)
            assert np.array_equal(o.partition_digest(gk, gp, goff), o.partition_digest(ok, op, ooff))
            assert hj.partition_layout(P.REL_S) == "exact"


@pytest.mark.parametrize("cfg,nR,nS", [(dict(bits1=5, bits2=4), 1 << 16, 1 << 20), (dict(bits1=8, bits2=7), 1 << 18, 3 << 20),
                                       (None, 1 << 22, 1 << 23),
                                       # 16 / 17 radix bits: the sample histogram no longer fits one workgroup's LDS (sliced sampling pass)
                                       (dict(bits1=9, bits2=7), 1 << 20, 1 << 24), (dict(bits1=9, bits2=8), 1 << 20, 3 << 22),
                                       (dict(bits1=8, bits2=8), 1 << 21, 1 << 24),
                                       # a 512-way SECOND pass under skew (one line per child, the hot child's overflow leaves tuple by tuple)
                                       (dict(bits1=6, bits2=9), 1 << 20, 3 << 22),
                                       # few heads: 9 radix bits + log2(16 heads) < 16, so the tables hold FULL keys (since round 6 every default
                                       # shape takes 16-bit tags): the full-key kernels over list items
                                       (dict(bits1=5, bits2=4, lds_heads=16, lds_capacity=600), 1 << 16, 1 << 20)])
def test_sampled_path_for_a_skewed_probe_side(P, cfg, nR, nS):
    """A skewed relation on the probe side: the first join finds its slots overflowing, samples the key distribution once and
    from then on partitions it with the histogram-free passes at per-digit capacities (layout 'sampled': a partition is a list
    of ranges, the heavy digit is spread over many pass-2 workgroups).  Count, aggregate and the materialised multiset must
    equal the oracle's in both materialising paths; introspection still hands out oracle-identical gap-free partitions; data
    that changes under the binding so that the capacities no longer hold falls back to the exact passes."""
    import torch
    rng = np.random.default_rng(92)
    R = rng.permutation(nR).astype(np.int32)
    u = rng.random(nS)
    S = np.where(u < 0.40, R[3], np.where(u < 0.55, R[11], np.where(u < 0.60, R[12], R[rng.integers(0, nR, nS)]))).astype(np.int32)
    Pr = np.arange(nR, dtype=np.int32)
    Ps = (np.arange(nS, dtype=np.int64) * 7 % 1000003).astype(np.int32)
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=True)
    dR, dPr, dS, dPs = (torch.from_numpy(x).cuda() for x in (R, Pr, S, Ps))
    with P.HashJoin(0) as hj:
        if cfg:
            hj.configure(**cfg)
        hj.bind_device(P.REL_R, dR, dPr)
        hj.bind_device(P.REL_S, dS, dPs)
        for i in range(3):     # overflow -> sample -> sampled passes; then twice from the remembered tables
            hj.bind_device(P.REL_S, dS, dPs)
            assert hj.join() == (em, eagg), i
            assert hj.partition_layout(P.REL_S) == "sampled" and hj.partition_layout(P.REL_R) == "slotted"
        k, pr, ps = hj.join_materialize()
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        if em <= 2_000_000:
            for a, b in zip(sorted_triples(k, pr, ps), sorted_triples(*o.join_materialize(R, Pr, S, Ps))):
                assert np.array_equal(a, b)
        hj.partition(P.REL_R)
        hj.partition(P.REL_S)
        k, pr, ps = hj.join_materialize(cap=em)                       # one probe, no count before it
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        # small probe chunks: runs of whole ranges are closed when the next range would not fit, long ranges are cut
        hj.configure(**dict(cfg or {}, probe_chunk=3000))
        assert hj.join() == (em, eagg) and hj.partition_layout(P.REL_S) == "sampled"
        k, pr, ps = hj.join_materialize(cap=em)
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        # tiny chunks: most partitions of a wave have ranges longer than a chunk (k_join_expand writes their chunk items wave-wide)
        hj.configure(**dict(cfg or {}, probe_chunk=257))
        assert hj.join() == (em, eagg) and hj.partition_layout(P.REL_S) == "sampled"
        k, pr, ps = hj.join_materialize(cap=em)
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        hj.configure(**(cfg or {}))
        assert hj.join() == (em, eagg)
        # introspection: gap-free partitions identical to the oracle's (the relation is redone with the exact passes for it)
        c = hj.config()
        bits = c["bits1"] + c["bits2"]
        gk, gp, goff = hj.partitions(P.REL_S, nS)
        ok, op, ooff = o.radix_partition(S, Ps, 0, bits)
        assert np.array_equal(goff, ooff) and np.array_equal(o.partition_digest(gk, gp, goff), o.partition_digest(ok, op, ooff))
        assert hj.join() == (em, eagg) and hj.partition_layout(P.REL_S) == "sampled"
        # the data changes under the binding: another key is heavy now, the sampled capacities overflow -> exact passes
        S2 = np.where(u < 0.45, R[1000], R[rng.integers(0, nR, nS)]).astype(np.int32)
        dS.copy_(torch.from_numpy(S2).cuda())
        em2, eagg2, _ = o.join_count(R, Pr, S2, Ps, checksum=False)
        for _ in range(2):
            assert hj.join() == (em2, eagg2)
            assert hj.partition_layout(P.REL_S) == "exact"


@pytest.mark.parametrize("cfg,nR,nS,both", [(dict(bits1=5, bits2=4, build_side=2), 1 << 16, 1 << 20, False),
                                            (dict(bits1=8, bits2=7, build_side=2), 1 << 18, 3 << 20, False),
                                            (dict(build_side=2), 1 << 22, 1 << 24, False),
                                            (dict(bits1=6, bits2=5, build_side=2, lds_capacity=512, lds_heads=128), 1 << 20, 1 << 21, True),
                                            (dict(bits1=8, bits2=8, build_side=2), 1 << 18, 1 << 21, False)])
def test_skewed_build_side_one_launch_passes_and_flipped_roles(P, cfg, nR, nS, both):
    """the SKEWED relation builds (build_side=2: S).  After first contact S is partitioned by the histogram-free
    passes at sampled capacities like a skewed probe side — no k_hist — and its partitions, lists of ranges, are built into the
    LDS table piece by piece (general work items); a build partition that does not fit the table while the other side's is
    smaller is joined with the roles flipped (the reference's jp.cu:929-1003): the heavy hitter becomes many items instead of
    one workgroup looping over hundreds of table chunks.  Count, aggregate and the materialised (key, payR, payS) multiset —
    payloads in the right columns whichever side built — against the oracle; `both`: R is skewed too (lists on both sides);
    8+8 bits: 16 radix bits, where the sampling pass counts the partition ids in two slices (one workgroup's LDS holds 2^15 counters)."""
    import torch
    rng = np.random.default_rng(94)
    R = rng.permutation(nR).astype(np.int32)
    u = rng.random(nS)
    S = np.where(u < 0.40, R[3], np.where(u < 0.55, R[11], np.where(u < 0.60, R[12], R[rng.integers(0, nR, nS)]))).astype(np.int32)
    if both:
        R[rng.random(nR) < 0.30] = -5                    # a heavy hitter on the other side as well (a key S does not hold: no quadratic output)
        R[100:400] = R[11]                               # and duplicates of one of S's heavy keys: 300 x 15 % of S more matches
    Pr = np.arange(nR, dtype=np.int32)
    Ps = (np.arange(nS, dtype=np.int64) * 7 % 1000003 + (1 << 24)).astype(np.int32)      # disjoint from payR: a swapped column shows
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=True)
    dR, dPr, dS, dPs = (torch.from_numpy(x).cuda() for x in (R, Pr, S, Ps))
    with P.HashJoin(0) as hj:
        hj.configure(**cfg)
        hj.bind_device(P.REL_R, dR, dPr)
        hj.bind_device(P.REL_S, dS, dPs)
        c0 = hj.config()
        assert c0["build_side"] == 2
        sampled_ok = c0["bits2"] > 0 and c0["bits1"] + c0["bits2"] <= 17     # default bits follow the SMALLER relation (2^22: 9+1)
        for i in range(3):     # overflow -> sample -> sampled passes + general items; then twice from the remembered tables
            assert hj.join() == (em, eagg), i
            assert hj.partition_layout(P.REL_S) == ("sampled" if sampled_ok else "exact"), i
        hj.enable_timings(2)
        hj.timings_reset()
        assert hj.join() == (em, eagg)
        kt = hj.timings()
        hj.enable_timings(0)
        if sampled_ok:
            assert "k_hist" not in kt or kt["k_hist"]["launches"] == 0, kt   # no histogram after first contact, on either side
            assert hj.partition_layout(P.REL_R) == ("sampled" if both else "slotted")
        k, pr, ps = hj.join_materialize()
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        if em <= 3_000_000:
            for a, b in zip(sorted_triples(k, pr, ps), sorted_triples(*o.join_materialize(R, Pr, S, Ps))):
                assert np.array_equal(a, b)
        assert np.array_equal(R[pr], k) and int(ps.min()) >= (1 << 24)  # payR / payS in their own columns whichever side built
        hj.partition_both()
        k, pr, ps = hj.join_materialize(cap=em)                       # one probe on fresh partitions, no count before it
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        hj.configure(**dict(cfg, probe_chunk=3000))
        assert hj.join() == (em, eagg)
        k, pr, ps = hj.join_materialize(cap=em)
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        hj.configure(**cfg)
        assert hj.join() == (em, eagg)
        # introspection still hands out oracle-identical gap-free partitions of the build relation
        c = hj.config()
        gk, gp, goff = hj.partitions(P.REL_S, nS)
        ok, op, ooff = o.radix_partition(S, Ps, 0, c["bits1"] + c["bits2"])
        assert np.array_equal(goff, ooff) and np.array_equal(o.partition_digest(gk, gp, goff), o.partition_digest(ok, op, ooff))
        assert hj.join() == (em, eagg)


def test_late_materialize_with_a_skewed_build_side(P):
    """join_partitioned_varpayload semantics with the skewed relation building: the late-materialising kernel takes plain work items,
    so a sampled build side is redone with the exact passes for it."""
    import torch
    rng = np.random.default_rng(95)
    nR, nS = 1 << 16, 1 << 20
    R = rng.permutation(nR).astype(np.int32)
    S = np.where(rng.random(nS) < 0.5, R[5], R[rng.integers(0, nR, nS)]).astype(np.int32)
    Dr = rng.integers(-1000, 1000, (2, nR)).astype(np.int32)
    Ds = rng.integers(-1000, 1000, (3, nS)).astype(np.int32)
    idx = {int(k): i for i, k in enumerate(R)}
    rows = np.array([idx[int(k)] for k in S], dtype=np.int64)
    expect = (int(Dr[:, rows].astype(np.int64).sum()) + int(Ds.astype(np.int64).sum())) % (1 << 64)
    with P.HashJoin(0) as hj:
        hj.configure(bits1=5, bits2=4, build_side=2)
        hj.load_host(P.REL_R, R, payload="rowid")
        hj.load_host(P.REL_S, S, payload="rowid")
        assert hj.join()[0] == nS and hj.join()[0] == nS
        assert hj.partition_layout(P.REL_S) == "sampled"
        dDr, dDs = torch.from_numpy(Dr).cuda(), torch.from_numpy(Ds).cuda()
        hj.partition_both()
        m, s = hj.join_late_materialize(dDr, 2, nR, dDs, 3, nS)
        assert m == nS and s == expect


def test_stream_probe_segments_that_overflow_are_redone(P):
    """The count-only streaming loop never blocks the host: every segment's result and overflow flag are parked on the device
    and read once at the end; segments whose histogram-free slots overflowed (half of every segment is one key here)
    contributed nothing and are redone through the blocking path.  Uniform segments in between take the fast road."""
    rng = np.random.default_rng(93)
    nR, nS, seg = 1 << 16, 1 << 21, 1 << 19
    R = rng.permutation(nR).astype(np.int32)
    S = R[rng.integers(0, nR, nS)].astype(np.int32)
    S[: seg // 2] = R[5]                      # segment 0: skewed
    S[2 * seg: 2 * seg + seg // 2] = R[9]     # segment 2: skewed; segments 1 and 3: uniform
    Ps = np.arange(nS, dtype=np.int32)
    em, eagg, _ = o.join_count(R, np.arange(nR, dtype=np.int32), S, Ps, checksum=False)
    with P.HashJoin(0) as hj:
        hj.configure(bits1=5, bits2=4)
        hj.load_host(P.REL_R, R, np.arange(nR, dtype=np.int32))
        for _ in range(2):
            assert hj.join_stream_probe(S, Ps, segment_tuples=seg) == (em, eagg)
        assert hj.join_stream_probe(S, None, "rowid", segment_tuples=seg) == (em, eagg)


# ---- the heavy-hitter bypass (round 6): hj_join / hj_join_and_materialize join the probe side's hottest keys in pass 1 ----

def _zipf_fk(P, nR, nS, theta, seed):
    g = P.generator
    g.seed_generator(seed)
    R = g.create_relation_unique(None, nR, nR)
    S = g.create_relation_zipf(None, nS, nR, theta)
    return R, S


@pytest.mark.parametrize("theta,expect_mode", [(0.5, 0), (1.0, 1), (1.5, 1)])
@pytest.mark.parametrize("cfg", [dict(bits1=8, bits2=7), dict(bits1=9, bits2=6), dict(bits1=9, bits2=9)])
def test_heavy_hitter_bypass_zipf(P, theta, expect_mode, cfg):
    """PK-FK with Zipf foreign keys from the reference's generator stream.  theta 1.0 / 1.5: the top keys cover more than a tenth of S,
    pass 1 joins them itself (hot_stats mode 1 / 2) — count, aggregate (signed payload products mod 2^64) and the materialised multiset
    against the oracle; theta 0.5: the top 1024 keys cover ~6 %: the look at the keys says no, the plain sampled path runs.  9+9 bits:
    with the bypass the sampled path also takes 18 radix bits (two 512-way passes; without it such a relation goes to the exact passes).  Then the
    call sequences around it: a materialising probe after hj_join (the relation is partitioned again, whole), a count after
    hj_join_and_materialize, introspection, a capacity that is too small."""
    import torch
    nR, nS = 1 << 18, 3 << 20
    R, S = _zipf_fk(P, nR, nS, theta, 777)
    rng = np.random.default_rng(5)
    Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
    Ps = rng.integers(-2**31, 2**31 - 1, nS).astype(np.int32)
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=True)
    dR, dPr, dS, dPs = (torch.from_numpy(x).cuda() for x in (R, Pr, S, Ps))
    with P.HashJoin(0) as hj:
        hj.configure(**cfg)
        hj.bind_device(P.REL_R, dR, dPr)
        hj.bind_device(P.REL_S, dS, dPs)
        for i in range(3):       # overflow -> key sample + table -> bypass; then twice from what was learned
            assert hj.join() == (em, eagg), i
            st = hj.hot_stats()
            assert st["mode"] == expect_mode, (i, st)
            assert hj.partition_layout(P.REL_S) == ("sampled" if (expect_mode or cfg["bits1"] + cfg["bits2"] <= 17) else "exact")
        if expect_mode:
            assert st["keys"] > 100 and 0.1 < st["share"] < 1.0 and st["matches"] > 0.5 * st["share"] * nS, st
            assert st["matches"] < em
        # the count again on the same partitions (what pass 1 counted is still on the device)
        assert hj.join_count() == (em, eagg)
        # a materialising probe wants every tuple in a partition: S is partitioned again without the bypass
        k, pr, ps = hj.join_materialize(cap=em)
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        assert hj.hot_stats()["mode"] == 0
        # one call, the output columns known to pass 1: the hot keys' tuples are written from there
        k, pr, ps = hj.join_and_materialize(cap=em)
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        assert hj.hot_stats()["mode"] == (2 if expect_mode else 0)
        for a, b in zip(sorted_triples(k, pr, ps), sorted_triples(*o.join_materialize(R, Pr, S, Ps))):
            assert np.array_equal(a, b)
        # ... and a count on partitions whose hot tuples went to an output: partitioned again, whole
        assert hj.join_count() == (em, eagg)
        # too small a capacity: nothing beyond it is written, the true size comes back
        bufs = [torch.full((em // 2 + 64,), -7, dtype=torch.int32, device="cuda:0") for _ in range(3)]
        with pytest.raises(P.HJError) as ei:
            hj.join_and_materialize_into(bufs[0], bufs[1], bufs[2], em // 2)
        assert ei.value.code == P.ECAPACITY and str(em) in str(ei.value)
        assert all(bool((b[em // 2:] == -7).all()) for b in bufs)
        # introspection after a bypassing join: oracle-identical gap-free partitions
        assert hj.join() == (em, eagg)
        c = hj.config()
        gk, gp, goff = hj.partitions(P.REL_S, nS)
        ok, op, ooff = o.radix_partition(S, Ps, 0, c["bits1"] + c["bits2"])
        assert np.array_equal(goff, ooff) and np.array_equal(o.partition_digest(gk, gp, goff), o.partition_digest(ok, op, ooff))
        assert hj.join() == (em, eagg) and hj.hot_stats()["mode"] == expect_mode


@pytest.mark.parametrize("skewed_rel", [1, 0])
def test_heavy_hitter_bypass_only_takes_keys_the_other_side_holds_once(P, skewed_rel):
    """Four hot keys in the skewed relation: A and D are unique in the other relation (bypassed), B is held three times there (its
    tuples take the ordinary path and match three times each), C not at all (ordinary path, no match).  Both bindings of the skewed
    relation (as S and as R: the payload columns must not swap).  Then the other relation changes under its binding so that A is no
    longer unique: the slots sized without A's tuples overflow, the call falls back and still answers right."""
    import torch
    rng = np.random.default_rng(96)
    nB, nP = 1 << 18, 3 << 20
    Bk = rng.permutation(nB).astype(np.int32) + 10            # the PK side: keys 10 .. nB+9
    A, Bdup, D = int(Bk[3]), int(Bk[11]), int(Bk[12])
    C = -12345                                                 # a key the PK side does not hold
    Bk[100:102] = Bdup                                         # B three times
    u = rng.random(nP)
    Pk = np.where(u < 0.30, A, np.where(u < 0.45, Bdup, np.where(u < 0.55, C, np.where(u < 0.62, D, Bk[rng.integers(200, nB, nP)])))).astype(np.int32)
    Bp = np.arange(nB, dtype=np.int32)
    Pp = np.arange(nP, dtype=np.int32) + (1 << 24)             # disjoint from the other column: a swapped payload column shows
    rels = {1 - skewed_rel: (Bk, Bp), skewed_rel: (Pk, Pp)}
    (Rk, Rp), (Sk, Sp) = rels[0], rels[1]
    em, eagg, echk = o.join_count(Rk, Rp, Sk, Sp, checksum=True)
    dev = {r: tuple(torch.from_numpy(x).cuda() for x in rels[r]) for r in (0, 1)}
    with P.HashJoin(0) as hj:
        hj.configure(bits1=8, bits2=7)
        hj.bind_device(P.REL_R, *dev[0])
        hj.bind_device(P.REL_S, *dev[1])
        for i in range(3):
            assert hj.join() == (em, eagg), i
        st = hj.hot_stats()
        assert st["mode"] == 1 and st["keys"] == 2 and 0.30 < st["share"] < 0.45, st     # A and D only
        assert abs(st["matches"] - int(((Pk == A) | (Pk == D)).sum())) == 0, st
        k, pr, ps = hj.join_and_materialize(cap=em)
        assert hj.hot_stats()["mode"] == 2
        assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
        for a, b in zip(sorted_triples(k, pr, ps), sorted_triples(*o.join_materialize(Rk, Rp, Sk, Sp))):
            assert np.array_equal(a, b)
        lo, hi = (pr, ps) if skewed_rel == 1 else (ps, pr)
        assert int(lo.max()) < (1 << 24) <= int(hi.min())      # payR / payS in their own columns
        # the PK side changes under its binding: A twice now.  A's tuples are no longer bypassed and come back into slots sized
        # without them: overflow -> the plain sampled path (planned from the data as it is now)
        Bk2 = Bk.copy()
        Bk2[150] = A
        dev[1 - skewed_rel][0].copy_(torch.from_numpy(Bk2).cuda())
        rels2 = {1 - skewed_rel: (Bk2, Bp), skewed_rel: (Pk, Pp)}
        em2, eagg2, echk2 = o.join_count(rels2[0][0], rels2[0][1], rels2[1][0], rels2[1][1], checksum=True)
        assert em2 > em
        for i in range(2):
            assert hj.join() == (em2, eagg2), i
        k, pr, ps = hj.join_and_materialize(cap=em2)
        assert len(k) == em2 and o.triples_checksum(k, pr, ps) == echk2


def test_heavy_hitter_bypass_under_a_captured_step(P):
    """hj_config.graph: the step with the bypass (memset, k_hot_build, the bypassing pass 1) replays from a hipGraph."""
    import torch
    nR, nS = 1 << 18, 3 << 20
    R, S = _zipf_fk(P, nR, nS, 1.0, 4242)
    Pr, Ps = np.arange(nR, dtype=np.int32), (np.arange(nS, dtype=np.int64) * 7 % 1000003).astype(np.int32)
    em, eagg, _ = o.join_count(R, Pr, S, Ps, checksum=False)
    dR, dPr, dS, dPs = (torch.from_numpy(x).cuda() for x in (R, Pr, S, Ps))
    with P.HashJoin(0) as hj:
        hj.configure(bits1=8, bits2=7, graph=1)
        hj.bind_device(P.REL_R, dR, dPr)
        hj.bind_device(P.REL_S, dS, dPs)
        for i in range(5):
            assert hj.join() == (em, eagg), i
        assert hj.hot_stats()["mode"] == 1


# ---- the look before the first optimistic attempt (round 6) ----

@pytest.mark.parametrize("kind", ["zipf", "uniform", "sorted", "low_bits_constant"])
def test_look_before_the_first_attempt(P, monkeypatch, kind):
    """A relation nothing is known about gets 2^16 of its keys counted by pass-1 and pass-2 digit before its first histogram-free attempt
    (relations of >= 2^26 tuples by default; HJ_SKEW_PROBE lowers the bar here).  Heavy skew — a Zipf(1.0) probe side, keys whose low bits never
    change — goes to the sampled path at once: no failed attempt in the first call's breakdown; uniform and sorted keys stay on the plain passes
    (the sample positions are hashed: a fixed stride over sorted keys would put every sample into one digit).  Results against the oracle
    either way, and the old sequence (attempt, flag, redo) with the look switched off."""
    import torch
    nR, nS = 1 << 18, 3 << 20
    rng = np.random.default_rng(11)
    if kind == "zipf":
        R, S = _zipf_fk(P, nR, nS, 1.0, 778)
    elif kind == "uniform":
        R = rng.permutation(nR).astype(np.int32); S = rng.integers(0, nR, nS).astype(np.int32)
    elif kind == "sorted":
        R = np.arange(nR, dtype=np.int32); S = np.arange(nS, dtype=np.int32)
    else:
        nR = 1 << 20   # (large enough to be looked at itself: its pass-2 digit is as degenerate as S's)
        R = (rng.permutation(nR) * 128).astype(np.int32); S = (rng.integers(0, nR, nS) * 128).astype(np.int32)
    Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
    Ps = rng.integers(-2**31, 2**31 - 1, nS).astype(np.int32)
    em, eagg, echk = o.join_count(R, Pr, S, Ps, checksum=True)
    dR, dPr, dS, dPs = (torch.from_numpy(x).cuda() for x in (R, Pr, S, Ps))
    skewed = kind in ("zipf", "low_bits_constant")
    for probe in ("20", "0"):
        monkeypatch.setenv("HJ_SKEW_PROBE", probe)
        with P.HashJoin(0) as hj:
            hj.configure(bits1=8, bits2=7)
            hj.bind_device(P.REL_R, dR, dPr)
            hj.bind_device(P.REL_S, dS, dPs)
            assert hj.join() == (em, eagg)
            bd = hj.last_call_breakdown()
            lay = hj.partition_layout(P.REL_S)
            if probe == "20":
                assert bd["failed_optimistic_attempt_ms"] == 0, bd
                assert (lay != "slotted") == skewed, lay
                if kind == "zipf":
                    assert lay == "sampled" and hj.hot_stats()["mode"] == 1
            elif skewed:
                assert bd["failed_optimistic_attempt_ms"] > 0 and lay != "slotted", (bd, lay)
            for _ in range(2):
                assert hj.join() == (em, eagg)
            k, pr, ps = hj.join_and_materialize(cap=em)
            assert len(k) == em and o.triples_checksum(k, pr, ps) == echk
