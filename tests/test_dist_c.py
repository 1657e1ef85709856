"""The multi-GPU join behind the C ABI (include/hj_dist.h, csrc/hj_dist.hip): level-0 split -> sliced fixed-size exchange ->
local passes + join -> all-reduce, C++ host code throughout.  On a one-GPU box the ranks of one process share cuda:0 and the
exchange runs over the in-process device-copy transport (RCCL refuses duplicate GPUs): the slicing, the segment tables, the
flag gathering and the exact fallback are the code that runs over RCCL on a multi-GPU node; RCCL itself is exercised at
world size 1 (communicator, grouped send/recv of a rank's own share, all-gather, all-reduce) and, where two GPUs are
visible, at world size 2."""
from importlib import import_module

import numpy as np
import pytest

from hjtest import pkg
from oracle import pyoracle as o

pytestmark = pytest.mark.gpu


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _inputs(nR, nS, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "unique":            # PK-FK: R unique, S draws from R's keys
        R = rng.permutation(max(nR, 1) * 3)[:nR].astype(np.int32)
        S = R[rng.integers(0, max(nR, 1), nS)].astype(np.int32) if nR else rng.integers(0, 100, nS).astype(np.int32)
    elif kind == "dups":            # duplicates on both sides, negative keys
        R = rng.integers(-5000, 5000, nR).astype(np.int32)
        S = rng.integers(-5000, 5000, nS).astype(np.int32)
    else:                           # one key holds 40 % of S: its owner's slots overflow -> every rank goes exact
        R = rng.permutation(nR * 2)[:nR].astype(np.int32)
        S = R[rng.integers(0, nR, nS)].astype(np.int32)
        S[rng.random(nS) < 0.4] = R[0]
    return R, S


def _run(devices, R, S, cuts=None, dist_cfg=None, ctx_cfg=None, payload="rowid"):
    """Rank r gets a contiguous cut of R and of S (cuts: fractions, default even); returns (matches, agg, stats per rank)."""
    import torch
    P = pkg()
    D = import_module(P.__name__ + ".dist")
    G = len(devices)
    cuts = cuts or [(i + 1) / G for i in range(G)]
    Pr = np.arange(len(R), dtype=np.int32) if payload == "rowid" else np.ones(len(R), np.int32)
    Ps = (np.arange(len(S), dtype=np.int32) * 3 - 7).astype(np.int32) if payload == "rowid" else np.ones(len(S), np.int32)
    with D.GroupJoin(devices) as g:
        if dist_cfg:
            g.configure(**dist_cfg)
        keep = []
        for r in range(G):
            if ctx_cfg:
                g.context(r).configure(**ctx_cfg)
            lo = [int(round((cuts[r - 1] if r else 0) * len(X))) for X in (R, S)]
            hi = [int(round(cuts[r] * len(X))) for X in (R, S)]
            cols = []
            for X, Px, a, b in ((R, Pr, lo[0], hi[0]), (S, Ps, lo[1], hi[1])):
                dev = torch.device("cuda", devices[r])
                cols += [torch.from_numpy(np.ascontiguousarray(X[a:b])).to(dev), torch.from_numpy(np.ascontiguousarray(Px[a:b])).to(dev)]
            keep.append(cols)
            g.bind(r, P.REL_R, cols[0], cols[1])
            g.bind(r, P.REL_S, cols[2], cols[3])
        out = [g.join() for _ in range(2)]          # twice: buffers, events and learned state are reused
        stats = [g.stats(r) for r in range(G)]
        transport = g.transport
    assert out[0] == out[1]
    em, eagg, _ = o.join_count(R, Pr, S, Ps, checksum=False)
    assert out[0] == (em, eagg), (out[0], (em, eagg), stats)
    assert sum(s["received"][0] for s in stats) == len(R) and sum(s["received"][1] for s in stats) == len(S), stats
    return stats, transport


@pytest.mark.parametrize("world", [2, 3])
def test_sliced_exchange_ranks_share_one_gpu(world):
    """World sizes 2 and 3 (not a power of two) on cuda:0: the sliced fixed-size path (two-pass radix bits forced so that
    it applies at test sizes), 1 to 5 slices, uneven local sizes, an empty rank."""
    R, S = _inputs(300_000, 700_001, 5, "unique")
    for slices in (1, 3, 5):
        stats, transport = _run([0] * world, R, S, dist_cfg=dict(slices=slices), ctx_cfg=dict(bits1=5, bits2=4))
        assert transport == "device-copy" and all(s["path"] == "sliced" for s in stats), stats
    cuts = [0.5, 0.5, 1.0][:world] if world == 3 else [0.2, 1.0]       # rank 1 of 3 holds nothing; 20/80 at world 2
    stats, _ = _run([0] * world, R, S, cuts=cuts, dist_cfg=dict(slices=4), ctx_cfg=dict(bits1=5, bits2=4))
    assert all(s["path"] == "sliced" for s in stats)
    R, S = _inputs(200_000, 200_000, 6, "dups")
    stats, _ = _run([0] * world, R, S, dist_cfg=dict(slices=2), ctx_cfg=dict(bits1=4, bits2=3))
    assert all(s["path"] == "sliced" for s in stats)


@pytest.mark.parametrize("world", [2, 3])
def test_skew_sends_every_rank_to_the_exact_path(world):
    """One key holds 40 % of S: its owner's slots overflow somewhere; the flags are summed with the result, every rank
    repeats the join on the exact path (same answer), and the next call on the same columns goes there directly."""
    R, S = _inputs(100_000, 600_000, 7, "heavy")
    stats, _ = _run([0] * world, R, S, dist_cfg=dict(slices=3), ctx_cfg=dict(bits1=5, bits2=4))
    assert all(s["path"] == "exact" for s in stats), stats
    stats, _ = _run([0] * world, R, S, dist_cfg=dict(exact_only=True))
    assert all(s["path"] == "exact" for s in stats)
    # ... where the shards are dealt to the GPUs by size once skew has been seen (or on request): the heavy key's shard no longer
    # shares a GPU with an average load.  Received tuples per rank, hash sharding vs size-aware:
    tot = lambda st: [a + b for a, b in (s["received"] for s in st)]
    imb = lambda st: max(tot(st)) / (sum(tot(st)) / len(st))
    plain, _ = _run([0] * world, R, S, dist_cfg=dict(exact_only=True))
    sized, _ = _run([0] * world, R, S, dist_cfg=dict(exact_only=True, balance_size=True))
    assert all(s["balanced"] for s in sized) and not any(s["balanced"] for s in plain)
    assert imb(sized) < imb(plain) and imb(sized) < 1.12, (tot(plain), tot(sized))


def test_small_relations_take_the_exact_path():
    """Single-pass sizes have no histogram-free pass to slice into: exact-count exchange, default radix bits."""
    R, S = _inputs(5_000, 12_345, 8, "unique")
    stats, _ = _run([0, 0], R, S)
    assert all(s["path"] == "exact" for s in stats)
    stats, _ = _run([0, 0], np.empty(0, np.int32), S)                   # an empty relation
    assert all(s["path"] == "exact" for s in stats)


def test_rccl_world1_through_the_c_abi():
    """RCCL itself, as far as one GPU goes: communicator from ncclCommInitAll, the rank's own share sent to itself through
    the grouped ncclSend/ncclRecv of every slice (self_via_link), all-gather of sizes, all-reduce of the result; both paths."""
    R, S = _inputs(400_000, 900_000, 9, "unique")
    stats, transport = _run([0], R, S, dist_cfg=dict(slices=3, self_via_link=True), ctx_cfg=dict(bits1=5, bits2=4))
    assert transport == "rccl" and stats[0]["path"] == "sliced"
    stats, _ = _run([0], R, S, dist_cfg=dict(slices=3), ctx_cfg=dict(bits1=5, bits2=4))
    assert stats[0]["path"] == "sliced"
    stats, _ = _run([0], R, S, dist_cfg=dict(exact_only=True, self_via_link=True))
    assert stats[0]["path"] == "exact"


def test_default_geometry_at_2p24():
    """Default radix bits and slice count at a size where the sliced path applies by itself (2^24 per rank, 2 ranks)."""
    import torch
    P = pkg()
    D = import_module(P.__name__ + ".dist")
    n = 1 << 24
    with D.GroupJoin([0, 0]) as g:
        keep = []
        for r in range(2):
            hj = g.context(r)
            cols = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4)]
            hj.gen_unique(cols[0], n, r * n, 2 * n, 1)
            hj.gen_unique(cols[2], n, r * n, 2 * n, 2)
            hj.fill_payload(cols[1], n, "ones")
            hj.fill_payload(cols[3], n, "ones")
            hj.sync()
            keep.append(cols)
            g.bind(r, P.REL_R, cols[0], cols[1])
            g.bind(r, P.REL_S, cols[2], cols[3])
        assert g.join() == (2 * n, 2 * n)
        st = [g.stats(r) for r in range(2)]
    assert all(s["path"] == "sliced" for s in st) and sum(s["received"][0] for s in st) == 2 * n, st


def _bind_group(g, P, world, R, S, misalign_rank=None):
    """Even cuts of R and S over the ranks of a group on cuda:0; misalign_rank: that rank's S key column starts 4 bytes off a
    16-byte boundary (hj_dist_rank_join refuses it before its first collective)."""
    import torch
    keep = []
    for r in range(world):
        cols = []
        for X in (R, S):
            a, b = len(X) * r // world, len(X) * (r + 1) // world
            k = torch.from_numpy(np.ascontiguousarray(X[a:b])).cuda()
            cols += [k, torch.ones_like(k)]
        if r == misalign_rank:
            pad = torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), cols[2]])
            cols[2] = pad[1:]
        keep.append(cols)
        g.bind(r, P.REL_R, cols[0], cols[1], n=int(cols[0].numel()))
        g.bind(r, P.REL_S, cols[2], cols[3], n=int(cols[2].numel()))
    return keep


@pytest.mark.parametrize("world,stall", [(2, 1), (3, 0)])
def test_a_stalled_peer_hits_the_deadline_instead_of_hanging(world, stall):
    """no wait of the multi-GPU path blocks for ever.  One rank stops taking part in the exchange of S's
    first slice for 2.5 deadlines (hj_dist_config.test_stall_rank); the others give up at the deadline with a message that says
    who waited for whom and where, the stalled rank finds the group aborted when it comes back, hj_dist_join returns an error —
    and the group refuses further joins (a communicator with a collective that was given up is not reused)."""
    import time
    P = pkg()
    D = import_module(P.__name__ + ".dist")
    R, S = _inputs(200_000, 400_000, 21, "unique")
    with D.GroupJoin([0] * world) as g:
        for r in range(world):
            g.context(r).configure(bits1=5, bits2=4)
        keep = _bind_group(g, P, world, R, S)
        g.configure(slices=3, timeout_ms=1000)
        assert g.join()[0] == len(S)                       # healthy first: the deadline does not fire on a working group
        g.configure(slices=3, timeout_ms=1000, test_stall_rank=stall + 1)
        t0 = time.time()
        with pytest.raises(P.HJError) as ei:
            g.join()
        dt = time.time() - t0
        msg = str(ei.value)
        assert "deadline" in msg and "rank" in msg and ("waited for rank(s) %d" % stall) in msg, msg
        assert 0.9 < dt < 8.0, dt                          # the deadline (1 s) and the stalled rank's return (2.7 s), not minutes
        g.configure(slices=3, timeout_ms=1000)
        with pytest.raises(P.HJError) as ei2:
            g.join()
        assert "aborted" in str(ei2.value)
        del keep
    with D.GroupJoin([0] * world) as g:                    # a fresh group on the same device works
        for r in range(world):
            g.context(r).configure(bits1=5, bits2=4)
        keep = _bind_group(g, P, world, R, S)
        assert g.join()[0] == len(S)


def test_a_failing_rank_takes_every_rank_out_of_the_join():
    """a rank that fails locally must not leave its peers inside a collective.  Rank 1's S keys are not 16-byte aligned:
    it fails before its first collective; ranks 0 and 2 are waiting in the all-gather of the sizes and leave at once — long before
    the 60-s deadline — with an error that names rank 1."""
    import time
    P = pkg()
    D = import_module(P.__name__ + ".dist")
    R, S = _inputs(100_000, 300_000, 22, "unique")
    with D.GroupJoin([0, 0, 0]) as g:
        keep = _bind_group(g, P, 3, R, S, misalign_rank=1)
        g.configure(timeout_ms=60_000)
        t0 = time.time()
        with pytest.raises(P.HJError) as ei:
            g.join()
        assert time.time() - t0 < 20.0
        assert "rank 1" in str(ei.value) and "aligned" in str(ei.value), str(ei.value)
        del keep


def test_transport_is_selectable():
    """hj_dist_create_transport: device copies on request (what a multi-GPU node A/Bs against RCCL's kernels), RCCL refused for
    ranks that share a device, unknown names refused; world 1 runs over either."""
    P = pkg()
    D = import_module(P.__name__ + ".dist")
    R, S = _inputs(150_000, 350_000, 23, "unique")
    for transport, devices, expect in (("copy", [0], "device-copy"), ("rccl", [0], "rccl"), ("device-copy", [0, 0], "device-copy"), (None, [0, 0], "device-copy")):
        with D.GroupJoin(devices, transport=transport) as g:
            assert g.transport == expect
            for r in range(len(devices)):
                g.context(r).configure(bits1=5, bits2=4)
            keep = _bind_group(g, P, len(devices), R, S)
            g.configure(slices=2, self_via_link=True)
            assert g.join()[0] == len(S)
            st = g.stats(0)
            assert st["path"] == "sliced" and st["exchange_ms"] > 0
            # tuples among the link bytes: what the rank sent to others (nothing at world 1)
            if len(devices) == 1:
                assert st["payload_bytes"] == 0
            else:
                assert 0 < st["payload_bytes"] <= st["link_bytes"], st
            del keep
    for bad in (("rccl", [0, 0]), ("smoke-signals", [0])):
        with pytest.raises(P.HJError):
            D.GroupJoin(bad[1], transport=bad[0])


def test_more_ranks_than_gpus_is_refused():
    P = pkg()
    D = import_module(P.__name__ + ".dist")
    with pytest.raises(P.HJError):
        D.GroupJoin(list(range(_gpu_count() + 1)))   # rank r on device r: one device too many


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs on one node")
def test_rccl_world2_through_the_c_abi():
    """Two GPUs: the same calls over real links (skipped on the one-GPU boxes of this pool)."""
    R, S = _inputs(3_000_000, 7_000_001, 10, "unique")
    stats, transport = _run([0, 1], R, S, dist_cfg=dict(slices=4), ctx_cfg=dict(bits1=7, bits2=5), payload="ones")
    assert transport == "rccl" and all(s["path"] == "sliced" for s in stats)
    R, S = _inputs(200_000, 900_000, 11, "heavy")
    stats, _ = _run([0, 1], R, S, ctx_cfg=dict(bits1=5, bits2=4))
    assert all(s["path"] == "exact" for s in stats)
