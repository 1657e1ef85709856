"""Worker for tests/test_dist.py: runs ShardedJoin over gloo (CPU) with a stand-in engine whose
shard_split / join are numpy + the oracle, so the exchange logic of the multi-GPU path is exercised
end to end at world_size 2 without a GPU.  With HJ_DIST_GPU=1 the engine is the real HashJoin."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
from oracle import pyoracle as o  # noqa: E402


class OracleEngine:
    """Same interface as HashJoin for the three calls ShardedJoin makes (CPU tensors)."""

    def __init__(self, pkg):
        self.pkg = pkg
        self.rel = {}

    def shard_split(self, keys, pays, n, nshards, out_keys, out_pays):
        k, p = keys[:n].numpy(), pays[:n].numpy()
        owner = np.array([self.pkg.shard_of(int(x), nshards) for x in k], dtype=np.int64) if n else np.empty(0, np.int64)
        order = np.argsort(owner, kind="stable")
        out_keys[:n] = torch.from_numpy(k[order])
        out_pays[:n] = torch.from_numpy(p[order])
        return [int((owner == s).sum()) for s in range(nshards)]

    def shard_count(self, keys, n, nshards):
        k = keys[:n].numpy()
        owner = np.array([self.pkg.shard_of(int(x), nshards) for x in k], dtype=np.int64) if n else np.empty(0, np.int64)
        return np.bincount(owner, minlength=nshards).tolist()

    def shard_split_ordered(self, keys, pays, n, nshards, position, out_keys, out_pays):
        k, p = keys[:n].numpy(), pays[:n].numpy()
        pos = np.asarray(position, dtype=np.int64)
        where = pos[np.array([self.pkg.shard_of(int(x), nshards) for x in k], dtype=np.int64)] if n else np.empty(0, np.int64)
        order = np.argsort(where, kind="stable")
        out_keys[:n] = torch.from_numpy(k[order])
        out_pays[:n] = torch.from_numpy(p[order])
        return np.bincount(where, minlength=nshards).tolist()

    def bind_device(self, rel, keys, pays, n):
        self.rel[rel] = (keys, pays, n)      # like the real engine: the data is read at partition time

    def _snap(self, rel):
        k, p, n = self.rel[rel]
        return k[:n].numpy().copy(), p[:n].numpy().copy()

    def digest_pairs(self, keys, pays, n):
        k, p = keys[:n].numpy(), pays[:n].numpy()
        return int(o.partition_digest(k, p, np.array([0, n], np.uint64))[0]) if n else 0

    def partition(self, rel):
        self.rel[("part", rel)] = self._snap(rel)   # by now the exchange of this relation must be complete

    def join_count(self):
        (rk, rp), (sk, sp) = self.rel[("part", 0)], self.rel[("part", 1)]
        m, agg, _ = o.join_count(rk, rp, sk, sp, checksum=False)
        return m, agg

    def join_materialize(self):
        (rk, rp), (sk, sp) = self.rel[("part", 0)], self.rel[("part", 1)]
        return o.join_materialize(rk, rp, sk, sp)


def main():
    gpu = os.environ.get("HJ_DIST_GPU") == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    pkg = graft.load_package()
    if gpu:
        one = os.environ.get("HJ_DIST_ONE_GPU") == "1"   # every rank on cuda:0, host-staged exchange over gloo
        idx = 0 if one else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(idx)
        dev = torch.device("cuda", idx)
        if one:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        engine = pkg.HashJoin(dev.index, stream=torch.cuda.current_stream().cuda_stream)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo")
        engine = OracleEngine(pkg)
    from importlib import import_module
    dj = import_module(pkg.__name__ + ".dist").ShardedJoin(engine, pkg, dev, balance=os.environ.get("HJ_DIST_BALANCE", "hash"))
    dj.force_exchange = os.environ.get("HJ_DIST_FORCE_EXCHANGE") == "1"
    if "HJ_DIST_CHUNK" in os.environ:       # force several point-to-point chunks per peer
        dj.CHUNK = int(os.environ["HJ_DIST_CHUNK"])

    if os.environ.get("HJ_DIST_BIG") and gpu:
        # full-size exchange: 2^30 tuples per relation per rank generated on the device (at world size 2 a peer's share of
        # a column is 2^29 elements = 2 GiB, the size at which a single RCCL message was seen corrupted: the chunked
        # point-to-point path must carry it).  Keys = two permutations of [0, world * 2^30): closed-form count.
        n = 1 << int(os.environ["HJ_DIST_BIG"])
        cols = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4)]
        engine.gen_unique(cols[0], n, rank * n, world * n, 1)
        engine.gen_unique(cols[2], n, rank * n, world * n, 2)
        engine.fill_payload(cols[1], n, "ones")
        engine.fill_payload(cols[3], n, "ones")
        engine.sync()
        res = [dj.join(cols[0], cols[1], cols[2], cols[3], verify=True) for _ in range(2)]
        if rank == 0:
            print("RESULT " + json.dumps({"got": res, "expect": [[world * n, world * n]] * 2}))
        dist.destroy_process_group()
        return

    # every rank generates the same global relations and keeps its own slice
    rng = np.random.default_rng(123)
    nR, nS = 20_000, 50_001
    if "HJ_DIST_N" in os.environ:           # GPU runs: large enough for several chunks per peer
        nR, nS = (int(x) for x in os.environ["HJ_DIST_N"].split(","))
    R = rng.integers(-5000, 5000, nR).astype(np.int32)
    S = rng.integers(-5000, 5000, nS).astype(np.int32)
    if os.environ.get("HJ_DIST_SKEW"):      # one key holds 30 % of S: the heavy hitter's shard must not drag its GPU down
        S[rng.random(nS) < 0.3] = 1234
    Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
    Ps = np.arange(nS, dtype=np.int32)

    def sl(a, n):
        lo, hi = n * rank // world, n * (rank + 1) // world
        return torch.from_numpy(a[lo:hi].copy()).to(dev)

    if os.environ.get("HJ_DIST_MATERIALIZE"):
        # the sharded materialising join: every rank keeps the tuples of the partitions it owns; the union is the oracle's multiset
        # (checked through the order-independent checksum, a sum of per-tuple mixes mod 2^64, and the sizes), a rank only holds its shard
        gm, (k, pr, ps), sizes = dj.join_materialize(sl(R, nR), sl(Pr, nR), sl(S, nS), sl(Ps, nS))
        own_ok = all(pkg.shard_of(int(x), world) == rank for x in k[:200]) if dj.balance == "hash" else True
        chk = dj._allreduce_u64([o.triples_checksum(k, pr, ps) if len(k) else 0, int(own_ok)])
        if rank == 0:
            em, _, echk = o.join_count(R, Pr, S, Ps, checksum=True)
            print("RESULT " + json.dumps({"got": [gm, sum(sizes), chk[0], chk[1]], "expect": [em, em, echk, world], "sizes": sizes}))
        dist.destroy_process_group()
        return

    res = []
    for i in range(2):  # twice: buffers are reused across steps; the first with the exchange check on
        res.append(dj.join(sl(R, nR), sl(Pr, nR), sl(S, nS), sl(Ps, nS), verify=(i == 0)))
    mine = torch.tensor(list(dj.last_received), dtype=torch.int64, device=dj.cdev if hasattr(dj, "cdev") else dev)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    recv = [[int(x) for x in t.tolist()] for t in allr]
    # an empty local slice on one rank must work too
    e = torch.empty(0, dtype=torch.int32, device=dev)
    if rank == 0:
        res.append(dj.join(e, e, sl(S, nS), sl(Ps, nS)))
    else:
        res.append(dj.join(sl(R, nR), sl(Pr, nR), sl(S, nS), sl(Ps, nS)))
    if rank == 0:
        em, eagg, _ = o.join_count(R, Pr, S, Ps, checksum=False)
        lo = nR // world
        em3, eagg3, _ = o.join_count(R[lo:], Pr[lo:], S, Ps, checksum=False)
        print("RESULT " + json.dumps({"got": res, "expect": [[em, eagg], [em, eagg], [em3, eagg3]], "received": recv}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
