"""Multi-GPU join: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-GPU; its structural analogue is the co-processing path, which radix-splits
both relations 16 ways on the host and joins each level-0 partition independently
(src/hash_join_clustered_probe.cu:1256-1266, 1503-1618).  Here the level-0 split runs on each GPU
(hj_shard_split: hash of the key → owner GPU), the shards are exchanged with ONE all-to-all-v per
column — xGMI is point-to-point, every ordered GPU pair has its own link, so nothing is relayed —
and each GPU then runs the unchanged local path (partition + build/probe) on what it received.
Only the 64-bit match count / aggregate is all-reduced.

`engine` is a HashJoin (HIP).  Tests drive the same exchange logic over gloo with a stand-in engine.
"""
import torch
import torch.distributed as dist


class ShardedJoin:
    VIRTUAL = 8   # balance="size": virtual shards per GPU

    def __init__(self, engine, pkg, device, group=None, balance="hash"):
        """balance="hash": GPU g owns hash shard g (uniform keys).  balance="size": 8 virtual shards per GPU, assigned to
        GPUs by their global size (longest-processing-time first over |R|+|S| per shard) so that a heavy hitter's shard
        shares its GPU with few others — the reference's size-aware placement idea (partition-primitives.cu:307-468)."""
        assert balance in ("hash", "size")
        self.balance = balance
        self.e = engine
        self.pkg = pkg
        self.dev = device
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # a second communicator for the tiny control messages (split sizes, final all-reduce): its
        # collectives are not queued behind the multi-GiB column exchanges of the data communicator
        self.ctl = dist.new_group(ranks=list(range(self.world))) if group is None else group
        self._buf = {}
        self.last_received = (0, 0)
        self.force_exchange = False   # tests: run split + exchange even at world size 1
        # A transport that cannot move device memory (gloo): columns are staged through host memory.  Slow, but it lets
        # the real engine run at world size > 1 where RCCL is not available (two ranks sharing one GPU in the tests).
        self.staged = device.type == "cuda" and dist.get_backend(group) == "gloo"
        self.cdev = torch.device("cpu") if self.staged else device   # where the small control tensors live

    def _get(self, name, n):
        """Reusable int32 column of at least n elements (HBM is plentiful: keep, do not re-allocate)."""
        # positions inside the partition kernels are 32-bit: a receive side that skew has grown beyond that is refused here, with a
        # message, instead of failing later inside hj_partition
        if n >= (1 << 32) - (1 << 20):
            raise RuntimeError("rank %d would hold %d tuples of one relation (limit 2^32 - 2^20 per GPU): use more GPUs or balance='size'"
                               % (self.rank, n))
        t = self._buf.get(name)
        if t is None or t.numel() < n:
            t = torch.empty(max(int(n * 1.02) + 1024, 1024), dtype=torch.int32, device=self.dev)
            self._buf[name] = t
        return t

    # RCCL (2.26, ROCm 7) returns wrong data for a single all-to-all message of 2 GiB or more (measured:
    # 2^28 int32 elements fine, 2^29 corrupted — tools/experiments/rccl_2gib_repro.py), and at 2 GPUs a peer's share of a 2^30-
    # tuple column is exactly 2^29 elements.  Columns therefore travel as grouped point-to-point
    # sends/receives of at most CHUNK elements (512 MiB): one ncclGroup per column = one all-to-all-v.
    CHUNK = 1 << 27

    def exchange_async(self, cols, send_counts):
        """All-to-all-v of several columns that share one split: cols = {name: tensor[sum(send_counts)]}.
        Returns ({name: tensor}, n_received, [work handles]); the columns are valid after work.wait()."""
        sc = torch.tensor(send_counts, dtype=torch.int64, device=self.cdev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.ctl)
        recv_counts = [int(x) for x in rc.tolist()]
        total = sum(recv_counts)
        soff = [0]
        for c in send_counts:
            soff.append(soff[-1] + int(c))
        roff = [0]
        for c in recv_counts:
            roff.append(roff[-1] + c)
        out, works = {}, []
        for name, t in cols.items():
            r = self._get("recv_" + name, total)
            if self.staged:
                torch.cuda.current_stream(self.dev).synchronize()
                th, rh = t[:soff[-1]].cpu(), torch.empty(total, dtype=torch.int32)
                hops = []
                for step in range(self.world):
                    p, q = (self.rank + step) % self.world, (self.rank - step) % self.world
                    if step == 0:
                        rh[roff[p]:roff[p] + int(send_counts[p])] = th[soff[p]:soff[p + 1]]
                        continue
                    if int(send_counts[p]):
                        hops.append(dist.P2POp(dist.isend, th[soff[p]:soff[p + 1]], p, group=self.group))
                    if recv_counts[q]:
                        hops.append(dist.P2POp(dist.irecv, rh[roff[q]:roff[q + 1]], q, group=self.group))
                for wk in (dist.batch_isend_irecv(hops) if hops else []):
                    wk.wait()
                r[:total].copy_(rh)
                out[name] = r
                continue
            ops = []
            for step in range(self.world):
                p = (self.rank + step) % self.world      # rank-staggered peer order
                q = (self.rank - step) % self.world
                if step == 0:                             # own share: a local copy, no link involved
                    n = int(send_counts[p])
                    if n:
                        r[roff[p]:roff[p] + n].copy_(t[soff[p]:soff[p] + n], non_blocking=True)
                    continue
                for c0 in range(0, int(send_counts[p]), self.CHUNK):
                    c1 = min(c0 + self.CHUNK, int(send_counts[p]))
                    ops.append(dist.P2POp(dist.isend, t[soff[p] + c0:soff[p] + c1], p, group=self.group))
                for c0 in range(0, recv_counts[q], self.CHUNK):
                    c1 = min(c0 + self.CHUNK, recv_counts[q])
                    ops.append(dist.P2POp(dist.irecv, r[roff[q] + c0:roff[q] + c1], q, group=self.group))
            if ops:
                works.extend(dist.batch_isend_irecv(ops))
            out[name] = r
        return out, total, works

    def exchange(self, cols, send_counts):
        out, total, works = self.exchange_async(cols, send_counts)
        for w in works:
            w.wait()
        return out, total

    @staticmethod
    def assign_by_size(sizes, world):
        """Longest-processing-time-first assignment of shards to GPUs; deterministic (ties by index) so that every
        rank computes the same map.  Returns (owner[v], position[v]) with shards ordered by (owner, v)."""
        load = [0] * world
        owner = [0] * len(sizes)
        for v in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
            g = min(range(world), key=lambda j: (load[j], j))
            owner[v] = g
            load[g] += sizes[v]
        order = sorted(range(len(sizes)), key=lambda v: (owner[v], v))
        position = [0] * len(sizes)
        for pos, v in enumerate(order):
            position[v] = pos
        return owner, position

    def _split(self, k, p, n, out_k, out_p, plan):
        """Level-0 split of one relation: per-GPU send counts.  plan = None (hash shard g -> GPU g) or (owner, position)."""
        if plan is None:
            return self.e.shard_split(k, p, n, self.world, out_k, out_p)
        owner, position = plan
        per_pos = self.e.shard_split_ordered(k, p, n, len(owner), position, out_k, out_p)
        counts = [0] * self.world
        for v, pos in enumerate(position):
            counts[owner[v]] += per_pos[pos]
        return counts

    def _allreduce_u64(self, vals):
        """Sum 64-bit values over the ranks mod 2^64 (as 32-bit halves: the int64 SUM cannot overflow)."""
        halves = []
        for v in vals:
            halves += [v & 0xFFFFFFFF, v >> 32]
        t = torch.tensor(halves, dtype=torch.int64, device=self.cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.ctl)
        h = [int(x) for x in t.tolist()]
        mask = (1 << 64) - 1
        return [(h[2 * i] + (h[2 * i + 1] << 32)) & mask for i in range(len(vals))]

    def join_materialize(self, Rk, Rp, Sk, Sp):
        """The sharded MATERIALISING join of this driver (what hj_dist_rank_join_materialize is behind the C ABI): after the exchange
        every rank materialises the (key, payR, payS) tuples of the partitions it owns and keeps them; the sizes are all-gathered.
        Returns (global matches, (key, payR, payS) host columns of this rank's share, [tuples of every rank])."""
        return self.join(Rk, Rp, Sk, Sp, materialize=True)

    def _finish(self, materialize):
        e = self.e
        if not materialize:
            m, agg = e.join_count()                               # unchanged single-GPU build+probe
            gm, ga = self._allreduce_u64([m, agg])
            return gm, ga
        k, pr, ps = e.join_materialize()                          # this rank's share stays with it (the output stays sharded)
        t = torch.tensor([len(k)], dtype=torch.int64, device=self.cdev)
        allt = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(allt, t, group=self.ctl)
        sizes = [int(x.item()) for x in allt]
        return sum(sizes), (k, pr, ps), sizes

    def join(self, Rk, Rp, Sk, Sp, verify=False, materialize=False):
        """Local slices of R and S (int32 device columns) → (global matches, global sum payR*payS mod 2^64).

        verify=True (first warm-up step of bench.py): the order-independent (key,payload) digest of everything
        sent must equal the digest of everything received, summed over the ranks — a corrupted exchange fails
        here at once instead of feeding garbage (one giant duplicate partition) to the join.

        Pipeline: split R | exchange R ‖ split S | exchange S ‖ partition R | partition S | build+probe.
        The column exchanges are asynchronous on the data communicator; xGMI moves them while the CUs
        run the next local step (an exchange needs ~150 GB/s per link, the local passes need HBM)."""
        e, w = self.e, self.world
        nR, nS = int(Rk.numel()), int(Sk.numel())
        if w == 1 and not self.force_exchange:
            # one GPU owns every key: the level-0 split is the identity and nothing crosses a link — the local
            # slices ARE the received relations (no split pass, no copy)
            e.bind_device(self.pkg.REL_R, Rk, Rp, nR)
            e.bind_device(self.pkg.REL_S, Sk, Sp, nS)
            e.partition(self.pkg.REL_R)
            e.partition(self.pkg.REL_S)
            self.last_received = (nR, nS)
            return self._finish(materialize)
        plan = None
        if self.balance == "size":
            # which GPU owns which virtual shard must be the same for R and S: count both first (keys only, no data
            # movement), add up over the ranks, assign by size
            ns = w * self.VIRTUAL
            local = [a + b for a, b in zip(e.shard_count(Rk, nR, ns), e.shard_count(Sk, nS, ns))]
            t = torch.tensor(local, dtype=torch.int64, device=self.cdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.ctl)
            plan = self.assign_by_size([int(x) for x in t.tolist()], w)
            self.last_plan = plan
        okR, opR = self._get("split_kR", nR), self._get("split_pR", nR)
        cR = self._split(Rk, Rp, nR, okR, opR, plan)             # level-0 split, one contiguous run per owner
        gotR, totR, workR = self.exchange_async({"kR": okR, "pR": opR}, cR)
        okS, opS = self._get("split_kS", nS), self._get("split_pS", nS)
        cS = self._split(Sk, Sp, nS, okS, opS, plan)             # runs while R is on the links
        gotS, totS, workS = self.exchange_async({"kS": okS, "pS": opS}, cS)
        for wk in workR:
            wk.wait()
        e.bind_device(self.pkg.REL_R, gotR["kR"], gotR["pR"], totR)
        e.bind_device(self.pkg.REL_S, gotS["kS"], gotS["pS"], totS)   # sizes fix the radix bits for both
        e.partition(self.pkg.REL_R)                              # runs while S is on the links
        for wk in workS:
            wk.wait()
        if verify and hasattr(e, "digest_pairs"):
            sent = (e.digest_pairs(Rk, Rp, nR) + e.digest_pairs(Sk, Sp, nS)) & ((1 << 64) - 1)
            got = (e.digest_pairs(gotR["kR"], gotR["pR"], totR) + e.digest_pairs(gotS["kS"], gotS["pS"], totS)) & ((1 << 64) - 1)
            n_sent, n_got, d_sent, d_got = self._allreduce_u64([nR + nS, totR + totS, sent, got])
            if (n_sent, d_sent) != (n_got, d_got):
                raise RuntimeError("all-to-all exchange corrupted the relations: sent %d tuples (digest %#x), received %d (digest %#x)"
                                   % (n_sent, d_sent, n_got, d_got))
        e.partition(self.pkg.REL_S)
        self.last_received = (totR, totS)
        return self._finish(materialize)


# ---------------------------------------------------------------------------------------------------------------------
# The same exchange behind the C ABI (include/hj_dist.h, csrc/hj_dist.hip): C++ host code calling RCCL directly, sliced
# so that split(i+1) || exchange(i) || local pass-1(i-1) overlap, fixed-size messages (no count comes back to the host).
# ---------------------------------------------------------------------------------------------------------------------
import ctypes as _C

from . import _lib as _hjlib
from .join import HJError as _HJError, HashJoin as _HashJoin, _dev_ptr


def _stats_dict(s):
    return {"received": [int(s.received[0]), int(s.received[1])], "link_bytes": int(s.link_bytes), "payload_bytes": int(s.payload_bytes),
            "path": "sliced" if s.path == 0 else "exact", "slices": int(s.slices), "spans_per_slice": int(s.spans_per_slice),
            "slot_capacity": [int(s.slot_capacity[0]), int(s.slot_capacity[1])],
            "split_ms": [float(s.split_ms[0]), float(s.split_ms[1])], "pass1_ms": [float(s.pass1_ms[0]), float(s.pass1_ms[1])],
            "pass2_join_ms": float(s.pass2_join_ms), "first_split_ms": float(s.first_split_ms), "last_pass1_ms": float(s.last_pass1_ms),
            "wall_ms": float(s.wall_ms), "early_pass2_join_ms": float(s.early_pass2_join_ms), "probe_groups": int(s.probe_groups), "balanced": bool(s.balanced),
            "exchange_ms": float(s.exchange_ms), "materializing": bool(s.materializing), "materialized": int(s.materialized)}


class GroupJoin:
    """hj_dist: ONE process drives every rank (one context + one host thread per rank).  devices[r] = HIP device of rank r;
    distinct devices talk over RCCL, ranks sharing a device over the in-process device-copy transport (tests)."""

    def __init__(self, devices, transport=None):
        """transport: None / "auto" (RCCL for distinct devices, device copies for shared ones; $HJ_DIST_TRANSPORT), "rccl", "copy"."""
        self._L = _hjlib.lib()
        arr = (_C.c_int * len(devices))(*devices)
        h = _C.c_void_p()
        rc = self._L.hj_dist_create_transport(_C.byref(h), len(devices), arr, transport.encode() if transport else None)
        if rc:
            raise _HJError(rc, "hj_dist_create_transport(%r, %r) failed (fewer GPUs visible than ranks, no GPU, or a transport the devices "
                               "do not allow)" % (list(devices), transport))
        self._h = h
        self.world = len(devices)
        self._keep = {}

    def _ck(self, rc):
        if rc:
            raise _HJError(rc, (self._L.hj_dist_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.hj_dist_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def transport(self):
        return self._L.hj_dist_transport(self._h).decode()

    def set_transport(self, transport):
        """hj_dist_set_transport: the same group (contexts, bindings, buffers) over "rccl" or "copy"."""
        self._ck(self._L.hj_dist_set_transport(self._h, transport.encode() if transport else None))

    def context(self, rank):
        """The rank's HashJoin context (borrowed: owned by the group) — configure radix bits, generate inputs with it."""
        hj = _HashJoin.__new__(_HashJoin)
        hj._L = self._L
        hj._h = _C.c_void_p(self._L.hj_dist_context(self._h, rank))
        hj._keep = {}
        hj.close = lambda: None   # never destroyed from here
        return hj

    def configure(self, slices=0, exact_only=False, self_via_link=False, phantom_world=0, single_group=False, balance_size=False,
                  timeout_ms=0):
        cfg = _hjlib.DistConfig(slices=slices, exact_only=int(exact_only), self_via_link=int(self_via_link), phantom_world=phantom_world,
                                single_group=int(single_group), balance_size=int(balance_size), timeout_ms=int(timeout_ms))
        self._ck(self._L.hj_dist_configure(self._h, _C.byref(cfg)))

    def bind(self, rank, rel, keys, pays, n=None):
        n = int(keys.numel()) if n is None else n
        self._keep[(rank, rel)] = (keys, pays)
        self._ck(self._L.hj_dist_bind(self._h, rank, rel, _dev_ptr(keys), _dev_ptr(pays), n))

    def join(self):
        m, a = _C.c_uint64(), _C.c_uint64()
        self._ck(self._L.hj_dist_join(self._h, _C.byref(m), _C.byref(a)))
        return m.value, a.value

    def bind_output(self, rank, key, payR, payS, cap=None):
        """Rank `rank` writes the (key, payR, payS) tuples of the partitions it owns into these device columns (hj_dist_bind_output)."""
        cap = int(key.numel()) if cap is None else int(cap)
        self._keep[(rank, "out")] = (key, payR, payS)
        self._ck(self._L.hj_dist_bind_output(self._h, rank, _dev_ptr(key), _dev_ptr(payR), _dev_ptr(payS), cap))

    def join_materialize(self, agg=True):
        """hj_dist_join_materialize: (global matches, global aggregate or None, [tuples written by every rank]).  HJError with
        code HJ_ECAPACITY when some rank's output did not fit; .n_out then still holds the sizes."""
        m, a = _C.c_uint64(), _C.c_uint64()
        n_out = (_C.c_uint64 * self.world)()
        rc = self._L.hj_dist_join_materialize(self._h, _C.byref(m), _C.byref(a) if agg else None, n_out)
        self.n_out = [int(x) for x in n_out]
        self.last_matches = m.value
        self._ck(rc)
        return m.value, (a.value if agg else None), self.n_out

    def stats(self, rank):
        s = _hjlib.DistStats()
        self._ck(self._L.hj_dist_get_stats(self._h, rank, _C.byref(s)))
        return _stats_dict(s)


class RankJoin:
    """hj_dist_rank: one process per GPU.  `engine` is this rank's HashJoin; the 128-byte RCCL id made by rank 0 is handed
    to the others through torch.distributed (any backend: it is a broadcast of 128 bytes over the control plane)."""

    def __init__(self, engine, rank, world, group=None):
        self._L = _hjlib.lib()
        self.e = engine
        idbuf = (_C.c_ubyte * 128)()
        if rank == 0 and self._L.hj_dist_unique_id(_C.cast(idbuf, _C.c_void_p)):
            idbuf = (_C.c_ubyte * 128)()          # all zeros = "rank 0 could not make an id": every rank raises below, nobody waits
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
        if not any(raw):
            raise _HJError(-2, "hj_dist_unique_id failed on rank 0 (librccl not loadable?)")
        idbuf = (_C.c_ubyte * 128).from_buffer_copy(raw)
        h = _C.c_void_p()
        rc = self._L.hj_dist_rank_create(_C.byref(h), engine._h, rank, world, _C.cast(idbuf, _C.c_void_p))
        if rc:
            raise _HJError(rc, "hj_dist_rank_create(rank %d of %d) failed" % (rank, world))
        self._h = h
        self.world = world
        self.last_received = (0, 0)

    def _ck(self, rc):
        if rc:
            raise _HJError(rc, (self._L.hj_dist_rank_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.hj_dist_rank_destroy(self._h)
            self._h = None

    def configure(self, slices=0, exact_only=False, self_via_link=False, phantom_world=0, single_group=False, balance_size=False,
                  timeout_ms=0):
        cfg = _hjlib.DistConfig(slices=slices, exact_only=int(exact_only), self_via_link=int(self_via_link), phantom_world=phantom_world,
                                single_group=int(single_group), balance_size=int(balance_size), timeout_ms=int(timeout_ms))
        self._ck(self._L.hj_dist_rank_configure(self._h, _C.byref(cfg)))

    def join(self, Rk, Rp, Sk, Sp, verify=False):
        m, a = _C.c_uint64(), _C.c_uint64()
        self._ck(self._L.hj_dist_rank_join(self._h, _dev_ptr(Rk), _dev_ptr(Rp), int(Rk.numel()), _dev_ptr(Sk), _dev_ptr(Sp),
                                           int(Sk.numel()), _C.byref(m), _C.byref(a)))
        st = self.stats()
        self.last_received = tuple(st["received"])
        return m.value, a.value

    def join_materialize(self, Rk, Rp, Sk, Sp, out_key, out_payR, out_payS, cap=None, agg=False):
        """hj_dist_rank_join_materialize (collective): this rank's share of the output goes to its own device columns.
        Returns (global matches, global aggregate or None, tuples this rank wrote, [tuples of every rank])."""
        cap = int(out_key.numel()) if cap is None else int(cap)
        m, a, n = _C.c_uint64(), _C.c_uint64(), _C.c_uint64()
        n_all = (_C.c_uint64 * self.world)()
        self._ck(self._L.hj_dist_rank_join_materialize(self._h, _dev_ptr(Rk), _dev_ptr(Rp), int(Rk.numel()), _dev_ptr(Sk), _dev_ptr(Sp),
                                                       int(Sk.numel()), _dev_ptr(out_key), _dev_ptr(out_payR), _dev_ptr(out_payS), cap,
                                                       _C.byref(n), n_all, _C.byref(m), _C.byref(a) if agg else None))
        st = self.stats()
        self.last_received = tuple(st["received"])
        return m.value, (a.value if agg else None), n.value, [int(x) for x in n_all]

    def stats(self):
        s = _hjlib.DistStats()
        self._ck(self._L.hj_dist_rank_get_stats(self._h, _C.byref(s)))
        return _stats_dict(s)
