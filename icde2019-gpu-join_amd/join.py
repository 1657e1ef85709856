"""Host-side mirror of the reference's join interface on top of the C ABI (include/hj.h).

    HashJoin            one context per GPU; the steps of outOfGPU_Join1_payload
                        (src/hash_join_clustered_probe.cu:802-994) as methods
    hashJoinClusteredProbe(R, S)   the reference entry point itself (hjcp.cu:2062-2073), called
                        through the same `args` block main.cu fills (src/common-host.h:39-52)

numpy arrays are host columns; anything with .data_ptr() (torch tensors) is taken as an HBM-resident
column.  There is no CPU fallback: without libhj.so + a GPU every call raises.
"""
import ctypes as C

import numpy as np

from . import _lib

REL_R, REL_S = 0, 1
EINVAL, EHIP, ENOMEM, ECAPACITY, EIO = -1, -2, -3, -4, -5   # include/hj.h
PAYLOAD_ONES, PAYLOAD_ROWID, PAYLOAD_GIVEN = 0, 1, 2
_PAYLOAD = {"ones": PAYLOAD_ONES, "rowid": PAYLOAD_ROWID, "given": PAYLOAD_GIVEN}


class HJError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libhj error %d: %s" % (code, msg))
        self.code = code


def _dev_ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, int):
        return C.c_void_p(x)
    raise TypeError("expected a device tensor or a raw device pointer, got %r" % type(x))


def _host_i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.c_void_p)


class HashJoin:
    """One join context on one GPU (src/main.cu:93: one device per process)."""

    def __init__(self, device=0, stream=None):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.hj_create(C.byref(h), device)
        if rc:
            raise HJError(rc, "hj_create(device=%d) failed: no usable GPU (the HIP path has no CPU fallback)" % device)
        self._h = h
        self._keep = {}
        if stream is not None:
            self.set_stream(stream)

    # -- plumbing ---------------------------------------------------------------------------------
    def _ck(self, rc):
        if rc:
            raise HJError(rc, (self._L.hj_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.hj_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream):
        """stream: a raw hipStream_t value, e.g. torch.cuda.current_stream().cuda_stream (0 = HIP's
        default stream); None = back to the context's private stream."""
        own = C.c_void_p(-1 & (2 ** 64 - 1))  # HJ_OWN_STREAM
        self._ck(self._L.hj_set_stream(self._h, own if stream is None else C.c_void_p(int(stream))))

    def configure(self, bits1=0, bits2=0, force_bits=False, build_side=0, lds_capacity=0, lds_heads=0,
                  probe_chunk=0, exact_only=False, graph=False):
        cfg = _lib.Config(bits1=bits1, bits2=bits2, force_bits=int(force_bits), build_side=build_side,
                          lds_capacity=lds_capacity, lds_heads=lds_heads, probe_chunk=probe_chunk,
                          exact_only=int(exact_only), graph=int(graph))
        self._ck(self._L.hj_configure(self._h, C.byref(cfg)))

    def config(self):
        cfg = _lib.Config()
        self._ck(self._L.hj_get_config(self._h, C.byref(cfg)))
        return {k: getattr(cfg, k) for k, _ in _lib.Config._fields_ if not k.startswith("reserved")}

    def sync(self):
        self._ck(self._L.hj_sync(self._h))

    # -- relations --------------------------------------------------------------------------------
    def load_host(self, rel, keys, pays=None, payload="ones"):
        keys, kp = _host_i32(keys)
        mode = _PAYLOAD[payload] if pays is None else PAYLOAD_GIVEN
        pp = None
        if pays is not None:
            pays, pp = _host_i32(pays)
            assert len(pays) == len(keys)
        self._ck(self._L.hj_load_host(self._h, rel, kp, pp, len(keys), mode))

    def bind_device(self, rel, keys, pays, n=None):
        n = int(keys.numel()) if n is None else n
        self._keep[rel] = (keys, pays)  # keep the caller's tensors alive while bound
        self._ck(self._L.hj_bind_device(self._h, rel, _dev_ptr(keys), _dev_ptr(pays), n))

    # -- the path ---------------------------------------------------------------------------------
    def partition(self, rel):
        self._ck(self._L.hj_partition(self._h, rel))

    def partition_both(self):
        """hj_partition(R) + hj_partition(S), S's passes on a second stream beside R's."""
        self._ck(self._L.hj_partition_both(self._h))

    def partition_layout(self, rel):
        """'slotted' if the histogram-free passes produced the partitions, 'sampled' if their variable-capacity form for a
        skewed probe side did, 'exact' if the histogram passes did."""
        v = C.c_int()
        self._ck(self._L.hj_partition_layout(self._h, rel, C.byref(v)))
        return {0: "exact", 1: "slotted", 2: "sampled"}[v.value]

    def join_count(self):
        m, a = C.c_uint64(), C.c_uint64()
        self._ck(self._L.hj_join_count(self._h, C.byref(m), C.byref(a)))
        return m.value, a.value

    def join(self):
        m, a = C.c_uint64(), C.c_uint64()
        self._ck(self._L.hj_join(self._h, C.byref(m), C.byref(a)))
        return m.value, a.value

    def last_call_breakdown(self):
        """Host-side breakdown of the last join() call (hj_last_call_breakdown), ms."""
        a, f, s, tt = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        n = C.c_uint32()
        self._ck(self._L.hj_last_call_breakdown(self._h, C.byref(a), C.byref(n), C.byref(f), C.byref(s), C.byref(tt)))
        return {"allocation_ms": round(a.value, 3), "allocations": n.value, "failed_optimistic_attempt_ms": round(f.value, 3),
                "sample_and_plan_ms": round(s.value, 3), "total_ms": round(tt.value, 3),
                "rest_ms (the step that answered)": round(tt.value - a.value - f.value - s.value, 3)}

    def join_late_materialize(self, d_Dr, ncolR, strideR, d_Ds, ncolS, strideS):
        """Row-id payloads + gather of extra columns on every match (jp.cu:1420-1557)."""
        m, a = C.c_uint64(), C.c_uint64()
        self._ck(self._L.hj_join_late_materialize(self._h, _dev_ptr(d_Dr), ncolR, strideR, _dev_ptr(d_Ds), ncolS, strideS,
                                                  C.byref(m), C.byref(a)))
        return m.value, a.value

    def join_nonpartitioned(self, kind):
        """Comparison baselines: 0 = perfect array (jp.cu:628-668), 1 = global chained table (jp.cu:681-742)."""
        m, a = C.c_uint64(), C.c_uint64()
        self._ck(self._L.hj_join_nonpartitioned(self._h, kind, C.byref(m), C.byref(a)))
        return m.value, a.value

    def join_stream_probe(self, S, Ps=None, payload="ones", segment_tuples=0):
        """S stays on the host and is streamed through HBM in segments (hjcp.cu:1684-1984); R must be loaded."""
        S, kp = _host_i32(S)
        mode = _PAYLOAD[payload] if Ps is None else PAYLOAD_GIVEN
        pp = None
        if Ps is not None:
            Ps, pp = _host_i32(Ps)
        m, a = C.c_uint64(), C.c_uint64()
        self._ck(self._L.hj_join_stream_probe(self._h, kp, pp, len(S), segment_tuples, mode, C.byref(m), C.byref(a)))
        return m.value, a.value

    def join_stream_probe_materialize(self, S, Ps=None, payload="ones", segment_tuples=0, cap=None, out=None):
        """Streaming probe side with materialisation (hjcp.cu:1917-1961): returns host columns (key, payR, payS).
        out = three preallocated host int32 arrays (e.g. views of pinned torch tensors) of at least cap elements."""
        S, kp = _host_i32(S)
        mode = _PAYLOAD[payload] if Ps is None else PAYLOAD_GIVEN
        pp = None
        if Ps is not None:
            Ps, pp = _host_i32(Ps)
        if cap is None:
            cap = self.join_stream_probe(S, Ps, payload, segment_tuples)[0]
        if out is None:
            out = [np.empty(max(cap, 1), np.int32) for _ in range(3)]
        n, a = C.c_uint64(), C.c_uint64()
        self._ck(self._L.hj_join_stream_probe_materialize(self._h, kp, pp, len(S), segment_tuples, mode,
                                                          out[0].ctypes.data_as(C.c_void_p), out[1].ctypes.data_as(C.c_void_p),
                                                          out[2].ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(a)))
        return tuple(o[:n.value] for o in out), a.value

    def join_coprocess(self, R, Pr, S, Ps, level0_parts=0, host_threads=0):
        """Both relations host-resident: host level-0 split + per-partition GPU joins (hjcp.cu:1000-1680)."""
        R, rk = _host_i32(R)
        S, sk = _host_i32(S)
        rp = sp = None
        if Pr is not None:
            Pr, rp = _host_i32(Pr)
        if Ps is not None:
            Ps, sp = _host_i32(Ps)
        m, a = C.c_uint64(), C.c_uint64()
        self._ck(self._L.hj_join_coprocess(self._h, rk, rp, len(R), sk, sp, len(S), level0_parts, host_threads,
                                           C.byref(m), C.byref(a)))
        return m.value, a.value

    def coprocess_numa(self):
        """(NUMA nodes of the host, node closest to the GPU, CPUs of that node the split's workers were bound to)."""
        a, b, d = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.hj_coprocess_numa(self._h, C.byref(a), C.byref(b), C.byref(d)))
        return a.value, b.value, d.value

    def coprocess_groups(self):
        """Residency groups of the last join_coprocess call (runs of level-0 pairs uploaded and joined together)."""
        v = C.c_uint32()
        self._ck(self._L.hj_coprocess_groups(self._h, C.byref(v)))
        return v.value

    def host_split_throughput(self):
        v = C.c_double()
        self._ck(self._L.hj_host_split_throughput(self._h, C.byref(v)))
        return v.value

    def join_materialize_into(self, d_key, d_payR, d_payS, cap):
        n = C.c_uint64()
        self._ck(self._L.hj_join_materialize(self._h, _dev_ptr(d_key), _dev_ptr(d_payR), _dev_ptr(d_payS), cap,
                                             C.byref(n)))
        return n.value

    def join_and_materialize_into(self, d_key, d_payR, d_payS, cap):
        """hj_join_and_materialize: partition both + ONE probe in one call (a skewed probe side's hot keys are written by pass 1)."""
        n = C.c_uint64()
        self._ck(self._L.hj_join_and_materialize(self._h, _dev_ptr(d_key), _dev_ptr(d_payR), _dev_ptr(d_payS), cap, C.byref(n)))
        return n.value

    def join_and_materialize(self, cap):
        """Host numpy columns (key, payR, payS) of hj_join_and_materialize."""
        bufs = [self.device_malloc(max(cap, 1) * 4) for _ in range(3)]
        try:
            n = self.join_and_materialize_into(bufs[0], bufs[1], bufs[2], cap)
            return tuple(self.to_host(b, n, np.int32) for b in bufs)
        finally:
            for b in bufs:
                self.device_free(b)

    def hot_stats(self):
        """The heavy-hitter bypass of the last join() / join_and_materialize(): mode (0 none, 1 counted, 2 written), keys, share, matches."""
        m, k, s, x = C.c_int(), C.c_uint32(), C.c_double(), C.c_uint64()
        self._ck(self._L.hj_hot_stats(self._h, C.byref(m), C.byref(k), C.byref(s), C.byref(x)))
        return {"mode": m.value, "keys": k.value, "share": s.value, "matches": x.value}

    def debug_set_stamps(self, d_join, d_part2):
        """Experiments (libhj_stamps.so): device buffers for per-workgroup timelines of the count kernel / pass 2."""
        self._ck(self._L.hj_debug_set_stamps(self._h, _dev_ptr(d_join), _dev_ptr(d_part2)))

    def reload_knobs(self):
        """Experiments: re-read the HJ_* environment knobs (the library reads them once, in hj_create)."""
        self._ck(self._L.hj_reload_knobs(self._h))

    def join_materialize(self, cap=None):
        """Returns host numpy columns (key, payR, payS); sizes the output with a count run if cap is None."""
        if cap is None:
            cap = self.join_count()[0]
        bufs = [self.device_malloc(max(cap, 1) * 4) for _ in range(3)]
        try:
            n = self.join_materialize_into(bufs[0], bufs[1], bufs[2], cap)
            return tuple(self.to_host(b, n, np.int32) for b in bufs)
        finally:
            for b in bufs:
                self.device_free(b)

    # -- memory helpers ---------------------------------------------------------------------------
    def device_malloc(self, nbytes):
        p = C.c_void_p()
        self._ck(self._L.hj_device_malloc(self._h, C.byref(p), nbytes))
        return p.value

    def device_free(self, ptr):
        self._ck(self._L.hj_device_free(self._h, C.c_void_p(ptr)))

    def to_host(self, dptr, count, dtype):
        out = np.empty(count, dtype=dtype)
        p = dptr.value if isinstance(dptr, C.c_void_p) else (dptr.data_ptr() if hasattr(dptr, "data_ptr") else dptr)
        self._ck(self._L.hj_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes))
        return out

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.device_malloc(max(arr.nbytes, 16))
        self._ck(self._L.hj_memcpy_h2d(self._h, C.c_void_p(p), arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return p

    # -- introspection ----------------------------------------------------------------------------
    def partition_pointers(self, rel):
        k, p, o, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._ck(self._L.hj_get_partitions(self._h, rel, C.byref(k), C.byref(p), C.byref(o), C.byref(n)))
        return k.value, p.value, o.value, n.value

    def partitions(self, rel, n):
        """Host copies (keys, pays, offsets) of the partitioned relation of n tuples."""
        k, p, o, nparts = self.partition_pointers(rel)
        off = self.to_host(o, nparts + 1, np.uint64)
        return self.to_host(k, n, np.int32), self.to_host(p, n, np.int32), off

    def verify_partitions(self, rel, with_digests=False):
        bad = C.c_uint64()
        dg = None
        _, _, _, nparts = self.partition_pointers(rel)
        if with_digests:
            dg = self.device_malloc(nparts * 8)
        try:
            self._ck(self._L.hj_verify_partitions(self._h, rel, C.byref(bad), C.c_void_p(dg) if dg else None))
            digests = self.to_host(dg, nparts, np.uint64) if dg else None
        finally:
            if dg:
                self.device_free(dg)
        return bad.value, digests

    def enable_timings(self, level=1):
        """Per-kernel HIP-event timing: 0 off (default), 1 the data-moving kernels, 2 every launch."""
        self._ck(self._L.hj_enable_timings(self._h, level))

    def timings_reset(self):
        self._ck(self._L.hj_timings_reset(self._h))

    def timings(self):
        arr = (_lib.KernelTime * 32)()
        n = C.c_uint32()
        self._ck(self._L.hj_timings(self._h, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): {"launches": arr[i].launches, "total_ms": arr[i].total_ms,
                                        "last_ms": arr[i].last_ms} for i in range(min(n.value, 32))}

    # -- multi-GPU level-0 split / synthesis / digests ------------------------------------------------
    def shard_split(self, keys, pays, n, nshards, out_keys, out_pays):
        counts = (C.c_uint64 * nshards)()
        self._ck(self._L.hj_shard_split(self._h, _dev_ptr(keys), _dev_ptr(pays), n, nshards, _dev_ptr(out_keys),
                                        _dev_ptr(out_pays), counts))
        return [int(c) for c in counts]

    def shard_count(self, keys, n, nshards):
        counts = (C.c_uint64 * nshards)()
        self._ck(self._L.hj_shard_count(self._h, _dev_ptr(keys), n, nshards, counts))
        return [int(c) for c in counts]

    def shard_split_ordered(self, keys, pays, n, nshards, position, out_keys, out_pays):
        """position[v] = output position of shard v; returns the tuple counts per output position."""
        counts = (C.c_uint64 * nshards)()
        pos = (C.c_uint32 * nshards)(*[int(x) for x in position])
        self._ck(self._L.hj_shard_split_ordered(self._h, _dev_ptr(keys), _dev_ptr(pays), n, nshards, pos, _dev_ptr(out_keys),
                                                _dev_ptr(out_pays), counts))
        return [int(c) for c in counts]

    def gen_unique(self, d_keys, n, first, domain, seed):
        self._ck(self._L.hj_gen_unique(self._h, _dev_ptr(d_keys), n, first, domain, seed))

    def gen_zipf(self, d_keys, n, first, alphabet, theta, seed):
        self._ck(self._L.hj_gen_zipf(self._h, _dev_ptr(d_keys), n, first, alphabet, theta, seed))

    def fill_payload(self, d_pays, n, payload="ones", first_rowid=0):
        self._ck(self._L.hj_fill_payload(self._h, _dev_ptr(d_pays), n, _PAYLOAD[payload], first_rowid))

    def ubench(self, kind, in_k, in_p, out_k, out_p, n, reps=5):
        """HBM ceiling of a radix pass's access pattern with no partitioning work: kind 'copy' or 'line_scatter'; 'read' / 'write':
        both columns streamed in only / out only (the two ends of a kernel's read:write mix); 'copy1' / 'pairs' / 'pairs_scatter' /
        'soa_to_pairs_scatter' (round 6, the layout gate): the same bytes as ONE array per side — in_p must be in_k + n, out_p out_k + n.
        Returns GB/s (bytes read + written per second)."""
        ms, nb = C.c_double(), C.c_uint64()
        self._ck(self._L.hj_ubench(self._h, _UBENCH_KINDS[kind], _dev_ptr(in_k), _dev_ptr(in_p), _dev_ptr(out_k),
                                   _dev_ptr(out_p), n, reps, C.byref(ms), C.byref(nb)))
        return nb.value / (ms.value * 1e-3) / 1e9

    def digest_pairs(self, d_keys, d_pays, n):
        d = C.c_uint64()
        self._ck(self._L.hj_digest_pairs(self._h, _dev_ptr(d_keys), _dev_ptr(d_pays), n, C.byref(d)))
        return d.value

    def digest_triples(self, d_key, d_pr, d_ps, n):
        d = C.c_uint64()
        self._ck(self._L.hj_digest_triples(self._h, _dev_ptr(d_key), _dev_ptr(d_pr), _dev_ptr(d_ps), n, C.byref(d)))
        return d.value


_UBENCH_KINDS = {"copy": 0, "line_scatter": 1, "read": 2, "write": 3, "copy1": 4, "pairs": 5, "pairs_scatter": 6, "soa_to_pairs_scatter": 7}


def host_join(R, Pr, S, Ps, threads=0):
    """The library's own CPU radix join (a reported baseline; no GPU): (matches, agg, seconds)."""
    R, rp = _host_i32(R)
    S, sp = _host_i32(S)
    prp = psp = None
    if Pr is not None:
        Pr, prp = _host_i32(Pr)
    if Ps is not None:
        Ps, psp = _host_i32(Ps)
    m, a, dt = C.c_uint64(), C.c_uint64(), C.c_double()
    rc = _lib.lib().hj_host_join(rp, prp, len(R), sp, psp, len(S), threads, C.byref(m), C.byref(a), C.byref(dt))
    if rc:
        raise HJError(rc, "hj_host_join")
    return m.value, a.value, dt.value


def host_split(keys, pays, parts, threads=0):
    """The host level-0 split on its own (no GPU): (out_keys, out_pays, offsets[parts+1], GB/s)."""
    keys, kp = _host_i32(keys)
    pp = None
    if pays is not None:
        pays, pp = _host_i32(pays)
    ok, op = np.empty(len(keys) + 16, np.int32), np.empty(len(keys) + 16, np.int32)
    # 64-byte-aligned views so that the non-temporal line stores are taken
    def aligned(a):
        sh = (-a.ctypes.data) % 64 // 4
        return a[sh:sh + len(keys)]
    ok, op = aligned(ok), aligned(op)
    off = np.zeros(parts + 1, np.uint64)
    gbs = C.c_double()
    rc = _lib.lib().hj_host_split(kp, pp, len(keys), parts, threads, ok.ctypes.data_as(C.c_void_p), op.ctypes.data_as(C.c_void_p),
                                  off.ctypes.data_as(_lib.u64p), C.byref(gbs))
    if rc:
        raise HJError(rc, "hj_host_split")
    return ok, op, off, gbs.value


def host_split_blocks(keys, pays, parts, threads=0):
    """The one-pass host split hj_join_coprocess runs (no GPU): (out_keys, out_pays | None, block_part, block_start, block_count, GB/s).
    The output columns are staging columns with holes: only the blocks are meaningful."""
    keys, kp = _host_i32(keys)
    pp = None
    if pays is not None:
        pays, pp = _host_i32(pays)
    L = _lib.lib()
    cap = L.hj_host_split_blocks_capacity(len(keys), parts, threads)

    def aligned():
        a = np.zeros(cap + 16, np.int32)
        sh = (-a.ctypes.data) % 64 // 4
        return a[sh:sh + cap]
    ok = aligned()
    op = aligned() if pays is not None else None
    maxb = cap // 256 + 1
    bp, bs, bc = np.zeros(maxb, np.uint32), np.zeros(maxb, np.uint64), np.zeros(maxb, np.uint32)
    nb, gbs = C.c_uint64(), C.c_double()
    rc = L.hj_host_split_blocks(kp, pp, len(keys), parts, threads, ok.ctypes.data_as(C.c_void_p),
                                op.ctypes.data_as(C.c_void_p) if op is not None else None, cap, bp.ctypes.data_as(C.c_void_p),
                                bs.ctypes.data_as(C.c_void_p), bc.ctypes.data_as(C.c_void_p), maxb, C.byref(nb), C.byref(gbs))
    if rc:
        raise HJError(rc, "hj_host_split_blocks")
    m = nb.value
    return ok, op, bp[:m], bs[:m], bc[:m], gbs.value


def shard_of(key, nshards):
    return _lib.lib().hj_shard_of(int(np.int32(key)), nshards)


def hashJoinClusteredProbe(R, S):
    """The reference's entry point (src/main.cu:291 `input.alg.joinAlg(&joinArgs,&time)`), called the
    way main.cu calls it: host key columns in an `args` block.  Returns hj_last_result as a dict."""
    L = _lib.lib()
    R = np.ascontiguousarray(R, np.int32)
    S = np.ascontiguousarray(S, np.int32)
    a = _lib.Args()
    a.R = R.ctypes.data_as(_lib.i32p)
    a.R_els = len(R)
    a.S = S.ctypes.data_as(_lib.i32p)
    a.S_els = len(S)
    a.threadsNum, a.sharedMem, a.pivotsNum = 32, 30 << 10, 1
    ret = L.hashJoinClusteredProbe(C.byref(a), None)
    res = _lib.LastResult()
    L.hj_reference_last_result(C.byref(res))
    out = {k: getattr(res, k) for k in ("matches", "agg", "materialized", "status")}
    out["partition_ms"] = list(res.partition_ms)
    out["join_ms"] = list(res.join_ms)
    out["return"] = ret
    return out
