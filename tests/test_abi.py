"""The C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares.
No compute calls here (no GPU needed)."""
import ctypes
import os
import re

import pytest

from hjtest import ROOT, has_gpu, pkg


def _declared_functions():
    names = set()
    for h in ("hj.h", "hj_reference_abi.h", "hj_dist.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
            n = m.group(1)
            if n.startswith("hj_") or n == "hashJoinClusteredProbe":
                names.add(n)
    return names


def test_library_builds_and_exports_every_declared_symbol():
    p = pkg()
    L = p._lib.lib()
    declared = _declared_functions()
    assert len(declared) >= 35
    for n in declared:
        assert hasattr(L, n), "libhj.so does not export %s" % n
    # and the Python binding covers exactly the header
    assert declared == set(p._lib.SIGNATURES), declared ^ set(p._lib.SIGNATURES)
    assert b"gfx950" in L.hj_version()


def test_code_object_is_gfx950():
    so = pkg()._lib.LIB_PATH
    blob = open(so, "rb").read()
    assert b"gfx950" in blob and b"k_scatter" in blob and b"k_join" in blob


def test_struct_layout_matches_reference_args_block():
    # src/common-host.h:39-52: int* S; size_t S_els; char[50]; int* R; size_t R_els; char[50]; int; uint; uint
    A = pkg()._lib.Args
    assert A.S.offset == 0 and A.S_els.offset == 8 and A.S_filename.offset == 16
    assert A.R.offset == 72 and A.R_els.offset == 80 and A.R_filename.offset == 88
    assert A.threadsNum.offset == 140 and A.sharedMem.offset == 144 and A.pivotsNum.offset == 148
    assert ctypes.sizeof(A) == 152


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    p = pkg()
    with pytest.raises(p.HJError):
        p.HashJoin(0)
    # the reference entry point reports the failure instead of falling back to a CPU join
    import numpy as np
    r = p.hashJoinClusteredProbe(np.arange(8, dtype=np.int32), np.arange(8, dtype=np.int32))
    assert r["status"] != 0 and r["matches"] == 0


def _build_c_client(tmp_path, name="c_abi_client"):
    """gcc (not hipcc), C99, only include/*.h: the ABI is consumable from plain C."""
    import subprocess
    p = pkg()
    p._lib.build()
    exe = str(tmp_path / name)
    libdir = os.path.dirname(p._lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", name + ".c"), "-o", exe, "-L", libdir, "-lhj",
                           "-Wl,-rpath," + libdir])
    return exe


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_c_client_builds_and_fails_loudly_without_gpu(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_client_on_gpu(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_client ok: 150000 matches" in r.stdout


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_c_dist_client_builds_and_fails_loudly_without_gpu(tmp_path):
    """include/hj_dist.h from plain C (gcc -std=c99 -Werror): builds against the library; without a GPU the group cannot be made."""
    import subprocess
    r = subprocess.run([_build_c_client(tmp_path, "c_abi_dist_client")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_dist_client_on_gpu(tmp_path):
    """The multi-GPU ABI driven from plain C: three ranks on cuda:0, count-only and materialising, every output tuple checked."""
    import subprocess
    r = subprocess.run([_build_c_client(tmp_path, "c_abi_dist_client")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_dist_client ok: 180000 matches over 3 ranks (device-copy transport)" in r.stdout


def test_host_write_combining_split_parity():
    """hj_host_split (the co-processing path's host level-0 partitioner: per-thread software write-combining lines,
    non-temporal AVX2 stores — partition-primitives.cu:40-125,129-232) against numpy: offsets, and per partition the
    exact (key,payload) multiset.  Pure host code: runs without a GPU."""
    import numpy as np
    p = pkg()
    rng = np.random.default_rng(17)
    for n, parts, threads in ((0, 4, 2), (1, 1, 1), (15, 16, 3), (1000, 7, 4), (100_003, 16, 8), (250_000, 64, 5), (60_000, 1000, 2)):
        keys = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
        keys[: n // 3] = rng.integers(0, 40, n // 3)            # a skewed third
        pays = np.arange(n, dtype=np.int32)
        ok, op, off, gbs = p.host_split(keys, pays, parts, threads)
        owner = np.array([p.shard_of(int(k), parts) for k in keys], dtype=np.int64) if n else np.empty(0, np.int64)
        cnt = np.bincount(owner, minlength=parts)
        assert off[0] == 0 and np.array_equal(np.diff(off.astype(np.int64)), cnt)
        assert np.array_equal(keys[op], ok)                         # payload travels with its key
        assert sorted(op.tolist()) == list(range(n))                # a permutation: nothing lost, nothing duplicated
        for q in range(parts):
            seg = op[int(off[q]):int(off[q + 1])]
            assert np.all(owner[seg] == q)
        # keys-only / payload = ones variants
        ok2, op2, off2, _ = p.host_split(keys, None, parts, threads)
        assert np.array_equal(off2, off) and np.all(op2 == 1) and np.array_equal(np.sort(ok2), np.sort(keys))


def test_host_radix_join_parity():
    """hj_host_join (bench.py's `cpu_baseline.best_effort`: the library's own host code as a CPU radix join — one-pass block split, then
    cache-sized chained tables, hash_join_clustered_probe.cu:2013-2059's build/probe per piece) against the oracle: match count and
    aggregate with signed payloads, duplicates on both sides, keys one side lacks, empty sides, payload = ones.  Pure host code."""
    import numpy as np
    from oracle import pyoracle as o
    p = pkg()
    rng = np.random.default_rng(23)
    for nR, nS, dup, threads in ((0, 10, False, 2), (7, 0, False, 1), (5, 3, True, 3), (1000, 4000, True, 4), (100_003, 400_001, False, 8),
                                 (1 << 19, 1 << 20, True, 5)):
        R = (rng.integers(-2**31, 2**31 - 1, nR) if dup else rng.permutation(max(nR, 1))[:nR]).astype(np.int32)
        if dup and nR >= 3:
            R[: nR // 3] = R[nR // 3: 2 * (nR // 3)][: nR // 3]
        S = (R[rng.integers(0, nR, nS)] if nR else rng.integers(0, 9, nS)).astype(np.int32)
        S[::7] = -5                                              # a heavy key R (almost surely) lacks
        Pr = rng.integers(-2**31, 2**31 - 1, nR).astype(np.int32)
        Ps = rng.integers(-2**31, 2**31 - 1, nS).astype(np.int32)
        em, eagg, _ = o.join_count(R, Pr, S, Ps, checksum=False)
        m, a, dt = p.host_join(R, Pr, S, Ps, threads)
        assert (m, a) == (em, eagg), (nR, nS, threads)
        em1, eagg1, _ = o.join_count(R, None, S, None, checksum=False)
        assert p.host_join(R, None, S, None, threads)[:2] == (em1, eagg1)
        assert p.host_join(S, Ps, R, Pr, threads)[:2] == (em, eagg)   # the larger side first: the smaller side of a pair builds


def test_host_one_pass_block_split_parity():
    """hj_host_split_blocks (what hj_join_coprocess runs: one pass, partitions as lists of blocks taken from per-worker arenas — the
    reference's bucket chains, join-primitives.cu:138-192, on the host) against numpy: every block holds tuples of its partition only,
    blocks do not overlap and stay inside the capacity, and the blocks together are exactly the input (key, payload) multiset."""
    import numpy as np
    p = pkg()
    rng = np.random.default_rng(23)
    for n, parts, threads in ((0, 4, 2), (1, 1, 1), (15, 16, 3), (5000, 7, 4), (300_007, 16, 8), (250_000, 64, 5), (70_000, 1000, 2),
                              (1_200_000, 16, 3)):
        keys = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
        keys[: n // 3] = rng.integers(0, 40, n // 3)            # a skewed third
        pays = np.arange(n, dtype=np.int32)
        for with_pay in (True, False):
            ok, op, bp, bs, bc, gbs = p.host_split_blocks(keys, pays if with_pay else None, parts, threads)
            assert int(bc.sum()) == n
            assert np.all(bc > 0) and np.all(bp < parts)
            order = np.lexsort((bs, bp))
            assert np.array_equal(order, np.arange(len(bp)))        # sorted by (partition, start)
            by_addr = np.argsort(bs)
            ends = bs[by_addr].astype(np.int64) + bc[by_addr]
            assert np.all(ends[:-1] <= bs[by_addr][1:].astype(np.int64)) and (len(ends) == 0 or ends[-1] <= len(ok))
            got_k = np.concatenate([ok[int(s):int(s) + int(c)] for s, c in zip(bs, bc)]) if len(bs) else np.empty(0, np.int32)
            owner = np.repeat(bp, bc)
            if n:
                assert np.array_equal(np.array([p.shard_of(int(k), parts) for k in got_k[:: max(1, n // 20000)]]), owner[:: max(1, n // 20000)])
            if with_pay:
                got_p = np.concatenate([op[int(s):int(s) + int(c)] for s, c in zip(bs, bc)]) if len(bs) else np.empty(0, np.int32)
                assert np.array_equal(keys[got_p], got_k)               # payload travels with its key
                assert np.array_equal(np.sort(got_p), np.arange(n))     # a permutation: nothing lost, nothing duplicated
            else:
                assert op is None and np.array_equal(np.sort(got_k), np.sort(keys))
    # what the split publishes while it runs (the ranges hj_join_coprocess uploads beside the split): checked inside the library under
    # hj_host_split_debug_progress(1) (a debug symbol, not in the header) — every published range was final when published and consists
    # of whole, full blocks
    p._lib.lib().hj_host_split_debug_progress(1)
    try:
        published = 0
        for n, parts, threads in ((2_000_000, 16, 4), (1_500_000, 3, 7), (900_000, 200, 2)):
            keys = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
            keys[: n // 4] = 5                                     # one partition far ahead of the others
            ok, op, bp, bs, bc, covered = p.host_split_blocks(keys, None, parts, threads)
            assert int(bc.sum()) == n and 0 <= covered <= n
            published += covered
        assert published > 0
    finally:
        p._lib.lib().hj_host_split_debug_progress(0)
    # the staging capacity stays in proportion: at most an eighth more than the tuples once there is a block's worth per (worker,
    # partition), and a small input does not pay for the threads and partitions it was offered (one worker per 2^16 tuples)
    cap_of = p._lib.lib().hj_host_split_blocks_capacity
    assert cap_of(0, 4096, 64) <= 1 << 20 and cap_of(1000, 16, 16) <= 8192
    for n, parts, threads in ((1 << 20, 16, 16), (1 << 27, 16, 16), (1 << 31, 16, 64), (3_000_001, 16, 7)):
        assert n <= cap_of(n, parts, threads) <= n + n // 8 + 2 * parts * 256 * threads, (n, parts, threads)
    # too small a capacity is refused, not overrun
    L = p._lib.lib()
    import ctypes as C
    nb = C.c_uint64()
    k = np.zeros(1000, np.int32)
    o = np.zeros(1000, np.int32)
    assert L.hj_host_split_blocks(k.ctypes.data_as(C.c_void_p), None, 1000, 4, 2, o.ctypes.data_as(C.c_void_p), None, 1000,
                                  None, None, None, 0, C.byref(nb), None) == p.ECAPACITY


def test_timinginfo_layout_matches_reference(tmp_path):
    """struct timingInfo (src/common.h:101-119): unsigned n; timeval start[5], end[5]; 8 doubles; 4 unsigned counters.
    Offsets on x86-64 (timeval = 16 bytes): the header a maintainer links against must agree field by field."""
    import subprocess
    fields = ["n", "start", "end", "greaterTime", "reduce_usecs", "fixPositions_usecs", "scatter_usecs", "copy_usecs",
              "bitonic_usecs", "total_usecs", "greaterEventTime", "greaterCallsNum", "bitonicCallsNum", "reduceCallsNum",
              "fixPositionsCallsNum"]
    expect = [0, 8, 88, 168, 176, 184, 192, 200, 208, 216, 224, 232, 236, 240, 244]
    src = tmp_path / "ti.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hj_reference_abi.h"\nint main(void){\n' +
                   "".join('printf("%%zu\\n", offsetof(timingInfo, %s));\n' % f for f in fields) +
                   'printf("%zu\\n", sizeof(timingInfo)); printf("%zu\\n", sizeof(args)); return 0; }\n')
    exe = tmp_path / "ti"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[:-2] == expect and got[-2] == 248 and got[-1] == 152
    ref = "/root/reference/src/common.h"
    if os.path.exists(ref):   # build container: the reference's own declaration order, field by field
        text = open(ref).read()
        body = text[text.index("typedef struct timingInfo {"):text.index("} timingInfo;")]
        names = re.findall(r"\b(\w+)(?:\[5\])?(?:\s*=\s*\d+)?;", body)
        assert names == fields, names
