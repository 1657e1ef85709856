"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/oracle.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HJ_ASAN=1 (tests/test_asan.py): the AddressSanitizer/UBSan build of the oracle (`make -C oracle asan`)
_ASAN = os.environ.get("HJ_ASAN") == "1"
_SO = os.path.join(_HERE, "liboracle_asan.so" if _ASAN else "liboracle.so")
_lib = None

_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_SO)], stdout=subprocess.DEVNULL)
    ref = os.environ.get("HJ_REFERENCE", "/root/reference")
    if os.path.isdir(ref) and (force or not os.path.exists(os.path.join(_HERE, "_ref", "refgen"))
                               or not os.path.exists(os.path.join(_HERE, "_ref", "refjoin"))):
        subprocess.check_call(["make", "-C", _HERE, "ref", "REF=" + ref], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        L = _lib
        L.o_seed_generator.argtypes = [C.c_uint]
        L.o_random_gen.argtypes = [_i32p, C.c_uint64, C.c_int64]
        L.o_random_unique_gen.argtypes = [_i32p, C.c_uint64, C.c_int64, C.c_uint]
        L.o_knuth_shuffle.argtypes = [_i32p, C.c_uint64]
        L.o_fk_from_pk.argtypes = [_i32p, C.c_uint64, _i32p, C.c_uint64]
        L.o_gen_zipf.argtypes = [C.c_uint64, C.c_uint, C.c_double, _i32p]
        L.o_create_relation_n.argtypes = [_i32p, _i32p, C.c_uint64, C.c_uint64]
        L.o_read_bin.argtypes = [C.c_char_p, _i32p, C.c_uint64]
        L.o_read_bin.restype = C.c_int
        L.o_write_bin.argtypes = [C.c_char_p, _i32p, C.c_uint64]
        L.o_write_bin.restype = C.c_int
        L.o_radix_partition.argtypes = [_i32p, _i32p, C.c_uint64, C.c_uint32, C.c_uint32, _i32p, _i32p, _u64p]
        L.o_partition_digest.argtypes = [_i32p, _i32p, _u64p, C.c_uint64, _u64p]
        L.o_mix_triple.argtypes = [C.c_int32] * 3
        L.o_mix_triple.restype = C.c_uint64
        L.o_mix_pair.argtypes = [C.c_int32] * 2
        L.o_mix_pair.restype = C.c_uint64
        L.o_join_count.argtypes = [_i32p, _i32p, C.c_uint64, _i32p, _i32p, C.c_uint64, _u64p, _u64p, _u64p]
        L.o_join_materialize.argtypes = [_i32p, _i32p, C.c_uint64, _i32p, _i32p, C.c_uint64, _i32p, _i32p, _i32p, C.c_uint64]
        L.o_join_materialize.restype = C.c_uint64
        L.o_triples_checksum.argtypes = [_i32p, _i32p, _i32p, C.c_uint64]
        L.o_triples_checksum.restype = C.c_uint64
        L.o_joinCpu.argtypes = [_i32p, C.c_uint64, _i32p, C.c_uint64, C.c_int, _u64p, C.POINTER(C.c_uint32)]
        L.o_radix_join_omp.argtypes = [_i32p, _i32p, C.c_uint64, _i32p, _i32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _u64p]
        L.o_radix_join_omp.restype = C.c_uint64
        L.o_max_threads.restype = C.c_int
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i32p)


def _u(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


# ---- generators -------------------------------------------------------------------------------
def seed_generator(seed):
    lib().o_seed_generator(seed)


def random_gen(n, maxid):
    a = np.empty(n, np.int32)
    lib().o_random_gen(_p(a), n, maxid)
    return a


def random_unique_gen(n, maxid, time_seed):
    a = np.empty(n, np.int32)
    lib().o_random_unique_gen(_p(a), n, maxid, time_seed)
    return a


def fk_from_pk(pk, nfk):
    pk = np.ascontiguousarray(pk, np.int32)
    a = np.empty(nfk, np.int32)
    lib().o_fk_from_pk(_p(a), nfk, _p(pk), len(pk))
    return a


def gen_zipf(n, alphabet, theta):
    a = np.empty(n, np.int32)
    lib().o_gen_zipf(n, alphabet, theta, _p(a))
    return a


def create_relation_n(a, times):
    a = np.ascontiguousarray(a, np.int32)
    out = np.empty(len(a) * times, np.int32)
    lib().o_create_relation_n(_p(a), _p(out), len(a), times)
    return out


def read_bin(path, n):
    a = np.empty(n, np.int32)
    rc = lib().o_read_bin(path.encode(), _p(a), n)
    if rc:
        raise IOError("o_read_bin(%s) -> %d" % (path, rc))
    return a


def write_bin(path, a):
    a = np.ascontiguousarray(a, np.int32)
    rc = lib().o_write_bin(path.encode(), _p(a), len(a))
    if rc:
        raise IOError("o_write_bin(%s) -> %d" % (path, rc))


# ---- partition ----------------------------------------------------------------------------------
def radix_partition(keys, pays, shift, bits):
    keys = np.ascontiguousarray(keys, np.int32)
    pays = np.ascontiguousarray(pays, np.int32)
    ok, op = np.empty_like(keys), np.empty_like(pays)
    off = np.empty((1 << bits) + 1, np.uint64)
    lib().o_radix_partition(_p(keys), _p(pays), len(keys), shift, bits, _p(ok), _p(op), _u(off))
    return ok, op, off


def partition_digest(keys, pays, offsets):
    keys = np.ascontiguousarray(keys, np.int32)
    pays = np.ascontiguousarray(pays, np.int32)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    d = np.empty(len(offsets) - 1, np.uint64)
    lib().o_partition_digest(_p(keys), _p(pays), _u(offsets), len(offsets) - 1, _u(d))
    return d


# ---- join ---------------------------------------------------------------------------------------
def join_count(R, Pr, S, Ps, checksum=True):
    R = np.ascontiguousarray(R, np.int32)
    S = np.ascontiguousarray(S, np.int32)
    Pr = None if Pr is None else np.ascontiguousarray(Pr, np.int32)
    Ps = None if Ps is None else np.ascontiguousarray(Ps, np.int32)
    m, a, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib().o_join_count(_p(R), _p(Pr), len(R), _p(S), _p(Ps), len(S), C.byref(m), C.byref(a),
                       C.byref(c) if checksum else None)
    return m.value, a.value, (c.value if checksum else None)


def join_materialize(R, Pr, S, Ps):
    R = np.ascontiguousarray(R, np.int32)
    S = np.ascontiguousarray(S, np.int32)
    Pr = None if Pr is None else np.ascontiguousarray(Pr, np.int32)
    Ps = None if Ps is None else np.ascontiguousarray(Ps, np.int32)
    m, _, _ = join_count(R, Pr, S, Ps, checksum=False)
    k, pr, ps = (np.empty(m, np.int32) for _ in range(3))
    n = lib().o_join_materialize(_p(R), _p(Pr), len(R), _p(S), _p(Ps), len(S), _p(k), _p(pr), _p(ps), m)
    assert n == m
    return k, pr, ps


def triples_checksum(k, pr, ps):
    k, pr, ps = (np.ascontiguousarray(x, np.int32) for x in (k, pr, ps))
    return lib().o_triples_checksum(_p(k), _p(pr), _p(ps), len(k))


def sort_triples(k, pr, ps):
    """Canonical order of an output multiset: by (key,payR,payS) as uint32 bit patterns."""
    ku, pru, psu = (np.asarray(x, np.int32).view(np.uint32) for x in (k, pr, ps))
    o = np.lexsort((psu, pru, ku))
    return np.asarray(k)[o], np.asarray(pr)[o], np.asarray(ps)[o]


def joinCpu(R, S, threads=1):
    R = np.ascontiguousarray(R, np.int32)
    S = np.ascontiguousarray(S, np.int32)
    s, g = C.c_uint64(), C.c_uint32()
    lib().o_joinCpu(_p(R), len(R), _p(S), len(S), threads, C.byref(s), C.byref(g))
    return s.value, g.value


def radix_join_omp(R, Pr, S, Ps, bits1, bits2, threads):
    R = np.ascontiguousarray(R, np.int32)
    S = np.ascontiguousarray(S, np.int32)
    Pr = None if Pr is None else np.ascontiguousarray(Pr, np.int32)
    Ps = None if Ps is None else np.ascontiguousarray(Ps, np.int32)
    a = C.c_uint64()
    m = lib().o_radix_join_omp(_p(R), _p(Pr), len(R), _p(S), _p(Ps), len(S), bits1, bits2, threads, C.byref(a))
    return m, a.value


def max_threads():
    return lib().o_max_threads()


def mix_triple(k, pr, ps):
    return lib().o_mix_triple(k, pr, ps)


def mix_pair(k, p):
    return lib().o_mix_pair(k, p)


# ---- the reference generator (this container only) ------------------------------------------------
def refgen_path():
    p = os.path.join(_HERE, "_ref", "refgen")
    return p if os.path.exists(p) else None
