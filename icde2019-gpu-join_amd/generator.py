"""Host mirror of the generator_ETHZ drop-in (csrc/gen_ethz.cpp; src/generator_ETHZ.cuh:11-23):
same function names as the reference's generator API, numpy in/out, explicit seed."""
import ctypes as C

import numpy as np

from . import _lib


def _out(n):
    a = np.empty(n, np.int32)
    return a, a.ctypes.data_as(C.c_void_p)


def _fn(path):
    return path.encode() if path else None


def _ck(rc, what):
    if rc:
        raise IOError("%s failed with code %d" % (what, rc))


def seed_generator(seed):
    """gen.cu:23-27 seed_generator, also fixing the time(NULL) seed of random_unique_gen; 0 = time(NULL)."""
    _lib.lib().hj_gen_set_seed(seed)


def create_relation_unique(filename, num_tuples, maxid):
    a, p = _out(num_tuples)
    _ck(_lib.lib().hj_create_relation_unique(_fn(filename), p, num_tuples, maxid), "create_relation_unique")
    return a


def create_relation_nonunique(filename, num_tuples, maxid):
    a, p = _out(num_tuples)
    _ck(_lib.lib().hj_create_relation_nonunique(_fn(filename), p, num_tuples, maxid), "create_relation_nonunique")
    return a


def create_relation_zipf(filename, num_tuples, maxid, zipf_param):
    a, p = _out(num_tuples)
    _ck(_lib.lib().hj_create_relation_zipf(_fn(filename), p, num_tuples, maxid, zipf_param), "create_relation_zipf")
    return a


def create_relation_fk_from_pk(filename, fk_tuples, pkrel):
    pk = np.ascontiguousarray(pkrel, np.int32)
    a, p = _out(fk_tuples)
    _ck(_lib.lib().hj_create_relation_fk_from_pk(_fn(filename), p, fk_tuples, pk.ctypes.data_as(C.c_void_p), len(pk)),
        "create_relation_fk_from_pk")
    return a


def create_relation_n(in_relation, n):
    src = np.ascontiguousarray(in_relation, np.int32)
    a, p = _out(len(src) * n)
    _ck(_lib.lib().hj_create_relation_n(src.ctypes.data_as(C.c_void_p), p, len(src), n), "create_relation_n")
    return a


def readFromFile(filename, num_tuples):
    a, p = _out(num_tuples)
    _ck(_lib.lib().hj_read_relation(filename.encode(), p, num_tuples), "readFromFile(%s)" % filename)
    return a


def writeToFile(filename, relation):
    r = np.ascontiguousarray(relation, np.int32)
    _ck(_lib.lib().hj_write_relation(filename.encode(), r.ctypes.data_as(C.c_void_p), len(r)), "writeToFile")
