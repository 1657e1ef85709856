"""ctypes binding of libhj.so (include/hj.h).  No fallback: if the library is missing it is built
with hipcc, and if that fails the import fails."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# HJ_ASAN=1 (CPU test leg only, tests/test_asan.py): the host-only AddressSanitizer/UBSan build of the plain-C++ half of the
# library (generator drop-in, host write-combining split, shard function) and of the driver (`make asan`).  It has no GPU
# entry points: only the symbols it exports are bound.
ASAN = os.environ.get("HJ_ASAN") == "1"
LIB_PATH = os.path.join(_HERE, "libhj_host_asan.so" if ASAN else "libhj.so")
BENCH_PATH = os.path.join(_HERE, "bench_asan" if ASAN else "bench")

i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class Config(C.Structure):
    _fields_ = [("bits1", C.c_uint32), ("bits2", C.c_uint32), ("force_bits", C.c_uint32),
                ("build_side", C.c_uint32), ("lds_capacity", C.c_uint32), ("lds_heads", C.c_uint32),
                ("probe_chunk", C.c_uint32), ("exact_only", C.c_uint32), ("reserved1", C.c_uint32),
                ("reserved0", C.c_uint32), ("graph", C.c_uint32), ("reserved", C.c_uint32 * 5)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint32), ("total_ms", C.c_float),
                ("last_ms", C.c_float)]


class Args(C.Structure):  # include/hj_reference_abi.h  (src/common-host.h:39-52)
    _fields_ = [("S", i32p), ("S_els", C.c_size_t), ("S_filename", C.c_char * 50),
                ("R", i32p), ("R_els", C.c_size_t), ("R_filename", C.c_char * 50),
                ("threadsNum", C.c_int), ("sharedMem", C.c_uint), ("pivotsNum", C.c_uint)]


class DistConfig(C.Structure):  # include/hj_dist.h
    _fields_ = [("slices", C.c_uint32), ("exact_only", C.c_uint32), ("self_via_link", C.c_uint32), ("phantom_world", C.c_uint32),
                ("single_group", C.c_uint32), ("balance_size", C.c_uint32), ("timeout_ms", C.c_uint32), ("reserved", C.c_uint32)]


class DistStats(C.Structure):
    _fields_ = [("received", C.c_uint64 * 2), ("link_bytes", C.c_uint64), ("payload_bytes", C.c_uint64), ("path", C.c_uint32),
                ("slices", C.c_uint32), ("spans_per_slice", C.c_uint32), ("slot_capacity", C.c_uint32 * 2),
                ("split_ms", C.c_float * 2), ("pass1_ms", C.c_float * 2), ("pass2_join_ms", C.c_float),
                ("first_split_ms", C.c_float), ("last_pass1_ms", C.c_float), ("wall_ms", C.c_float), ("early_pass2_join_ms", C.c_float), ("probe_groups", C.c_uint32),
                ("balanced", C.c_uint32), ("exchange_ms", C.c_float), ("materializing", C.c_uint32), ("materialized", C.c_uint64),
                ("reserved", C.c_uint32 * 1)]


class LastResult(C.Structure):
    _fields_ = [("matches", C.c_ulonglong), ("agg", C.c_ulonglong), ("materialized", C.c_ulonglong),
                ("partition_ms", C.c_double * 2), ("join_ms", C.c_double * 2), ("status", C.c_int)]


# every symbol include/hj.h and include/hj_reference_abi.h declare: (restype, argtypes)
SIGNATURES = {
    "hj_version": (C.c_char_p, []),
    "hj_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "hj_destroy": (C.c_int, [vp]),
    "hj_error": (C.c_char_p, [vp]),
    "hj_set_stream": (C.c_int, [vp, vp]),
    "hj_configure": (C.c_int, [vp, C.POINTER(Config)]),
    "hj_get_config": (C.c_int, [vp, C.POINTER(Config)]),
    "hj_sync": (C.c_int, [vp]),
    "hj_load_host": (C.c_int, [vp, C.c_int, vp, vp, C.c_uint64, C.c_int]),
    "hj_bind_device": (C.c_int, [vp, C.c_int, vp, vp, C.c_uint64]),
    "hj_partition": (C.c_int, [vp, C.c_int]),
    "hj_partition_both": (C.c_int, [vp]),
    "hj_shard_count": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, u64p]),
    "hj_shard_split_ordered": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), vp, vp, u64p]),
    "hj_enable_timings": (C.c_int, [vp, C.c_int]),
    "hj_last_call_breakdown": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hj_join_stream_probe_materialize": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_int, vp, vp, vp, C.c_uint64, u64p, u64p]),
    "hj_host_split": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, u64p, C.POINTER(C.c_double)]),
    "hj_coprocess_numa": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hj_host_split_throughput": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "hj_host_split_blocks_capacity": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32]),
    "hj_host_split_blocks": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p,
                                       C.POINTER(C.c_double)]),
    "hj_coprocess_groups": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
    "hj_host_join": (C.c_int, [vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_uint32, u64p, u64p, C.POINTER(C.c_double)]),
    "hj_ubench": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), u64p]),
    "hj_partition_layout": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "hj_join_count": (C.c_int, [vp, u64p, u64p]),
    "hj_join_materialize": (C.c_int, [vp, vp, vp, vp, C.c_uint64, u64p]),
    "hj_join": (C.c_int, [vp, u64p, u64p]),
    "hj_join_and_materialize": (C.c_int, [vp, vp, vp, vp, C.c_uint64, u64p]),
    "hj_hot_stats": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_double), u64p]),
    "hj_reload_knobs": (C.c_int, [vp]),
    "hj_debug_set_stamps": (C.c_int, [vp, vp, vp]),
    "hj_join_late_materialize": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64, vp, C.c_uint32, C.c_uint64, u64p, u64p]),
    "hj_join_nonpartitioned": (C.c_int, [vp, C.c_int, u64p, u64p]),
    "hj_join_stream_probe": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_int, u64p, u64p]),
    "hj_join_coprocess": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]),
    "hj_device_malloc": (C.c_int, [vp, C.POINTER(vp), C.c_uint64]),
    "hj_device_free": (C.c_int, [vp, vp]),
    "hj_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "hj_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "hj_get_partitions": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u64p]),
    "hj_timings_reset": (C.c_int, [vp]),
    "hj_timings": (C.c_int, [vp, C.POINTER(KernelTime), C.c_uint32, C.POINTER(C.c_uint32)]),
    "hj_shard_split": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, u64p]),
    "hj_shard_of": (C.c_uint32, [C.c_int32, C.c_uint32]),
    "hj_gen_unique": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "hj_gen_zipf": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_uint64]),
    "hj_fill_payload": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_uint64]),
    "hj_digest_pairs": (C.c_int, [vp, vp, vp, C.c_uint64, u64p]),
    "hj_digest_triples": (C.c_int, [vp, vp, vp, vp, C.c_uint64, u64p]),
    "hj_verify_partitions": (C.c_int, [vp, C.c_int, u64p, vp]),
    "hj_gen_set_seed": (None, [C.c_uint64]),
    "hj_create_relation_unique": (C.c_int, [C.c_char_p, vp, C.c_uint64, C.c_int64]),
    "hj_create_relation_nonunique": (C.c_int, [C.c_char_p, vp, C.c_uint64, C.c_int64]),
    "hj_create_relation_zipf": (C.c_int, [C.c_char_p, vp, C.c_uint64, C.c_int64, C.c_double]),
    "hj_create_relation_fk_from_pk": (C.c_int, [C.c_char_p, vp, C.c_uint64, vp, C.c_uint64]),
    "hj_create_relation_n": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64]),
    "hj_read_relation": (C.c_int, [C.c_char_p, vp, C.c_uint64]),
    "hj_write_relation": (C.c_int, [C.c_char_p, vp, C.c_uint64]),
    "hj_dist_create": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]),
    "hj_dist_create_transport": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.c_char_p]),
    "hj_dist_set_transport": (C.c_int, [vp, C.c_char_p]),
    "hj_dist_destroy": (C.c_int, [vp]),
    "hj_dist_error": (C.c_char_p, [vp]),
    "hj_dist_world": (C.c_int, [vp]),
    "hj_dist_transport": (C.c_char_p, [vp]),
    "hj_dist_context": (vp, [vp, C.c_int]),
    "hj_dist_configure": (C.c_int, [vp, C.POINTER(DistConfig)]),
    "hj_dist_bind": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_uint64]),
    "hj_dist_join": (C.c_int, [vp, u64p, u64p]),
    "hj_dist_bind_output": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_uint64]),
    "hj_dist_join_materialize": (C.c_int, [vp, u64p, u64p, u64p]),
    "hj_dist_get_stats": (C.c_int, [vp, C.c_int, C.POINTER(DistStats)]),
    "hj_dist_unique_id": (C.c_int, [vp]),
    "hj_dist_rank_create": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int, vp]),
    "hj_dist_rank_destroy": (C.c_int, [vp]),
    "hj_dist_rank_error": (C.c_char_p, [vp]),
    "hj_dist_rank_configure": (C.c_int, [vp, C.POINTER(DistConfig)]),
    "hj_dist_rank_join": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, u64p, u64p]),
    "hj_dist_rank_join_materialize": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p, u64p, u64p, u64p]),
    "hj_dist_rank_get_stats": (C.c_int, [vp, C.POINTER(DistStats)]),
    "hj_dist_debug_stall_rank": (C.c_int, [C.c_int]),          # debug symbols (tests): not declared in the headers
    "hj_host_split_debug_progress": (C.c_int, [C.c_int]),
    "hashJoinClusteredProbe": (C.c_uint, [C.POINTER(Args), vp]),
    "hj_reference_last_result": (None, [C.POINTER(LastResult)]),
}

_lib = None


def build(force=False):
    """Compile libhj.so and bench for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not (os.path.exists(LIB_PATH) and os.path.exists(BENCH_PATH)):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "asan" if ASAN else "all"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)   # always the in-tree build: no override
        for name, (res, argt) in SIGNATURES.items():
            if ASAN and not hasattr(_lib, name):
                continue  # a GPU entry point: not part of the host-only sanitizer build
            f = getattr(_lib, name)  # AttributeError if the header and the library disagree
            f.restype = res
            f.argtypes = argt
    return _lib
