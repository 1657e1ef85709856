"""One library, one workload, no package import: ctypes on the entry points rounds 5 and 6 share (hj_create, hj_device_malloc, hj_gen_*,
hj_bind_device, hj_join, hj_join_count, hj_join_materialize).  Usage: round_ab.py <libhj.so> <uniform LOG2N | zipf LOG2R LOG2S> [steps]
Prints ms per hj_join step (wall clock around `steps` calls; the call returns the count, so it is synchronous) and ms per materialising probe."""
import ctypes as C, sys, time
L = C.CDLL(sys.argv[1])
vp, u64 = C.c_void_p, C.c_uint64
L.hj_create.argtypes = [C.POINTER(vp), C.c_int]
L.hj_device_malloc.argtypes = [vp, C.POINTER(vp), u64]
L.hj_gen_unique.argtypes = [vp, vp, u64, u64, u64, u64]
L.hj_gen_zipf.argtypes = [vp, vp, u64, u64, u64, C.c_double, u64]
L.hj_fill_payload.argtypes = [vp, vp, u64, C.c_int, u64]
L.hj_bind_device.argtypes = [vp, C.c_int, vp, vp, u64]
L.hj_join.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
L.hj_join_count.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
L.hj_join_materialize.argtypes = [vp, vp, vp, vp, u64, C.POINTER(u64)]
L.hj_sync.argtypes = [vp]
L.hj_error.argtypes = [vp]; L.hj_error.restype = C.c_char_p
h = vp()
assert L.hj_create(C.byref(h), 0) == 0
def ck(rc):
    if rc: raise SystemExit("hj error %d: %s" % (rc, L.hj_error(h).decode()))
def dmalloc(nbytes):
    p = vp(); ck(L.hj_device_malloc(h, C.byref(p), nbytes)); return p
kind = sys.argv[2]
if kind == "uniform":
    nR = nS = 1 << int(sys.argv[3]); rest = sys.argv[4:]
else:
    nR, nS = 1 << int(sys.argv[3]), 1 << int(sys.argv[4]); rest = sys.argv[5:]
steps = int(rest[0]) if rest else 20
Rk, Rp, Sk, Sp = dmalloc(nR * 4), dmalloc(nR * 4), dmalloc(nS * 4), dmalloc(nS * 4)
if kind == "uniform":
    ck(L.hj_gen_unique(h, Rk, nR, 0, nR, 1)); ck(L.hj_gen_unique(h, Sk, nS, 0, nS, 2))
else:
    ck(L.hj_gen_unique(h, Rk, nR, 0, nR, 3)); ck(L.hj_gen_zipf(h, Sk, nS, 0, nR, 1.0, 4))
ck(L.hj_fill_payload(h, Rp, nR, 0, 0)); ck(L.hj_fill_payload(h, Sp, nS, 0, 0))
ck(L.hj_sync(h))
ck(L.hj_bind_device(h, 0, Rk, Rp, nR)); ck(L.hj_bind_device(h, 1, Sk, Sp, nS))
m, a = u64(), u64()
for _ in range(5): ck(L.hj_join(h, C.byref(m), C.byref(a)))
t0 = time.perf_counter()
for _ in range(steps): ck(L.hj_join(h, C.byref(m), C.byref(a)))
ms = (time.perf_counter() - t0) * 1e3 / steps
# the materialising probe on the partitions the last step left (the round-5 shape of the measurement: partition + count, then one probe)
cap = m.value
ok, pr, ps = dmalloc(cap * 4), dmalloc(cap * 4), dmalloc(cap * 4)
n = u64()
for _ in range(2): ck(L.hj_join_materialize(h, ok, pr, ps, cap, C.byref(n)))
t0 = time.perf_counter()
for _ in range(steps): ck(L.hj_join_materialize(h, ok, pr, ps, cap, C.byref(n)))
mms = (time.perf_counter() - t0) * 1e3 / steps
fused = ""
try:
    f = L.hj_join_and_materialize   # round 6: partition both + the one probe in one call (round 5's equivalent: hj_join + hj_join_materialize)
    f.argtypes = [vp, vp, vp, vp, u64, C.POINTER(u64)]
    for _ in range(3): ck(f(h, ok, pr, ps, cap, C.byref(n)))
    t0 = time.perf_counter()
    for _ in range(steps): ck(f(h, ok, pr, ps, cap, C.byref(n)))
    fused = "   hj_join_and_materialize %.3f ms" % ((time.perf_counter() - t0) * 1e3 / steps)
except AttributeError:
    fused = "   (partition + count + probe = %.3f ms)" % (ms + mms)
print("%s matches %d agg %d  hj_join %.3f ms/step = %.1f Gtuples/s   hj_join_materialize (probe only) %.3f ms  (%d written)"
      % (" ".join(sys.argv[2:5 if kind != "uniform" else 4]), m.value, a.value, ms, (nR + nS) / ms / 1e6, mms, n.value) + fused, flush=True)
