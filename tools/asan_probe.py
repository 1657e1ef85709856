import sys, ctypes as C, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as g
p = g.load_package()
print(p._lib.LIB_PATH)
L = p._lib.lib()
keys = np.arange(1000, dtype=np.int32)
outk = np.empty(500, np.int32)   # too small on purpose
off = np.empty(5, np.uint64)
gbs = C.c_double()
L.hj_host_split(keys.ctypes.data_as(C.c_void_p), None, 1000, 4, 2, outk.ctypes.data_as(C.c_void_p), None, off.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(gbs))
print("no error?!")
