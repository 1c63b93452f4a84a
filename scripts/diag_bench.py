import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine
L = engine.load_library()
L.jaicov_debug_diag_bench.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
for dbg, name in [(0, "full"), (1, "no chol16"), (2, "no panels (load/store+inverse only)"), (4, "no inverse phase"), (7, "I/O only"), (3, "inverse+io"), (5, "mfma panels + io")]:
    ms = C.c_double()
    L.jaicov_debug_diag_bench(dbg, 200, C.byref(ms))
    L.jaicov_debug_diag_bench(dbg, 200, C.byref(ms))
    print(f"dbg={dbg} {name:40s} {ms.value*1e3:.1f} us/launch (incl. 128 KB H2D copy)")
