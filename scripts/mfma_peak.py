import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine
import numpy as np
L = engine.load_library()
L.jaicov_debug_mfma_peak.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
for blocks, iters in [(1024, 4000), (2048, 20000), (1024, -4000), (2048, -20000), (2048, -60000)]:
    ms = C.c_double(); tf = C.c_double()
    L.jaicov_debug_mfma_peak(blocks, iters, C.byref(ms), C.byref(tf))
    print(f"blocks={blocks} iters={iters}: {ms.value:.2f} ms  {tf.value:.1f} TFLOP/s fp64 MFMA")
