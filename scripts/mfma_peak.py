import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine
import numpy as np
L = engine.load_library()
L.jaicov_debug_mfma_peak.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
for blocks, iters in [(256, 2000), (512, 2000), (1024, 2000), (2048, 4000), (2048, 20000)]:
    ms = C.c_double(); tf = C.c_double()
    L.jaicov_debug_mfma_peak(blocks, iters, C.byref(ms), C.byref(tf))
    print(f"blocks={blocks} iters={iters}: {ms.value:.2f} ms  {tf.value:.1f} TFLOP/s fp64 MFMA")
# big GEMM rates of the production kernel
rng = np.random.default_rng(0)
for (M, N, K) in [(4096, 4096, 512), (8192, 8192, 512), (8192, 8192, 2048), (16384, 16384, 512)]:
    A = rng.normal(size=(M, K)); B = rng.normal(size=(N, K)); Cm = np.zeros((M, N))
    _, ms = engine.dense_gemm(0, 0, A, B, Cm, M, N, K, alpha=-1.0, beta=1.0, repeats=5)
    print(f"gemm NT {M}x{N}x{K} beta=1: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
    _, ms = engine.dense_gemm(0, 0, A, B, Cm, M, N, K, alpha=-1.0, beta=1.0, lower_only=True, repeats=5)
    print(f"   lower-only: {ms:.3f} ms {1.0*M*(N+128)*K/ms/1e9:.1f} TFLOP/s")
