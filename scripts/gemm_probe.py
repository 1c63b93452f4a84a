import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine
import numpy as np
rng = np.random.default_rng(0)
for (M, N, K) in [(8192, 8192, 128), (8192, 8192, 256), (8192, 8192, 512), (8192, 8192, 1024), (8192, 8192, 2048)]:
    A = rng.normal(size=(M, K)); B = rng.normal(size=(N, K)); Cm = np.zeros((M, N))
    for beta in (0.0, 1.0):
        _, ms = engine.dense_gemm(0, 0, A, B, Cm, M, N, K, alpha=-1.0, beta=beta, repeats=5)
        print(f"gemm {M}x{N}x{K} beta={beta}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
