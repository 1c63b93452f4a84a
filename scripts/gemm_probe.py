"""fp64 MFMA GEMM rate against K, with and without the staggered start (exploration helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine
import numpy as np
rng = np.random.default_rng(0)
for (M, N, K) in [(8192, 8192, 512), (8192, 8192, 1024), (8192, 8192, 4096), (12800, 12800, 512)]:
    A = rng.normal(size=(M, K)); B = rng.normal(size=(N, K)); Cm = np.zeros((M, N))
    for stag in (0, 496, 0, 496):
        os.environ["JAICOV_GEMM_STAGGER"] = str(stag)
        for lower in (0, 1):
            _, ms = engine.dense_gemm(0, 0, A, B, Cm, M, N, K, alpha=-1.0, beta=1.0, lower_only=lower, repeats=5)
            fl = (M * (M + 128.0) if lower else 2.0 * M * N) * K
            print(f"gemm {M}x{N}x{K} lower={lower} stagger={stag}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s", flush=True)
