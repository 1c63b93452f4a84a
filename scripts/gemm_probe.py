"""fp64 MFMA GEMM rate: compact against strided operands (as inside the factorisation), K = 512 (exploration helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine
import numpy as np
rng = np.random.default_rng(0)
M = 12800; K = 512
for lda in (512, 15232, 16384, 15232 + 16):
    A = np.zeros((M, lda)); A[:, :K] = rng.normal(size=(M, K))
    Cm = np.zeros((M, 15232 if lda != 512 else M))
    for lower in (0, 1):
        _, ms = engine.dense_gemm(0, 0, A, A, Cm, M, M, K, alpha=-1.0, beta=1.0, lower_only=lower, repeats=100)
        fl = (M * (M + 128.0) if lower else 2.0 * M * M) * K
        print(f"gemm {M}x{M}x{K} lda={lda} ldc={Cm.shape[1]} lower={lower}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s", flush=True)
