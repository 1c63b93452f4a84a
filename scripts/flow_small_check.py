"""Debugging aid: DenseSolver::potrf on small orders against LAPACK, tile by tile (dataflow factorisation forced):
python scripts/flow_small_check.py 384 640.  Prints max |L - L_ref| per 128 x 128 tile of the lower triangle."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("JAICOV_FLOW_MIN_BLOCKS", "1")
import numpy as np
from bundle_adjustment_amd import engine

L = engine.load_library()
L.jaicov_debug_potrf_factor.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
for n in [int(a) for a in sys.argv[1:]] or [384]:
    rng = np.random.default_rng(n + 1)
    G = rng.normal(size=(n, n + 20))
    S = G @ G.T / n + np.eye(n)
    out = np.zeros((n, n))
    rc = L.jaicov_debug_potrf_factor(n, np.ascontiguousarray(S).ctypes.data, out.ctypes.data)
    ref = np.linalg.cholesky(S)
    nb = n // 128
    print(n, "rc", rc)
    for i in range(nb):
        print("  " + " ".join("%8.1e" % np.abs(np.tril(out)[128 * i:128 * i + 128, 128 * j:128 * j + 128] - ref[128 * i:128 * i + 128, 128 * j:128 * j + 128]).max() for j in range(i + 1)), flush=True)
