"""How many 128-tile GEMM workgroups (256 threads, 73.7 KB LDS, <= 256 VGPRs) are resident per CU / shader engine in an
ordinary (unmasked) launch?  Uses the per-workgroup trace of jaicov_debug_gemm_trace."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine
L = engine.load_library()
M, K = 8192, 2048
tiles = (M // 128) ** 2
out = np.zeros((tiles, 8), np.int64)
L.jaicov_debug_gemm_trace.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
assert L.jaicov_debug_gemm_trace(M, K, 0, out.ctypes.data) == 0
t0 = out[:, 0].min()
s, e = (out[:, 0] - t0) / 100.0, (out[:, 3] - t0) / 100.0
hw = out[:, 4]; xcc = out[:, 5] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 10 + cu
for tq in (100.0, 300.0, 600.0):
    live = (s <= tq) & (e > tq)
    k, c = np.unique(key[live], return_counts=True)
    ses = {}
    for kk, cc in zip(k, c):
        ses.setdefault(int(kk) // 10, [0, 0]); ses[int(kk) // 10][0] += 1; ses[int(kk) // 10][1] += int(cc)
    print(f"t={tq:.0f} us: resident {live.sum()} on {len(k)} CUs; per-CU histogram {dict(zip(*np.unique(c, return_counts=True)))}")
    print("   per (xcc, se, sh): " + " ".join(f"{a}:{v[0]}/{v[1]}" for a, v in sorted(ses.items())[:8]))
