"""Per-workgroup timeline of the trailing-update GEMM (debug export jaicov_debug_gemm_trace)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine
L = engine.load_library()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lower = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tm = M // 128; tiles = tm * (tm + 1) // 2 if lower else tm * tm
out = np.zeros((tiles, 8), np.int64)
L.jaicov_debug_gemm_trace.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
rc = L.jaicov_debug_gemm_trace(M, K, lower, out.ctypes.data)
assert rc == 0, rc
t0 = out[:, 0].min()
us = (out[:, :4] - t0) / 100.0
print(f"tiles={tiles} span={us[:,3].max():.1f} us")
pro = us[:, 1] - us[:, 0]; loop = us[:, 2] - us[:, 1]; epi = us[:, 3] - us[:, 2]
for name, v in (("prologue", pro), ("loop", loop), ("epilogue", epi), ("total", us[:, 3] - us[:, 0])):
    print(f"{name:9s} mean={v.mean():7.2f} p10={np.percentile(v,10):7.2f} p50={np.percentile(v,50):7.2f} p90={np.percentile(v,90):7.2f} max={v.max():7.2f}")
mhz = out[:, 6] / np.maximum(out[:, 2] - out[:, 1], 1) * 100.0
print(f"shader clock in the k loop: mean {mhz.mean():.0f} MHz  p10 {np.percentile(mhz,10):.0f}  p90 {np.percentile(mhz,90):.0f}")
hw = out[:, 4]; xcc = out[:, 5] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 10 + cu
uniq = np.unique(key)
print("distinct CU keys:", len(uniq))
# timeline of the busiest CU
k0 = uniq[0]
sel = np.where(key == k0)[0]
sel = sel[np.argsort(us[sel, 0])]
for i in sel[:20]:
    print(f"  wg {i:5d} start={us[i,0]:8.1f} loop={us[i,1]:8.1f}..{us[i,2]:8.1f} end={us[i,3]:8.1f}")
# start-time histogram: are workgroups in lockstep?
st = np.sort(us[:, 0])
print("start times (every 64th):", np.round(st[::64][:40], 1))
