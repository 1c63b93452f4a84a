"""Stand-alone timing (and, for the dataflow factorisation, the per-task timeline) of DenseSolver::potrf on a synthetic SPD
matrix: python scripts/flow_trace.py [n=15104] [reps=5] [trace=1].  JAICOV_POTRF_LEGACY=1 times the stream-scheduled one."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine

L = engine.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15104
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
want_trace = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nb = n // 128
cap = (nb * (nb + 3) // 2 + 8) * 8 * 4
tr = np.zeros(cap, np.int64)
ms = np.zeros(reps)
nt = C.c_int(0)
L.jaicov_debug_potrf_bench.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_int)]
rc = L.jaicov_debug_potrf_bench(n, reps, ms.ctypes.data, tr.ctypes.data if want_trace else None, cap, C.byref(nt))
assert rc == 0, rc
flops = n ** 3 / 3.0
print(f"n={n} nb={nb} tasks={nt.value} ms per factorisation: {np.round(ms, 3)}  best {flops / ms.min() / 1e9:.1f} TFLOP/s  median {flops / np.median(ms) / 1e9:.1f}")
if want_trace and nt.value:
    t = tr[: nt.value * 8].reshape(-1, 8)
    t0 = t[:, 0].min()
    us = (t[:, :4] - t0) / 100.0
    wait = t[:, 4] / 100.0
    span = us[:, 3].max()
    busy = (us[:, 3] - us[:, 0]).sum()
    slots = len(np.unique(t[:, 6]))
    print(f"span {span:.0f} us, workgroups seen {slots}, sum of task time {busy / slots:.0f} us per workgroup, of which waiting {wait.sum() / slots:.0f} us")
    # tasks are column-major: find the diagonal tasks = first task of each column
    idx = 0
    col_end = []
    for j in range(nb):
        cnt = nb + 1 - j
        col_end.append(us[idx:idx + cnt, 3].max())
        idx += cnt
    col_end = np.array(col_end)
    d = np.diff(col_end)
    print("column completion (us) every 8th:", np.round(col_end[::8]).astype(int))
    print(f"per-column advance: first 16 mean {d[:16].mean():.1f} us, middle mean {d[nb // 2 - 8: nb // 2 + 8].mean():.1f}, last 16 mean {d[-16:].mean():.1f}")
    q = np.linspace(0, span, 11)
    act = [(np.minimum(us[:, 3], b) - np.maximum(us[:, 0], a)).clip(0).sum() / (b - a) for a, b in zip(q[:-1], q[1:])]
    wt = []
    print("mean resident tasks per tenth of the span:", np.round(act, 0))
    pro = us[:, 1] - us[:, 0]
    print(f"C-tile load: mean {pro.mean():.1f} us p90 {np.percentile(pro, 90):.1f};  finish (after updates): mean {(us[:, 3] - us[:, 2]).mean():.1f} us")
