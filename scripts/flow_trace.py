"""Stand-alone timing (and, for the dataflow factorisation, the per-task timeline) of DenseSolver::potrf on a synthetic SPD
matrix: python scripts/flow_trace.py [n=15104] [reps=5] [trace=1].  JAICOV_FACTOR_FORM=streams times the stream-scheduled one."""
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
    ct = tr[nt.value * 8: (nt.value + nb) * 8].reshape(-1, 8)
    if ct[:, 0].any():      # chain kernel: per block column potrf start, factor done, operands there, solve done, update done
        cu = (ct[:, :8] - ct[0, 0]) / 100.0
        per = np.diff(cu[:, 0])
        for name, sl in (("columns 1-15", slice(1, 16)), ("middle 16", slice(nb // 2 - 8, nb // 2 + 8)), ("last 16", slice(nb - 18, nb - 2))):
            print(f"chain workgroup {name}: period {per[sl].mean():.1f} us = potrf {(cu[:, 1] - cu[:, 0])[sl].mean():.1f} + wait for the two tiles {(cu[:, 2] - cu[:, 1])[sl].mean():.1f}"
                  f" + load/solve {(cu[:, 3] - cu[:, 2])[sl].mean():.1f} + update {(cu[:, 4] - cu[:, 3])[sl].mean():.1f}"
                  f"   [potrf: factor {(cu[:, 7] - cu[:, 0])[sl].mean():.1f}, store issue {(cu[:, 1] - cu[:, 7])[sl].mean():.1f}; load/solve: loads+drain {(cu[:, 5] - cu[:, 2])[sl].mean():.1f}, products {(cu[:, 6] - cu[:, 5])[sl].mean():.1f}, store issue {(cu[:, 3] - cu[:, 6])[sl].mean():.1f}]")
        print("chain: potrf start of every 8th column (us):", np.round(cu[::8, 0]).astype(int))
    t_all = tr[: nt.value * 8].reshape(-1, 8)
    # the task list the solver ran (split update ranges: partial-sum tasks sit between the tiles' own tasks, which stay column-major)
    sm, sf = C.c_int(1), C.c_int(0)
    L.jaicov_debug_flow_split(nb, C.byref(sm), C.byref(sf))
    form = os.environ.get("JAICOV_FACTOR_FORM", "")
    second = 2 if (form == "chain3" or (form != "chain2" and nb < 80)) else 0
    tl = np.zeros((nt.value, 4), np.int32)
    L.jaicov_debug_flow_tasks2.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_int]
    ntl = L.jaicov_debug_flow_tasks2(nb, nb + 1, 1, 1, second, sm.value, sf.value, tl.ctypes.data, nt.value)
    is_part = (tl[:, 3] & (1 << 21)) != 0 if ntl == nt.value else np.zeros(nt.value, bool)
    if is_part.any():
        pt = t_all[is_part]
        pus = (pt[:, :4] - t_all[:, 0].min()) / 100.0
        psteps = (tl[is_part, 3] & 0xfff) - (tl[is_part, 2] & 0xfff)
        print(f"split update ranges: {sm.value} pieces from block column {sf.value}: {int(is_part.sum())} partial-sum tasks, {psteps.mean():.1f} steps each, "
              f"{((pus[:, 2] - pus[:, 1]) / psteps).mean():.2f} us per step, waiting {pt[:, 4].mean() / 100.0:.1f} us per task; first starts at {pus[:, 0].min():.0f} us, last ends at {pus[:, 3].max():.0f} us")
    t = t_all[~is_part]
    t0 = t_all[:, 0].min()
    us = (t[:, :4] - t0) / 100.0
    wait = t[:, 4] / 100.0
    span = us[:, 3].max()
    busy = ((t_all[:, 3] - t_all[:, 0]) / 100.0).sum()
    slots = len(np.unique(t_all[:, 6] & 0xffff))
    print(f"span {span:.0f} us, workgroups seen {slots}, sum of task time {busy / slots:.0f} us per workgroup, of which waiting {t_all[:, 4].sum() / 100.0 / slots:.0f} us")
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
    us_all = (t_all[:, :4] - t0) / 100.0
    act = [(np.minimum(us_all[:, 3], b) - np.maximum(us_all[:, 0], a)).clip(0).sum() / (b - a) for a, b in zip(q[:-1], q[1:])]
    wt = []
    print("mean resident tasks per tenth of the span:", np.round(act, 0))
    # which CUs ran the workgroups (HW_ID: cu 11:8, sh 12, se 15:13; XCC id separately)
    hw = t[:, 7] & 0xffffffff; xcc = (t[:, 7] >> 32) & 0xf; key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 10 + ((hw >> 8) & 0xf)
    wg_cu = {}
    runs = (t[:, 6] >> 16) & 0xffff; w_upd = (t[:, 6] >> 32) / 100.0
    for b, k in zip(t[:, 6] & 0xffff, key):
        wg_cu[int(b)] = int(k)
    cus = np.unique(list(wg_cu.values()), return_counts=True)
    print(f"distinct CUs {len(cus[0])}; workgroups per CU histogram {dict(zip(*np.unique(cus[1], return_counts=True)))}; per XCC {np.bincount(np.array(list(wg_cu.values())) // 1000, minlength=8)}")
    per = {}
    for k, c in zip(*cus):
        per.setdefault(int(k) // 1000, []).append(int(c))
    print("per XCC: CUs / CUs with one workgroup / by shader engine (se*2+sh: CUs, workgroups):")
    for x in sorted(per):
        ses = {}
        for k, c in zip(*cus):
            if int(k) // 1000 == x:
                e = (int(k) % 1000) // 10
                ses.setdefault(e, [0, 0]); ses[e][0] += 1; ses[e][1] += int(c)
        print(f"  xcc {x}: {len(per[x])} CUs, {per[x].count(1)} single; " + " ".join(f"{e}:{v[0]}/{v[1]}" for e, v in sorted(ses.items())))
    steps = np.array([nb + 1 - 0] * 0)
    upd_us = (t[:, 2] - t[:, 1]) / 100.0
    mhz = t[:, 5] / np.maximum(t[:, 2] - t[:, 1], 1) * 100.0
    sel = upd_us > 200
    print(f"shader clock in the update phase (tasks > 200 us): mean {mhz[sel].mean():.0f} MHz p10 {np.percentile(mhz[sel], 10):.0f} p90 {np.percentile(mhz[sel], 90):.0f}")
    if True:
        ks = np.concatenate([np.full(nb + 1 - j, j) for j in range(nb)])
        mid = (ks > 40) & (ks < 80)
        fin_us = us[:, 3] - us[:, 2]
        rowi = np.concatenate([np.arange(j, nb + 1) for j in range(nb)])
        offd = mid & (rowi != ks)
        print(f"tiles of columns 41-79: waiting {wait[offd].mean():.0f} us per task, after the updates {fin_us[offd].mean():.0f} us; "
              f"diagonal tiles: start -> handed over {(us[:, 3] - us[:, 0])[mid & (rowi == ks)].mean():.0f} us vs off-diagonal start -> updates done {(us[:, 2] - us[:, 0])[offd].mean():.0f} us")
        # how far behind its column's diagonal tile does a tile start / end its updates?
        starts = np.cumsum([0] + [nb + 1 - j for j in range(nb)])
        lag_s = np.concatenate([us[starts[j]:starts[j + 1], 0] - us[starts[j], 0] for j in range(nb)])
        dready = np.concatenate([np.full(nb + 1 - j, us[starts[j], 3]) for j in range(nb)])
        print(f"   start lag behind the diagonal tile: mean {lag_s[offd].mean():.0f} us, max {lag_s[offd].max():.0f}; updates done relative to diagonal tile handed over: mean {(us[:, 2] - dready)[offd].mean():.0f} us, p10 {np.percentile((us[:, 2] - dready)[offd], 10):.0f}, p90 {np.percentile((us[:, 2] - dready)[offd], 90):.0f}")
        print(f"   runs per task {runs[offd].mean():.1f} (max {runs[offd].max()}), waiting during the updates {w_upd[offd].mean():.0f} us, waiting for inv(L_jj) {(wait - w_upd)[offd].mean():.0f} us")
        d_ij = (rowi - ks)
        for lo, hi in ((1, 2), (2, 8), (8, 30), (30, 200)):
            m2 = offd & (d_ij >= lo) & (d_ij < hi)
            if m2.any():
                print(f"   rows j+{lo}..j+{hi - 1}: waiting during updates {w_upd[m2].mean():.0f} us, runs {runs[m2].mean():.1f}, start lag {lag_s[m2].mean():.0f} us")
        steps = np.maximum(((tl[~is_part, 3] & 0xfff) - (tl[~is_part, 2] & 0xfff)) if ntl == nt.value else ks, 1)
        print(f"update phase per block column, tiles of columns 41-79: {(upd_us[mid] / steps[mid]).mean():.2f} us incl. waits ({(wait[mid] / steps[mid]).mean():.2f} us waiting)")
        # chain: (j,j) handed to the diagonal kernel -> L[j+1][j] final -> (j+1,j+1) handed over
        starts = np.cumsum([0] + [nb + 1 - j for j in range(nb)])
        a = np.array([us[starts[j], 3] for j in range(nb - 1)])
        b = np.array([us[starts[j] + 1, 3] for j in range(nb - 1)])
        c = np.array([us[starts[j + 1], 3] for j in range(nb - 1)])
        for name, sl in (("columns 1-15", slice(1, 16)), ("middle 16", slice(nb // 2 - 8, nb // 2 + 8)), ("last 16", slice(nb - 18, nb - 2))):
            print(f"chain {name}: diag kernel + solve of L[j+1][j] {np.mean((b - a)[sl]):.1f} us, last update + store of (j+1,j+1) {np.mean((c - b)[sl]):.1f} us")
    pro = us[:, 1] - us[:, 0]
    print(f"C-tile load: mean {pro.mean():.1f} us p90 {np.percentile(pro, 90):.1f};  finish (after updates): mean {(us[:, 3] - us[:, 2]).mean():.1f} us")
    if os.environ.get("FLOW_TRACE_WGS"):
        blk = (t[:, 6] & 0xffff).astype(int)
        first = {}
        for row, b in zip(t, blk):
            if b not in first or row[0] < first[b][0]:
                first[b] = (row[0], int(row[7] >> 32) & 0xf, int(row[7]) & 0xffffffff)
        seen = sorted(first)
        missing = [b for b in range(512) if b not in first]
        print("workgroups that never ran a task:", missing)
        late = sorted(first.items(), key=lambda kv: -kv[1][0])[:12]
        print("latest first task starts (workgroup, us after the first, xcc, se, cu):", [(b, round((v[0] - t0) / 100.0), v[1], (v[2] >> 13) & 7, (v[2] >> 8) & 0xf) for b, v in late])
        for x in (0, 1):
            print(f"xcc {x}: workgroups", sorted(b for b, v in first.items() if v[1] == x))
    if os.environ.get("FLOW_TRACE_PATH") and ct[:, 0].any():
        # the dependency path into the chain workgroup's wait, column by column: when did potrf(c) end, when were the finished tiles of block column c - 1
        # that the two tiles need there, when were the two tiles' updates done and stored
        tix = {(int(a), int(b)): q for q, (a, b) in enumerate(zip(rowi, ks))}
        lo, hi = [int(v) for v in os.environ["FLOW_TRACE_PATH"].split(":")]
        print("c: potrf(c) start, end | tile (c+1,c-1): updates done, final | tile (c+1,c): start, C loaded, updates done, stored, runs, waited | diag (c+1,c+1): stored | chain: operands there, solve done, update done   [us after potrf(c) start]")
        cu = (ct[:, :8] - t0) / 100.0      # same origin as the tasks
        for c in range(lo, hi):
            z = cu[c, 0]
            a = tix.get((c + 1, c - 1)); b = tix[(c + 1, c)]; d = tix[(c + 1, c + 1)]
            print(f"{c}: 0 {cu[c, 1] - z:.0f} | {us[a, 2] - z:.0f} {us[a, 3] - z:.0f} | {us[b, 0] - z:.0f} {us[b, 1] - z:.0f} {us[b, 2] - z:.0f} {us[b, 3] - z:.0f} runs {runs[b]} waited {wait[b]:.0f} | {us[d, 3] - z:.0f} | {cu[c, 2] - z:.0f} {cu[c, 3] - z:.0f} {cu[c, 4] - z:.0f}")
    if os.environ.get("FLOW_TRACE_ROW") and ct[:, 0].any():
        # one block row near the end: link by link, when is L[r][k] final, when has tile (r, k+1) its last update, when is potrf(k+1) done
        r, lo, hi = [int(v) for v in os.environ["FLOW_TRACE_ROW"].split(":")]
        print(f"row {r}: k | potrf(k) done | tile (r,k): updates done, final (finish = wait for the inverse + product + store) | step of (r,k+1) after L[r][k] | after potrf(k) done")
        for k in range(lo, hi):
            a = tix[(r, k)]; b = tix.get((r, k + 1))
            fk = cu[k, 1]
            print(f"  {k}: {fk:.0f} | {us[a, 2]:.0f} {us[a, 3]:.0f} (finish {us[a, 3] - us[a, 2]:.0f}) | {(us[b, 2] - us[a, 3]) if b is not None else 0:.0f} | updates done {us[a, 2] - fk:.0f}, final {us[a, 3] - fk:.0f} after potrf(k)")
