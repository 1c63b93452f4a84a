"""Event-driven model of the dataflow Cholesky (csrc/cholflow.hip) for designing task orders on the CPU.

Slots = resident workgroups, two per CU; every slot draws the next ticket of the task list when it is free.  A task: load
the tile, apply its block columns in order as their operand tiles become final, finish (diagonal tile -> the diagonal
kernel, a serial server; off-diagonal -> wait for inv(L_jj), multiply, store; partial visit -> store).  A step costs
STEP_PAIRED us when the other workgroup of the CU is computing at that moment, STEP_ALONE otherwise (the matrix pipe is
shared).  Calibrated against scripts/flow_trace.py on MI355X (order 15104: 22.7 ms, column advance 122 / 251 / 95 us
early / middle / late).   python scripts/flow_sim.py [nb=118]
"""
import heapq
import sys
from collections import defaultdict

import numpy as np

FIN = 1 << 20
STEP_PAIRED, STEP_ALONE = 31.5, 17.0
LOAD, STORE, TRSM_EXTRA, DIAG, HOP = 5.0, 6.0, 4.0, 58.0, 3.0


def schedule_left(nb, rows, w=1):
    t = []
    for j0 in range(0, nb, w):
        j1 = min(nb, j0 + w)
        for j in range(j0, j1):
            for i in range(j, j1):
                t.append((i, j, 0, j | FIN))
        for i in range(j1, rows):
            for j in range(j0, j1):
                t.append((i, j, 0, j | FIN))
    return t


def schedule_hybrid(nb, rows, E, pw=4, spread=1.0):
    """The first E panels of pw columns right-looking (partial visits of the trailing tiles, interleaved with the next panel's
    columns), the rest left-looking."""
    t = []

    def panel_cols(s):          # tasks of panel s, one list per column
        k0 = pw * s
        return [[(i, c, k0, c | FIN) for i in range(c, rows)] for c in range(pw * s, min(nb, pw * (s + 1)))]

    for col in panel_cols(0):
        t += col
    for s in range(E):
        k0, k1 = pw * s, pw * (s + 1)
        nxt0, nxt1 = pw * (s + 1), min(nb, pw * (s + 2))
        if k1 >= nb:
            break
        A = [(i, j, k0, k1) for j in range(nxt0, nxt1) for i in range(j, rows)]
        B = [(i, j, k0, k1) for i in range(nxt1, rows) for j in range(nxt1, min(i, nb - 1) + 1)]
        t += A
        cols = panel_cols(s + 1) if s + 1 < E else [[(i, c, k1, c | FIN) for i in range(c, rows)] for c in range(nxt0, nxt1)]
        n = len(cols)
        per = int(len(B) / n * spread) if n else len(B)
        pos = 0
        for c in range(n):
            t += cols[c]
            take = B[pos:pos + per] if c < n - 1 else B[pos:]
            t += take
            pos += len(take)
    start = pw * (E + 1)
    for j in range(start, nb):
        for i in range(j, rows):
            t.append((i, j, pw * E, j | FIN))
    return t


def simulate(tasks, nb, rows, slots=510, verbose=False, label=""):
    done = {}                       # (i, k) -> time final
    applied = {}                    # (i, j) -> (k, time)
    waiters = defaultdict(list)     # key -> [slot]
    computing = [False] * slots     # slot is in a step right now (for the partner's speed)
    state = [None] * slots          # per slot: dict of the running task
    ev = [(0.0, s) for s in range(slots)]
    heapq.heapify(ev)
    nxt = 0
    diag_free = 0.0
    busy = wait = 0.0
    end = 0.0
    col_end = np.zeros(nb)

    def flag_set(key, t):
        for s in waiters.pop(key, []):
            heapq.heappush(ev, (t + HOP, s))

    def set_done(i, k, t):
        done[(i, k)] = t
        flag_set(("d", i, k), t)
        end_t[0] = max(end_t[0], t)

    end_t = [0.0]
    pending_diag = []
    while ev:
        t, s = heapq.heappop(ev)
        st = state[s]
        computing[s] = False
        if st is None:
            if nxt >= len(tasks):
                continue
            i, j, k0, w = tasks[nxt]
            nxt += 1
            st = state[s] = dict(i=i, j=j, k=k0, k0=k0, k1=w & (FIN - 1), fin=bool(w & FIN), phase="load", t0=t, wait_from=None)
        if st["wait_from"] is not None:
            wait += t - st["wait_from"]
            st["wait_from"] = None
        i, j = st["i"], st["j"]
        if st["phase"] == "load":
            if st["k0"] > 0:
                a = applied.get((i, j))
                if a is None or a[0] < st["k0"] or a[1] > t:
                    if a is not None and a[0] >= st["k0"]:
                        heapq.heappush(ev, (a[1] + HOP, s))
                    else:
                        waiters[("a", i, j, st["k0"])].append(s)
                    st["wait_from"] = t
                    continue
            st["phase"] = "upd"
            heapq.heappush(ev, (t + LOAD, s))
            continue
        if st["phase"] == "upd":
            k = st["k"]
            if k < st["k1"]:
                need = [(i, k), (j, k)]
                blocked = False
                for key in need:
                    d = done.get(key)
                    if d is None:
                        waiters[("d",) + key].append(s)
                        blocked = True
                        break
                    if d > t:
                        heapq.heappush(ev, (d + HOP, s))
                        blocked = True
                        break
                if blocked:
                    st["wait_from"] = t
                    continue
                dur = STEP_PAIRED if computing[s ^ 1] else STEP_ALONE
                computing[s] = True
                busy += dur
                st["k"] = k + 1
                heapq.heappush(ev, (t + dur, s))
                continue
            # updates done
            if not st["fin"]:
                tt = t + STORE
                applied[(i, j)] = (st["k1"], tt)
                flag_set(("a", i, j, st["k1"]), tt)
                state[s] = None
                heapq.heappush(ev, (tt, s))
                continue
            if i == j:
                tt = t + STORE
                d0 = max(tt + HOP, diag_free)
                diag_free = d0 + DIAG
                set_done(j, j, diag_free)
                state[s] = None
                heapq.heappush(ev, (tt, s))
                continue
            st["phase"] = "solve"
            heapq.heappush(ev, (t + STORE, s))
            continue
        if st["phase"] == "solve":
            d = done.get((j, j))
            if d is None:
                waiters[("d", j, j)].append(s)
                st["wait_from"] = t
                continue
            if d > t:
                heapq.heappush(ev, (d + HOP, s))
                st["wait_from"] = t
                continue
            dur = (STEP_PAIRED if computing[s ^ 1] else STEP_ALONE) + TRSM_EXTRA + STORE
            computing[s] = True
            busy += dur
            st["phase"] = "final"
            heapq.heappush(ev, (t + dur, s))
            continue
        if st["phase"] == "final":
            set_done(i, j, t)
            col_end[j] = max(col_end[j], t)
            state[s] = None
            heapq.heappush(ev, (t, s))
            continue
    end = max(end_t[0], diag_free)
    if verbose:
        d = np.diff(col_end[:-1])
        print(f"{label:44s} span {end / 1e3:6.2f} ms  busy {busy / (end * slots):.2f} wait {wait / (end * slots):.2f}  "
              f"column advance first16 {d[:16].mean():4.0f} mid {d[nb // 2 - 8:nb // 2 + 8].mean():4.0f} last16 {d[-16:].mean():4.0f}  tasks {len(tasks)}")
    return end


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 118
    rows = nb + 1
    simulate(schedule_left(nb, rows, 1), nb, rows, verbose=True, label="left-looking column-major")
    simulate(schedule_left(nb, rows, 4), nb, rows, verbose=True, label="left-looking, 4 columns interleaved")
    for E in (2, 4, 6, 8, 12):
        for spread in (1.0,):
            simulate(schedule_hybrid(nb, rows, E, 4, spread), nb, rows, verbose=True, label=f"hybrid: {E} panels of 4 right-looking")
    for E in (2, 4):
        simulate(schedule_hybrid(nb, rows, E, 8), nb, rows, verbose=True, label=f"hybrid: {E} panels of 8 right-looking")


def schedule_prefill(nb, rows, G=4, D=6, slots=510, chain=125.0, cmax=40, frac=1.0, order="near"):
    """Column-major left-looking, with the slot time the chain leaves idle in the early columns filled by partial visits:
    after the tasks of column c, tiles of columns >= c + D get the block columns [applied, b) that are final by then
    (b = multiple of G, b <= c), as many tiles as fit into the idle time of one chain link."""
    t = []
    applied = {}
    for c in range(nb):
        for i in range(c, rows):
            t.append((i, c, applied.get((i, c), 0), c | FIN))
        if c >= cmax:
            continue
        b = (c // G) * G
        if b <= 0:
            continue
        own = (rows - c) * max(c - 0, 1) * STEP_PAIRED          # this column's own work (upper estimate)
        idle = max(0.0, slots * chain - own) * frac
        cand = []
        cols = range(c + D, nb) if order == "near" else range(nb - 1, c + D - 1, -1)
        for j in cols:
            for i in range(j, rows):
                a = applied.get((i, j), 0)
                if a < b:
                    cand.append((i, j, a))
            if len(cand) * (STEP_PAIRED * G + 12) > idle:
                break
        for (i, j, a) in cand:
            cost = (b - a) * STEP_PAIRED + 12
            if idle < cost:
                break
            idle -= cost
            t.append((i, j, a, b))
            applied[(i, j)] = b
    return t


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 118
    rows = nb + 1
    print("---- prefill")
    for G in (2, 4, 8):
        for D in (4, 8):
            for frac in (0.5, 1.0, 1.5):
                for cmax in (24, 40):
                    simulate(schedule_prefill(nb, rows, G, D, cmax=cmax, frac=frac), nb, rows, verbose=True, label=f"prefill G={G} D={D} frac={frac} cmax={cmax}")
