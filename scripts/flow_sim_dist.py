"""Event-driven model of the dataflow Cholesky distributed over N GPUs (the design that is not built).

Block columns are dealt cyclically: column j belongs to GPU j mod N, which runs the tasks of its columns (left-looking, in
column order) on its own pool of workgroup slots and its own diagonal-block server.  A finished tile becomes visible to its
owner at once and to every other GPU XGMI microseconds later (peer store of 128 KB over one xGMI link + flag).  Everything
else is scripts/flow_sim.py's model of one GPU (per-task serial block-column steps, two workgroups sharing a CU's matrix
pipe, the diagonal server as a serial resource), with its constants re-calibrated to the chain form on MI355X (order
15 104 on one GPU: 22.0 ms).  The model says what the dependency structure allows, not what a fabric under load delivers.

    python scripts/flow_sim_dist.py [nb=118]
"""
import heapq
import sys
from collections import defaultdict

import numpy as np

STEP_PAIRED, STEP_ALONE = 30.3, 17.0            # one block-column step of a tile (two workgroups per CU / one)
LOAD, STORE, TRSM_EXTRA, HOP = 5.0, 6.0, 4.0, 3.0
DIAG = 26.0                                     # the diagonal server's share of a chain link, set so that the period of a
                                                # chain-bound column on one GPU is the measured 88-90 us
XGMI = 4.0                                      # extra latency until a tile (and its flag) is visible on another GPU


def simulate(nb, rows, ngpu, slots=496, xgmi=XGMI, verbose=True):
    owner = lambda j: j % ngpu
    tasks = [[] for _ in range(ngpu)]
    for j in range(nb):
        for i in range(j, rows):
            tasks[owner(j)].append((i, j))
    done = {}                                    # (i, k) -> (time final, producing gpu)
    waiters = defaultdict(list)
    nxt = [0] * ngpu
    diag_free = [0.0] * ngpu
    computing = [[False] * slots for _ in range(ngpu)]
    state = [[None] * slots for _ in range(ngpu)]
    ev = [(0.0, g, s) for g in range(ngpu) for s in range(slots)]
    heapq.heapify(ev)
    busy = 0.0
    end = 0.0
    col_end = np.zeros(nb)

    def seen(key, g):                            # when GPU g can use tile `key`
        d = done.get(key)
        if d is None:
            return None
        return d[0] + (0.0 if d[1] == g else xgmi)

    def set_done(i, k, t, g):
        nonlocal end
        done[(i, k)] = (t, g)
        end = max(end, t)
        for (gg, s) in waiters.pop((i, k), []):
            heapq.heappush(ev, (t + HOP + (0.0 if gg == g else xgmi), gg, s))

    while ev:
        t, g, s = heapq.heappop(ev)
        st = state[g][s]
        computing[g][s] = False
        if st is None:
            if nxt[g] >= len(tasks[g]):
                continue
            i, j = tasks[g][nxt[g]]
            nxt[g] += 1
            st = state[g][s] = dict(i=i, j=j, k=0, phase="load")
        i, j = st["i"], st["j"]
        if st["phase"] == "load":
            st["phase"] = "upd"
            heapq.heappush(ev, (t + LOAD, g, s))
            continue
        if st["phase"] == "upd":
            k = st["k"]
            if k < j:
                blocked = False
                for key in ((i, k), (j, k)):
                    v = seen(key, g)
                    if v is None:
                        waiters[key].append((g, s)); blocked = True; break
                    if v > t:
                        heapq.heappush(ev, (v + HOP, g, s)); blocked = True; break
                if blocked:
                    continue
                dur = STEP_PAIRED if computing[g][s ^ 1] else STEP_ALONE
                computing[g][s] = True
                busy += dur
                st["k"] = k + 1
                heapq.heappush(ev, (t + dur, g, s))
                continue
            if i == j:                           # to this GPU's diagonal server
                tt = t + STORE
                d0 = max(tt + HOP, diag_free[g])
                diag_free[g] = d0 + DIAG
                set_done(j, j, diag_free[g], g)
                col_end[j] = diag_free[g]
                state[g][s] = None
                heapq.heappush(ev, (tt, g, s))
                continue
            st["phase"] = "solve"
            heapq.heappush(ev, (t + STORE, g, s))
            continue
        if st["phase"] == "solve":
            v = seen((j, j), g)
            if v is None:
                waiters[(j, j)].append((g, s)); continue
            if v > t:
                heapq.heappush(ev, (v + HOP, g, s)); continue
            dur = (STEP_PAIRED if computing[g][s ^ 1] else STEP_ALONE) + TRSM_EXTRA + STORE
            computing[g][s] = True
            busy += dur
            st["phase"] = "final"
            heapq.heappush(ev, (t + dur, g, s))
            continue
        if st["phase"] == "final":
            set_done(i, j, t, g)
            state[g][s] = None
            heapq.heappush(ev, (t, g, s))
            continue
    span = max(end, max(diag_free))
    if verbose:
        d = np.diff(col_end)
        print(f"N = {ngpu}: {span / 1e3:6.2f} ms   matrix pipes busy {busy / (span * slots * ngpu):.2f}   "
              f"column period first 16 {d[:16].mean():4.0f} us, middle {d[nb // 2 - 8:nb // 2 + 8].mean():4.0f}, last 16 {d[-16:].mean():4.0f}")
    return span


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 118
    rows = nb + 1
    base = None
    for n in (1, 2, 4, 8):
        t = simulate(nb, rows, n)
        base = base or t
    print("with 10 us instead of 4 us per cross-GPU hop:")
    for n in (2, 4, 8):
        simulate(nb, rows, n, xgmi=10.0)
