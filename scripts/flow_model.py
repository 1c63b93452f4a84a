"""List-scheduling model of the dataflow factorisation's task list (cholflow.hip, chain form) with round 4's measured costs: what the
span is sensitive to, and whether early partial visits of the late columns' tiles (any placement) shorten it.
    python scripts/flow_model.py [nb=118]        (CPU, seconds)
Tickets are drawn in list order by whichever of the resident workgroups is free; a task applies its block-column steps one after the other,
each as soon as both operand tiles are final; the chain workgroup is a serial server with a fixed cost per block column.  The measured
factorisation (22.1 ms at nb = 118) has ~1.5 ms that this model does not: the product with inv(L_jj) at the shared matrix pipe, flag hops,
fragmented runs."""
import heapq
import sys

import numpy as np

P = dict(STEP=31.5,     # us per 128^3 update step with two workgroups per CU (scripts/flow_trace.py: 33.97 incl. waits, columns 41-79)
         CLOAD=5.0, CSTORE=5.0, FINISH=12.0,
         CHAIN=60.0,    # potrf 27 + solve 16 + update 11 + hand-over
         INV=25.0,      # inverse of the diagonal block after its potrf
         WGS=496)


def base_task(i, j, nb):
    if i == j:
        return max(j - 1, 0), False
    if i == j + 1 and i < nb:
        return j, False
    return j, True


def schedule(nb, rows, split=0, frac=None, j_from=0):
    """the product's list; split > 0: tile (i, j >= j_from) gets early partial visits of `split` steps, the visit [a, a + split) placed after
    block column a + split - 1 (frac None) or a fraction `frac` of the way from there to column j"""
    cols = [[] for _ in range(nb)]
    extra = [[] for _ in range(nb)]
    for j in range(nb):
        for i in range(j, rows):
            k1, fin = base_task(i, j, nb)
            a = 0
            if split and j >= j_from:
                while k1 - a > split + split // 2:
                    first = a + split - 1
                    at = first if frac is None else first + int(frac * (j - 1 - first))
                    if at >= j - 1:
                        break
                    extra[at].append((i, j, a, a + split, False))
                    a += split
            cols[j].append((i, j, a, k1, fin))
    tasks = []
    for j in range(nb):
        tasks += cols[j] + extra[j]
    return tasks


def simulate(nb, rows, tasks, p=P):
    done = np.full((rows + 1, nb), np.inf)
    applied, part = {}, {}
    diag_done = np.full(nb, np.inf)
    inv_done = np.full(nb, np.inf)
    free = [0.0] * p["WGS"]
    heapq.heapify(free)
    diag_done[0] = 30.0
    inv_done[0] = diag_done[0] + p["INV"]
    state = {"at": 0}

    def chain_to(c):
        while state["at"] < c:
            n = state["at"] + 1
            t = max(diag_done[n - 1], part[(n, n - 1)], part[(n, n)])
            done[n][n - 1] = t + 17.0
            diag_done[n] = t + p["CHAIN"]
            inv_done[n] = diag_done[n] + p["INV"]
            state["at"] = n

    end = 0.0
    for (i, j, k0, k1, fin) in tasks:
        t = heapq.heappop(free)
        if k0 > 0:
            t = max(t, applied[(i, j)][1])
        t += p["CLOAD"]
        for k in range(k0, k1):
            if done[i][k] == np.inf or done[j][k] == np.inf:
                chain_to(min(k + 1, nb - 1))
            t = max(t, done[i][k], done[j][k]) + p["STEP"]
        if fin:
            chain_to(j)
            t = max(t + p["CSTORE"], inv_done[j]) + p["FINISH"]
            done[i][j] = t
        else:
            t += p["CSTORE"]
            applied[(i, j)] = (k1, t)
            if (i == j and k1 == max(j - 1, 0)) or (i == j + 1 and k1 == j):
                part[(i, j)] = t
        end = max(end, t)
        heapq.heappush(free, t)
    chain_to(nb - 1)
    return max(end, diag_done[nb - 1]), diag_done


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 118
    rows = nb + 1
    base = schedule(nb, rows)
    steps = sum(t[3] - t[2] for t in base)
    e, dd = simulate(nb, rows, base)
    print(f"nb={nb}: {len(base)} tasks, {steps} steps = {steps * P['STEP'] / P['WGS'] / 1000:.2f} ms of work per workgroup; span {e / 1000:.2f} ms")
    print("chain reaches every 8th block column at (us):", np.round(dd[::8]).astype(int))
    print("sensitivity of the span (ms):")
    for name, vals in (("CHAIN", (60.0, 40.0, 20.0)), ("INV", (25.0, 5.0)), ("FINISH", (12.0, 4.0)), ("STEP", (31.5, 25.0)), ("WGS", (496, 992))):
        for v in vals:
            q = dict(P); q[name] = v
            print(f"  {name} = {v}: {simulate(nb, rows, base, q)[0] / 1000:.2f}")
    print("early partial visits of `split` steps (placement: right behind the last column they need / part of the way to their own column):")
    for split in (16, 32, 48):
        for frac in (None, 0.3, 0.5, 0.7):
            for jf in (0, 64):
                t = schedule(nb, rows, split, frac, jf)
                print(f"  split {split}, placement {frac}, tiles of columns >= {jf}: {len(t)} tasks, span {simulate(nb, rows, t)[0] / 1000:.2f} ms")
