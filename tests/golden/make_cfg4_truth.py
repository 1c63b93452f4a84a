#!/usr/bin/env python
"""Second fixture at the headline size: the EXACT solution (to ~1e-12) of the oracle's own systems, so that forward errors can
be stated.  At config 4 the Jacobi-scaled normal matrix has cond ~ 1e9: any backward-stable fp64 solver -- the reference's
packed Bunch-Kaufman dspsv as much as the engine's Cholesky -- is only accurate to ~1e-7 relative, and two such solvers differ
from each other by that much (tests/golden/cfg4/cfg4_oracle.npz vs the GPU: 2e-8 .. 4e-7).  Here the oracle's solution is
refined with residuals accumulated in x87 extended precision (oracle_residual_ld; NOT part of the reference's algorithm) until it
stops moving; tests/test_gpu_cfg4_golden.py then measures oracle and GPU against that.

    python tests/golden/make_cfg4_truth.py        (about 40 min on one core: two assemblies + two dsptrf; no inverse)

Writes tests/golden/cfg4/cfg4_truth.npz: dx1_true, dx2_true (both passes of make_cfg4_golden.py), 16 exact columns of Qxx of the
final pass, and the oracle's own errors against them.
"""
import concurrent.futures as cf
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_cfg4_golden as g  # noqa: E402  (same scene, same assembly, same column sample)

orc = g.orc
N_QCOLS = 16


def refine(L, U, A0, fac, ipiv, b, x, label, iters=4):
    """x <- x + solve(fac, b - A0 x) with the residual in extended precision; returns x and the history of max|dx|/max|x|"""
    hist = []
    r = np.zeros(U)
    for it in range(iters):
        L.oracle_residual_ld(U, orc._p(A0), orc._p(x), orc._p(b), orc._p(r))
        L.oracle_dsptrs(U, orc._p(fac), ipiv.ctypes.data_as(orc._pi), orc._p(r))
        x = x + r
        hist.append(float(np.abs(r).max() / np.abs(x).max()))
        g.log(f"{label}: refinement {it}: correction {hist[-1]:.3e}")
        if hist[-1] < 1e-14:
            break
    return x, hist


def main():
    out_dir = g.OUT
    orc.build()
    L = orc.lib()
    fp = g.scene.config(os.environ.get("GOLDEN_CONFIG", "cfg4"))
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    o = orc.Oracle(fp)
    cols = g.sample_columns(fp)
    rng = np.random.Generator(np.random.Philox(g.PROBE_SEED + 2))
    qcols = np.sort(rng.choice(cols, size=min(N_QCOLS, cols.size), replace=False))
    t = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=int(os.environ.get("GOLDEN_THREADS", "6"))) as ex:
        weights = list(ex.map(lambda b: o.block_weight(s2, b), range(fp.n_image_blocks)))
    g.log(f"block weights {time.perf_counter() - t:.1f} s")
    res = {}
    values = fp.values.copy()
    for p in (1, 2):
        times = []
        N, n = g.assemble(o, fp, values, s2, weights, times)
        V = o.finalize(values, N, n, 0.0, False)
        o.precondition(V, N, n)
        A0 = N.copy(); b = n.copy()
        ipiv = np.zeros(U, np.int32)
        info = L.oracle_dsptrf(U, orc._p(N), ipiv.ctypes.data_as(orc._pi))
        assert info == 0
        y = b.copy()
        L.oracle_dsptrs(U, orc._p(N), ipiv.ctypes.data_as(orc._pi), orc._p(y))
        y_or = y.copy()
        y, hist = refine(L, U, A0, N, ipiv, b, y, f"pass {p} dx")
        dx_true, dx_or = V * y, V * y_or
        res[f"dx{p}_true"] = dx_true
        res[f"dx{p}_oracle_err"] = np.array([np.abs(dx_or - dx_true).max() / np.abs(dx_true).max()])
        g.log(f"pass {p}: oracle dspsv vs exact: {res[f'dx{p}_oracle_err'][0]:.3e}")
        if p == 2:
            Qc = np.zeros((qcols.size, U)); Qc_or = np.zeros((qcols.size, U))
            for a, c in enumerate(qcols):
                e = np.zeros(U); e[c] = 1.0
                yq = e.copy()
                L.oracle_dsptrs(U, orc._p(N), ipiv.ctypes.data_as(orc._pi), orc._p(yq))
                Qc_or[a] = V * yq * V[c]
                yq, _ = refine(L, U, A0, N, ipiv, e, yq, f"Qxx column {c}", iters=3)
                Qc[a] = V * yq * V[c]
            res["qcols"] = qcols; res["Qcols_true"] = Qc
            sd = np.sqrt(np.abs(np.array([Qc[a, c] for a, c in enumerate(qcols)])))
            res["Qcols_oracle_err"] = np.array([np.abs(Qc_or - Qc).max() / np.abs(Qc).max()])
            g.log(f"final pass: oracle Qxx columns (dsptrs) vs exact: {res['Qcols_oracle_err'][0]:.3e}")
        else:
            values, _ = o.update(values, np.load(os.path.join(out_dir, "cfg4_oracle.npz"))["dx1"] if os.path.exists(os.path.join(out_dir, "cfg4_oracle.npz")) else dx_or)
        del N, A0
    np.savez(os.path.join(out_dir, "cfg4_truth.npz"), **res)
    g.log("done")


if __name__ == "__main__":
    main()
