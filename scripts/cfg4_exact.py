"""Accuracy of the step and of Qxx at config 4/5 against the EXACT solutions of the systems the DEVICE assembled (VERDICT r2,
next 1): separates what the device's solver contributes from what the rounding of its assembly contributes (the floor).

Run on the GPU box (~10 min, most of it two packed Bunch-Kaufman factorisations on two host threads):
    python scripts/cfg4_exact.py [config]          -> gpurun_out/cfg4_exact.json  (copied to profiles/r03_cfg4_accuracy.json)

At the parameters of the fixture's final pass (start values + the oracle's first step) the engine assembles
  F: the full system (MatrixInversion.FULL, order U)            and solves / inverts it,
  R: the EO-reduced system (the product path of every LM pass)  and solves / inverts it,
each with and without iterative refinement.  The host then computes, with the oracle's dsptrf of the Jacobi-scaled system as the
preconditioner and residuals of the UNSCALED system in long double (oracle_residual_ld), the exact step and exact columns of the
inverse of each -- exact to ~1e-13, certified by the size of the last correction.
"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as orc

N_QCOLS = int(os.environ.get("EXACT_QCOLS", "12"))


def log(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", flush=True)


rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())


class ExactSolver:
    """Exact solutions of  A x = b  for A symmetric, packed 'U', order k (d = 0): dsptrf of V A V once, refinement in long double."""

    def __init__(self, A_packed, k, label):
        self.L = orc.lib(); self.k = k; self.label = label
        self.A0 = np.ascontiguousarray(A_packed[:k * (k + 1) // 2])
        idx = np.arange(k, dtype=np.int64)
        dg = self.A0[idx * (idx + 3) // 2]
        self.V = np.where(dg > 1.1102230246251565e-16, 1.0 / np.sqrt(np.abs(dg) + (dg <= 0)), 1.0)   # BA:825-828
        self.F = self.A0.copy()
        self.L.oracle_precondition(k, orc._p(self.V), orc._p(self.F), orc._p(np.zeros(k)))           # NES:82-91
        self.ipiv = np.zeros(k, np.int32)
        t = time.time()
        info = self.L.oracle_dsptrf(k, orc._p(self.F), self.ipiv.ctypes.data_as(orc._pi))
        assert info == 0, info
        log(f"{label}: dsptrf of order {k}: {time.time() - t:.0f} s")

    def bk(self, b):
        """the reference's algorithm: dspsv on the scaled system, un-scaled (BA:266-297)"""
        y = self.V * b
        self.L.oracle_dsptrs(self.k, orc._p(self.F), self.ipiv.ctypes.data_as(orc._pi), orc._p(y))
        return self.V * y

    def exact(self, b, x0=None, iters=5):
        x = self.bk(b) if x0 is None else x0.copy()
        r = np.zeros(self.k); last = None
        for _ in range(iters):
            self.L.oracle_residual_ld(self.k, orc._p(self.A0), orc._p(x), orc._p(b), orc._p(r))
            dx = self.bk(r)
            x = x + dx
            last = float(np.abs(dx).max() / np.abs(x).max())
            if last < 1e-15:
                break
        return x, last


def packed_column(ap, k, c):
    """column c of the symmetric matrix in packed 'U' (k leading rows)"""
    col = np.empty(k)
    col[:c + 1] = ap[c * (c + 1) // 2: c * (c + 1) // 2 + c + 1]
    r = np.arange(c + 1, k, dtype=np.int64)
    col[c + 1:] = ap[c + r * (r + 1) // 2]
    return col


def host_part(NF, nF, U, NR, nR, k, qcols_F, qcols_R, out, truth=None):
    """NF, nF: full system (packed, order U); NR, nR: reduced system (order k).  Fills out['exact'] in two threads."""
    res = {}

    def work(tag, A, b, order, qcols):
        es = ExactSolver(A, order, tag)
        x, last = es.exact(b)
        res[tag] = {"dx": x, "dx_last_correction": last, "dx_bk": es.bk(b), "Qcols": {}, "Qlast": 0.0}
        for c in qcols:
            e = np.zeros(order); e[c] = 1.0
            q, lastq = es.exact(e)
            res[tag]["Qcols"][int(c)] = q
            res[tag]["Qlast"] = max(res[tag]["Qlast"], lastq)
        log(f"{tag}: exact step (last correction {last:.1e}) and {len(qcols)} exact columns of the inverse (last correction {res[tag]['Qlast']:.1e})")

    th = [threading.Thread(target=work, args=("F", NF, nF, U, qcols_F)), threading.Thread(target=work, args=("R", NR, nR, k, qcols_R))]
    for t in th: t.start()
    while any(t.is_alive() for t in th):
        time.sleep(45); log("... host factorisations / refinements running")
    for t in th: t.join()
    return res


def main():
    from bundle_adjustment_amd import engine, scene
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    fp = scene.config(name); U, s2 = fp.n_unknowns, fp.sigma2apriori
    G = os.path.join(ROOT, "tests", "golden", "cfg4")
    have_fix = name == "cfg4" and os.path.exists(os.path.join(G, "cfg4_truth.npz"))
    values = fp.values.copy()
    if have_fix:
        z = np.load(os.path.join(G, "cfg4_oracle.npz")); tr = np.load(os.path.join(G, "cfg4_truth.npz"))
        cols = fp.slot_columns(); m = cols >= 0
        values[m] += z["dx1"][cols[m]]
        qF = [int(c) for c in tr["qcols"][:N_QCOLS]]
    else:
        qF = [int(c) for c in np.random.default_rng(3).choice(U, min(N_QCOLS, U), replace=False)]
    dev = {}
    for refinement in (-1, 0):
        eng = engine.Engine(fp, refinement=refinement)
        eng.set_parameters(values)
        eng.prepare_inverse(engine.INVERT_FULL); eng.build(s2, 0.0)
        if refinement == 0:
            NF, nF = eng.get_normal()
        dxF = eng.solve(engine.INVERT_FULL)
        QF = eng.get_cofactor()
        colsF = {c: packed_column(QF, U, c) for c in qF}
        diagF = QF[np.arange(U, dtype=np.int64) * (np.arange(U, dtype=np.int64) + 3) // 2].copy()
        del QF
        eng.prepare_inverse(engine.INVERT_REDUCED); eng.build(s2, 0.0)
        k = eng.reduced_order()
        if refinement == 0:
            NRf, nRf = eng.get_normal()
            NR = NRf[:k * (k + 1) // 2].copy(); nR = nRf[:k].copy(); del NRf
        dxR = eng.solve(engine.INVERT_REDUCED)
        qR = [c for c in qF if c < k]
        QR = eng.get_cofactor()
        colsR = {c: packed_column(QR, k, c) for c in qR}
        del QR
        t = eng.timings()
        dev[refinement] = dict(dxF=dxF, dxR=dxR, colsF=colsF, colsR=colsR, diagF=diagF, solve_ms=float(t["solve"]))
        eng.close()
        log(f"device, refinement {refinement}: done (reduced order {k})")
    ex = host_part(NF, nF, U, NR, nR, k, qF, qR, None)
    out = {"config": name, "U": U, "reduced_order": k, "exact_last_correction": {t: ex[t]["dx_last_correction"] for t in ex},
           "exact_Q_last_correction": {t: ex[t]["Qlast"] for t in ex}}
    for tag, key, order in (("F", "dxF", U), ("R", "dxR", k)):
        e = ex[tag]["dx"]
        out[f"step_{tag}"] = {"device_unrefined_vs_exact": rel(dev[-1][key][:order], e), "device_refined_vs_exact": rel(dev[0][key][:order], e),
                              "reference_algorithm_dspsv_on_the_device_system_vs_exact": rel(ex[tag]["dx_bk"], e)}
    out["step_exact_R_vs_exact_F"] = rel(ex["R"]["dx"], ex["F"]["dx"][:k])          # what the Schur assembly's rounding costs
    def qerr(devcols, excols, order):
        qmax = max(np.abs(v).max() for v in excols.values())
        cm = max(np.abs(devcols[c][:order] - excols[c][:order]).max() for c in excols) / qmax
        dg = max(abs(devcols[c][c] / excols[c][c] - 1.0) for c in excols)
        return {"columns_rel_to_max": float(cm), "diag_rel": float(dg)}
    out["Q_F_device_vs_exact_inverse_of_device_N"] = qerr(dev[0]["colsF"], ex["F"]["Qcols"], U)
    out["Q_R_device_vs_exact_inverse_of_device_reduced_N"] = qerr(dev[0]["colsR"], ex["R"]["Qcols"], k)
    out["Q_exact_R_vs_exact_F_block"] = qerr(ex["R"]["Qcols"], {c: ex["F"]["Qcols"][c][:k] for c in ex["R"]["Qcols"]}, k)
    if have_fix:
        tq = {int(c): tr["Qcols_true"][a] for a, c in enumerate(tr["qcols"]) if int(c) in ex["F"]["Qcols"]}
        out["FLOOR_Q_exact_inverse_of_device_N_vs_exact_inverse_of_oracle_N"] = qerr(ex["F"]["Qcols"], tq, U)
        out["Q_F_device_vs_exact_inverse_of_oracle_N"] = qerr(dev[0]["colsF"], tq, U)
        out["Q_R_device_vs_exact_inverse_of_oracle_N"] = qerr(dev[0]["colsR"], {c: tq[c][:k] for c in tq if c < k}, k)
        out["FLOOR_step_exact_of_device_F_vs_exact_of_oracle"] = rel(ex["F"]["dx"], tr["dx2_true"])
        out["step_R_device_refined_vs_exact_of_oracle"] = rel(dev[0]["dxR"][:k], tr["dx2_true"][:k])
        out["diagQ_F_device_vs_oracle_dsptri"] = float(np.abs(dev[0]["diagF"] / z["diagQ"] - 1.0).max())
    out["refinement_ms"] = dev[0]["solve_ms"] - dev[-1]["solve_ms"]
    log(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cfg4_exact.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
