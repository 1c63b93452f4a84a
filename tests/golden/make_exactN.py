#!/usr/bin/env python
"""GROUND TRUTH for the accuracy of the cofactor matrix: the normal equations assembled in extended precision ("exact N") and
their exact solution / inverse columns, at the parameter values of the oracle fixtures.

    python tests/golden/make_exactN.py cfg3 converged       (~1 min; round 5: config 3 proper, ordinary 2 x 2 weights)
    python tests/golden/make_exactN.py cfg3b converged      (~2 min, 8 threads)
    python tests/golden/make_exactN.py cfg4 converged       (~25 min, 8 threads, ~12 GB)
    python tests/golden/make_exactN.py cfg4 pass2

Why: at config 4 the per-image dispersions have cond(D) ~ 1e7 and the Jacobi-scaled normal matrix cond ~ 1e9.  An fp64 inverse
of D (the reference: dpptrf + dpptri, DirectlyObservedParameterGroup.java:82-86) is only good to ~1e-11 .. 1e-9, which the
inverse of N amplifies to ~1e-7 on Qxx: the device's Qxx and the oracle's (= the reference algorithm's) differ by 1.2e-7 .. 2.5e-7
and the round-3 verdict asked WHOSE error that is.  oracle/ba_exact.c assembles N = sum A'(sigma0^2 inv D)A with inv(D), all products
and all sums in x87 extended precision (64-bit mantissa), rounded once to a (hi, lo) pair of doubles; here that system is solved
exactly (fp64 Cholesky of V N V as preconditioner, residuals of the UNSCALED (hi + lo) system in extended precision, until the
correction is < 1e-13 or stops falling: cond . 2^-64, a few 1e-12) for the right-hand side n and for the unit vectors of the fixtures' 400 sample columns.

Written to tests/golden/<cfg>/<cfg>_exactN[_pass2].npz:
    n_exact, Nv_exact (probe of make_cfg4_golden.py), dx_true, sample_cols, Qsample_true (400 x 400), qcols + Qcols_true (8 whole
    columns), P_rows_exact (16 rows of sigma0^2 inv(D) of image blocks 0 and 1),
and to the .json beside it the ORACLE's (= reference algorithm's) own errors against this truth, measured the way the GPU tests
measure the device's: Qxx correlation-scaled over the sample, the sampled variances, the step, N.v, n, and inv(D) per block --
plus a binary128 certificate of the extended-precision inverse (max |I - D X| over 32 rows).
Nothing here is the reference's arithmetic; it is the yardstick both implementations are held against.
"""
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_cfg4_golden as g  # noqa: E402  (scene, probes, column sample, block-fair assembly of the oracle)

orc = g.orc
N_QCOLS = 8


def unpack_lower(ap, U, scale=None):
    """packed 'U' column-major == row-major lower triangle -> dense array with the LOWER triangle filled"""
    S = np.zeros((U, U))
    off = 0
    for r in range(U):
        S[r, :r + 1] = ap[off:off + r + 1]
        off += r + 1
    if scale is not None:
        S *= scale[:, None]
        S *= scale[None, :]
    return S


def main():
    cfg, point = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "converged")
    scene_name = {"cfg3": "cfg3", "cfg3b": "cfg3_block", "cfg4": "cfg4"}[cfg]
    out_dir = os.path.join(HERE, cfg)
    orc.build()
    L = orc.lib()
    fp = g.scene.config(scene_name)
    assert fp.rank_defect == 0, "the truth machinery assumes an SPD system (no datum border)"
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    o = orc.Oracle(fp)
    probe = g.probe_vector(U)
    # --- the parameter values of the fixture's inverting pass, and the oracle's results there
    if point == "converged":
        z = np.load(os.path.join(out_dir, f"{cfg}_converged.npz"))
        values = z["values"].copy()        # after the last update (6.8e-12 mm at config 4): the same point for every purpose here
        tag = ""
    else:
        assert cfg == "cfg4"
        z = np.load(os.path.join(out_dir, "cfg4_oracle.npz"))
        values, _ = o.update(fp.values.copy(), z["dx1"])
        tag = "_pass2"
    cols = z["sample_cols"].astype(np.int64)
    meta = {"config": scene_name, "point": point, "U": int(U), "threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count()))}

    # --- exact assembly
    t = time.perf_counter()
    Nh, Nl, nh, nl = o.exact_accumulate(values, s2)
    g.log(f"extended-precision assembly {time.perf_counter() - t:.1f} s; |lo|/|hi| = {np.abs(Nl).max() / np.abs(Nh).max():.2e}")
    Nv = np.zeros(U)
    L.oracle_matvec_ld2(U, orc._p(Nh), orc._p(Nl), orc._p(probe), orc._p(Nv))

    # --- the oracle's own assembly at the same values (fp64, dpptrf + dpptri weights, the "fair" two-product form)
    t = time.perf_counter()
    import concurrent.futures as cf
    nkey, vkey = ("n", "Nv") if point == "converged" else ("n2", "Nv2")
    if nkey in z.files and vkey in z.files:            # the fixture holds the oracle's n and N.v of this pass
        no, Nvo = z[nkey], z[vkey]
        wts = {b: o.block_weight(s2, b) for b in range(min(4, fp.n_image_blocks))}
    elif fp.n_image_blocks == 0:                       # ordinary image groups only (config 3 proper): the oracle's literal stacking (PDF:475-505)
        No, no, _ = o.build(values, s2, 0.0)
        wts = {}
        Nvo = g.packed_matvec(No, probe)
        meta["oracle_N_err_max_entry"] = float(np.abs(No - Nh).max() / np.abs(Nh).max())
        del No
    else:
        with cf.ThreadPoolExecutor(max_workers=6) as ex:
            wts = dict(enumerate(ex.map(lambda b: o.block_weight(s2, b), range(fp.n_image_blocks))))
        No, no = g.assemble(o, fp, values, s2, [wts[b] for b in range(fp.n_image_blocks)], [])
        g.log(f"oracle assembly {time.perf_counter() - t:.1f} s")
        Nvo = g.packed_matvec(No, probe)
        meta["oracle_N_err_max_entry"] = float(np.abs(No - Nh).max() / np.abs(Nh).max())
        del No
    # n vanishes at the converged point (it is the gradient): measure it against the size of its terms, max |N| . |dx| ~ max |N.v| here
    meta["oracle_n_err_vs_Nv"] = float(np.abs(no - nh).max() / np.abs(Nv).max())
    meta["oracle_Nv_err"] = float(np.abs(Nvo - Nv).max() / np.abs(Nv).max())
    g.log(f"oracle vs exact: n (against max|N.v|) {meta['oracle_n_err_vs_Nv']:.2e}, N.v {meta['oracle_Nv_err']:.2e}")

    # --- inv(D): oracle (dpptrf + dpptri of D / sigma0^2) vs extended precision, per block; binary128 certificate
    perr = []
    P_rows = []
    for b in range(min(4, fp.n_image_blocks)):
        Ph, Pl = o.exact_block_weight(s2, b)
        perr.append(float(np.abs(wts[b] - Ph).max() / np.abs(Ph).max()))
        if b < 2:
            P_rows.append(Ph[:16].copy())
        if b == 0:
            m = Ph.shape[0]
            D = fp.blk_disp[fp.blk_disp_offset[0]:fp.blk_disp_offset[0] + m * m].copy()
            meta["cond_D_block0"] = float(np.linalg.cond(D.reshape(m, m)))
            res_q = L.oracle_inverse_residual_q(m, orc._p(D), orc._p(Ph), orc._p(Pl), s2, 0, 32)
            res_q_or = L.oracle_inverse_residual_q(m, orc._p(D), orc._p(wts[0]), None, s2, 0, 32)
            meta["binary128_residual_exact_P"] = float(res_q)
            meta["binary128_residual_oracle_P"] = float(res_q_or)
            g.log(f"block 0: cond(D) {meta['cond_D_block0']:.2e}; max|I - D P / s2| over 32 rows in binary128: extended {res_q:.2e}, oracle fp64 {res_q_or:.2e}")
    meta["oracle_P_err_blocks"] = perr
    g.log(f"oracle inv(D) vs extended precision, blocks 0..3: {perr}")
    del wts

    # --- exact solutions: fp64 Cholesky of V N_hi V as preconditioner, extended residuals of the unscaled (hi + lo) system
    idx = np.arange(U, dtype=np.int64)
    diag = Nh[idx * (idx + 3) // 2]
    V = np.where(diag > 2.0 ** -53, 1.0 / np.sqrt(diag), 1.0)
    t = time.perf_counter()
    S = unpack_lower(Nh, U, V)
    cfac = sl.cho_factor(S, lower=True, overwrite_a=True, check_finite=False)
    g.log(f"preconditioner: Cholesky of order {U} {time.perf_counter() - t:.1f} s")

    def solve_exact(Bh, Bl, label):
        """rows of Bh (+ Bl) are right-hand sides; returns the exact solutions (rows) of (N_hi + N_lo) x = b"""
        q = Bh.shape[0]
        X = (sl.cho_solve(cfac, (Bh * V[None, :]).T, check_finite=False).T) * V[None, :]
        R = np.zeros_like(X)
        prev = np.inf
        for it in range(8):
            L.oracle_residual_ld2(U, orc._p(Nh), orc._p(Nl), q, orc._p(X), orc._p(Bh), orc._p(Bl) if Bl is not None else None,
                                  orc._p(R))
            dX = (sl.cho_solve(cfac, (R * V[None, :]).T, check_finite=False).T) * V[None, :]
            X = np.ascontiguousarray(X + dX)
            corr = float((np.abs(dX).max(axis=1) / np.abs(X).max(axis=1)).max())
            g.log(f"{label}: refinement {it}: largest relative correction {corr:.2e}")
            if corr < 1e-13 or corr > 0.5 * prev:      # converged, or at the floor of the extended residual (cond . 2^-64)
                break
            prev = corr
        return X, corr

    dx_true, c = solve_exact(nh[None, :].copy(), nl[None, :].copy(), "step")
    dx_true = dx_true[0]
    meta["dx_last_correction"] = c
    E = np.zeros((cols.size, U))
    E[np.arange(cols.size), cols] = 1.0
    t = time.perf_counter()
    Qc, c = solve_exact(E, None, "Qxx columns")
    g.log(f"{cols.size} exact columns of Qxx {time.perf_counter() - t:.1f} s")
    meta["Qcols_last_correction"] = c
    Qs_true = Qc[:, cols]
    Qs_true = 0.5 * (Qs_true + Qs_true.T)
    rng = np.random.Generator(np.random.Philox(g.PROBE_SEED + 7))
    pick = np.sort(rng.choice(cols.size, size=N_QCOLS, replace=False))

    # --- the oracle's (reference algorithm's) errors against the truth, measured as the GPU tests measure the device's
    ref = z["Qsample"]
    sd = np.sqrt(np.abs(np.diag(Qs_true)))
    meta["oracle_Qsample_err"] = float((np.abs(ref - Qs_true) / np.outer(sd, sd)).max())
    meta["oracle_diag_err"] = float(np.abs(z["diagQ"][cols] / np.diag(Qs_true) - 1.0).max())
    dx_or = z["dx_final"] if "dx_final" in z.files else (z["dx2"] if "dx2" in z.files else None)
    if dx_or is not None and point == "pass2":
        meta["oracle_dx_err"] = float(np.abs(dx_or - dx_true).max() / np.abs(dx_true).max())
    g.log(f"ORACLE vs truth: Qxx sample (correlation-scaled) {meta['oracle_Qsample_err']:.2e}, sampled variances {meta['oracle_diag_err']:.2e}")

    np.savez_compressed(os.path.join(out_dir, f"{cfg}_exactN{tag}.npz"), n_exact=nh, Nv_exact=Nv, dx_true=dx_true,
                        sample_cols=cols, Qsample_true=Qs_true, qcols=cols[pick], Qcols_true=Qc[pick],
                        P_rows_exact=np.stack(P_rows) if P_rows else np.zeros(0))
    with open(os.path.join(out_dir, f"{cfg}_exactN{tag}.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    g.log("done")


if __name__ == "__main__":
    main()
