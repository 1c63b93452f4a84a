#!/usr/bin/env python
"""The oracle's estimateModel() run to TERMINATION on a config-3-sized scene with config 4's dense per-image dispersions
(`scene.config("cfg3_block")`: 100 images x 1 000 points, 400 points per image, m = 800, U = 3 614; the EO-reduced order 3 014 takes the
dataflow factorisation on the device).  Writes tests/golden/cfg3b/cfg3b_converged.{npz,json}.

    python tests/golden/make_cfg3b_golden.py        (about 15 min on one core: ~70 s of assembly per pass)

The loop is walked pass by pass like make_cfg4_converged.py (BundleAdjustment.java:228-355, MatrixInversion.FULL, no damping): the
image groups through the "fair" two-product form (oracle_block_fair), shared groups through oracle_accumulate, dspsv, and dsptri in
the pass after max|dx| <= sqrt(eps)."""
import concurrent.futures as cf
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_cfg4_golden as g  # noqa: E402

orc = g.orc
SQRT_EPS = 1.0536712127723509e-8


def main():
    out_dir = os.path.join(HERE, "cfg3b")
    os.makedirs(out_dir, exist_ok=True)
    orc.build()
    fp = g.scene.config("cfg3_block")
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    o = orc.Oracle(fp)
    probe = g.probe_vector(U)
    cols = g.sample_columns(fp)
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        weights = list(ex.map(lambda b: o.block_weight(s2, b), range(fp.n_image_blocks)))
    values = fp.values.copy(); hist = []; is_est = False
    while True:
        p = len(hist) + 1
        res = g.one_pass(o, fp, values, s2, weights, is_est, probe, f"pass {p}" + (" (final)" if is_est else ""))
        if is_est:
            omega = o.omega(values, s2, res["dx"])
        values, mx = o.update(values, res["dx"])
        hist.append(mx)
        g.log(f"pass {p}: max|dx| = {mx:.6e}")
        if is_est:
            break
        if mx <= SQRT_EPS:
            is_est = True
        assert p < 40
    Q = res["Q"]
    dof = fp.degree_of_freedom
    meta = {"config": "cfg3_block", "U": int(U), "passes": len(hist), "iteration_step": len(hist) - 1, "state": 1,
            "max_abs_dx": [float(m) for m in hist], "sqrt_eps": SQRT_EPS, "omega": float(omega), "degree_of_freedom": int(dof),
            "sigma2aposteriori": float(abs(omega / dof)), "sigma2apriori": float(s2), "qxx_frobenius": g.packed_fro(Q, U),
            "probe_seed": g.PROBE_SEED, "reference": "BundleAdjustment.java:228-355 run to termination; MathExtension.java:338-366"}
    np.savez_compressed(os.path.join(out_dir, "cfg3b_converged.npz"), values=values, diagQ=g.packed_diag(Q, U), sample_cols=cols,
                        Qsample=g.packed_sub(Q, cols), Qv=g.packed_matvec(Q, probe))
    with open(os.path.join(out_dir, "cfg3b_converged.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    g.log("done: " + json.dumps(meta["max_abs_dx"]))


if __name__ == "__main__":
    main()
