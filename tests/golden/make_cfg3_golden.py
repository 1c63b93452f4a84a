#!/usr/bin/env python
"""Runs the CPU oracle's estimateModel() to TERMINATION at BASELINE config 3 (100 images x 1 000 points, full interior set, 2x2
correlated image points, U = 3 614) and writes ``tests/golden/cfg3/cfg3_converged.npz`` (+ ``.json``).

    python tests/golden/make_cfg3_golden.py        (about 3 min on one core)

oracle_estimate (ba_oracle.c) = BundleAdjustment.java:203-387 with MatrixInversion.FULL: passes until max|dx| <= sqrt(eps)
(BA:327-335), then the final pass with dspsv + dsptri (MathExtension.java:338-366), Omega (BA:430), last update (BA:450-462).
The same loop is walked here pass by pass as well (oracle.step), to store the history of max|dx| -- and the two must agree
bit for bit.  Stored: converged parameter values, iteration count, Omega, sigma0^2, diag(Qxx), a 400 x 400 sample of Qxx, Qxx.v.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_cfg4_golden as g  # noqa: E402  (probe vector, column sample, packed helpers)

orc = g.orc
SQRT_EPS = 1.0536712127723509e-8


def main():
    out_dir = os.path.join(HERE, "cfg3")
    os.makedirs(out_dir, exist_ok=True)
    orc.build()
    fp = g.scene.config("cfg3")
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    o = orc.Oracle(fp)
    t = time.perf_counter()
    values, Q, res = o.estimate(invert=True)
    t_est = time.perf_counter() - t
    g.log(f"oracle_estimate: state {res.state}, iterations {res.iterations}, max|dx| {res.max_abs_dx:.3e}, {t_est:.1f} s")
    assert res.state == 1
    # the same loop pass by pass, for the history
    v = fp.values.copy(); hist = []; is_est = False
    while True:
        dx, Qs, _, _ = o.step(v, s2, 0.0, is_est)
        v, mx = o.update(v, dx)
        hist.append(mx)
        if is_est:
            break
        if mx <= SQRT_EPS:
            is_est = True
    assert np.array_equal(v, values) and np.array_equal(Qs, Q), "the stepwise loop and oracle_estimate disagree"
    probe = g.probe_vector(U)
    cols = g.sample_columns(fp)
    dof = fp.degree_of_freedom
    meta = {"config": "cfg3", "U": int(U), "passes": len(hist), "iteration_step": int(res.iterations), "state": int(res.state),
            "max_abs_dx": [float(m) for m in hist], "sqrt_eps": SQRT_EPS, "omega": float(res.omega),
            "degree_of_freedom": int(dof), "sigma2aposteriori": float(abs(res.omega / dof)), "sigma2apriori": float(s2),
            "qxx_frobenius": g.packed_fro(Q, U), "probe_seed": g.PROBE_SEED, "seconds_estimate": t_est,
            "reference": "BundleAdjustment.java:203-387 run to termination; MathExtension.java:338-366"}
    np.savez_compressed(os.path.join(out_dir, "cfg3_converged.npz"), values=values, diagQ=g.packed_diag(Q, U), sample_cols=cols,
                        Qsample=g.packed_sub(Q, cols), Qv=g.packed_matvec(Q, probe))
    with open(os.path.join(out_dir, "cfg3_converged.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    g.log("done: " + json.dumps(meta["max_abs_dx"]))


if __name__ == "__main__":
    main()
