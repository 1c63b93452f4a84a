"""The whole loop, run to the REFERENCE's termination criterion at BASELINE configs 3 and 4, against the oracle run to the same
criterion (missing 1 + 2).

BundleAdjustment.java:327-353: passes run until max|dx| <= sqrt(eps) = 1.0537e-8, then ONE more pass with the inverse
(BA:250-281).  Fixtures: tests/golden/cfg3/cfg3_converged.* (make_cfg3_golden.py: oracle_estimate, 27 s) and
tests/golden/cfg4/cfg4_converged.* (make_cfg4_converged.py: the oracle's run at the headline size continued from the two
passes of cfg4_oracle.npz to termination, ~2 h of one core).

What is asserted: state ERROR_FREE_ESTIMATION, the SAME number of passes as the reference algorithm took, last max|dx| below the
criterion, the converged parameters within 1e-9 (north_star) of the oracle's, Omega, sigma0^2 and every diagonal entry of Qxx at
the converged point.  Default and deterministic assembly, MatrixInversion FULL and REDUCED.
"""
import json
import os

import numpy as np
import pytest

from bundle_adjustment_amd import engine, scene

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SQRT_EPS = 1.0536712127723509e-8       # Math.sqrt(Constant.EPS)


def load(cfg):
    p = os.path.join(G, cfg, f"{cfg}_converged")
    if not os.path.exists(p + ".npz"):
        pytest.skip(f"{p}.npz not generated")
    return dict(np.load(p + ".npz")), json.load(open(p + ".json"))


def relative_parameter_error(fp, got, ref):
    """object coordinates and projection centres against the extent of the object (2 000 mm), every other parameter against its
    own magnitude (floor 1.0) -- the measure test_gpu_fullsize.py uses"""
    P3, I6 = 3 * fp.n_points, 6 * fp.n_images
    den = np.maximum(np.abs(ref), 1.0)
    den[:P3] = 2000.0
    den[-I6:].reshape(-1, 6)[:, :3] = 2000.0
    return float((np.abs(got - ref) / den).max())


def truth(cfg):
    """tests/golden/<cfg>/<cfg>_exactN.*: the normal equations of the converged point assembled in extended precision and inverted
    exactly (make_exactN.py) -- and what the ORACLE's dspsv + dsptri result is worth against it (round 4)."""
    p = os.path.join(G, cfg, f"{cfg}_exactN")
    if not os.path.exists(p + ".npz"):
        pytest.skip(f"{p}.npz not generated")
    return dict(np.load(p + ".npz")), json.load(open(p + ".json"))


def run_to_termination(fp, z, meta, invert, deterministic, qtol, check_values=1e-9, exact=None, ttol=None):
    eng = engine.Engine(fp, deterministic=deterministic)
    values, res = eng.estimate(invert=invert)
    try:
        assert res.state == 1                                              # ERROR_FREE_ESTIMATION
        assert res.iterations == meta["iteration_step"], (res.iterations, meta["iteration_step"], res.max_abs_dx)
        assert res.max_abs_dx <= SQRT_EPS
        err = relative_parameter_error(fp, values, z["values"])
        assert err < check_values, err
        assert abs(res.omega - meta["omega"]) <= 1e-9 * meta["omega"]
        k = eng.cofactor_order()
        assert k == (fp.n_unknowns if invert == engine.INVERT_FULL else eng.reduced_order())
        Q = eng.get_cofactor()
        idx = np.arange(k, dtype=np.int64)
        dq = float(np.abs(Q[idx * (idx + 3) // 2] / z["diagQ"][:k] - 1.0).max())
        del Q
        cols = z["sample_cols"]; keep = cols < k
        Qs = eng.get_cofactor_sub(cols[keep].astype(np.int32))
        ref = z["Qsample"][np.ix_(keep, keep)]
        sd = np.sqrt(np.abs(np.diag(ref)))
        cq = float((np.abs(Qs - ref) / np.outer(sd, sd)).max())
        assert dq < qtol and cq < qtol, (dq, cq)
        if exact is not None:
            # WHOSE error is the difference to the oracle?  Against the exact inverse of the exactly assembled system the device
            # must be (i) within ttol, (ii) no further away than the reference algorithm is, and (iii) its distance to the oracle
            # must be explained by the oracle's own error (triangle inequality)
            t, tm = exact
            assert np.array_equal(t["sample_cols"], cols)
            tr = t["Qsample_true"][np.ix_(keep, keep)]
            sdt = np.sqrt(np.abs(np.diag(tr)))
            dev_t = float((np.abs(Qs - tr) / np.outer(sdt, sdt)).max())
            orc_t = float((np.abs(ref - tr) / np.outer(sdt, sdt)).max())
            assert dev_t < ttol, dev_t
            assert dev_t <= orc_t, (dev_t, orc_t)
            assert cq <= orc_t + ttol, (cq, orc_t)
            print(f"   against the exact inverse: device {dev_t:.2e}, oracle (reference algorithm) {orc_t:.2e}, device vs oracle {cq:.2e}")
        return err, dq, cq, res
    finally:
        eng.close()


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("invert", [engine.INVERT_FULL, engine.INVERT_REDUCED])
def test_config3_runs_to_the_references_termination(invert, deterministic):
    z, meta = load("cfg3")
    fp = scene.config("cfg3")
    # cond(V N V) ~ 4e8 here.  Through round 4 Qxx was 2e-9 (diag) / 3e-9 (correlation-scaled sample) from the oracle's dsptri, and the truth
    # fixture of round 5 (tests/golden/cfg3/cfg3_exactN.*, ordinary 2 x 2 weights: no dispersion is inverted here) shows whose error that
    # was: the reference algorithm is 4.7e-10 from the exact inverse, the Cholesky-based inverse 2-3e-9 -- north_star's 1e-9 missed by the
    # DEVICE at this config.  With the Newton-Schulz step on the inverse (engine option inverse_refinement, default for orders <= 8192) the
    # device is at the rounding of Q's entries, and its distance to the oracle is the oracle's own error: asserted below 1e-9.
    err, dq, cq, res = run_to_termination(fp, z, meta, invert, deterministic, qtol=1e-9, exact=truth("cfg3"), ttol=1e-10)
    print(f"cfg3 invert={invert} det={deterministic}: iterations {res.iterations}, max|dx| {res.max_abs_dx:.2e}, parameters {err:.2e}, "
          f"diag Qxx {dq:.2e}, sample {cq:.2e}")


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("invert", [engine.INVERT_FULL, engine.INVERT_REDUCED])
def test_config3_with_dense_dispersions_runs_to_the_references_termination(invert, deterministic):
    """Config 3's size with config 4's dense per-image dispersions (U = 3 614, cond ~ 4e8): the EO pre-elimination, the dataflow
    factorisation of the reduced order 3 014, the refinement and (FULL) the expansion of Qxx, at a size where the oracle's whole run is
    a 15-minute fixture (tests/golden/make_cfg3b_golden.py)."""
    z, meta = load("cfg3b")
    fp = scene.config("cfg3_block")
    # Qxx against the oracle's dsptri: achieved 9e-9 .. 1.0e-8, of which 8.6e-9 .. 9.8e-9 are the ORACLE's distance from the exact
    # inverse (cfg3b_exactN.json); the device's own distance: 5e-10 .. 7e-10 (asserted < 2e-9)
    err, dq, cq, res = run_to_termination(fp, z, meta, invert, deterministic, qtol=2e-8, exact=truth("cfg3b"), ttol=2e-9)
    print(f"cfg3b invert={invert} det={deterministic}: iterations {res.iterations}, max|dx| {res.max_abs_dx:.2e}, parameters {err:.2e}, "
          f"diag Qxx {dq:.2e}, sample {cq:.2e}")


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("invert", [engine.INVERT_FULL, engine.INVERT_REDUCED])
def test_config4_runs_to_the_references_termination(cfg4_scene, invert, deterministic):
    z, meta = load("cfg4")
    # Qxx at cond ~ 1e9 against the oracle's dsptri: achieved 2.0e-7 .. 2.3e-7 -- the ORACLE's own distance from the exact inverse
    # of the exactly assembled system is 2.1e-7 .. 2.3e-7 (cfg4_exactN.json: the reference's fp64 dpptrf + dpptri weights at
    # cond(D) = 2e7); the device's distance from that truth: 1.7e-8 .. 2.2e-8 (asserted < 5e-8)
    err, dq, cq, res = run_to_termination(cfg4_scene, z, meta, invert, deterministic, qtol=3e-7, exact=truth("cfg4"), ttol=5e-8)
    print(f"cfg4 invert={invert} det={deterministic}: iterations {res.iterations}, max|dx| {res.max_abs_dx:.2e}, parameters {err:.2e}, "
          f"diag Qxx {dq:.2e}, sample {cq:.2e}")


@pytest.mark.parametrize("cfg", ["cfg3", "cfg3b", "cfg4"])
def test_host_estimate_model_runs_to_the_references_termination(cfg, request):
    """The same through the object API a JAICOV user sees: Camera / Image / ObjectCoordinate graph (tests/scene_graph.py, incl.
    Image.setDispersion for config 4's dense per-image dispersions) -> BundleAdjustment.estimateModel() with MatrixInversion.FULL
    (host/jaicov.cpp: BA:203-387 on the engine).  The host numbers the points in the order the images meet them (BA:667-782), a
    permutation of the scene's numbering: parameters are compared through the objects, Qxx through UnknownParameter.getColumn()."""
    import scene_graph
    from bundle_adjustment_amd import host_api as H
    z, meta = load(cfg)
    fp = request.getfixturevalue("cfg4_scene") if cfg == "cfg4" else scene.config("cfg3_block" if cfg == "cfg3b" else cfg)
    ba, cam, pts, images, _ = scene_graph.object_graph(H, fp)
    ba.setInvertNormalEquation(H.MatrixInversion.FULL)
    state = ba.estimateModel()
    assert state == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
    assert ba.getIterations() == meta["iteration_step"]
    got = scene_graph.adjusted_values(H, fp, cam, pts, images)
    err = relative_parameter_error(fp, got, z["values"])
    assert err < 1e-9, err
    assert abs(ba.getOmega() - meta["omega"]) <= 1e-9 * meta["omega"]
    assert abs(ba.getVarianceFactorAposteriori() - meta["sigma2aposteriori"]) <= 1e-9 * meta["sigma2aposteriori"]
    # sample of Qxx: scene column -> host column through the objects
    scene_cols = fp.slot_columns()
    host_col = np.full(fp.n_unknowns, -1, np.int64)
    ups = [c() for p in pts for c in (p.getX, p.getY, p.getZ)]
    for s, up in enumerate(ups):
        if scene_cols[s] >= 0:
            host_col[scene_cols[s]] = up.getColumn()
    cols = z["sample_cols"]
    keep = host_col[cols] >= 0                       # the object points among the sampled columns
    Qs = np.asarray(ba.cofactorSub([int(c) for c in host_col[cols[keep]]]))
    ref = z["Qsample"][np.ix_(keep, keep)]
    sd = np.sqrt(np.abs(np.diag(ref)))
    cq = float((np.abs(Qs - ref) / np.outer(sd, sd)).max())
    print(f"{cfg} host estimateModel: iterations {ba.getIterations()}, parameters {err:.2e}, Qxx sample ({int(keep.sum())} point columns) {cq:.2e}")
    assert cq < {"cfg3": 2e-8, "cfg3b": 2e-8, "cfg4": 3e-7}[cfg]       # cfg3b / cfg4: the oracle's own error, see the tests above
