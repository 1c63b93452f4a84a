"""Pins the CPU oracle (oracle/ba_oracle.c): golden Jacobian rows, LAPACK cross-checks, loop behaviour."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg

import helpers
from bundle_adjustment_amd import scene
from bundle_adjustment_amd.problem import full_to_packed, packed_to_full


def test_machine_eps(oracle_mod):
    # Constant.java:68-75 ends at 2^-53; sqrt -> 1.0536712127723509e-8
    assert oracle_mod.lib().oracle_eps() == 2.0 ** -53
    assert np.sqrt(2.0 ** -53) == 1.0536712127723509e-08


@pytest.mark.parametrize("name", ["pinhole", "radial", "full", "tangential2"])
def test_rows_match_symbolic_golden(oracle_mod, name):
    """Residual and every Jacobian entry vs sympy/mpmath derivatives of the model function.
    Tolerance 1e-11 relative to the row's largest entry: fp64 evaluation of ~100-flop expressions."""
    sets = helpers.load_golden_rows()
    dist, cases = sets[name]["dist"], sets[name]["cases"]
    fp = helpers.problem_from_cases(dist, cases)
    o = oracle_mod.Oracle(fp)
    nd = len(dist)
    worst = 0.0
    for i, c in enumerate(cases):
        w, A, P, diag = o.rows(fp.values, i)
        assert diag
        np.testing.assert_allclose(w, c["w"], rtol=0, atol=1e-12)
        for r, key in enumerate(("Ax", "Ay")):
            ref = np.array(c[key])
            got = A[r, :12 + nd]
            # entry-wise: relative to the entry where it is well scaled, else to the row's magnitude
            err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
            worst = max(worst, err.max())
    assert worst < 1e-10, worst


@pytest.mark.parametrize("name", ["zernike_x", "zernike_y", "zernike_gradient", "zernike_mixed", "zernike_high_x", "zernike_high_y", "zernike_example_gradient"])
def test_zernike_rows_match_symbolic_golden(oracle_mod, name):
    """ZernikeDistortionModelFactory.java:41-227 restated in the oracle vs sympy/mpmath derivatives of
    dx = z Z(xs, ys) (X), dy = z Z (Y), (dx, dy) = z grad Z (Gradient) for polynomials of even radial order
    (tests/golden/make_zernike_golden.py explains why only those have an independent truth)."""
    sets = helpers.load_golden_rows("jacobian_rows_zernike.json")
    dist, cases = sets[name]["dist"], sets[name]["cases"]
    fp = helpers.problem_from_cases(dist, cases)
    o = oracle_mod.Oracle(fp)
    nd = len(dist)
    worst = 0.0
    for i, c in enumerate(cases):
        w, A, P, diag = o.rows(fp.values, i)
        np.testing.assert_allclose(w, c["w"], rtol=0, atol=1e-12)
        for r, key in enumerate(("Ax", "Ay")):
            ref = np.array(c[key])
            err = np.abs(A[r, :12 + nd] - ref) / np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
            worst = max(worst, err.max())
    assert worst < 1e-9, worst


def test_zernike_polynomial_terms(oracle_mod):
    """n, m of the single index (Schwiegerling Eq. 2:100/101) and Z_4^0 = sqrt(5/pi) (6 r^4 - 6 r^2 + 1) through the own
    column of a ZERNIKE_X coefficient (= Z itself, ZDF:213-220)."""
    sets = helpers.load_golden_rows("jacobian_rows_zernike.json")
    case = sets["zernike_x"]["cases"][0]
    fp = helpers.problem_from_cases([[7, 12]], [dict(case, dist_values=[0.0])])
    w, A, _, _ = oracle_mod.Oracle(fp).rows(fp.values, 0)
    # undistorted image coordinates: with z = 0 the residual is obs - (x0 + xs)
    xs = case["obs"][0] - w[0] - case["io"][0]
    ys = case["obs"][1] - w[1] - case["io"][1]
    rho2 = (xs * xs + ys * ys) / case["r0"] ** 2
    Z40 = np.sqrt(5.0 / np.pi) * (6.0 * rho2 ** 2 - 6.0 * rho2 + 1.0)
    assert abs(A[0, 12] - Z40) < 1e-13 * max(1.0, abs(Z40)) and A[1, 12] == 0.0


def test_weights_2x2(oracle_mod):
    sets = helpers.load_golden_rows()
    fp = helpers.problem_from_cases(sets["radial"]["dist"], sets["radial"]["cases"], sigma=7e-4, rho=0.3)
    o = oracle_mod.Oracle(fp)
    s2 = 2.5e-7
    w, A, P, diag = o.rows(fp.values, 3, sigma2=s2)
    assert not diag
    D = np.array([[7e-4 ** 2, 0.3 * 7e-4 ** 2], [0.3 * 7e-4 ** 2, 7e-4 ** 2]])
    np.testing.assert_allclose(P, s2 * np.linalg.inv(D), rtol=1e-13)


@pytest.mark.parametrize("n", [1, 2, 5, 33, 120])
def test_dspsv_dsptri_vs_lapack(oracle_mod, n):
    """Packed Bunch-Kaufman restatement vs scipy's LAPACK dsysv on the unpacked matrix, with a zero leading block
    like the datum border (BundleAdjustment.java:493-635)."""
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n)); M = M + M.T
    d = min(6, n // 3)
    M[:d, :d] = 0.0
    b = rng.normal(size=n)
    ap = full_to_packed(M); x = b.copy()
    info = oracle_mod.lib().oracle_solve(n, ap.ctypes.data_as(oracle_mod._pd), x.ctypes.data_as(oracle_mod._pd), 1)
    assert info == 0
    _, _, xref, info2 = scipy.linalg.lapack.dsysv(M, b, lower=0)
    assert info2 == 0
    np.testing.assert_allclose(x, xref, rtol=1e-9, atol=1e-12 * np.abs(xref).max())
    Q = packed_to_full(ap, n)
    assert np.abs(Q @ M - np.eye(n)).max() < 1e-9 * np.linalg.cond(M)


def test_dsptrf_pivots_match_dsytf2(oracle_mod):
    """Same pivot sequence and factor as LAPACK's unblocked dsytf2 (the full-storage twin of dsptrf)."""
    if not hasattr(scipy.linalg.lapack, "dsytf2"):
        pytest.skip("scipy build without dsytf2")
    rng = np.random.default_rng(7)
    n = 40
    M = rng.normal(size=(n, n)); M = M + M.T; M[:6, :6] = 0
    ap = full_to_packed(M); ipiv = np.zeros(n, np.int32)
    info = oracle_mod.lib().oracle_dsptrf(n, ap.ctypes.data_as(oracle_mod._pd), ipiv.ctypes.data_as(oracle_mod._pi))
    ldu, piv, info2 = scipy.linalg.lapack.dsytf2(M, lower=0)
    assert info == info2 == 0
    np.testing.assert_array_equal(ipiv, piv)
    np.testing.assert_allclose(packed_to_full(ap, n)[np.triu_indices(n)], ldu[np.triu_indices(n)], rtol=1e-10, atol=1e-12)


def test_dpptrf_dpptri(oracle_mod):
    rng = np.random.default_rng(3)
    m = 45
    L = np.tril(rng.normal(0, 0.1, (m, m)), -1) + np.eye(m)
    D = L @ L.T * 1e-4
    P = np.zeros((m, m))
    info = oracle_mod.lib().oracle_dispersion_to_weight(m, D.ctypes.data_as(oracle_mod._pd), 2.5e-7,
                                                        P.ctypes.data_as(oracle_mod._pd))
    assert info == 0
    np.testing.assert_allclose(P, 2.5e-7 * np.linalg.inv(D), rtol=1e-9)
    bad = -np.eye(3)
    assert oracle_mod.lib().oracle_dispersion_to_weight(3, bad.ctypes.data_as(oracle_mod._pd), 1.0,
                                                        P.ctypes.data_as(oracle_mod._pd)) > 0


@pytest.mark.parametrize("name", ["tiny", "tiny_block", "tiny_free"])
def test_normal_equations_match_dense_algebra(oracle_mod, name):
    """N = A'PA, n = A'Pw assembled by the a10 restatement vs dense numpy algebra on the same rows."""
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    U = fp.n_unknowns
    s2 = fp.sigma2apriori
    N, n = o.accumulate(fp.values, s2)
    Nf = packed_to_full(N, U)
    # dense reference from the rows
    Nref = np.zeros((U, U)); nref = np.zeros(U)
    in_block = np.zeros(fp.n_image_points, bool)
    for b in range(fp.n_image_blocks):
        in_block[fp.blk_ip_begin[b]:fp.blk_ip_begin[b + 1]] = True
    def cols_of(ip):
        img, pt = fp.ip_image[ip], fp.ip_point[ip]
        return np.concatenate([fp.point_col[pt], fp.io_col[0], fp.eo_col[img], fp.dist_col])
    for ip in range(fp.n_image_points):
        if in_block[ip]:
            continue
        w, A, P, _ = o.rows(fp.values, ip, s2)
        Ad = np.zeros((2, U)); c = cols_of(ip); k = c.size
        Ad[:, c[c >= 0]] = A[:, :k][:, c >= 0]
        Nref += Ad.T @ P @ Ad; nref += Ad.T @ P @ w
    for b in range(fp.n_image_blocks):
        lo, hi = fp.blk_ip_begin[b], fp.blk_ip_begin[b + 1]
        m = 2 * (hi - lo)
        Ad = np.zeros((m, U)); wv = np.zeros(m)
        for ip in range(lo, hi):
            w, A, _, _ = o.rows(fp.values, ip, s2)
            c = cols_of(ip); k = c.size
            Ad[2 * (ip - lo):2 * (ip - lo) + 2, c[c >= 0]] = A[:, :k][:, c >= 0]
            wv[2 * (ip - lo):2 * (ip - lo) + 2] = w
        D = fp.blk_disp[fp.blk_disp_offset[b]:fp.blk_disp_offset[b] + m * m].reshape(m, m)
        Pm = s2 * np.linalg.inv(D)
        Nref += Ad.T @ Pm @ Ad; nref += Ad.T @ Pm @ wv
    for s in range(fp.n_scale_bars):
        a, b_ = fp.sb_point_a[s], fp.sb_point_b[s]
        dv = fp.values[3 * b_:3 * b_ + 3] - fp.values[3 * a:3 * a + 3]
        ln = np.linalg.norm(dv); e = dv / ln
        Ad = np.zeros(U); Ad[fp.point_col[a]] = -e; Ad[fp.point_col[b_]] = e
        p = s2 / fp.sb_var[s]
        Nref += p * np.outer(Ad, Ad); nref += Ad * p * (fp.sb_length[s] - ln)
    sc = fp.slot_columns()
    for g in range(fp.n_direct_groups):
        lo, hi = fp.dg_row_begin[g], fp.dg_row_begin[g + 1]
        m = hi - lo
        Ad = np.zeros((m, U)); Ad[np.arange(m), sc[fp.dg_slot[lo:hi]]] = 1
        wv = fp.dg_obs[lo:hi] - fp.values[fp.dg_slot[lo:hi]]
        if fp.dg_disp_offset[g] >= 0:
            D = fp.dg_disp[fp.dg_disp_offset[g]:fp.dg_disp_offset[g] + m * m].reshape(m, m)
            Pm = s2 * np.linalg.inv(D)
        else:
            Pm = np.diag(s2 / fp.dg_var[lo:hi])
        Nref += Ad.T @ Pm @ Ad; nref += Ad.T @ Pm @ wv
    scale = np.sqrt(np.outer(np.diag(Nref), np.diag(Nref))) + 1e-300
    d = fp.rank_defect
    assert (np.abs(Nf - Nref)[d:, d:] / scale[d:, d:]).max() < 1e-9
    np.testing.assert_allclose(n, nref, rtol=1e-8, atol=1e-9 * np.abs(nref).max())


@pytest.mark.parametrize("name,dof_lo,dof_hi", [("tiny", 0.5, 1.6), ("tiny_block", 0.5, 2.0), ("tiny_free", 0.5, 1.6)])
def test_estimate_converges(oracle_mod, name, dof_lo, dof_hi):
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    v, Q, res = o.estimate()
    assert res.state == 1                      # ERROR_FREE_ESTIMATION
    assert 2 <= res.iterations <= 15
    assert res.max_abs_dx <= np.sqrt(2.0 ** -53)
    ratio = res.omega / fp.degree_of_freedom / fp.sigma2apriori
    assert dof_lo < ratio < dof_hi, ratio
    # Qxx is the inverse of the bordered matrix at the solution
    N, n, V = o.build(v, fp.sigma2apriori)
    K = packed_to_full(N, fp.n_unknowns)
    Qf = packed_to_full(Q, fp.n_unknowns)
    Vd = V[:, None] * V[None, :]
    assert np.abs((Qf / Vd) @ (K * Vd) - np.eye(fp.n_unknowns)).max() < 1e-5
    if fp.rank_defect:
        d = fp.rank_defect
        # inner constraints hold: B dx = 0 is equivalent to B Qxx = 0 on the unknown block
        B = K[:d, d:]
        assert np.abs(B @ Qf[d:, d:]).max() < 1e-9 * np.abs(Qf[d:, d:]).max()


def test_levenberg_marquardt_path(oracle_mod):
    fp = scene.config("tiny")
    o = oracle_mod.Oracle(fp)
    v0, _, r0 = o.estimate()
    v1, _, r1 = o.estimate(lam0=1.0)
    assert r1.state == 1 and r1.iterations > r0.iterations
    np.testing.assert_allclose(v1, v0, rtol=0, atol=1e-6)


def test_sharded_accumulation_adds_up(oracle_mod):
    """N = sum over ranks of N_rank when images are sharded."""
    fp = scene.config("tiny_block")
    o = oracle_mod.Oracle(fp)
    N, n = o.accumulate(fp.values, fp.sigma2apriori)
    h = fp.n_images // 2
    N0, n0 = o.accumulate(fp.values, fp.sigma2apriori, 0, h, True)
    N1, n1 = o.accumulate(fp.values, fp.sigma2apriori, h, fp.n_images, False)
    np.testing.assert_allclose(N0 + N1, N, rtol=1e-12, atol=1e-14 * np.abs(N).max())
    np.testing.assert_allclose(n0 + n1, n, rtol=1e-12, atol=1e-14 * np.abs(n).max())


@pytest.mark.parametrize("name", ["tiny", "tiny_block", "tiny_free"])
def test_reduced_and_pre_elimination_restatements_agree_with_full(oracle_mod, name):
    """MatrixInversion.REDUCED / PRE_ELIMINATION (BundleAdjustment.java:261-267, 283-291, 1197-1453) restated literally
    (reduceNormalEquationSystem, solve on the leading numRows, extractReducedParameters): same iteration count, same
    adjusted parameters, same Omega and the same leading numRows x numRows block of Qxx as MatrixInversion.FULL."""
    from bundle_adjustment_amd import scene
    from bundle_adjustment_amd.problem import packed_to_full
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    v1, Q1, r1 = o.estimate(invert=1)
    k, U = o.reduced_rows(), fp.n_unknowns
    assert k == fp.rank_defect + 3 * fp.n_points + int((fp.io_col >= 0).sum() + (fp.dist_col >= 0).sum()) == int(fp.eo_col.min())
    Qf = packed_to_full(Q1, U)[:k, :k]
    for mode in (2, 3):
        v, Q, r = o.estimate(invert=mode)
        assert r.state == 1 and r.iterations == r1.iterations
        assert (np.abs(v - v1) / np.maximum(np.abs(v1), 1e-3)).max() < 1e-10
        assert abs(r.omega - r1.omega) <= 1e-10 * r1.omega
        Qm = packed_to_full(Q, U)[:k, :k]
        assert np.abs(Qm - Qf).max() <= 1e-9 * np.abs(Qf).max()


def test_reduced_mode_leaves_the_unsolved_rhs_in_the_eo_step(oracle_mod):
    """in the last pass of MatrixInversion.REDUCED the exterior-orientation entries of dx are V_c^2 n_c
    (the unsolved, twice-scaled right-hand side), not the solved step."""
    from bundle_adjustment_amd import scene
    fp = scene.config("tiny_block")
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    N, n, V = o.build(fp.values, s2)
    n0 = n.copy()
    o.precondition(V, N, n)
    o.reduce(N, n, False)
    k = o.reduced_rows()
    assert o.L.oracle_solve(k, oracle_mod._p(N), oracle_mod._p(n), 1) == 0
    o.precondition(V, N, n)
    np.testing.assert_allclose(n[k:], V[k:] ** 2 * n0[k:], rtol=1e-15)
    dx_full, _, _, _ = o.step(fp.values, s2)
    np.testing.assert_allclose(n[:k], dx_full[:k], rtol=0, atol=1e-9 * np.abs(dx_full[:k]).max())
    assert np.abs(n[k:] - dx_full[k:]).max() > 1e-3 * np.abs(dx_full[k:]).max()      # far from the solution away from convergence


# ---- ground truth (oracle/ba_exact.c): NOT the reference's arithmetic, the yardstick for the accuracy study ---------------------
def test_extended_precision_weight_is_certified_in_binary128(oracle_mod):
    """sigma0^2 inv(D) in x87 extended precision + one compensated Newton step, against a binary128 residual: an ill-conditioned
    dispersion (cond ~ 1e5; config 4's: 2e7) where the fp64 dpptrf + dpptri of the reference (DOPG:82-86) loses 5 digits."""
    L = oracle_mod.lib()
    fp = scene.make_scene(3, 300, 300, dist=scene.DIST_RADIAL, weights="block", n_control=4)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    Ph, Pl = o.exact_block_weight(s2, 0)
    P64 = o.block_weight(s2, 0)
    m = Ph.shape[0]
    D = fp.blk_disp[fp.blk_disp_offset[0]:fp.blk_disp_offset[0] + m * m].copy()
    cond = np.linalg.cond(D.reshape(m, m))
    assert cond > 1e4
    pd = oracle_mod._p
    r_exact = L.oracle_inverse_residual_q(m, pd(D), pd(Ph), pd(Pl), s2, 0, 64)      # 64 rows: binary128 is software arithmetic
    r_fp64 = L.oracle_inverse_residual_q(m, pd(D), pd(P64), None, s2, 0, 64)
    assert r_exact < 1e-14 and r_exact < 1e-3 * r_fp64, (r_exact, r_fp64, cond)
    # the fp64 inverse agrees with it to cond * eps, no better
    err = np.abs(P64 - Ph).max() / np.abs(Ph).max()
    assert 1e-16 < err < 1e-9, err


def test_every_weight_matrix_is_certified_by_binary128_probes(oracle_mod):
    """The row-wise certificate above costs m^3 software multiplications and therefore looks at 64 rows of one matrix.  The probe form
    (oracle_inverse_residual_probe_q: |v - D P v / sigma0^2| for random sign vectors, binary128) covers every row for 2 m^2 per vector:
    here every block of a scene, and in the committed tests/golden/cfg4/cfg4_weight_certificate.json (make_weight_certificate.py) all
    500 weight matrices of config 4 -- the ones the truth fixture cfg4_exactN.npz is assembled from."""
    import json
    import os
    L = oracle_mod.lib()
    pd = oracle_mod._p
    fp = scene.make_scene(3, 300, 300, dist=scene.DIST_RADIAL, weights="block", n_control=4)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    for b in range(fp.n_image_blocks):
        Ph, Pl = o.exact_block_weight(s2, b)
        P64 = o.block_weight(s2, b)
        m = Ph.shape[0]
        D = np.ascontiguousarray(fp.blk_disp[fp.blk_disp_offset[b]:fp.blk_disp_offset[b] + m * m])
        r_exact = L.oracle_inverse_residual_probe_q(m, pd(D), pd(Ph), pd(Pl), s2, 2, b + 1)
        r_fp64 = L.oracle_inverse_residual_probe_q(m, pd(D), pd(P64), None, s2, 2, b + 1)
        assert r_exact < 2e-15 and r_exact < 1e-3 * r_fp64, (b, r_exact, r_fp64)
        # the probe sees what the exact row-wise residual sees (a row's 2-norm against its largest entry: within m^(1/2))
        rows = L.oracle_inverse_residual_q(m, pd(D), pd(Ph), pd(Pl), s2, 0, 8)
        assert rows <= r_exact * 4 and r_exact <= rows * 4 * m ** 0.5, (rows, r_exact)
    meta = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg4", "cfg4_weight_certificate.json")))
    assert meta["blocks"] == 500 and len(meta["per_block_residual_exact"]) == 500 and meta["order_min"] == meta["order_max"] == 1000
    assert meta["residual_exact_max"] < 1e-13                                   # every one of the 500 is an inverse to binary128's witness
    assert meta["residual_fp64_min"] > 1000 * meta["residual_exact_max"]         # ... and every fp64 dpptri weight is visibly not
    assert 1e-12 < meta["fp64_weight_error_median"] < 1e-9                      # cond(D) eps: what separates the oracle's N from the exact one


@pytest.mark.parametrize("name", ["tiny", "tiny_block", "tiny_free"])
def test_extended_precision_assembly_agrees_with_the_restatement(oracle_mod, name):
    """oracle_exact_accumulate (every group kind: ordinary 2 x 2 / diagonal image points, dense image blocks, scale bar, dense and
    diagonal directly observed groups) against oracle_accumulate on well-conditioned scenes: the same N and n to fp64 rounding; the
    (hi, lo) pair carries the extra bits."""
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    N, n = o.accumulate(fp.values, s2)
    Nh, Nl, nh, nl = o.exact_accumulate(fp.values, s2)
    assert np.abs(N - Nh).max() <= 2e-14 * np.abs(Nh).max()
    assert np.abs(n - nh).max() <= 2e-13 * np.abs(nh).max()
    assert np.abs(Nl).max() <= 2.0 ** -52 * np.abs(Nh).max() and np.abs(Nl).max() > 0
    # extended residual of the (hi, lo) system: r = b - (N_hi + N_lo) x
    L = oracle_mod.lib()
    U = fp.n_unknowns
    x = np.random.default_rng(1).standard_normal(U)
    y = np.zeros(U)
    L.oracle_matvec_ld2(U, oracle_mod._p(Nh), oracle_mod._p(Nl), oracle_mod._p(x), oracle_mod._p(y))
    full = packed_to_full(Nh, U) + packed_to_full(Nl, U)
    np.testing.assert_allclose(y, full @ x, rtol=1e-12, atol=1e-12 * np.abs(y).max())


def test_exact_fixtures_say_the_reference_algorithm_is_the_noisier_side():
    """tests/golden/*/..._exactN.json (make_exactN.py): what the truth fixtures hold about the ORACLE (= the reference's dpptrf + dpptri
    weights, fp64 stacking, dspsv + dsptri): its Qxx is 1e-8 (config 3 size) / 2.3e-7 (config 4) from the exact inverse of the exactly
    assembled system -- the GPU tests hold the device against the same truth (test_gpu_termination.py, test_gpu_cfg4_golden.py)."""
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for cfg, lo, hi in (("cfg3b", 2e-9, 5e-8), ("cfg4", 5e-8, 1e-6)):
        meta = json.load(open(os.path.join(g, cfg, f"{cfg}_exactN.json")))
        z = np.load(os.path.join(g, cfg, f"{cfg}_exactN.npz"))
        assert lo < meta["oracle_Qsample_err"] < hi
        assert meta["binary128_residual_exact_P"] < 1e-3 * meta["binary128_residual_oracle_P"]
        assert meta["Qcols_last_correction"] < 2e-10          # the truth itself is converged to ~cond * 2^-64
        Q = z["Qsample_true"]
        assert np.abs(Q - Q.T).max() == 0 and np.all(np.diag(Q) > 0)


def test_strips_layout_scene_has_locality_and_adjusts(oracle_mod):
    """scene layout 'strips' (round 4: a block flown in strips, the scene behind `cfg4_local`): neighbouring points -- neighbouring columns
    through the first-seen numbering of BA:667-782 -- share most of their images, where the SURVEY 8(d) scene's random sub-sampling leaves
    ~10 %; and it is a well-posed adjustment (the oracle's loop converges on it)."""
    def shared(fp):
        pc = np.asarray(fp.point_col).reshape(-1, 3)[:, 0]
        order = np.argsort(pc)
        imgs = [set() for _ in range(fp.n_points)]
        for ip, pt in zip(fp.ip_image.tolist(), fp.ip_point.tolist()):
            imgs[pt].add(ip)
        s = [len(imgs[int(order[i])] & imgs[int(order[i + 1])]) / max(1, min(len(imgs[int(order[i])]), len(imgs[int(order[i + 1])])))
             for i in range(fp.n_points - 1)]
        return float(np.mean(s))
    kw = dict(dist=scene.DIST_RADIAL, weights="block", n_control=6, control_dense=True)
    local = scene.make_scene(60, 600, 60, layout="strips", **kw)
    rand = scene.make_scene(60, 600, 60, layout="sphere", **kw)
    sl, sr = shared(local), shared(rand)
    assert sl > 0.5 and sl > 1.5 * sr, (sl, sr)          # config 4's size: 0.53 against 0.135
    again = scene.make_scene(60, 600, 60, layout="strips", **kw)
    assert np.array_equal(again.ip_x, local.ip_x) and np.array_equal(again.blk_disp, local.blk_disp)      # seeded, reproducible
    # (ordinary weights for the convergence check: the oracle's literal loop nest over a joint dispersion is Theta(m^2 k^2))
    plain = scene.make_scene(60, 600, 60, layout="strips", dist=scene.DIST_RADIAL, weights="diag", n_control=6)
    v, _, r = oracle_mod.Oracle(plain).estimate(invert=False)
    assert r.state == 1 and r.max_abs_dx <= 1.0536712127723509e-8
