"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.

Tolerances (north_star): indexing bit-exact; estimated coordinates and covariances within 1e-9 relative.
Row-level and matrix-level checks are tighter (1e-11 .. 1e-12), see each test.
"""
import numpy as np
import pytest

import helpers
from bundle_adjustment_amd import engine, scene
from bundle_adjustment_amd.problem import full_to_packed, packed_to_full

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alay,blay,kmode,lower", [(0, 0, 0, False), (0, 0, 0, True), (0, 1, 3, False), (0, 1, 1, False),
                                                    (1, 1, 2, True), (1, 0, 0, False)])
def test_gemm_f64_family(alay, blay, kmode, lower):
    rng = np.random.default_rng(5)
    M, N, K = 256, 384 if not lower else 256, 256
    if kmode in (1, 2):
        K = M
    if kmode == 3:
        K = N
    Aop = rng.normal(size=(M, K)); Bop = rng.normal(size=(K, N)); C0 = rng.normal(size=(M, N))
    # triangular operands for the restricted-k modes (the kernel skips the zero part)
    tr = np.arange(M)[:, None] // 128; tk = np.arange(K)[None, :] // 128
    if kmode == 1:
        Aop = np.where(tk <= tr, Aop, 0.0)
    if kmode == 2:
        Aop = np.where(tk >= tr, Aop, 0.0)
    if kmode == 3:
        Bop = np.where((np.arange(K)[:, None] // 128) >= (np.arange(N)[None, :] // 128), Bop, 0.0)
    A = Aop if alay == 0 else np.ascontiguousarray(Aop.T)
    B = np.ascontiguousarray(Bop.T) if blay == 0 else Bop
    ref = 0.75 * Aop @ Bop - 0.5 * C0
    got, _ = engine.dense_gemm(alay, blay, A, B, C0, M, N, K, alpha=0.75, beta=-0.5, lower_only=lower, kmode=kmode)
    if lower:
        mask = (np.arange(M)[:, None] // 128) >= (np.arange(N)[None, :] // 128)
        np.testing.assert_allclose(got[mask], ref[mask], rtol=0, atol=1e-11)
        np.testing.assert_array_equal(got[~mask], C0[~mask])
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)


@pytest.mark.parametrize("n", [5, 128, 300, 1000])
def test_dense_spd_solve_and_inverse(n):
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n + 20))
    S = G @ G.T / n + np.eye(n)
    b = rng.normal(size=(3, n))
    x, ap, _ = engine.dense_spd_solve_packed(full_to_packed(S), b, invert=True)
    ref = np.linalg.solve(S, b.T).T
    np.testing.assert_allclose(x, ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(packed_to_full(ap, n), np.linalg.inv(S), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n", [100, 200, 384, 500, 640, 1100, 2100, 4200])
def test_dense_spd_solve_one_right_hand_side(n, monkeypatch):
    """ONE right-hand side takes the polling-wave form of the backward chain (backsolve_chain8_kernel, dense.hip; by default from 24
    block columns on, here from one): 1, 2, 3, 4, 5, 9, 17 and 33 block columns -- no predecessor, fewer predecessors than
    pre-multiplied blocks, streams of odd and even length, one and two workgroups per block column, and streams longer than the ring
    between the polling wave and the streaming waves (8 slots)."""
    monkeypatch.setenv("JAICOV_CHAIN8_MIN_NB", "1")
    rng = np.random.default_rng(7 * n)
    G = rng.normal(size=(n, n + 20))
    S = G @ G.T / n + np.eye(n)
    b = rng.normal(size=(1, n))
    x, _, _ = engine.dense_spd_solve_packed(full_to_packed(S), b)
    np.testing.assert_allclose(x, np.linalg.solve(S, b.T).T, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("n", [128, 300, 1000, 2100])
def test_dataflow_factorisation_on_small_orders(n, monkeypatch):
    """The dataflow Cholesky (csrc/cholflow.hip) is the default from 24 block columns on; forced here on 1, 3, 8 and 17
    block columns (ragged last block included), against LAPACK."""
    monkeypatch.setenv("JAICOV_FLOW_MIN_BLOCKS", "1")
    rng = np.random.default_rng(n + 1)
    G = rng.normal(size=(n, n + 20))
    S = G @ G.T / n + np.eye(n)
    b = rng.normal(size=(3, n))
    x, ap, _ = engine.dense_spd_solve_packed(full_to_packed(S), b, invert=True)
    np.testing.assert_allclose(x, np.linalg.solve(S, b.T).T, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(packed_to_full(ap, n), np.linalg.inv(S), rtol=1e-9, atol=1e-12)
    with pytest.raises(engine.EngineError) as ei:
        S[n // 2, n // 2] = -1.0
        engine.dense_spd_solve_packed(full_to_packed(S), b)
    assert ei.value.code == 1


@pytest.mark.parametrize("form", ["chain", "chain2", "chain3", "two_step", "one_kernel", "streams"])
@pytest.mark.parametrize("n", [384, 1152, 2304])
def test_factor_tile_by_tile(n, form, monkeypatch):
    """The Cholesky factor itself, every 128 x 128 tile against LAPACK, under every form of the factorisation (the chain
    form's workgroups each own particular tiles: a solve / inverse check alone can hide which one is wrong)."""
    import ctypes as C
    monkeypatch.setenv("JAICOV_FLOW_MIN_BLOCKS", "1")
    if form != "chain":
        monkeypatch.setenv("JAICOV_FACTOR_FORM", form)          # dense.hip factor_form(): the non-default forms are named here
    lib = engine.load_library()
    lib.jaicov_debug_potrf_factor.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n + 20))
    S = np.ascontiguousarray(G @ G.T / n + np.eye(n))
    out = np.zeros((n, n))
    assert lib.jaicov_debug_potrf_factor(n, S.ctypes.data, out.ctypes.data) == 0
    np.testing.assert_allclose(np.tril(out), np.linalg.cholesky(S), rtol=0, atol=1e-12)


@pytest.mark.parametrize("form", ["chain", "chain3", "two_step", "one_kernel"])
@pytest.mark.parametrize("split", ["2:0", "3:8"])
def test_factor_tile_by_tile_with_split_update_ranges(form, split, monkeypatch):
    """Round 5: the tasks of the late block columns hand the first part(s) of their update range to partial-sum tasks and add the sums
    before they store / finish their tile (cholflow.hip, FLOW_PART; the default from 80 block columns on).  Forced here on 30 block
    columns (2:0 = two pieces from the first column that is long enough, 16; 3:8 = three pieces, from column 24) under the forms that
    run the tile kernel, tile by tile against LAPACK -- the diagonal and subdiagonal tiles of the chain forms are split like the others."""
    import ctypes as C
    monkeypatch.setenv("JAICOV_FLOW_MIN_BLOCKS", "1")
    monkeypatch.setenv("JAICOV_FLOW_SPLIT", split)
    if form != "chain":
        monkeypatch.setenv("JAICOV_FACTOR_FORM", form)
    lib = engine.load_library()
    lib.jaicov_debug_potrf_factor.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    n = 3840
    rng = np.random.default_rng(n + len(form))
    G = rng.normal(size=(n, n + 20))
    S = np.ascontiguousarray(G @ G.T / n + np.eye(n))
    out = np.zeros((n, n))
    assert lib.jaicov_debug_potrf_factor(n, S.ctypes.data, out.ctypes.data) == 0
    np.testing.assert_allclose(np.tril(out), np.linalg.cholesky(S), rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [6400, 10112, 10496])
def test_factor_tile_by_tile_at_the_orders_between(n):
    """50 block columns: the chain form with its third workgroup (default below 80 block columns since round 4's last sweep); 79: the largest
    order that runs so; 82: two chain workgroups and, since round 5, split update ranges from block column 41 on (the default rule).
    Held tile by tile against LAPACK like the small orders above."""
    import ctypes as C
    lib = engine.load_library()
    lib.jaicov_debug_potrf_factor.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, 64))
    S = np.ascontiguousarray(G @ G.T / 64 + np.diag(1.0 + rng.random(n)))
    out = np.zeros((n, n))
    assert lib.jaicov_debug_potrf_factor(n, S.ctypes.data, out.ctypes.data) == 0
    np.testing.assert_allclose(np.tril(out), np.linalg.cholesky(S), rtol=0, atol=1e-11)


def test_gather_with_partner_ranges_longer_than_a_wave():
    """A block flown in strips (scene layout "strips": neighbouring points are neighbouring columns) with dense per-image dispersions and 150
    points per image: an image's partners of a row point fill SEVERAL 64-lane segments of one strip, and most (image, strip) pairs are empty.
    That is the regime the deterministic gather's pass-major distribution was built for (round 5: segments dealt to the waves pass by pass,
    the turn word counts segments, images without a partner in the strip are never visited) and the small random scenes never reach (at
    most 60 points per image).  The oracle's literal stacking needs minutes at this size, so the deterministic (pass-major) engine is held
    to the arrival-order engine -- the image-major loop, which the other tests of this file hold to the oracle: the reduced N and n to summation-
    order rounding, the step and Omega; and to itself: the same bits on a rebuild."""
    fp = scene.make_scene(40, 600, 150, layout="strips", dist=scene.DIST_FULL, weights="block", n_control=8, control_dense=True)
    s2, U = fp.sigma2apriori, fp.n_unknowns
    pc = fp.point_col.reshape(-1, 3)[:, 0]
    longest = 0
    for img in range(fp.n_images):
        cols = pc[fp.ip_point[fp.ip_image == img]]
        longest = max(longest, int(np.bincount(cols[cols >= 0] // 256).max()))
    assert longest > 64                                  # more partners of one image in 256 columns than a wave has lanes
    out = {}
    for det in (True, False):
        eng = engine.Engine(fp, deterministic=det)
        eng.set_parameters(fp.values)
        eng.build(s2, 0.0)
        e0 = eng.reduced_order()
        assert e0 == U - 6 * fp.n_images               # the fused (EO-eliminating) gather ran
        N1, n1 = eng.get_normal()
        dx = eng.solve(False)
        om = eng.omega(s2, dx)
        eng.build(s2, 0.0)
        N2, n2 = eng.get_normal()
        if det:
            assert np.array_equal(N1, N2) and np.array_equal(n1, n2)
        out[det] = (N1[:e0 * (e0 + 1) // 2], n1[:e0], dx, om)
        eng.close()
    Nd, nd, dxd, omd = out[True]
    Na, na, dxa, oma = out[False]
    idx = np.arange(nd.size, dtype=np.int64)
    dg = np.sqrt(np.abs(Na[idx * (idx + 3) // 2])); dg[dg == 0] = 1.0
    r = np.repeat(idx, idx + 1)                          # row of every packed entry (row-major lower = packed 'U')
    c = np.concatenate([np.arange(k + 1) for k in range(nd.size)]) if nd.size < 4000 else None
    assert c is not None
    assert (np.abs(Nd - Na) / (dg[r] * dg[c])).max() < 1e-12
    np.testing.assert_allclose(nd, na, rtol=0, atol=1e-12 * np.abs(na).max())
    np.testing.assert_allclose(dxd, dxa, rtol=0, atol=1e-9 * np.abs(dxa).max())
    assert abs(omd - oma) <= 1e-10 * oma


def test_keep_rule_is_derived_from_measured_residency():
    """Round 5: which of the tile kernel's last workgroups take no ticket is no longer "the last eight of every XCD of an
    8 x 32 x 2 grid" but follows from a residency measurement beside stand-ins of the chain workgroups (cholflow.hip,
    flow_measure_residency): XCDs and shader engines seen, blocks dealt per XCD, how many of them stay queued where a CU is taken."""
    import ctypes as C
    lib = engine.load_library()
    out = (C.c_int * 8)()
    assert lib.jaicov_debug_flow_residency(out) == 0
    valid, n_xcd, n_se, dealt, resident, queued, keep, se_cap = list(out)
    assert valid == 1 and n_xcd >= 1 and n_se >= 1 and dealt >= resident > 0 and queued == dealt - resident and se_cap >= 1
    # the worst case of the geometry: the queue of such an XCD stops when the shader engine with the taken CU is offered one block too many
    leave = max(queued, dealt - n_se * min(se_cap, dealt)) if queued > 0 else 0
    assert keep == (n_xcd * (dealt - leave) if leave and dealt > leave else 0)
    if (n_xcd, n_se, dealt) == (8, 4, 64):          # the MI355X: 59 of 64 resident on an XCD whose reserved CU is taken (15 + 15 + 15 + 14), 5 queued
        # (59 / 5 in a fresh process, 56-58 / 6-8 when the first solver was measured at the end of a config-4 engine's creation: the dispatcher's
        # round-robin over the shader engines stood elsewhere; the rule is the worst case, 64 - 4 x 14 = 8 leave, whatever was seen)
        assert 56 <= resident <= 59 and se_cap == 14 and keep == 448


def test_dense_not_spd_reports_singular():
    S = -np.eye(130)
    with pytest.raises(engine.EngineError) as ei:
        engine.dense_spd_solve_packed(full_to_packed(S), np.ones((1, 130)))
    assert ei.value.code == 1


@pytest.mark.parametrize("name", ["pinhole", "radial", "full", "tangential2"])
def test_rows_match_golden_and_oracle(oracle_mod, name):
    sets = helpers.load_golden_rows()
    dist, cases = sets[name]["dist"], sets[name]["cases"]
    fp = helpers.problem_from_cases(dist, cases)
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    w, A = eng.get_rows(0, fp.n_image_points)
    o = oracle_mod.Oracle(fp)
    nd = len(dist)
    for i, c in enumerate(cases):
        wo, Ao, _, _ = o.rows(fp.values, i)
        np.testing.assert_allclose(w[i], wo, rtol=0, atol=1e-13)
        scale = np.abs(Ao[:, :12 + nd]).max(axis=1, keepdims=True)
        assert (np.abs(A[i, :, :12 + nd] - Ao[:, :12 + nd]) / np.maximum(np.abs(Ao[:, :12 + nd]), 1e-6 * scale)).max() < 1e-11
        for r, key in enumerate(("Ax", "Ay")):
            ref = np.array(c[key])
            err = np.abs(A[i, r, :12 + nd] - ref) / np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
            assert err.max() < 1e-10
    eng.close()


@pytest.mark.parametrize("name", ["tiny", "tiny_block", "tiny_free"])
def test_normal_equations_match_oracle(oracle_mod, name):
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    for lam in (0.0, 0.5):
        No, no, Vo = o.build(fp.values, s2, lam)
        eng = engine.Engine(fp)
        eng.set_parameters(fp.values)
        eng.prepare_inverse(True)          # full system (no EO pre-elimination), as the reference assembles it
        eng.build(s2, lam)
        N, n = eng.get_normal()
        U = fp.n_unknowns
        Nf, Nof = packed_to_full(N, U), packed_to_full(No, U)
        dg = np.sqrt(np.abs(np.diag(Nof))); dg[dg == 0] = 1.0
        assert (np.abs(Nf - Nof) / np.outer(dg, dg)).max() < 1e-11
        np.testing.assert_allclose(n, no, rtol=0, atol=1e-11 * np.abs(no).max())
        eng.close()


@pytest.mark.parametrize("form", ["t_vector", "no_fork", "materialise"])
def test_assembly_forms_give_the_same_system(form, monkeypatch):
    """The alternative forms of the dense-block assembly (JAICOV_ASSEMBLY_FORM, assemble.hip: vector form of T = Dinv [A_c | w],
    camera-side kernels in front of the gather instead of beside it, P' written out) assemble the same EO-reduced system as the
    default and the same step (the default is held to the oracle by the tests around this one)."""
    fp = scene.make_scene(12, 150, 90, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    s2 = fp.sigma2apriori
    res = []
    for on in (False, True):
        if on:
            monkeypatch.setenv("JAICOV_ASSEMBLY_FORM", form)
        eng = engine.Engine(fp)
        eng.set_parameters(fp.values)
        eng.build(s2, 0.5)
        N, n = eng.get_normal()
        res.append((N.copy(), n.copy(), eng.solve(False)))
        eng.close()
    (N0, n0, dx0), (N1, n1, dx1) = res
    np.testing.assert_allclose(N1, N0, rtol=0, atol=1e-12 * np.abs(N0).max())
    np.testing.assert_allclose(n1, n0, rtol=0, atol=1e-12 * np.abs(n0).max())
    np.testing.assert_allclose(dx1, dx0, rtol=0, atol=1e-10 * np.abs(dx0).max())


@pytest.mark.parametrize("name", ["tiny", "tiny_block", "tiny_free"])
@pytest.mark.parametrize("invert", [False, True])
def test_solve_matches_oracle(oracle_mod, name, invert):
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    dxo, Qo, _, _ = o.step(fp.values, s2, 0.0, invert)
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    eng.prepare_inverse(invert)
    eng.build(s2, 0.0)
    dx = eng.solve(invert)
    d = fp.rank_defect
    np.testing.assert_allclose(dx[d:], dxo[d:], rtol=0, atol=1e-9 * np.abs(dxo[d:]).max())
    if d:
        np.testing.assert_allclose(dx[:d], dxo[:d], rtol=0, atol=1e-6 * np.abs(dxo[:d]).max() + 1e-12)
    om_o = o.omega(fp.values, s2, dxo)
    assert abs(eng.omega(s2, dxo) - om_o) <= 1e-10 * om_o
    if invert:
        U = fp.n_unknowns
        Q = packed_to_full(eng.get_cofactor(), U); Qref = packed_to_full(Qo, U)
        sd = np.sqrt(np.abs(np.diag(Qref))); sd[sd == 0] = 1.0
        # the default assembly is deterministic since round 4 (fixed summation order): 1e-9 on every scene.  (With arrival-order
        # atomics tiny_block -- 5 images, 36 points, poorly conditioned -- moved by up to 1.2e-9 from run to run and was held to 1e-8.)
        tol = 1e-9
        assert (np.abs(Q - Qref)[d:, d:] / np.outer(sd, sd)[d:, d:]).max() < tol
        np.testing.assert_allclose(np.diag(Q)[d:], np.diag(Qref)[d:], rtol=tol)
        assert np.abs(Q - Qref).max() <= 1e-9 * np.abs(Qref).max()
        idx = np.array([d, d + 3, U - 1, d + 1], np.int32)
        np.testing.assert_array_equal(eng.get_cofactor_sub(idx), Q[np.ix_(idx, idx)])
        np.testing.assert_array_equal(eng.get_dispersion_sub(0.37, idx), 0.37 * Q[np.ix_(idx, idx)])   # writers: sigma2apost * Qxx
    eng.close()


@pytest.mark.parametrize("name", ["tiny", "tiny_block", "tiny_free"])
def test_estimate_matches_oracle(oracle_mod, name):
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    vo, Qo, ro = o.estimate()
    eng = engine.Engine(fp)
    v, r = eng.estimate()
    assert r.state == ro.state == 1
    assert r.iterations == ro.iterations
    scale = np.maximum(np.abs(vo), 1e-3)
    assert (np.abs(v - vo) / scale).max() < 1e-9
    assert abs(r.omega - ro.omega) <= 1e-9 * ro.omega
    U, d = fp.n_unknowns, fp.rank_defect
    Q = packed_to_full(eng.get_cofactor(), U); Qref = packed_to_full(Qo, U)
    # variances at the converged state: 1e-9 on every scene but tiny_block (4.3e-9), where two fp64 assemblies of the same normal matrix
    # give inverses that far apart (test_gpu_edge_cases.py holds that attribution with extended-precision inverses of both)
    qerr = float(np.abs(np.diag(Q)[d:] / np.diag(Qref)[d:] - 1.0).max())
    assert qerr < 1e-8, qerr
    eng.close()


@pytest.mark.parametrize("name", ["tiny", "tiny_block"])
@pytest.mark.parametrize("max_iter", [1, 2])
def test_iteration_limit_ends_like_the_reference(oracle_mod, name, max_iter):
    """maximalNumberOfIterations reached before the step falls below sqrt(eps) (BundleAdjustment.java:228, 344-355): the same state
    (NO_CONVERGENCE, -4), the same number of passes and the same last iterate as the oracle's restatement of that loop."""
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    vo, _, ro = o.estimate(max_iter=max_iter, invert=False)
    eng = engine.Engine(fp)
    v, r = eng.estimate(max_iter=max_iter, invert=engine.INVERT_NONE)
    assert r.state == ro.state and r.iterations == ro.iterations
    assert ro.state == -4, ro.state
    assert (np.abs(v - vo) / np.maximum(np.abs(vo), 1e-3)).max() < 1e-9
    assert abs(r.max_abs_dx - ro.max_abs_dx) <= 1e-8 * ro.max_abs_dx
    eng.close()


def test_estimate_lm_and_simulation(oracle_mod):
    fp = scene.config("tiny")
    o = oracle_mod.Oracle(fp)
    vo, _, ro = o.estimate(lam0=1.0)
    eng = engine.Engine(fp)
    v, r = eng.estimate(lam0=1.0)
    assert r.state == ro.state == 1 and r.iterations == ro.iterations
    assert (np.abs(v - vo) / np.maximum(np.abs(vo), 1e-3)).max() < 1e-9
    eng.close()


@pytest.mark.parametrize("name", ["tiny", "tiny_block", "tiny_free"])
@pytest.mark.parametrize("invert", [engine.INVERT_FULL, engine.INVERT_REDUCED])
def test_simulation_leaves_parameters_and_gives_the_oracles_cofactors(oracle_mod, name, invert):
    """EstimationType.SIMULATION (BundleAdjustment.java:830-831): n = 0 for the WHOLE system, so dx = 0 for every unknown --
    the exterior orientations the engine pre-eliminates (tiny_block) included -- the loop ends after its first two passes,
    Omega = 0 and Qxx is the cofactor matrix at the start values."""
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    vo, Qo, ro = o.estimate(simulation=True)
    np.testing.assert_array_equal(vo, fp.values)
    eng = engine.Engine(fp)
    v, r = eng.estimate(simulation=True, invert=invert)
    assert r.state == ro.state == 1 and r.iterations == ro.iterations
    np.testing.assert_array_equal(v, fp.values)            # bit-exact: nothing moved, EO included
    assert r.omega == 0.0 and r.max_abs_dx == 0.0
    U, d = fp.n_unknowns, fp.rank_defect
    k = eng.cofactor_order()
    Q = packed_to_full(eng.get_cofactor(), k); Qref = packed_to_full(Qo, U)[:k, :k]
    sd = np.sqrt(np.abs(np.diag(Qref))); sd[sd == 0] = 1.0
    assert (np.abs(Q - Qref)[d:, d:] / np.outer(sd, sd)[d:, d:]).max() < 1e-9
    # one simulated pass through the step-wise ABI as well: the step is exactly zero
    eng.set_parameters(fp.values)
    eng.build(fp.sigma2apriori, 0.0, simulation=True)
    dx = eng.solve(False)
    assert np.all(dx[d:] == 0.0)
    eng.close()


@pytest.mark.parametrize("name", ["tiny", "tiny_block"])
def test_reduced_reference_quirk_option_reproduces_the_references_last_pass(oracle_mod, name):
    """Engine option reduced_reference_quirk (BundleAdjustment.java:261-267, 273, 430, 450-461): a solve with
    MatrixInversion.REDUCED leaves V_c^2 n_c in the exterior-orientation entries of dx, which the reference then uses for
    Omega and the update.  Checked away from convergence (where the quirk is large) against the oracle's literal restatement,
    on the EO-pre-eliminated path (tiny_block) and on the full-order path (tiny)."""
    fp = scene.config(name)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    N, n, V = o.build(fp.values, s2)
    o.precondition(V, N, n)
    o.reduce(N, n, False)
    k, U, d = o.reduced_rows(), fp.n_unknowns, fp.rank_defect
    assert o.L.oracle_solve(k, oracle_mod._p(N), oracle_mod._p(n), 1) == 0
    o.precondition(V, N, n)
    dx_ref = n
    eng = engine.Engine(fp, reduced_reference_quirk=True)
    eng.set_parameters(fp.values)
    eng.prepare_inverse(engine.INVERT_REDUCED)
    eng.build(s2, 0.0)
    dx = eng.solve(engine.INVERT_REDUCED)
    np.testing.assert_allclose(dx[d:k], dx_ref[d:k], rtol=0, atol=1e-9 * np.abs(dx_ref[d:k]).max())
    np.testing.assert_allclose(dx[k:], dx_ref[k:], rtol=1e-9, atol=1e-12 * np.abs(dx_ref[k:]).max())
    om_ref = o.omega(fp.values, s2, dx_ref)
    assert abs(eng.omega(s2, dx) - om_ref) <= 1e-9 * om_ref
    # without the option the same call returns the solved step
    eng2 = engine.Engine(fp)
    eng2.set_parameters(fp.values)
    eng2.prepare_inverse(engine.INVERT_REDUCED)
    eng2.build(s2, 0.0)
    dx2 = eng2.solve(engine.INVERT_REDUCED)
    full, _, _, _ = o.step(fp.values, s2)
    np.testing.assert_allclose(dx2[k:], full[k:], rtol=0, atol=1e-9 * np.abs(full[k:]).max())
    eng.close(); eng2.close()


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_deterministic_assembly_gives_identical_bits(oracle_mod, lam):
    """Engine option `deterministic` (default on since round 4): the image groups are summed in a fixed order (camera block: partial
    sums added in block order; point x point blocks: the waves of a workgroup pass a turn word for their adds, image order; per-image
    reductions in wave order), so N, n and the step are the same BITS in every run -- and the comparison with the oracle can
    hold 1e-9 on the small, poorly conditioned scene where run-to-run noise of the default mode reaches 1.2e-9 in Qxx."""
    fp = scene.make_scene(12, 150, 80, dist=scene.DIST_FULL, weights="block", n_control=6, control_dense=True)
    s2 = fp.sigma2apriori
    res = []
    for rep in range(3):
        eng = engine.Engine(fp) if rep else engine.Engine(fp, deterministic=True)      # the default IS the deterministic form (round 4)
        eng.set_parameters(fp.values)
        eng.build(s2, lam)
        N, n = eng.get_normal()                      # the EO-reduced system (leading block) as assembled
        dx = eng.solve(False)
        eng.prepare_inverse(engine.INVERT_FULL)
        eng.build(s2, lam)
        Nf, nf = eng.get_normal()                    # the full system
        res.append((N, n, dx, Nf, nf))
        eng.close()
    def where(idx):      # (row, column) of packed 'U' indices
        c = (np.sqrt(8.0 * idx + 1.0) - 1.0) // 2
        c = c.astype(np.int64)
        return list(zip((idx - c * (c + 1) // 2).tolist(), c.tolist()))
    for r in res[1:]:
        for name, a, b in zip(("N reduced", "n reduced", "dx", "N full", "n full"), r, res[0]):
            bad = np.flatnonzero(a != b)
            assert bad.size == 0, (name, bad.size, where(bad[:8]) if name.startswith("N") else bad[:8].tolist(), a[bad[:4]], b[bad[:4]])
    No, no, _ = oracle_mod.Oracle(fp).build(fp.values, s2, lam)
    U = fp.n_unknowns
    Nf, Nof = packed_to_full(res[0][3], U), packed_to_full(No, U)
    dg = np.sqrt(np.abs(np.diag(Nof))); dg[dg == 0] = 1.0
    assert (np.abs(Nf - Nof) / np.outer(dg, dg)).max() < 1e-11
    # tiny_block, the scene whose default-mode tolerance had to be 1e-8 (test_solve_matches_oracle): 1e-9 holds deterministically
    fp2 = scene.config("tiny_block")
    o = oracle_mod.Oracle(fp2)
    dxo, Qo, _, _ = o.step(fp2.values, fp2.sigma2apriori, 0.0, True)
    eng = engine.Engine(fp2, deterministic=True)
    eng.set_parameters(fp2.values)
    eng.prepare_inverse(True)
    eng.build(fp2.sigma2apriori, 0.0)
    dx = eng.solve(True)
    U2 = fp2.n_unknowns
    Q = packed_to_full(eng.get_cofactor(), U2); Qref = packed_to_full(Qo, U2)
    sd = np.sqrt(np.abs(np.diag(Qref)))
    assert (np.abs(Q - Qref) / np.outer(sd, sd)).max() < 1e-9
    np.testing.assert_allclose(dx, dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    eng.close()
    # the arrival-order form (deterministic < 0: LDS / memory atomics, 0.3 ms per pass faster at config 4) stays a tested second path
    fast = engine.Engine(fp2, deterministic=False)
    fast.set_parameters(fp2.values)
    fast.build(fp2.sigma2apriori, 0.0)
    np.testing.assert_allclose(fast.solve(False), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    fast.close()


def test_interrupt_ends_the_loop_with_state_interrupt():
    """BundleAdjustment.interrupt() (BundleAdjustment.java:1455, polled at :240 and :320) = jaicov_neq_cancel."""
    fp = scene.config("tiny")
    eng = engine.Engine(fp)
    eng.cancel()
    v, r = eng.estimate()
    assert r.state == -1 and r.iterations == 1          # EstimationStateType.INTERRUPT after the first build
    np.testing.assert_array_equal(v, fp.values)         # BA:240 returns before the solve: nothing was updated
    v, r = eng.estimate()                               # the flag was cleared (BA:243 this.interrupt = false)
    assert r.state == 1
    eng.close()


def test_sharded_engines_sum_to_full(oracle_mod):
    fp = scene.config("tiny_block")
    s2 = fp.sigma2apriori
    full = engine.Engine(fp); full.set_parameters(fp.values); full.prepare_inverse(True); full.build(s2); N, n = full.get_normal()
    h = fp.n_images // 2
    a = engine.Engine(fp, image_range=(0, h), apply_shared=True)
    b = engine.Engine(fp, image_range=(h, fp.n_images), apply_shared=False)
    for e_ in (a, b):
        e_.set_parameters(fp.values); e_.prepare_inverse(True); e_.accumulate(s2)
    Na, na = a.get_normal(); Nb, nb = b.get_normal()
    np.testing.assert_allclose(Na + Nb, N, rtol=1e-12, atol=1e-13 * np.abs(N).max())
    np.testing.assert_allclose(na + nb, n, rtol=1e-12, atol=1e-13 * np.abs(n).max())
    for e_ in (full, a, b):
        e_.close()


def test_medium_config2_against_oracle(oracle_mod):
    """BASELINE config 2 (20 x 200, radial, diagonal W): full estimate vs oracle."""
    fp = scene.config("cfg2")
    o = oracle_mod.Oracle(fp)
    vo, Qo, ro = o.estimate()
    eng = engine.Engine(fp)
    v, r = eng.estimate()
    assert r.state == ro.state == 1 and r.iterations == ro.iterations
    assert (np.abs(v - vo) / np.maximum(np.abs(vo), 1e-3)).max() < 1e-9
    Q = eng.get_cofactor()
    U = fp.n_unknowns
    dq = np.diag(packed_to_full(Q, U)); dqo = np.diag(packed_to_full(Qo, U))
    np.testing.assert_allclose(dq, dqo, rtol=1e-9)
    eng.close()


def test_rccl_reduce_path_world1(oracle_mod):
    """The collective path on one GPU: accumulate -> all_reduce (backend nccl == RCCL) on the engine's device buffer ->
    finalize -> solve must reproduce the plain build + solve."""
    import os
    import torch
    import torch.distributed as dist
    from bundle_adjustment_amd import distributed
    fp = scene.config("tiny_block")
    s2 = fp.sigma2apriori
    ref = engine.Engine(fp); ref.set_parameters(fp.values); ref.build(s2); dx_ref = ref.solve(False); ref.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        lo, hi = distributed.partition_images(fp, 1)[0]
        eng = engine.Engine(fp, image_range=(lo, hi), apply_shared=True)
        eng.set_parameters(fp.values)
        dx = distributed.sharded_step(eng, dist, torch.device("cuda", 0), s2)
        np.testing.assert_allclose(dx, dx_ref, rtol=0, atol=1e-9 * np.abs(dx_ref).max())   # deterministic assembly on both sides (default since round 4)
        # the exchange of the pre-eliminated EO steps (all-reduce in place on the engine's device array, jaicov_neq_eo_step_buffer)
        # runs only with more than one rank: rehearsed here on one
        assert eng.reduced_order() < eng.U
        os.environ["JAICOV_FORCE_EO_EXCHANGE"] = "1"
        try:
            dx2 = distributed.sharded_step(eng, dist, torch.device("cuda", 0), s2)
        finally:
            del os.environ["JAICOV_FORCE_EO_EXCHANGE"]
        np.testing.assert_allclose(dx2, dx_ref, rtol=0, atol=1e-9 * np.abs(dx_ref).max())
        ptr, cnt = eng.eo_step_buffer()
        assert ptr and cnt == 6 * fp.n_images
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_eo_pre_elimination_matches_full_solve(oracle_mod, lam):
    """schur.hip: eliminating the exterior-orientation blocks per image (rank-6 downdate of each image's weight matrix)
    must give the same step as the reference's full bordered solve -- reduced order, identical dx incl. the EO part."""
    fp = scene.make_scene(8, 60, 40, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    dxo, _, _, _ = o.step(fp.values, s2, lam, False)
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    eng.build(s2, lam)
    assert eng.reduced_order() == fp.n_unknowns - 6 * fp.n_images      # the EO columns are gone from the system
    dx = eng.solve(False)
    np.testing.assert_allclose(dx, dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    assert abs(eng.omega(s2, dx) - o.omega(fp.values, s2, dxo)) <= 1e-9 * o.omega(fp.values, s2, dxo)
    with pytest.raises(engine.EngineError):
        eng.build(s2, lam); eng.solve(True)          # inverse needs the full system: prepare_inverse first
    eng.prepare_inverse(True); eng.build(s2, lam)
    assert eng.reduced_order() == fp.n_unknowns
    np.testing.assert_allclose(eng.solve(True), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    eng.close()


def test_eo_pre_elimination_sharded(oracle_mod):
    """Two engines over disjoint image ranges: reduced systems add up, each returns its own images' EO step."""
    fp = scene.make_scene(8, 60, 40, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    s2 = fp.sigma2apriori
    dxo, _, _, _ = oracle_mod.Oracle(fp).step(fp.values, s2, 0.0, False)
    full = engine.Engine(fp); full.set_parameters(fp.values); full.build(s2); N, n = full.get_normal()
    e0 = full.reduced_order()
    a = engine.Engine(fp, image_range=(0, 3), apply_shared=True)
    b = engine.Engine(fp, image_range=(3, fp.n_images), apply_shared=False)
    for e_ in (a, b):
        e_.set_parameters(fp.values); e_.accumulate(s2)
    Na, na = a.get_normal(); Nb, nb = b.get_normal()
    np.testing.assert_allclose(Na + Nb, N, rtol=1e-11, atol=1e-12 * np.abs(N).max())
    np.testing.assert_allclose(na + nb, n, rtol=1e-11, atol=1e-12 * np.abs(n).max())
    dx = full.solve(False)
    np.testing.assert_allclose(dx, dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    assert np.all(dx[e0:] != 0)
    for e_ in (full, a, b):
        e_.close()


@pytest.mark.parametrize("free_network", [False, True])
def test_reduced_inverse_is_the_block_of_the_full_cofactor(oracle_mod, free_network):
    """MatrixInversion.REDUCED / PRE_ELIMINATION (BA:261-267): the final pass inverts the system from which the exterior
    orientations were eliminated; that inverse is the (border, points, IO, distortion) block of the full Qxx = K^-1,
    which the oracle computes with the reference's dspsv + dsptri on the full bordered system."""
    if free_network:
        fp = scene.make_scene(8, 60, 40, dist=scene.DIST_FULL, weights="block", n_control=0, scale_bar=True)
    else:
        fp = scene.make_scene(8, 60, 40, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    dxo, Qo, _, _ = o.step(fp.values, s2, 0.0, True)
    U, d = fp.n_unknowns, fp.rank_defect
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    eng.prepare_inverse(engine.INVERT_REDUCED)
    eng.build(s2, 0.0)
    e0 = eng.reduced_order()
    assert e0 == U - 6 * fp.n_images
    dx = eng.solve(engine.INVERT_REDUCED)
    np.testing.assert_allclose(dx[d:], dxo[d:], rtol=0, atol=1e-9 * np.abs(dxo[d:]).max())
    assert eng.cofactor_order() == e0
    Q = packed_to_full(eng.get_cofactor(), e0)
    Qref = packed_to_full(Qo, U)[:e0, :e0]
    sd = np.sqrt(np.abs(np.diag(Qref))); sd[sd == 0] = 1.0
    assert (np.abs(Q - Qref)[d:, d:] / np.outer(sd, sd)[d:, d:]).max() < 1e-9
    assert np.abs(Q - Qref).max() <= 1e-9 * np.abs(Qref).max()
    idx = np.array([d, d + 3, e0 - 1, d + 1], np.int32)
    np.testing.assert_array_equal(eng.get_cofactor_sub(idx), Q[np.ix_(idx, idx)])
    # FULL afterwards still works and agrees on the block
    eng.prepare_inverse(engine.INVERT_FULL); eng.build(s2, 0.0); eng.solve(engine.INVERT_FULL)
    assert eng.cofactor_order() == U
    Qf = packed_to_full(eng.get_cofactor(), U)
    assert (np.abs(Qf[:e0, :e0] - Q)[d:, d:] / np.outer(sd, sd)[d:, d:]).max() < 1e-9
    eng.close()


@pytest.mark.parametrize("name", ["tiny_block", "mid_block"])
def test_dense_contraction_mode_matches_structure_aware_and_oracle(oracle_mod, name):
    """assembly_mode = 1 (densemode.hip): J'WJ of the jointly dispersed image groups as the dense contraction A'(PA) on the
    matrix cores -- the literal full-weight branch of stackNormalEquationSystem (PDF:486-498) -- must give the normal
    equations of the structure-aware path and of the oracle, and the same step."""
    fp = scene.config(name) if name == "tiny_block" else scene.make_scene(12, 150, 90, dist=scene.DIST_FULL, weights="block",
                                                                         n_control=5, control_dense=True)
    s2 = fp.sigma2apriori
    U = fp.n_unknowns
    dense = engine.Engine(fp, assembly_mode=1)
    dense.set_parameters(fp.values)
    dense.build(s2, 0.0)
    assert dense.reduced_order() == U            # no EO pre-elimination in this mode
    Nd, nd = dense.get_normal()
    ref = engine.Engine(fp)
    ref.set_parameters(fp.values)
    ref.prepare_inverse(engine.INVERT_FULL)      # full system from the structure-aware kernels
    ref.build(s2, 0.0)
    Ns, ns = ref.get_normal()
    Nf, Nsf = packed_to_full(Nd, U), packed_to_full(Ns, U)
    dg = np.sqrt(np.abs(np.diag(Nsf))); dg[dg == 0] = 1.0
    assert (np.abs(Nf - Nsf) / np.outer(dg, dg)).max() < 1e-11
    np.testing.assert_allclose(nd, ns, rtol=0, atol=1e-10 * np.abs(ns).max())
    if name == "tiny_block":
        No, no, _ = oracle_mod.Oracle(fp).build(fp.values, s2, 0.0)
        Nof = packed_to_full(No, U)
        assert (np.abs(Nf - Nof) / np.outer(dg, dg)).max() < 1e-11
    dxd, dxs = dense.solve(False), ref.solve(engine.INVERT_FULL)
    np.testing.assert_allclose(dxd, dxs, rtol=0, atol=1e-9 * np.abs(dxs).max())
    dense.set_profiling(True); dense.kernel_stats(reset=True)
    dense.build(s2, 0.0)
    ks = dense.kernel_stats()
    assert ks["dense_passes"] == 1 and ks["dense_gemm_ms"] > 0 and ks["dense_flops"] > 0
    dense.close(); ref.close()


def test_fp32_accumulate_contraction_is_measurably_worse(oracle_mod):
    """BASELINE config 5's precision sweep in miniature (scripts/precision_sweep.py runs it at full size): assembly_mode = 2
    contracts J'WJ with fp32 operands and fp32 MFMA accumulation.  On a 530-unknown scene the step moves by ~1e-3 of its
    size and diag Qxx by ~1e-2 -- four to eight orders of magnitude above the fp64 paths; at config 3/4 size the fp32
    normal matrix is no longer positive definite (profiles/r01_precision_sweep_*.json)."""
    fp = scene.make_scene(12, 150, 90, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    s2 = fp.sigma2apriori
    dxs = {}
    for mode in (1, 2):
        eng = engine.Engine(fp, assembly_mode=mode)
        eng.set_parameters(fp.values)
        eng.build(s2, 0.0)
        dxs[mode] = eng.solve(False)
        eng.close()
    dev = np.abs(dxs[2] - dxs[1]).max() / np.abs(dxs[1]).max()
    assert 1e-7 < dev < 1e-1
    dxo, _, _, _ = oracle_mod.Oracle(fp).step(fp.values, s2, 0.0, False)
    np.testing.assert_allclose(dxs[1], dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())


@pytest.mark.parametrize("factorisation", ["default", "streams", "dataflow_two_step", "dataflow_one_kernel"])
def test_config3_step_against_oracle(oracle_mod, factorisation, monkeypatch):
    """BASELINE config 3 (100 images x 1 000 points, full interior set, 2x2 correlated image points, U = 3 614): one pass
    against the oracle's packed Bunch-Kaufman solve, normal equations included.  At this order (29 block
    columns) the default is the dataflow factorisation in its chain form; the other cases force the
    stream-scheduled one, the dataflow form with the separate diagonal kernel and the one-kernel form."""
    if factorisation != "default":                      # one_kernel: the form taken when kernels cannot overlap (counter collection)
        monkeypatch.setenv("JAICOV_FACTOR_FORM", {"streams": "streams", "dataflow_two_step": "two_step", "dataflow_one_kernel": "one_kernel"}[factorisation])
    fp = scene.config("cfg3")
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    U = fp.n_unknowns
    No, no, _ = o.build(fp.values, s2, 0.0)
    dxo, _, _, _ = o.step(fp.values, s2, 0.0, False)
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    eng.prepare_inverse(engine.INVERT_FULL)        # the unreduced N, as the reference stacks it (round 4: the default build pre-eliminates the EO of ordinary images too)
    eng.build(s2, 0.0)
    assert eng.reduced_order() == U
    N, n = eng.get_normal()
    dgo = np.sqrt(np.abs(No[np.arange(U) * (np.arange(U) + 3) // 2])); dgo[dgo == 0] = 1.0
    r, c = np.triu_indices(U)                      # packed 'U' is column-major upper: index r + c(c+1)/2
    rel = np.abs(N - No) / (dgo[r[np.argsort(r + c * (c + 1) // 2)]] * dgo[c[np.argsort(r + c * (c + 1) // 2)]])
    assert rel.max() < 1e-11
    np.testing.assert_allclose(n, no, rtol=0, atol=1e-11 * np.abs(no).max())
    dx = eng.solve(False)                          # order 3 614: 29 block columns
    np.testing.assert_allclose(dx, dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    eng.prepare_inverse(engine.INVERT_NONE)        # the product path: order 3 014 after the elimination, 24 block columns
    eng.build(s2, 0.0)
    assert eng.reduced_order() == U - 6 * fp.n_images
    np.testing.assert_allclose(eng.solve(False), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    eng.close()


@pytest.mark.parametrize("name", ["zernike_x", "zernike_y", "zernike_gradient", "zernike_mixed", "zernike_high_x", "zernike_high_y", "zernike_example_gradient"])
def test_zernike_rows_match_golden_and_oracle(oracle_mod, name):
    """Zernike X / Y / Gradient rows (ZernikeDistortionModelFactory.java:41-227) of the HIP kernel vs the oracle and vs the
    symbolic golden vectors (even radial orders)."""
    sets = helpers.load_golden_rows("jacobian_rows_zernike.json")
    dist, cases = sets[name]["dist"], sets[name]["cases"]
    fp = helpers.problem_from_cases(dist, cases)
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    w, A = eng.get_rows(0, fp.n_image_points)
    o = oracle_mod.Oracle(fp)
    nd = len(dist)
    for i, c in enumerate(cases):
        wo, Ao, _, _ = o.rows(fp.values, i)
        np.testing.assert_allclose(w[i], wo, rtol=0, atol=1e-13)
        scale = np.abs(Ao[:, :12 + nd]).max(axis=1, keepdims=True)
        assert (np.abs(A[i, :, :12 + nd] - Ao[:, :12 + nd]) / np.maximum(np.abs(Ao[:, :12 + nd]), 1e-6 * scale)).max() < 1e-10
        for r, key in enumerate(("Ax", "Ay")):
            ref = np.array(c[key])
            assert (np.abs(A[i, r, :12 + nd] - ref) / np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())).max() < 1e-9
    eng.close()


def test_zernike_odd_orders_and_adjustment_match_oracle(oracle_mod):
    """Odd radial orders have no independent truth (the reference truncates their exponents, ZDF:107,176,178): kernel and
    oracle restate the same formulas and must agree with each other -- rows, normal equations and the step of a small
    adjustment whose camera carries one coefficient of every Zernike model next to the radial set."""
    from bundle_adjustment_amd.problem import DIST_ZERNIKE_X, DIST_ZERNIKE_Y, DIST_ZERNIKE_Z
    import dataclasses
    from test_gpu_edge_cases import renumber
    base = scene.make_scene(8, 60, 40, dist=scene.DIST_RADIAL, weights="block", n_control=5, control_dense=True)
    extra = [(DIST_ZERNIKE_X, 1, 2e-4), (DIST_ZERNIKE_X, 7, -1e-4), (DIST_ZERNIKE_Y, 2, 1e-4), (DIST_ZERNIKE_Y, 12, 5e-5),
             (DIST_ZERNIKE_Z, 6, 1e-4), (DIST_ZERNIKE_Z, 13, -5e-5)]
    nd = base.dist_kind.size
    s = 3 * base.n_points + 3 + nd
    fp = renumber(base, cam_dist_begin=np.array([0, nd + len(extra)], np.int32),
                  dist_kind=np.concatenate([base.dist_kind, [k for k, _, _ in extra]]).astype(np.int32),
                  dist_order=np.concatenate([base.dist_order, [o for _, o, _ in extra]]).astype(np.int32),
                  values=np.concatenate([base.values[:s], [v for _, _, v in extra], base.values[s:]]), truth=None)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    U = fp.n_unknowns
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    w, A = eng.get_rows(0, 40)
    k = 12 + nd + len(extra)
    for i in range(40):
        wo, Ao, _, _ = o.rows(fp.values, i)
        np.testing.assert_allclose(w[i], wo, rtol=0, atol=1e-13)
        scale = np.abs(Ao[:, :k]).max(axis=1, keepdims=True)
        assert (np.abs(A[i, :, :k] - Ao[:, :k]) / np.maximum(np.abs(Ao[:, :k]), 1e-6 * scale)).max() < 1e-10
    No, no, _ = o.build(fp.values, s2, 0.0)
    dxo, _, _, _ = o.step(fp.values, s2, 0.0, False)
    eng.prepare_inverse(engine.INVERT_FULL)
    eng.build(s2, 0.0)
    N, n = eng.get_normal()
    Nf, Nof = packed_to_full(N, U), packed_to_full(No, U)
    dg = np.sqrt(np.abs(np.diag(Nof))); dg[dg == 0] = 1.0
    assert (np.abs(Nf - Nof) / np.outer(dg, dg)).max() < 1e-10
    eng.prepare_inverse(engine.INVERT_NONE)
    eng.build(s2, 0.0)
    np.testing.assert_allclose(eng.solve(False), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    eng.close()


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_sharded_reduce_buffers_with_damping(oracle_mod, lam):
    """Two engines over disjoint image ranges, summed through their reduce buffers as the multi-GPU host does (here both on
    one GPU, summed with torch): with the EO pre-elimination the buffer also carries the diagonal corrections of the LM
    damping (BA:814-822 damps the UNREDUCED diagonal), so the damped step must equal the oracle's."""
    import torch
    from bundle_adjustment_amd.distributed import DeviceArray
    fp = scene.make_scene(8, 60, 40, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    s2 = fp.sigma2apriori
    dxo, _, _, _ = oracle_mod.Oracle(fp).step(fp.values, s2, lam, False)
    a = engine.Engine(fp, image_range=(0, 3), apply_shared=True)
    b = engine.Engine(fp, image_range=(3, fp.n_images), apply_shared=False)
    bufs = []
    for e_ in (a, b):
        e_.set_parameters(fp.values)
        e_.accumulate(s2, lam)
        ptr, cnt = e_.reduce_buffer()
        bufs.append(torch.as_tensor(DeviceArray(ptr, cnt), device="cuda:0"))
    e0 = a.reduced_order()
    assert bufs[0].numel() == e0 * (e0 + 1) // 2 + 2 * e0
    total = bufs[0] + bufs[1]
    bufs[0].copy_(total); bufs[1].copy_(total)
    torch.cuda.synchronize()
    dx = np.zeros(fp.n_unknowns)
    for e_ in (a, b):
        e_.finalize(s2, lam)
        part = e_.solve(False)
        np.testing.assert_allclose(part[:e0], dxo[:e0], rtol=0, atol=1e-9 * np.abs(dxo).max())
        dx[:e0] = part[:e0]
        dx[e0:] += part[e0:]                       # every engine returns its own images' EO step
    np.testing.assert_allclose(dx, dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    a.close(); b.close()
