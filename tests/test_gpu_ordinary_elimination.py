"""EO pre-elimination of images whose points are ORDINARY ImageCoordinate groups (diagonal / 2 x 2 weights), and the batched
inversion of the dense dispersions at engine creation (round 4).

reduceNormalEquationSystem (BundleAdjustment.java:1197-1342) eliminates the exterior orientation of every image; all three reference
examples run MatrixInversion.REDUCED (ExampleReport.java:89).  Through round 3 the device path of that elimination existed only for
images with a joint dispersion; an ordinary image is now served as an image block with a block-diagonal weight (engine option
ordinary_group_elimination, default on).  Everything is held to the oracle's full bordered solve (dspsv + dsptri at order U).
"""
import numpy as np
import pytest

from bundle_adjustment_amd import engine, scene
from bundle_adjustment_amd.problem import packed_to_full

pytestmark = pytest.mark.gpu

# Round 5: by default the engine eliminates ordinary images only when that saves two block columns of the factorisation (not on the small
# scenes used here: config 2 runs 0.79 ms per pass without against 0.86 with); ordinary_group_elimination = 1 forces the path at any size.
FORCE = 1


SCENES = {
    "tiny": lambda: scene.config("tiny"),                # 2 x 2 weights, control points
    "tiny_free": lambda: scene.config("tiny_free"),      # diagonal weights, free network (d = 6), scale bar
    "cfg2": lambda: scene.config("cfg2"),                # BASELINE config 2: 20 x 200, diagonal weights
}


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_ordinary_images_are_pre_eliminated(oracle_mod, name, lam):
    fp = SCENES[name]()
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    assert fp.n_image_blocks == 0
    o = oracle_mod.Oracle(fp)
    dxo, _, No, no = o.step(fp.values, s2, lam, False)
    eng = engine.Engine(fp, ordinary_group_elimination=FORCE)
    eng.set_parameters(fp.values)
    eng.build(s2, lam)
    assert eng.reduced_order() == U - 6 * fp.n_images                     # the EO columns are gone from the system
    dx = eng.solve(False)
    np.testing.assert_allclose(dx, dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    om_o = o.omega(fp.values, s2, dxo)
    assert abs(eng.omega(s2, dx) - om_o) <= 1e-9 * om_o
    # the unreduced system, as the reference stacks it (PDF:475-505), through the same block kernels
    eng.prepare_inverse(engine.INVERT_FULL)
    eng.build(s2, lam)
    assert eng.reduced_order() == U
    N, n = eng.get_normal()
    Nref, nref, _ = o.build(fp.values, s2, lam)
    d = fp.rank_defect
    Nf, Rf = packed_to_full(N, U), packed_to_full(Nref, U)
    np.testing.assert_allclose(Nf[d:, d:], Rf[d:, d:], rtol=0, atol=1e-11 * np.abs(Rf).max())
    np.testing.assert_allclose(n, nref, rtol=0, atol=1e-11 * np.abs(nref).max())
    eng.close()
    # arrival-order sums (deterministic off): the other instances of the gather with compact block-diagonal weights, both systems
    arr = engine.Engine(fp, deterministic=False, ordinary_group_elimination=FORCE)
    arr.set_parameters(fp.values)
    arr.build(s2, lam)
    assert arr.reduced_order() == U - 6 * fp.n_images
    np.testing.assert_allclose(arr.solve(False), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    arr.prepare_inverse(engine.INVERT_FULL)
    arr.build(s2, lam)
    Na, na = arr.get_normal()
    np.testing.assert_allclose(packed_to_full(Na, U)[d:, d:], Rf[d:, d:], rtol=0, atol=1e-11 * np.abs(Rf).max())
    arr.close()
    # the old path (ordinary groups one by one into the full-order system) stays available and agrees
    old = engine.Engine(fp, ordinary_group_elimination=-1)
    old.set_parameters(fp.values)
    old.build(s2, lam)
    assert old.reduced_order() == U
    np.testing.assert_allclose(old.solve(False), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    old.close()


@pytest.mark.parametrize("name", ["tiny", "tiny_free"])
def test_reduced_inverse_of_ordinary_images_is_the_block_of_the_full_cofactor(oracle_mod, name):
    """MatrixInversion.REDUCED (BA:261-267) and FULL (expanded from the reduced inverse) on ordinary image groups against the
    oracle's dspsv + dsptri of the full bordered system."""
    fp = SCENES[name]()
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    o = oracle_mod.Oracle(fp)
    dxo, Qo, _, _ = o.step(fp.values, s2, 0.0, True)
    Qo = packed_to_full(Qo, U)
    eng = engine.Engine(fp, ordinary_group_elimination=FORCE)
    eng.set_parameters(fp.values)
    for inv in (engine.INVERT_REDUCED, engine.INVERT_FULL_EXPANDED):
        eng.prepare_inverse(inv)
        eng.build(s2, 0.0)
        dx = eng.solve(inv)
        np.testing.assert_allclose(dx, dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
        k = eng.cofactor_order()
        assert k == (U - 6 * fp.n_images if inv == engine.INVERT_REDUCED else U)
        Q = packed_to_full(eng.get_cofactor(), k)
        np.testing.assert_allclose(Q, Qo[:k, :k], rtol=0, atol=1e-9 * np.abs(Qo).max())
    eng.close()


def test_estimate_on_ordinary_images_matches_oracle(oracle_mod):
    """The whole loop (jaicov_neq_estimate, MatrixInversion.REDUCED) on BASELINE config 2 with the elimination on and off."""
    fp = scene.config("cfg2")
    vo, Qo, ro = oracle_mod.Oracle(fp).estimate()
    for flag in (FORCE, -1):
        eng = engine.Engine(fp, ordinary_group_elimination=flag)
        v, r = eng.estimate(invert=engine.INVERT_REDUCED)
        assert r.state == 1 and r.iterations == ro.iterations
        assert np.abs(v - vo).max() <= 1e-9 * 2000.0
        assert abs(r.omega - ro.omega) <= 1e-9 * ro.omega
        k, kr = eng.cofactor_order(), fp.n_unknowns - 6 * fp.n_images
        assert k == (kr if flag == FORCE else fp.n_unknowns)                 # without the elimination REDUCED is served by the full inverse (jaicov_neq.h)
        assert eng.reduced_order() == k
        Q = packed_to_full(eng.get_cofactor(), k)[:kr, :kr]
        Qr = packed_to_full(Qo, fp.n_unknowns)[:kr, :kr]
        assert np.abs(Q - Qr).max() <= 2e-9 * np.abs(Qr).max()
        eng.close()


def test_problems_that_do_not_qualify_keep_the_full_order_path(oracle_mod):
    """A disqualifier -- here: a joint dispersion that covers only PART of its image's observations, so that the image's exterior
    orientation is shared by two observation groups -- leaves ALL ordinary groups outside the elimination: the decision is
    all-or-nothing over the whole problem (every rank of a sharded run must assemble a system of the same order)."""
    import dataclasses
    fp = scene.config("tiny_block")
    m0 = 2 * int(fp.blk_ip_begin[1] - fp.blk_ip_begin[0])
    D0 = fp.blk_disp[:m0 * m0].reshape(m0, m0)[4:, 4:]                   # the first two points of image 0 leave the block ...
    begin = fp.blk_ip_begin.copy(); begin[0] += 2
    off = fp.blk_disp_offset.copy(); off[1:] -= m0 * m0 - D0.size
    disp = np.concatenate([D0.ravel(), fp.blk_disp[m0 * m0:]])
    var_x, var_y = fp.ip_var_x.copy(), fp.ip_var_y.copy()                 # ... and are ordinary groups with their own variances
    fp2 = dataclasses.replace(fp, blk_ip_begin=begin, blk_disp_offset=off, blk_disp=disp, ip_var_x=var_x, ip_var_y=var_y,
                              n_observations=0).validate()
    o = oracle_mod.Oracle(fp2)
    s2 = fp2.sigma2apriori
    dxo, _, _, _ = o.step(fp2.values, s2, 0.0, False)
    eng = engine.Engine(fp2, ordinary_group_elimination=FORCE)
    eng.set_parameters(fp2.values)
    eng.build(s2, 0.0)
    assert eng.reduced_order() == fp2.n_unknowns
    np.testing.assert_allclose(eng.solve(False), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    eng.close()


@pytest.mark.parametrize("m_points", [8, 64, 70, 200, 330])
def test_batched_dispersion_inverse_matches_the_references_dpptri(oracle_mod, m_points):
    """DOPG:82-86 at engine creation: all dispersions of one padded order are inverted in one set of batched launches (batchinv.hip).
    jaicov_neq_get_block_weight returns inv(D) in the CALLER's observation order (the engine keeps blocks column-sorted);
    oracle_dispersion_to_weight = dpptrf + dpptri of D / sigma0^2.  Orders 4 .. 570: one to five diagonal blocks (ragged last level of the triangular inverse), two padded orders in one problem."""
    fp = scene.make_scene(6, int(m_points / 0.55) + 12, m_points, dist=scene.DIST_RADIAL, weights="block", n_control=4, control_dense=True)
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    eng = engine.Engine(fp, ordinary_group_elimination=FORCE)
    for b in range(fp.n_image_blocks):
        W = eng.get_block_weight(b) * s2
        ref = o.block_weight(s2, b)
        assert np.abs(W - ref).max() <= 1e-9 * np.abs(ref).max()
        assert np.abs(W - W.T).max() == 0.0
    ct = eng.create_timings()
    assert ct["create_ms"] > 0 and ct["dispersions_to_weights_ms"] > 0
    eng.close()


def test_not_positive_definite_dispersion_is_reported():
    fp = scene.make_scene(4, 40, 20, dist=scene.DIST_RADIAL, weights="block", n_control=4)
    m = 2 * int(fp.blk_ip_begin[2] - fp.blk_ip_begin[1])
    off = int(fp.blk_disp_offset[1])
    D = fp.blk_disp[off:off + m * m].reshape(m, m)
    D[3, 3] = -D[3, 3]
    with pytest.raises(engine.EngineError) as ei:
        engine.Engine(fp, ordinary_group_elimination=FORCE)
    assert ei.value.code == 1                                             # JAICOV_ERR_SINGULAR: MatrixNotSPDException (DOPG:85-86)


def test_ordinary_images_sharded(oracle_mod):
    """Two engines over disjoint image ranges of a scene WITHOUT joint dispersions: both take the elimination (the decision is made on
    the whole problem), their reduced systems add up to the single engine's, each back-substitutes its own images' EO step."""
    fp = scene.config("tiny")
    s2 = fp.sigma2apriori
    dxo, _, _, _ = oracle_mod.Oracle(fp).step(fp.values, s2, 0.0, False)
    full = engine.Engine(fp, ordinary_group_elimination=FORCE); full.set_parameters(fp.values); full.build(s2); N, n = full.get_normal()
    e0 = full.reduced_order()
    assert e0 == fp.n_unknowns - 6 * fp.n_images
    a = engine.Engine(fp, image_range=(0, 2), apply_shared=True, ordinary_group_elimination=FORCE)
    b = engine.Engine(fp, image_range=(2, fp.n_images), apply_shared=False, ordinary_group_elimination=FORCE)
    for e_ in (a, b):
        e_.set_parameters(fp.values); e_.accumulate(s2)
        assert e_.reduced_order() == e0
    Na, na = a.get_normal(); Nb, nb = b.get_normal()
    np.testing.assert_allclose(Na + Nb, N, rtol=1e-11, atol=1e-12 * np.abs(N).max())
    np.testing.assert_allclose(na + nb, n, rtol=1e-11, atol=1e-12 * np.abs(n).max())
    np.testing.assert_allclose(full.solve(False), dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    for e_ in (full, a, b):
        e_.close()


def test_simulation_on_ordinary_images_leaves_parameters_and_gives_the_oracles_cofactors(oracle_mod):
    """EstimationType.SIMULATION (BA:228-236, 830-831: no right-hand side, no update, cofactors only) through the elimination of
    ordinary image groups, REDUCED and FULL."""
    fp = scene.config("tiny")
    o = oracle_mod.Oracle(fp)
    vo, Qo, ro = o.estimate(simulation=True)
    U = fp.n_unknowns
    Qo = packed_to_full(Qo, U)
    for inv in (engine.INVERT_REDUCED, engine.INVERT_FULL):
        eng = engine.Engine(fp, ordinary_group_elimination=FORCE)
        v, r = eng.estimate(invert=inv, simulation=True)
        assert r.state == 1
        np.testing.assert_array_equal(v, fp.values)
        k = eng.cofactor_order()
        assert k == (U - 6 * fp.n_images if inv == engine.INVERT_REDUCED else U)
        Q = packed_to_full(eng.get_cofactor(), k)
        np.testing.assert_allclose(Q, Qo[:k, :k], rtol=0, atol=1e-9 * np.abs(Qo).max())
        eng.close()


def test_two_cameras_with_ordinary_images(oracle_mod):
    """The elimination of ordinary (2 x 2 weighted) image groups with two cameras of different distortion sets: N, n, the step and the
    variances against the oracle (test_gpu_edge_cases.check_against_oracle)."""
    from bundle_adjustment_amd.problem import DIST_RADIAL_AI
    import test_gpu_edge_cases as ec
    base = scene.make_scene(8, 60, 36, dist=scene.DIST_FULL, weights="2x2", n_control=5)
    P, I = base.n_points, base.n_images
    nd = base.dist_kind.size
    keep = (base.dist_kind <= 1) | (base.dist_kind == DIST_RADIAL_AI)
    nd1 = int(keep.sum())
    s_io = 3 * P
    v = base.values
    values = np.concatenate([v[:s_io], v[s_io:s_io + 3], v[s_io:s_io + 3], v[s_io + 3:s_io + 3 + nd],
                             v[s_io + 3:s_io + 3 + nd][keep], v[s_io + 3 + nd:]])
    image_camera = np.array([0] * (I // 2) + [1] * (I - I // 2), np.int32)
    two = ec.renumber(base, image_camera=image_camera, cam_dist_begin=np.array([0, nd, nd + nd1], np.int32),
                      cam_r0=np.array([base.cam_r0[0], base.cam_r0[0]]),
                      dist_kind=np.concatenate([base.dist_kind, base.dist_kind[keep]]).astype(np.int32),
                      dist_order=np.concatenate([base.dist_order, base.dist_order[keep]]).astype(np.int32),
                      values=values, truth=None)
    assert two.n_image_blocks == 0
    reduced = ec.check_against_oracle(oracle_mod, two, ordinary_group_elimination=FORCE)
    assert reduced


def test_a_large_ordinary_problem_is_eliminated_with_compact_weights():
    """The weights of an ordinary image are kept as 2 x 2 blocks (DevProblem::ip_w3), not as an m x m matrix with zeros: 140 images of 2 000
    points each would need 18 GB in the dense form (which is why such a problem used to fall back to the full-order system) and need
    6 MB now.  Held against the engine's own full-order path (no oracle at this size): same step, same Omega, and the bits of two builds agree."""
    fp = scene.make_scene(140, 4000, 2000, dist=scene.DIST_RADIAL, weights="2x2", n_control=6)
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    assert fp.n_image_blocks == 0 and 140 * (2 * 2000) ** 2 * 8 > 16 * 2 ** 30
    eng = engine.Engine(fp, ordinary_group_elimination=FORCE)
    eng.set_parameters(fp.values)
    eng.build(s2, 0.0)
    assert eng.reduced_order() == U - 6 * fp.n_images
    dx = eng.solve(False)
    om = eng.omega(s2, dx)
    eng.build(s2, 0.0)
    assert np.array_equal(eng.solve(False), dx)                            # deterministic assembly: the same bits
    eng.close()
    full = engine.Engine(fp, ordinary_group_elimination=-1)                # the full-order system, small-group assembly
    full.set_parameters(fp.values)
    full.build(s2, 0.0)
    assert full.reduced_order() == U
    dxf = full.solve(False)
    assert np.abs(dx - dxf).max() <= 1e-9 * np.abs(dxf).max()
    assert abs(om - full.omega(s2, dxf)) <= 1e-9 * om
    full.close()


@pytest.mark.parametrize("per_image, eliminated", [(2048, True), (2049, False)])
def test_the_largest_ordinary_image_that_is_served_as_a_block(per_image, eliminated):
    """The limit of the elimination for ordinary images: 2 048 observations per image (4 096 rows: the elimination kernels' per-image panel).
    At the limit the exterior orientations are eliminated; one observation more and the WHOLE problem keeps the full-order path
    (all-or-nothing, engine.hip).  Either way the step is the full-order engine's."""
    import dataclasses
    fp = scene.make_scene(4, 2300, 2200, dist=scene.DIST_RADIAL, weights="2x2", n_control=6)
    assert np.bincount(fp.ip_image).min() > per_image
    seen = np.bincount(fp.ip_point)
    keep = []
    for i in range(4):                                             # exactly per_image observations per image: drop those of the most-seen points
        idx = np.flatnonzero(fp.ip_image == i)
        drop = idx[np.argsort(-seen[fp.ip_point[idx]], kind="stable")[:idx.size - per_image]]
        seen[fp.ip_point[drop]] -= 1
        keep.append(np.setdiff1d(idx, drop))
    keep = np.concatenate(keep)
    fp = dataclasses.replace(fp, n_observations=0, **{k: getattr(fp, k)[keep].copy() for k in
                                                      ("ip_image", "ip_point", "ip_x", "ip_y", "ip_var_x", "ip_var_y", "ip_rho")}).validate()
    U, s2 = fp.n_unknowns, fp.sigma2apriori
    assert np.bincount(fp.ip_image).tolist() == [per_image] * 4 and np.bincount(fp.ip_point).min() >= 2
    eng = engine.Engine(fp, ordinary_group_elimination=FORCE)
    eng.set_parameters(fp.values)
    eng.build(s2, 0.0)
    assert eng.reduced_order() == (U - 6 * fp.n_images if eliminated else U)
    dx = eng.solve(False)
    om = eng.omega(s2, dx)
    eng.close()
    full = engine.Engine(fp, ordinary_group_elimination=-1)
    full.set_parameters(fp.values)
    full.build(s2, 0.0)
    dxf = full.solve(False)
    assert np.abs(dx - dxf).max() <= 1e-9 * np.abs(dxf).max()
    assert abs(om - full.omega(s2, dxf)) <= 1e-9 * om
    full.close()


def test_default_eliminates_ordinary_images_only_where_it_saves_block_columns():
    """Round 5's size rule (engine.hip, create): the default serves ordinary images as blocks when the 6 I exterior-orientation columns are
    at least two 128-column blocks of the factorisation.  BASELINE config 2 (order 726 -> 606: six block columns -> five) stays at full
    order -- its pass is 0.79 ms that way against 0.86 --, config 3 (3 614 -> 3 014: 29 -> 24) is eliminated, as is a block the size of
    the bundled example (115 images: ten block columns -> four)."""
    for name, want in (("cfg2", False), ("cfg3", True)):
        fp = scene.config(name)
        eng = engine.Engine(fp)
        eng.set_parameters(fp.values)
        eng.build(fp.sigma2apriori, 0.0)
        U = fp.n_unknowns
        assert ((U + 127) // 128 - (U - 6 * fp.n_images + 127) // 128 >= 2) == want
        assert eng.reduced_order() == (U - 6 * fp.n_images if want else U), name
        eng.close()
