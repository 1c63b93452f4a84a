"""GPU parity on the irregular inputs the reference's object model allows: fixed parameters (column = Integer.MAX_VALUE,
UnknownParameter.java:27 -- no column, but the VALUE still enters the model), several cameras with their own distortion
sets (Camera.java:45-83), jointly dispersed and ordinary images in one adjustment, ragged image sizes."""
import dataclasses

import numpy as np
import pytest

from bundle_adjustment_amd import engine, numbering, scene
from bundle_adjustment_amd.problem import packed_to_full

pytestmark = pytest.mark.gpu


def renumber(fp, *, point_fixed=None, io_fixed=None, dist_fixed=None, eo_fixed=None, **changes):
    """Same scene, new index contract (BundleAdjustment.java:667-782) after fixing parameters / changing cameras."""
    image_camera = changes.get("image_camera", fp.image_camera)
    cam_dist_begin = changes.get("cam_dist_begin", fp.cam_dist_begin)
    n_cameras = len(cam_dist_begin) - 1
    num = numbering.number_unknowns(fp.n_points, n_cameras, image_camera, fp.ip_point, cam_dist_begin,
                                    point_fixed=point_fixed, io_fixed=io_fixed, dist_fixed=dist_fixed, eo_fixed=eo_fixed,
                                    sb_point_a=fp.sb_point_a, sb_point_b=fp.sb_point_b, dg_slot=changes.get("dg_slot", fp.dg_slot))
    return dataclasses.replace(fp, n_unknowns=num["n_unknowns"], rank_defect=num["rank_defect"], datum_flags=num["datum_flags"],
                               point_col=num["point_col"], io_col=num["io_col"], dist_col=num["dist_col"], eo_col=num["eo_col"],
                               n_observations=0, **changes).validate()


def check_against_oracle(oracle_mod, fp, invert=True, **engine_options):
    o = oracle_mod.Oracle(fp)
    s2 = fp.sigma2apriori
    U, d = fp.n_unknowns, fp.rank_defect
    No, no, _ = o.build(fp.values, s2, 0.0)
    dxo, Qo, _, _ = o.step(fp.values, s2, 0.0, invert)
    eng = engine.Engine(fp, **engine_options)
    eng.set_parameters(fp.values)
    eng.prepare_inverse(engine.INVERT_FULL)
    eng.build(s2, 0.0)
    N, n = eng.get_normal()
    Nf, Nof = packed_to_full(N, U), packed_to_full(No, U)
    dg = np.sqrt(np.abs(np.diag(Nof))); dg[dg == 0] = 1.0
    assert (np.abs(Nf - Nof) / np.outer(dg, dg)).max() < 1e-11
    np.testing.assert_allclose(n, no, rtol=0, atol=1e-11 * np.abs(no).max())
    dx = eng.solve(engine.INVERT_FULL if invert else engine.INVERT_NONE)
    np.testing.assert_allclose(dx[d:], dxo[d:], rtol=0, atol=1e-9 * np.abs(dxo[d:]).max())
    if invert:
        q, qo = np.diag(packed_to_full(eng.get_cofactor(), U))[d:], np.diag(packed_to_full(Qo, U))[d:]
        qerr = float(np.abs(q / qo - 1.0).max())
        if qerr >= 1e-9:
            # Two fp64 assemblies of the same N differ in the last bits of its entries (the sums run in different orders), and the
            # variances of a system with cond ~ 1e7 move by 1e-9 ... 1e-8 for that (measured here: 1.2e-9, 2.8e-9, 8.0e-9).  So: the
            # device's inverse must be the exact inverse of ITS N to 1e-9 (Newton steps in extended precision give that inverse), and
            # its distance from the oracle's must be what the two N's exact inverses differ by -- not more.
            def newton_ld(K, X0):
                Kl, X = K.astype(np.longdouble), X0.astype(np.longdouble)
                I2 = 2 * np.eye(K.shape[0], dtype=np.longdouble)
                for _ in range(4):
                    Xn = X @ (I2 - Kl @ X)
                    last = float(np.abs(np.diag(Xn)[d:] / np.diag(X)[d:] - 1).max())
                    X = Xn
                assert last < 1e-12, last             # converged: the exact inverse of K to cond * 2^-64, a thousand times below what is tested
                return X
            Xo, Xd = packed_to_full(Qo, U), packed_to_full(eng.get_cofactor(), U)
            To, Td = newton_ld(Nof, Xo), newton_ld(Nf, Xd)
            dev_inversion = float(np.abs(np.diag(Xd)[d:] / np.diag(Td)[d:] - 1).max())
            oracle_inversion = float(np.abs(np.diag(Xo)[d:] / np.diag(To)[d:] - 1).max())
            assembly = float(np.abs(np.diag(Td)[d:] / np.diag(To)[d:] - 1).max())
            assert dev_inversion < 1e-9, (dev_inversion, oracle_inversion)
            assert qerr <= 1.2 * assembly + dev_inversion + oracle_inversion, (qerr, assembly, dev_inversion, oracle_inversion)
    # the default route of the engine (EO pre-elimination where the problem allows it) gives the same step
    eng.prepare_inverse(engine.INVERT_NONE)
    eng.build(s2, 0.0)
    dx2 = eng.solve(False)
    np.testing.assert_allclose(dx2[d:], dxo[d:], rtol=0, atol=1e-9 * np.abs(dxo[d:]).max())
    reduced = eng.reduced_order() < U
    eng.close()
    return reduced


@pytest.mark.parametrize("weights", ["diag", "block"])
def test_fixed_parameters(oracle_mod, weights):
    fp = scene.make_scene(7, 50, 30, dist=scene.DIST_FULL, weights=weights, n_control=5, control_dense=True)
    P, I, nd = fp.n_points, fp.n_images, fp.dist_kind.size
    pf = np.zeros((P, 3), bool); pf[3, 2] = True; pf[7, :] = True
    iof = np.zeros((1, 3), bool); iof[0, 0] = True
    df = np.zeros(nd, bool); df[1] = True; df[nd - 1] = True
    ef = np.zeros((I, 6), bool); ef[2, 5] = True; ef[0, :] = True
    # the fixed point must not be a control point (its directly observed rows would have no column)
    ctrl = set(int(s) // 3 for s in fp.dg_slot)
    assert 7 not in ctrl and 3 not in ctrl
    fx = renumber(fp, point_fixed=pf, io_fixed=iof, dist_fixed=df, eo_fixed=ef)
    assert fx.n_unknowns == fp.n_unknowns - (1 + 3 + 1 + 2 + 1 + 6)
    reduced = check_against_oracle(oracle_mod, fx)
    assert not reduced            # EO columns are no longer 6 per image: the full-order route is taken


def test_two_cameras_with_different_distortion_sets(oracle_mod):
    base = scene.make_scene(8, 60, 36, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    P, I = base.n_points, base.n_images
    nd = base.dist_kind.size
    keep = (base.dist_kind <= 1) | (base.dist_kind == 5)   # camera 1: affinity/shear (Cx, Cy) + radial A1-A3 only
    nd1 = int(keep.sum())
    s_io = 3 * P
    v = base.values
    values = np.concatenate([v[:s_io], v[s_io:s_io + 3], v[s_io:s_io + 3], v[s_io + 3:s_io + 3 + nd],
                             v[s_io + 3:s_io + 3 + nd][keep], v[s_io + 3 + nd:]])
    image_camera = np.array([0] * (I // 2) + [1] * (I - I // 2), np.int32)
    two = renumber(base, image_camera=image_camera, cam_dist_begin=np.array([0, nd, nd + nd1], np.int32),
                   cam_r0=np.array([base.cam_r0[0], base.cam_r0[0]]),
                   dist_kind=np.concatenate([base.dist_kind, base.dist_kind[keep]]).astype(np.int32),
                   dist_order=np.concatenate([base.dist_order, base.dist_order[keep]]).astype(np.int32),
                   values=values, truth=None)
    assert two.n_unknowns == base.n_unknowns + 3 + nd1
    reduced = check_against_oracle(oracle_mod, two)
    assert reduced                 # every image still has a dense group and six trailing EO columns


def test_dense_and_ordinary_images_in_one_adjustment(oracle_mod):
    base = scene.make_scene(8, 60, 36, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
    k = 5                                           # images 0..4 keep their joint dispersion, 5..7 fall back to sigma_x, sigma_y
    mixed = dataclasses.replace(base, blk_ip_begin=base.blk_ip_begin[:k + 1].copy(), blk_disp_offset=base.blk_disp_offset[:k].copy(),
                                blk_disp=base.blk_disp[:int(base.blk_disp_offset[k])].copy(), n_observations=0).validate()
    reduced = check_against_oracle(oracle_mod, mixed)
    assert reduced                 # round 4: the ordinary images are pre-eliminated too (block-diagonal weights, test_gpu_ordinary_elimination.py)


def test_ragged_images(oracle_mod):
    """Images with very different numbers of points (3 ... 40) in jointly dispersed groups."""
    base = scene.make_scene(9, 45, 45, dist=scene.DIST_RADIAL, weights="block", n_control=5, control_dense=True, min_rays=2)
    counts = np.diff(base.blk_ip_begin)
    want = np.minimum(counts, np.array([3, 45, 5, 17, 45, 9, 33, 4, 45])[:counts.size])
    keep = np.concatenate([np.arange(base.blk_ip_begin[i], base.blk_ip_begin[i] + want[i]) for i in range(counts.size)])
    disp, off = [], [0]
    for i in range(counts.size):
        m0, m1 = 2 * int(counts[i]), 2 * int(want[i])
        D = base.blk_disp[base.blk_disp_offset[i]:base.blk_disp_offset[i] + m0 * m0].reshape(m0, m0)[:m1, :m1]
        disp.append(D.ravel()); off.append(off[-1] + m1 * m1)
    rays = np.bincount(base.ip_point[keep], minlength=base.n_points)
    assert rays.min() >= 2, "scene too thin for this test: every point needs two rays"
    rag = dataclasses.replace(base, ip_image=base.ip_image[keep], ip_point=base.ip_point[keep], ip_x=base.ip_x[keep], ip_y=base.ip_y[keep],
                              ip_var_x=base.ip_var_x[keep], ip_var_y=base.ip_var_y[keep], ip_rho=base.ip_rho[keep],
                              blk_ip_begin=np.concatenate([[0], np.cumsum(want)]).astype(np.int32),
                              blk_disp_offset=np.array(off[:-1], np.int64), blk_disp=np.concatenate(disp), n_observations=0)
    rag = renumber(rag)
    reduced = check_against_oracle(oracle_mod, rag)
    assert reduced


def test_incomplete_distortion_model_is_rejected():
    """A lone Bx (TangentialDistortionModel always owns Bx and By, TDF:39-134) must not silently drop out of the model."""
    base = scene.make_scene(6, 40, 24, dist=scene.DIST_FULL, weights="diag", n_control=4)
    nd = base.dist_kind.size
    keep = np.arange(nd) < 3                       # Cx, Cy, Bx
    s = 3 * base.n_points + 3
    bad = renumber(base, cam_dist_begin=np.array([0, 3], np.int32), dist_kind=base.dist_kind[keep].copy(),
                   dist_order=base.dist_order[keep].copy(), truth=None,
                   values=np.concatenate([base.values[:s], base.values[s:s + nd][keep], base.values[s + nd:]]))
    with pytest.raises(engine.EngineError) as ei:
        engine.Engine(bad)
    assert ei.value.code == -1


def test_an_engine_holds_one_cu_masked_hardware_queue_and_the_pool_keeps_two():
    """A CU-masked stream is a hardware queue of its own, idle or not; with ~20 of them in a process the scheduler time-slices ALL queues and a
    factorisation of 94 block columns takes 29 ms instead of 19 (32 of them: 58 ms; scripts/queue_count_probe.py) -- and the bounded waits of
    the dataflow factorisation start to run out (round 5: the GPU suite's "rare stalls", DESIGN.md section 4 "Hardware queues").  Until then
    every solver held two such streams, every engine has two solvers, and the pool kept them all for the life of the process.  Now: the
    dataflow solver holds ONE (the chain workgroups'), the EO-reduced solver shares the full-order solver's, and the pool keeps at most two
    idle ones of a kind."""
    import ctypes as C
    lib = engine.load_library()

    def census():
        a = (C.c_int * 8)()
        lib.jaicov_debug_stream_census(a)
        return list(a)

    fp = scene.config("cfg3")              # 24 block columns after the elimination: the dataflow factorisation, full-order and reduced solver
    before = census()
    engs = [engine.Engine(fp) for _ in range(5)]
    for e in engs:
        e.set_parameters(fp.values)
        e.build(fp.sigma2apriori, 0.0)
        e.solve(False)
    alive = census()
    held = [alive[k] - alive[4 + k] for k in range(4)]                 # streams in the hands of the five engines, per kind
    assert held[2] == 0 and held[3] == 5, (before, alive)              # no masked stream for trailing updates, one for the chain workgroups each
    for e in engs:
        e.close()
    after = census()
    assert after[2] <= 2 and after[3] <= 2 and after[6] == after[2] and after[7] == after[3], after    # all idle, and no more than two of a kind left
