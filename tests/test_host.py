"""Host mirror of JAICOV's object API (C++ / pybind11): index contract (CPU) and estimateModel() on the GPU.

Config 1 of BASELINE.json: the bundled example block (115 images, 150 points, 9 972 image points, 1 scale bar)."""
import gzip
import os
import shutil

import numpy as np
import pytest

from bundle_adjustment_amd import numbering
from bundle_adjustment_amd.problem import packed_to_full

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example")


@pytest.fixture(scope="module")
def H():
    from bundle_adjustment_amd import host_api
    return host_api


@pytest.fixture()
def example_base(tmp_path):
    for f in ("ior", "eor", "obc", "scale"):
        shutil.copy(os.path.join(GOLDEN, f"example.{f}"), tmp_path)
    with gzip.open(os.path.join(GOLDEN, "example.phc.gz")) as src, open(tmp_path / "example.phc", "wb") as dst:
        dst.write(src.read())
    return str(tmp_path / "example")


@pytest.fixture()
def example_report(tmp_path):
    with gzip.open(os.path.join(GOLDEN, "example.htm.gz")) as src, open(tmp_path / "example.htm", "wb") as dst:
        dst.write(src.read())
    return str(tmp_path / "example.htm")


def example_adjustment(H, base, unit_weights=False):
    """ExampleReport.java:61-89 on the flat files: A3, Cx, Cy fixed (example.htm:83,86-87), datum = names <= 3 chars."""
    pr = H.read_aicon_flat(base)
    cam = pr.camera
    cam.getDistortionModel(H.DistortionModelType.RADIAL_DISTORTION).get(3).setColumn(H.COLUMN_FIXED)
    aff = cam.getDistortionModel(H.DistortionModelType.AFFINITY_AND_SHEAR)
    aff.getCx().setColumn(H.COLUMN_FIXED); aff.getCy().setColumn(H.COLUMN_FIXED)
    for p in pr.points():
        if len(p.getName()) > 3:
            p.setDatum(False)
    if unit_weights:   # config 1 "identity dispersion": all variances 1 -> sigma0^2 = 1, P = I
        for im in cam.images():
            for ic in im.coordinates():
                ic.getX().setVariance(1.0); ic.getY().setVariance(1.0)
        for s in pr.scaleBars():
            s.getLength().setVariance(1.0)
    ba = H.BundleAdjustment()
    ba.add(cam)
    for s in pr.scaleBars():
        ba.add(s)
    ba.setInvertNormalEquation(H.MatrixInversion.FULL)
    return pr, ba


def test_example_index_known_answers(H, example_base):
    """SURVEY.md Appendix B / example.htm:33-35,42: n = 19 945, u = 1 147, d = 6, dof = 18 804 and the column layout."""
    pr, ba = example_adjustment(H, example_base)
    ba.prepareUnknownParameters()
    ba.flatten()
    cam = pr.camera
    assert cam.getNumberOfImages() == 115 and len(pr.points()) == 150 and len(pr.scaleBars()) == 1
    assert ba.getNumberOfObservations() == 19945
    assert ba.getNumberOfUnknownParameters() == 1147
    assert ba.getNumberOfDatumConditions() == 6
    assert ba.getDegreeOfFreedom() == 18804
    assert ba.getDatumFlags() == 1 + 2 + 4 + 8 + 16 + 32          # tx,ty,tz,rx,ry,rz free; scale fixed by the scale bar
    io = cam.getInteriorOrientation()
    assert [io.getPrinciplePointX().getColumn(), io.getPrinciplePointY().getColumn(), io.getPrincipleDistance().getColumn()] == [456, 457, 458]
    tan = cam.getDistortionModel(H.DistortionModelType.TANGENTIAL_DISTORTION)
    rad = cam.getDistortionModel(H.DistortionModelType.RADIAL_DISTORTION)
    assert [tan.getBx().getColumn(), tan.getBy().getColumn()] == [459, 460]
    assert [rad.get(1).getColumn(), rad.get(2).getColumn(), rad.get(3).getColumn()] == [461, 462, H.COLUMN_FIXED]
    pts = ba.getObjectCoordinates()
    assert [p.getName() for p in pts[:6]] == ["6", "14", "15", "17", "18", "25"]
    assert [pts[0].getX().getColumn(), pts[0].getY().getColumn(), pts[0].getZ().getColumn()] == [6, 7, 8]
    assert pts[-1].getZ().getColumn() == 455
    imgs = cam.images()
    assert imgs[0].getExteriorOrientation().get(H.ParameterType.CAMERA_COORDINATE_X).getColumn() == 463
    assert imgs[-1].getExteriorOrientation().get(H.ParameterType.CAMERA_KAPPA).getColumn() == 1152
    assert sum(p.isDatum() for p in pts) == 66
    d = ba.flat()
    # rows: image points first (x then y, image-major), scale bar last
    assert imgs[0].coordinates()[0].getX().getRow() == 0 and pr.scaleBars()[0].getLength().getRow() == 19944
    assert abs(d["sigma2apriori"] - min(d["ip_var_x"].min(), d["ip_var_y"].min(), d["sb_var"].min(), 1.0)) == 0


def test_cpp_numbering_matches_python_numbering(H, example_base):
    pr, ba = example_adjustment(H, example_base)
    ba.prepareUnknownParameters(); ba.flatten()
    d = ba.flat()
    P = d["point_col"].size // 3
    num = numbering.number_unknowns(P, 1, d["image_camera"], d["ip_point"], d["cam_dist_begin"],
                                    dist_fixed=d["dist_col"] < 0, sb_point_a=d["sb_point_a"], sb_point_b=d["sb_point_b"])
    np.testing.assert_array_equal(num["point_col"].ravel(), d["point_col"])
    np.testing.assert_array_equal(num["io_col"].ravel(), d["io_col"])
    np.testing.assert_array_equal(num["dist_col"], d["dist_col"])
    np.testing.assert_array_equal(num["eo_col"].ravel(), d["eo_col"])
    assert num["n_unknowns"] == d["n_unknowns"] and num["datum_flags"] == d["datum_flags"]


@pytest.mark.parametrize("seed", range(12))
def test_rank_defect_literal_vs_closed_form(H, seed):
    """BundleAdjustment.detectRankDefect: the loop-by-loop C++ mirror vs the closed form used by the flat generators."""
    rng = np.random.default_rng(seed)
    T = H.DistortionModelType
    cam = H.Camera(1, 10.0, [T.RADIAL_DISTORTION])
    pts = [H.ObjectCoordinate(str(i), *rng.normal(0, 100, 3)) for i in range(8)]
    imgs = [cam.add(i) for i in range(3)]
    for im in imgs:
        for p in pts:
            im.add(p, 0.1, 0.2, 0.001, 0.001)
    # random fixed coordinates / angles
    point_fixed = np.zeros((8, 3), bool); eo_fixed = np.zeros((3, 6), bool)
    PT = H.ParameterType
    eo_types = [PT.CAMERA_COORDINATE_X, PT.CAMERA_COORDINATE_Y, PT.CAMERA_COORDINATE_Z, PT.CAMERA_OMEGA, PT.CAMERA_PHI, PT.CAMERA_KAPPA]
    for _ in range(rng.integers(0, 4)):
        i, a = rng.integers(0, 8), rng.integers(0, 3)
        [pts[i].getX(), pts[i].getY(), pts[i].getZ()][a].setColumn(H.COLUMN_FIXED); point_fixed[i, a] = True
    for _ in range(rng.integers(0, 3)):
        i, a = rng.integers(0, 3), rng.integers(0, 6)
        imgs[i].getExteriorOrientation().get(eo_types[a]).setColumn(H.COLUMN_FIXED); eo_fixed[i, a] = True
    ba = H.BundleAdjustment(); ba.add(cam)
    sb = []
    if rng.random() < 0.5:
        sb = [H.ScaleBar(pts[0], pts[1], 10.0, 0.01)]; ba.add(sb[0])
    obs, kinds = [], []
    for _ in range(rng.integers(0, 5)):
        i, a = int(rng.integers(0, 8)), int(rng.integers(0, 3))
        up = [pts[i].getX(), pts[i].getY(), pts[i].getZ()][a]
        if up.getColumn() == H.COLUMN_FIXED or any(o[1] is up for o in obs):
            continue
        o = H.ObservationParameter(up); o.setVariance(1e-4); obs.append((o, up)); kinds.append("XYZ"[a])
    grp = None
    if obs:
        grp = H.DirectlyObservedParameterGroup([o for o, _ in obs]); ba.add(grp)
    ba.prepareUnknownParameters()
    n_fixed = point_fixed.sum(0) + eo_fixed[:, :3].sum(0)
    flags = numbering.detect_rank_defect(bool(sb), kinds, n_fixed, eo_fixed[:, 3:].any(0))
    assert ba.getDatumFlags() == flags


def test_example_oracle_known_answers(H, example_base, oracle_mod):
    """Pins the oracle on the reference's only third-party known answers (SURVEY.md 8c ii): with the protocol's uniform
    a-priori sigma 0.0005 mm (example.htm:31) the adjustment started from AICON's adjusted values is a near fixed point
    (converges in <= 4 passes, coordinates move by micrometres) and the a-posteriori S0 reproduces AICON's 0.000405."""
    pr, ba = example_adjustment(H, example_base)
    for im in pr.camera.images():
        for ic in im.coordinates():
            ic.getX().setVariance(0.0005 ** 2); ic.getY().setVariance(0.0005 ** 2)
    ba.prepareUnknownParameters(); ba.flatten()
    from bundle_adjustment_amd.host_api import flat_problem
    fp = flat_problem(ba).validate()
    v, Q, res = oracle_mod.Oracle(fp).estimate(invert=False)
    assert res.state == 1 and res.iterations <= 4
    s0 = np.sqrt(res.omega / fp.degree_of_freedom)
    assert abs(s0 - 0.000405) < 1.5e-6, s0
    assert np.abs(v - fp.values)[:3 * fp.n_points].max() < 0.01     # mm


@pytest.mark.gpu
@pytest.mark.parametrize("unit_weights", [False, True])
def test_example_estimate_model_matches_oracle(H, example_base, oracle_mod, unit_weights):
    """BundleAdjustment.estimateModel() through the C++ host + C ABI on the MI355X vs the CPU oracle on the same flat
    problem: adjusted parameters 1e-9 relative, cofactor diagonal 1e-8, identical iteration count."""
    from bundle_adjustment_amd.host_api import flat_problem
    pr0, ba0 = example_adjustment(H, example_base, unit_weights)
    ba0.useCentroidedCoordinates(False)
    ba0.prepareUnknownParameters(); ba0.flatten()
    fp = flat_problem(ba0).validate()
    vo, Qo, ro = oracle_mod.Oracle(fp).estimate()
    pr, ba = example_adjustment(H, example_base, unit_weights)
    ba.useCentroidedCoordinates(False)
    events = []
    ba.addPropertyChangeListener(lambda n, a, b: events.append(n))
    state = ba.estimateModel()
    assert state == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
    assert ro.state == 1 and ba.getIterations() == ro.iterations
    assert "INVERT_NORMAL_EQUATION_MATRIX" in events and events[-1] == "ERROR_FREE_ESTIMATION"
    pts = ba.getObjectCoordinates()
    got = np.array([[p.getX().getValue(), p.getY().getValue(), p.getZ().getValue()] for p in pts]).ravel()
    ref = vo[:got.size]
    assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)).max() < 1e-9
    assert abs(ba.getOmega() - ro.omega) < 1e-8 * ro.omega
    assert abs(ba.getVarianceFactorAposteriori() - ro.omega / fp.degree_of_freedom) < 1e-8 * ro.omega / fp.degree_of_freedom
    Q = ba.getCofactorMatrix()
    U, d = fp.n_unknowns, fp.rank_defect
    dq = np.diag(packed_to_full(Q, U))[d:]; dqo = np.diag(packed_to_full(Qo, U))[d:]
    np.testing.assert_allclose(dq, dqo, rtol=1e-7)


@pytest.mark.gpu
def test_example_centroided_equals_uncentroided(H, example_base):
    """useCentroidedCoordinates (BundleAdjustment.java:115-201) must not change the adjusted coordinates."""
    res = []
    for cen in (True, False):
        pr, ba = example_adjustment(H, example_base)
        ba.useCentroidedCoordinates(cen)
        assert ba.estimateModel() == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
        res.append(np.array([[p.getX().getValue(), p.getY().getValue(), p.getZ().getValue()] for p in ba.getObjectCoordinates()]))
    assert np.abs(res[0] - res[1]).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["REDUCED", "PRE_ELIMINATION"])
def test_example_reduced_inversion_modes(H, example_base, mode):
    """MatrixInversion.REDUCED / PRE_ELIMINATION (BundleAdjustment.java:261-267,283-291): same adjusted parameters as FULL
    and the same cofactor block for the datum border, points, interior orientation and distortion (the leading
    numRows of Qxx).  The bundled block has no jointly dispersed image groups, so the engine serves the mode with the
    full inverse here; the pre-eliminated route is covered by test_gpu_parity.py::test_reduced_inverse_*."""
    res = {}
    for m in ("FULL", mode):
        pr, ba = example_adjustment(H, example_base)
        ba.setInvertNormalEquation(getattr(H.MatrixInversion, m))
        assert ba.estimateModel() == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
        pts = np.array([[p.getX().getValue(), p.getY().getValue(), p.getZ().getValue()] for p in ba.getObjectCoordinates()])
        res[m] = (pts, np.array(ba.getCofactorMatrix()), ba.getIterations())
    assert res["FULL"][2] == res[mode][2]
    np.testing.assert_allclose(res[mode][0], res["FULL"][0], rtol=0, atol=1e-9)
    k = 6 + 3 * len(res["FULL"][0]) + 7          # d + 3P + (c, x0, y0, A1, A2, Bx, By): numRows of BA:262
    n = k * (k + 1) // 2
    ref = res["FULL"][1][:n]
    np.testing.assert_allclose(res[mode][1][:n], ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())   # two runs: atomics reorder sums


def distortion_model_adjustment(H, base):
    """ExampleDistortionModel.java:68-121 on the flat files: camera with the three Zernike models beside the .ior file's, c fixed at 28,
    every radial coefficient fixed at 0, gradient model with the single indices 4, 12, 24, 40, 60 (Z_2^0 .. Z_10^0), REDUCED."""
    T = H.DistortionModelType
    pr = H.read_aicon_flat(base, [T.ZERNIKE_GRADIENT, T.ZERNIKE_X, T.ZERNIKE_Y])
    cam = pr.camera
    c = cam.getInteriorOrientation().getPrincipleDistance()
    c.setValue(28.0); c.setColumn(H.COLUMN_FIXED)
    for u in cam.getDistortionModel(T.RADIAL_DISTORTION).parameters():
        u.setValue(0.0); u.setColumn(H.COLUMN_FIXED)
    z = cam.getDistortionModel(T.ZERNIKE_GRADIENT)
    order = 0
    for i in range(1, 6):
        order += 4 * i
        z.add(order)
    ba = H.BundleAdjustment()
    ba.add(cam)
    for s in pr.scaleBars():
        ba.add(s)
    ba.setInvertNormalEquation(H.MatrixInversion.REDUCED)
    return pr, ba


def test_third_example_index_contract_and_oracle(H, example_base, oracle_mod):
    """The reference's third example (ExampleDistortionModel.java): what prepareUnknownParameters makes of it -- 450 point + 2 interior
    (c fixed) + Cx, Cy, Bx, By (IORFileReader.java:95-206 sets what the .ior file carries free) + 5 Zernike gradient coefficients, A1..A3
    fixed at 0, + 690 exterior = 1 151 unknowns, all 150 points in the datum (rank defect 6: the scale comes from the bar) -- and the CPU
    oracle's run of it: converges, the Zernike set takes over what c = 28 (instead of 28.8) and the zeroed radial set leave, the
    a-posteriori sigma0 stays where the radial model put it (x 1.5)."""
    from bundle_adjustment_amd.host_api import flat_problem
    from bundle_adjustment_amd.problem import DIST_ZERNIKE_Z
    pr, ba = distortion_model_adjustment(H, example_base)
    ba.useCentroidedCoordinates(False)
    ba.prepareUnknownParameters(); ba.flatten()
    fp = flat_problem(ba).validate()
    zs = [(int(k), int(o), int(c)) for k, o, c in zip(fp.dist_kind, fp.dist_order, fp.dist_col) if k == DIST_ZERNIKE_Z]
    assert [o for _, o, _ in zs] == [4, 12, 24, 40, 60] and all(c >= 0 for _, _, c in zs)
    from bundle_adjustment_amd.problem import DIST_RADIAL_AI
    assert [int(c) for k, c in zip(fp.dist_kind, fp.dist_col) if k == DIST_RADIAL_AI] == [-1, -1, -1]      # A1, A2, A3 fixed (at 0)
    assert all(int(c) >= 0 for k, c in zip(fp.dist_kind, fp.dist_col) if k not in (DIST_RADIAL_AI, DIST_ZERNIKE_Z))   # Cx, Cy, Bx, By stay free
    assert int(fp.io_col[0][2]) == -1 and int(fp.io_col[0][0]) >= 0 and int(fp.io_col[0][1]) >= 0           # x0, y0 free, c fixed
    assert fp.rank_defect == 6 and fp.n_unknowns - fp.rank_defect == 450 + 2 + 4 + 5 + 6 * 115
    v, Q, res = oracle_mod.Oracle(fp).estimate(invert=2)                                           # MatrixInversion.REDUCED (BA:261-267)
    assert res.state == 1 and res.iterations <= 12, (res.state, res.iterations)
    s0 = np.sqrt(res.omega / fp.degree_of_freedom)
    _, ba1 = example_adjustment(H, example_base)
    ba1.useCentroidedCoordinates(False); ba1.prepareUnknownParameters(); ba1.flatten()
    fp1 = flat_problem(ba1).validate()
    _, _, r1 = oracle_mod.Oracle(fp1).estimate(invert=False)
    s1 = np.sqrt(r1.omega / fp1.degree_of_freedom)
    assert s0 < 1.5 * s1, (s0, s1)       # five radially symmetric Zernike terms describe this lens about as well as A1, A2 + B1, B2 did


@pytest.mark.gpu
def test_third_example_on_the_engine_matches_the_oracle(H, example_base, oracle_mod):
    """the only reference-shipped configuration of this path the suite did not run.  estimateModel() of the host mirror
    on the engine (EO pre-eliminated: reduced order U - 6 I) against the oracle's literal REDUCED run of the same flat problem: same pass
    count, adjusted parameters 1e-9, Omega, the standard deviations of the five Zernike coefficients and of x0, y0, the point variances."""
    from bundle_adjustment_amd.host_api import flat_problem
    from bundle_adjustment_amd.problem import DIST_ZERNIKE_Z
    pr0, ba0 = distortion_model_adjustment(H, example_base)
    ba0.useCentroidedCoordinates(False)
    ba0.prepareUnknownParameters(); ba0.flatten()
    fp = flat_problem(ba0).validate()
    vo, Qo, ro = oracle_mod.Oracle(fp).estimate(invert=2)
    pr, ba = distortion_model_adjustment(H, example_base)
    ba.useCentroidedCoordinates(False)
    state = ba.estimateModel()
    assert state == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
    assert ro.state == 1 and ba.getIterations() == ro.iterations
    pts = ba.getObjectCoordinates()
    got = np.array([[p.getX().getValue(), p.getY().getValue(), p.getZ().getValue()] for p in pts]).ravel()
    ref = vo[:got.size]
    assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)).max() < 1e-9
    T = H.DistortionModelType
    zern = pr.camera.getDistortionModel(T.ZERNIKE_GRADIENT)
    zslots = [j for j, k in enumerate(fp.dist_kind) if k == DIST_ZERNIKE_Z]
    base = 3 * fp.n_points + 3 * fp.n_cameras
    for j, order in zip(zslots, (4, 12, 24, 40, 60)):
        zv, zo = zern.get(order).getValue(), vo[base + j]
        assert abs(zv - zo) <= 1e-9 * max(abs(zo), 1e-3), (order, zv, zo)
    assert abs(ba.getOmega() - ro.omega) < 1e-8 * ro.omega
    # cofactors: the leading numRows block (border, points, x0, y0, the five coefficients) of the reduced inverse
    U, d = fp.n_unknowns, fp.rank_defect
    k = U - 6 * fp.n_images
    Q = packed_to_full(np.array(ba.getCofactorMatrix()), U)[:k, :k]
    Qr = packed_to_full(Qo, U)[:k, :k]
    cols = [int(fp.dist_col[j]) for j in zslots] + [int(fp.io_col[0][0]), int(fp.io_col[0][1])]
    for c in cols:
        assert d <= c < k
        assert abs(Q[c, c] - Qr[c, c]) <= 1e-8 * Qr[c, c], (c, Q[c, c], Qr[c, c])
    np.testing.assert_allclose(np.diag(Q)[d:], np.diag(Qr)[d:], rtol=1e-7)
    sc = np.sqrt(np.abs(np.diag(Qr)[d:]))          # (the datum border's rows have no variances to scale by)
    assert np.abs((Q[d:, d:] - Qr[d:, d:]) / np.outer(sc, sc)).max() < 1e-7


@pytest.mark.gpu
def test_native_example_distortion_model_program(example_base):
    """bundle-adjustment_amd/host/example_distortion_model = ExampleDistortionModel.java as a native program on the engine: the
    reference's listing (object points with uncertainties, interior orientation, every distortion parameter with `fixed` where it is, the
    statistics block)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "bundle-adjustment_amd", "host", "example_distortion_model")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "example_distortion_model"])
    out = subprocess.run([exe, example_base], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    assert "Bundle adjustment finished successfully..." in txt
    assert "Number of observations:           19945" in txt and "Number of unknown parameters:     1151" in txt
    assert "Degree of freedom:                18800" in txt
    lines = txt.splitlines()
    assert sum(1 for l in lines if l.rstrip().endswith("\to") or l.rstrip().endswith("\tn")) == 150     # one line per object point
    assert any(l.startswith("PRINCIPAL_DISTANCE") and "+28.0000000000 fixed" in l for l in lines)
    for order in (4, 12, 24, 40, 60):
        row = [l for l in lines if l.startswith(f"ZERNIKE_POLYNOMIAL_Z({order})")]
        assert len(row) == 1 and "fixed" not in row[0] and "+/-" in row[0], row
    for a in (1, 2, 3):
        row = [l for l in lines if l.startswith(f"RADIAL_POLYNOMIAL_A({a})")]
        assert len(row) == 1 and "+0.0000000000 fixed" in row[0], row


def test_zernike_models_in_the_object_api(H):
    """Camera with the three Zernike models next to the radial one (Camera.java:45-83 sorts the model types, DistortionModel
    ordinal = application order; ZernikeDistortionModel.java:36-60 parameter types): columns and flat descriptor."""
    from bundle_adjustment_amd.host_api import flat_problem
    from bundle_adjustment_amd.problem import DIST_RADIAL_AI, DIST_ZERNIKE_X, DIST_ZERNIKE_Y, DIST_ZERNIKE_Z
    T, PT = H.DistortionModelType, H.ParameterType
    cam = H.Camera(1, 13.488, [T.ZERNIKE_GRADIENT, T.RADIAL_DISTORTION, T.ZERNIKE_Y, T.ZERNIKE_X])   # any order: sorted
    cam.getDistortionModel(T.RADIAL_DISTORTION).add(1)
    zx, zy, zz = (cam.getDistortionModel(t) for t in (T.ZERNIKE_X, T.ZERNIKE_Y, T.ZERNIKE_GRADIENT))
    assert zx.add(4).getParameterType() == PT.ZERNIKE_POLYNOMIAL_X
    zx.add(7); zy.add(12); zz.add(3)
    assert zz.get(3).getParameterType() == PT.ZERNIKE_POLYNOMIAL_Z and zy.get(12).getParameterType() == PT.ZERNIKE_POLYNOMIAL_Y
    with pytest.raises(Exception):
        zx.add(4)                                    # order exists already (PolynomialDistortionModel.java:60-62)
    with pytest.raises(Exception):
        zy.add(0)                                    # ZernikeDistortionModel.java:67-68
    rng = np.random.default_rng(1)
    pts = [H.ObjectCoordinate(str(i), *rng.normal(0, 100, 3)) for i in range(6)]
    for i in range(3):
        im = cam.add(i)
        for p in pts:
            im.add(p, 0.1 * i, 0.2, 0.001, 0.001)
    ba = H.BundleAdjustment(); ba.add(cam)
    ba.prepareUnknownParameters(); ba.flatten()
    fp = flat_problem(ba)
    assert list(fp.dist_kind) == [DIST_RADIAL_AI, DIST_ZERNIKE_X, DIST_ZERNIKE_X, DIST_ZERNIKE_Y, DIST_ZERNIKE_Z]
    assert list(fp.dist_order) == [1, 4, 7, 12, 3]
    d = fp.rank_defect
    # columns: 18 point coordinates, then x0, y0, c, then the five coefficients in model order, then the EO blocks
    assert list(fp.dist_col) == [d + 18 + 3 + j for j in range(5)] and fp.cam_r0[0] == 13.488


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["FULL", "REDUCED"])
def test_native_example_program(example_base, mode):
    """bundle-adjustment_amd/host/example_flatfiles: the reference's ExampleFlatFiles as a native program on the engine (C++
    host mirror + C ABI, no Python in the loop): known answers of the bundled block (example.htm:31,33-35,42)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "bundle-adjustment_amd", "host", "example_flatfiles")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "example_flatfiles"])
    out = subprocess.run([exe, example_base, mode], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        if len(parts) >= 2:
            kv[" ".join(parts[:-1])] = parts[-1]
    assert kv["observations"] == "19945" and kv["unknown parameters"] == "1147"
    assert kv["datum conditions"] == "6" and kv["degree of freedom"] == "18804"
    # sigma0^2 a-priori = the smallest observation variance (BA:98,641), and the .phc file carries AICON's per-point sigmas:
    # sigma0 a-posteriori comes out at 1.2e-4; with the protocol's uniform a-priori sigma the oracle test above reproduces
    # AICON's 0.000405
    assert 5e-5 < float(kv["sigma0 a-posteriori"]) < 5e-4
    assert kv["iterations"] == "4"
    assert "ERROR_FREE_ESTIMATION" in out.stdout and "Estimation time" in out.stdout


def test_report_reader_known_answers(H, example_base, example_report):
    """AICONReportFileReader.java:117-390 restated natively: the bundled report gives the protocol's own
    n = 19 945, u = 1 147, d = 6, redundancy 18 804 (example.htm:33-35,42) with ExampleReport's datum choice, the fixed
    flags of A3, C1, C2 (example.htm:83,86-87), c = -Ck, and the same block as the flat files of the same project."""
    pr = H.read_aicon_report(example_report)
    cams = pr.cameras()
    assert len(cams) == 1 and cams[0].getId() == 1
    cam = cams[0]
    io = cam.getInteriorOrientation()
    assert io.getPrincipleDistance().getValue() == 28.78507 and io.getPrinciplePointX().getValue() == 0.01734892
    rad = cam.getDistortionModel(H.DistortionModelType.RADIAL_DISTORTION)
    aff = cam.getDistortionModel(H.DistortionModelType.AFFINITY_AND_SHEAR)
    assert rad.get(3).getColumn() == H.COLUMN_FIXED and rad.get(1).getColumn() == H.COLUMN_NOT_SET
    assert aff.getCx().getColumn() == H.COLUMN_FIXED and aff.getCy().getColumn() == H.COLUMN_FIXED
    assert aff.getCx().getValue() == -7.008010e-05
    assert len(pr.scaleBars()) == 1 and all(p.isDatum() for p in pr.points())
    # the same project as the flat files: images, points, rays per image
    fl = H.read_aicon_flat(example_base)
    rays = lambda c: {im.getId(): sorted(ic.getObjectCoordinate().getName() for ic in im.coordinates()) for im in c.images()}
    r_htm, r_flat = rays(cam), {k: v for k, v in rays(fl.camera).items() if v}
    assert r_htm == r_flat and sum(len(v) for v in r_htm.values()) == 9972
    used = {n for v in r_htm.values() for n in v}
    assert used == {p.getName() for p in pr.points()} and len(used) == 150
    flat_pts = {p.getName(): p for p in fl.points()}
    for p in pr.points():     # report: 4 decimals, .obc: more
        q = flat_pts[p.getName()]
        assert abs(p.getX().getValue() - q.getX().getValue()) < 1e-4 and abs(p.getZ().getValue() - q.getZ().getValue()) < 1e-4
    for p in pr.points():
        if len(p.getName()) > 3:
            p.setDatum(False)
    ba = H.BundleAdjustment()
    for c in cams:
        ba.add(c)
    for sb in pr.scaleBars():
        ba.add(sb)
    ba.prepareUnknownParameters()
    assert (ba.getNumberOfObservations(), ba.getNumberOfUnknownParameters(), ba.getNumberOfDatumConditions(),
            ba.getDegreeOfFreedom()) == (19945, 1147, 6, 18804)
    assert ba.getVarianceFactorApriori() == 0.0005 ** 2          # min(1, smallest variance), BA:98,641; example.htm:31


def test_report_reader_drops_malformed_lines(H, tmp_path):
    """parse() swallows what it cannot read (AICONReportFileReader.java:174-178): residual-flagged rays ('***'), rays of unknown
    points or images, a scale bar between unknown points, numbers with trailing garbage."""
    txt = """<h4><a name="interior_orientations">*** Innere Orientierungen ***</a></h4>
 Kamera/R0:               7     1.0e+001
 Ck       :  -2.0e+001   2.5e-004
 Xh       :   1.0e-002   fest
 A1       :  -1.0e-004   3.0e-008
 A9       :   5.0e-001   1.0e-003
 AZ2      :   2.0e-006   1.0e-008
<h4><a name="exterior_orientations">*** Aeussere Orientierungen ***</a></h4>
       3            7   10.0   20.0   30.0     0.01     0.02     0.03         81
           air  rad    0.1  0.2 -0.3   0.000028   0.000020   0.000075   0.000409   0.000411
       4            8   10.0   20.0   30.0     0.01     0.02     0.03         81
<h4><a name="object_points">*** Objektpunkte ***</a></h4>
P1        1.0       2.0      3.0    0.01    0.01    0.01       22        1
P2        4.0       5.0      6.0    0.01    0.01    0.01       22        1
P3        4.0x      5.0      6.0    0.01    0.01    0.01       22        1
<h4><a name="image_coordinates">*** Bildkoordinaten ***</a></h4>
P1      3         -0.5     -8.9   -0.001    0.007    0.0005    0.0004   1.00   1.00  3.44 19.67
P2      3          3.8     -8.6   -0.021    0.013    0.0005    0.0005   1.00   1.00 57.10 35.13  ***
P9      3          3.8     -8.6   -0.021    0.013    0.0005    0.0005   1.00   1.00 57.10 35.13
P2      4          3.8     -8.6   -0.021    0.013    0.0005    0.0005   1.00   1.00 57.10 35.13
<h4><a name="distances">*** Strecken ***</a></h4>
P1     P2        789.8480    -0.0117     0.1771     0.0100       3.71  ---
P1     P9        789.8480    -0.0117     0.1771     0.0100       3.71  ---
P1     P1        789.8480    -0.0117     0.1771     0.0100       3.71  ---
<h4><a name="Zusammenfassung">x</a> <a href="#Start">(zum Anfang)</a></h4>
P2      3          3.8     -8.6   -0.021    0.013    0.0005    0.0005   1.00   1.00 57.10 35.13
"""
    f = tmp_path / "r.htm"
    f.write_text(txt)
    pr = H.read_aicon_report(str(f))
    cam = pr.cameras()[0]
    assert cam.getId() == 7 and [im.getId() for im in cam.images()] == [3]          # image 4 names camera 8: dropped
    io = cam.getInteriorOrientation()
    assert io.getPrincipleDistance().getValue() == 20.0 and io.getPrinciplePointX().getColumn() == H.COLUMN_FIXED
    assert cam.getDistortionModel(H.DistortionModelType.RADIAL_DISTORTION).get(1).getValue() == -1.0e-4
    assert cam.getDistortionModel(H.DistortionModelType.DISTANCE_DISTORTION).get(2).getValue() == 2.0e-6
    im = cam.images()[0]
    eo = im.getExteriorOrientation()
    PT = H.ParameterType
    got = [eo.get(t).getValue() for t in (PT.CAMERA_COORDINATE_X, PT.CAMERA_COORDINATE_Y, PT.CAMERA_COORDINATE_Z, PT.CAMERA_OMEGA,
                                          PT.CAMERA_PHI, PT.CAMERA_KAPPA)]
    assert got == [10.0, 20.0, 30.0, 0.1, 0.2, -0.3]
    assert sorted(p.getName() for p in pr.points()) == ["P1", "P2"]                  # P3: 4.0x is no number
    assert [(ic.getObjectCoordinate().getName(), ic.getX().getValue()) for ic in im.coordinates()] == [("P1", -0.5)]
    assert im.coordinates()[0].getY().getVariance() == 0.0004 ** 2
    sb = pr.scaleBars()
    assert len(sb) == 1 and sb[0].getLength().getValue() == 789.848 and sb[0].getLength().getVariance() == 0.01 ** 2


@pytest.mark.gpu
def test_native_example_report_program(example_report):
    """bundle-adjustment_amd/host/example_report: the reference's ExampleReport (report reader, REDUCED inversion,
    ExampleReport.java:52-172) as a native program on the engine.  The report carries AICON's adjusted values and uniform
    a-priori sigmas, so the adjustment is a near fixed point and reproduces the protocol's S0 = 0.000405 (example.htm:31)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "bundle-adjustment_amd", "host", "example_report")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "example_report"])
    out = subprocess.run([exe, example_report], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = {}
    for line in out.stdout.splitlines():
        if ":" in line:
            k, v = line.split(":", 1)
            kv[k.strip()] = v.strip()
    assert kv["Number of observations"] == "19945" and kv["Number of unknown parameters"] == "1147"
    assert kv["Number of datum conditions"] == "6" and kv["Degree of freedom"] == "18804"
    apri, apost = (float(x) for x in kv["Variances of unit weight (ratio)"].split(":"))
    assert apri == 0.0005 ** 2 and abs(np.sqrt(apost) - 0.000405) < 1.5e-6
    assert int(kv["Iterations"]) <= 4
    rows = [l.split("\t") for l in out.stdout.splitlines() if l.count("\t") == 7]
    assert len(rows) == 150 and sum(r[7] == "d" for r in rows) == sum(len(r[0].strip()) <= 3 for r in rows)
    sig = np.array([[float(r[4]), float(r[5]), float(r[6])] for r in rows])
    assert np.all(sig > 0) and np.all(sig < 0.1)          # mm; the protocol lists 0.008 .. 0.02 (example.htm:1608ff)
    assert "PRINCIPAL_DISTANCE" in out.stdout and "RADIAL_POLYNOMIAL_A(3)" in out.stdout and "fixed" in out.stdout


def test_java_fixed_follows_the_java_formatter(H):
    """The text writers print with java.util.Formatter's %f (DefaultResultWriter.java:70,145): the shortest round-trip digits
    (Double.toString), rounded HALF_UP, zero padded -- where C's printf would go on with the binary expansion (0.1 ->
    0.1000000000000000055...) and round half-even on it (0.125 -> 0.12)."""
    cases = [((0.1, 20, False), "0.10000000000000000000"), ((1234.5678, 15, True), "+1234.567800000000000"),
             ((-1.23456789012345678e-7, 15, False), "-0.000000123456789"), ((0.125, 2, False), "0.13"), ((2.5, 0, False), "3"),
             ((0.99999, 2, True), "+1.00"), ((1e22, 2, False), "10000000000000000000000.00"), ((5e-324, 3, False), "0.000"),
             ((0.0, 3, True), "+0.000"), ((123456789.987654321, 15, False), "123456789.987654330000000"),
             ((-99.9999999999999999, 3, True), "-100.000"), ((float("nan"), 3, True), "NaN"), ((float("-inf"), 3, True), "-Infinity")]
    for args, want in cases:
        assert H.java_fixed(*args) == want, (args, H.java_fixed(*args))


def test_result_writers_without_cofactor(H, example_base, tmp_path):
    """MatlabResultWriter.java:60-222 / DefaultResultWriter.java:62-117 before any inversion (cofactor == null): variables,
    classes and struct fields of the .mat (read back with scipy), the .info listing, no `cov` fields, no dispersion, no .cxx."""
    import scipy.io
    pr, ba = example_adjustment(H, example_base)
    ba.prepareUnknownParameters()
    base = str(tmp_path / "result")
    H.MatlabResultWriter(base).export(ba)
    H.DefaultResultWriter(base).export(ba)
    m = scipy.io.loadmat(base + ".mat")
    assert m["variance_of_unit_weight_prio"].dtype == np.float64 and m["degree_of_freedom"].dtype == np.int32
    assert int(m["number_of_observations"][0, 0]) == 19945 and int(m["number_of_unknowns"][0, 0]) == 1147
    assert int(m["degree_of_freedom"][0, 0]) == 18804 and "dispersion" not in m
    c = m["coordinates"]
    assert c.shape == (1, 150) and c.dtype.names == ("name", "X", "Y", "Z", "covx", "covy", "covz")
    pts = ba.getObjectCoordinates()
    k = 1
    for i, p in enumerate(pts):
        e = c[0, i]
        assert e["name"][0] == p.getName() and e["X"][0, 0] == p.getX().getValue() and e["Z"][0, 0] == p.getZ().getValue()
        assert e["covx"].dtype == np.int32 and [int(e[f][0, 0]) for f in ("covx", "covy", "covz")] == [k, k + 1, k + 2]
        k += 3
    io = m["interior_orientations"]
    assert io.dtype.names == ("cam_id", "name", "value") and io.shape == (1, 3)
    assert [str(io[0, i]["name"][0]) for i in range(3)] == ["principal_point_x", "principal_point_y", "principal_distance"]
    assert io[0, 0]["cam_id"].dtype == np.int64 and io[0, 2]["value"][0, 0] == 28.78507
    di = m["distortion_parameters"]
    assert di.dtype.names == ("cam_id", "name", "value", "order") and di.shape == (1, 7)
    names = [(str(di[0, i]["name"][0]), int(di[0, i]["order"][0, 0])) for i in range(7)]
    assert names == [("affinity_and_shear_cx", -1), ("affinity_and_shear_cy", -1), ("tangential_distortion_bx", -1),
                     ("tangential_distortion_by", -1), ("radial_polynomial_a", 1), ("radial_polynomial_a", 2), ("radial_polynomial_a", 3)]
    lines = open(base + ".info").read().split("\n")
    assert len(lines) == 3 * 150 + 1 and lines[-1] == ""
    p0 = pts[0]
    assert lines[0] == "%25s\t%5s\t%35s\t%10d" % (p0.getName(), "X", H.java_fixed(p0.getX().getValue(), 15, False), 0)
    assert lines[4] == "%25s\t%5s\t%35s\t%10d" % (pts[1].getName(), "Y", H.java_fixed(pts[1].getY().getValue(), 15, False), 4)
    assert not os.path.exists(base + ".cxx")
    with pytest.raises(Exception):
        H.MatlabResultWriter("").export(ba)                      # "Error, export path cannot be null!"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["FULL", "REDUCED"])
def test_result_writers_export_the_device_gathered_dispersion(H, example_base, tmp_path, mode):
    """estimateModel() with a result writer set (BA:360-368): the .mat `dispersion` is the cofactor block of
    points + interior orientation + distortion (MatlabResultWriter.java:210-221, unscaled), the .cxx block is
    sigma2apost * Qxx of the points (DefaultResultWriter.java:139-147) -- both gathered on the device, compared with the
    host copy of the packed matrix."""
    import scipy.io
    base = str(tmp_path / "adjustment_results")
    results = {}
    for W in (H.MatlabResultWriter, H.DefaultResultWriter):
        pr, ba = example_adjustment(H, example_base)
        ba.setInvertNormalEquation(getattr(H.MatrixInversion, mode))
        events = []
        ba.addPropertyChangeListener(lambda n, a, b: events.append(n))
        w = W(base)
        ba.setAdjustmentResultWriter(w)
        assert ba.estimateModel() == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
        assert "EXPORT_ADJUSTMENT_RESULTS" in events and ba.hasCofactorMatrix()
        results[W] = (pr, ba)
    pr, ba = results[H.MatlabResultWriter]
    U = ba.getNumberOfUnknownParameters() + ba.getNumberOfDatumConditions()
    Q = packed_to_full(np.asarray(ba.getCofactorMatrix()), U)
    m = scipy.io.loadmat(base + ".mat")
    idx = []
    for p in ba.getObjectCoordinates():
        idx += [p.getX().getColumn(), p.getY().getColumn(), p.getZ().getColumn()]
    cam = pr.camera
    io = cam.getInteriorOrientation()
    par = [io.getPrinciplePointX(), io.getPrinciplePointY(), io.getPrincipleDistance()]
    for t in (H.DistortionModelType.AFFINITY_AND_SHEAR, H.DistortionModelType.TANGENTIAL_DISTORTION, H.DistortionModelType.RADIAL_DISTORTION):
        par += list(cam.getDistortionModel(t).parameters())
    cov = [int(m["interior_orientations"][0, i]["cov"][0, 0]) for i in range(3)] + \
          [int(m["distortion_parameters"][0, i]["cov"][0, 0]) for i in range(7)]
    nxt = len(idx) + 1
    for u, c in zip(par, cov):
        if 0 <= u.getColumn() < U:
            assert c == nxt
            idx.append(u.getColumn()); nxt += 1
        else:
            assert c == -1                                          # A3, Cx, Cy are fixed
    D = m["dispersion"]
    assert D.shape == (len(idx), len(idx)) and len(idx) == 3 * 150 + 7
    np.testing.assert_array_equal(D, Q[np.ix_(idx, idx)])
    assert float(m["variance_of_unit_weight_post"][0, 0]) == ba.getVarianceFactorAposteriori()
    # text writer: 450 x 450 block of the points, scaled
    pr2, ba2 = results[H.DefaultResultWriter]
    Q2 = packed_to_full(np.asarray(ba2.getCofactorMatrix()), U)
    pidx = idx[:450]
    rows = open(base + ".cxx").read().split("\n")
    assert len(rows) == 451 and rows[-1] == "" and all(len(r) == 450 * 37 for r in rows[:-1])
    T = np.array([[float(r[37 * j:37 * j + 35]) for j in range(450)] for r in rows[:-1]])
    np.testing.assert_allclose(T, ba2.getVarianceFactorAposteriori() * Q2[np.ix_(pidx, pidx)], rtol=0, atol=6e-16)
    assert rows[0][:35].lstrip()[0] in "+-"
    # a writer that cannot write: BA:362-367
    pr3, ba3 = example_adjustment(H, example_base)
    ba3.setAdjustmentResultWriter(H.MatlabResultWriter(str(tmp_path / "no_such_dir" / "x")))
    assert ba3.estimateModel() == H.EstimationStateType.EXPORT_ADJUSTMENT_RESULTS_FAILED


def _report_interior_precision(path):
    """Standard deviations and correlation matrix of the estimated interior-orientation parameters as printed by AICON 3D
    Studio in the bundled report (example.htm:77-98): third-party results for the same observations."""
    lines = open(path, encoding="latin-1").read().split("\n")
    start = next(i for i, l in enumerate(lines) if 'name="interior_orientations"' in l)
    sig, order, corr = {}, [], {}
    i = start + 1
    while not lines[i].startswith("Korrelation"):
        t = lines[i].replace(":", " ").split()
        if len(t) == 3 and "/" not in t[0] and t[2] != "fest":
            sig[t[0]] = float(t[2])
        i += 1
    i += 1
    while True:
        t = lines[i].split()
        i += 1
        if not t:
            if order:
                break
            continue
        if t[0] in sig and len(t) == len(order) + 2:
            order.append(t[0])
            for name, v in zip(order, t[1:]):
                corr[(t[0], name)] = corr[(name, t[0])] = float(v)
        else:
            break
    return sig, order, corr


def _check_interior_precision(H, cam, dispersion_of, report):
    """dispersion_of(list of columns) -> sigma2apost * Qxx block.  Interior orientation and distortion are invariant to the
    choice of the (minimal) datum, so AICON's values must come out although its datum differs (ExampleReport.java:71-82).
    The report lists Ck, the engine estimates c = -Ck: correlations with Ck change sign."""
    sig, order, corr = _report_interior_precision(report)
    assert order == ["Ck", "Xh", "Yh", "A1", "A2", "B1", "B2"]
    io = cam.getInteriorOrientation()
    rad = cam.getDistortionModel(H.DistortionModelType.RADIAL_DISTORTION)
    tan = cam.getDistortionModel(H.DistortionModelType.TANGENTIAL_DISTORTION)
    par = {"Ck": io.getPrincipleDistance(), "Xh": io.getPrinciplePointX(), "Yh": io.getPrinciplePointY(), "A1": rad.get(1),
           "A2": rad.get(2), "B1": tan.getBx(), "B2": tan.getBy()}
    C = dispersion_of([par[n].getColumn() for n in order])
    s = np.sqrt(np.diag(C))
    for k, n in enumerate(order):
        assert abs(s[k] / sig[n] - 1.0) < 2e-5, (n, s[k], sig[n])        # the report prints 7 digits
    R = C / np.outer(s, s)
    for a, na in enumerate(order):
        for b, nb in enumerate(order):
            flip = -1.0 if (na == "Ck") != (nb == "Ck") else 1.0
            assert abs(R[a, b] - flip * corr[(na, nb)]) < 6e-4, (na, nb, R[a, b], corr[(na, nb)])   # printed with 3 decimals


def test_oracle_reproduces_the_reports_interior_orientation_precision(H, example_report, oracle_mod):
    """Pins the oracle's covariance path (datum border, Bunch-Kaufman solve + inverse, a-posteriori variance factor;
    BA:493-635, MX:338-366, BA:1090-1093) on third-party numbers held by the reference's own example: the standard
    deviations (6 digits) and the correlation matrix (3 decimals) of c, x0, y0, A1, A2, B1, B2 in example.htm:77-98."""
    from bundle_adjustment_amd.host_api import flat_problem
    pr = H.read_aicon_report(example_report)
    cam = pr.cameras()[0]
    for p in pr.points():
        if len(p.getName()) > 3:
            p.setDatum(False)
    ba = H.BundleAdjustment()
    ba.add(cam)
    for sb in pr.scaleBars():
        ba.add(sb)
    ba.useCentroidedCoordinates(False)
    ba.prepareUnknownParameters(); ba.flatten()
    fp = flat_problem(ba).validate()
    v, Q, res = oracle_mod.Oracle(fp).estimate()
    assert res.state == 1
    s2 = res.omega / fp.degree_of_freedom
    Qf = packed_to_full(np.asarray(Q), fp.n_unknowns)
    _check_interior_precision(H, cam, lambda cols: s2 * Qf[np.ix_(cols, cols)], example_report)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["REDUCED", "FULL"])
def test_engine_reproduces_the_reports_interior_orientation_precision(H, example_report, mode):
    """The same third-party numbers from the engine: report reader -> estimateModel() on the MI355X -> dispersion block
    gathered on the device (the writers' path)."""
    pr = H.read_aicon_report(example_report)
    cam = pr.cameras()[0]
    for p in pr.points():
        if len(p.getName()) > 3:
            p.setDatum(False)
    ba = H.BundleAdjustment()
    ba.add(cam)
    for sb in pr.scaleBars():
        ba.add(sb)
    ba.setInvertNormalEquation(getattr(H.MatrixInversion, mode))
    assert ba.estimateModel() == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
    assert abs(np.sqrt(ba.getVarianceFactorAposteriori()) - 0.000405) < 1.5e-6
    _check_interior_precision(H, cam, lambda cols: ba.cofactorSub(cols, ba.getVarianceFactorAposteriori()), example_report)


def _report_point_and_station_precision(path):
    """sx, sy, sz of the object points (example.htm:1608ff) and of the projection centres (example.htm:108ff), 4 decimals."""
    import re
    lines = open(path, encoding="latin-1").read().split("\n")
    a = next(i for i, l in enumerate(lines) if 'name="exterior_orientations"' in l)
    b = next(i for i, l in enumerate(lines) if 'name="object_points"' in l)
    c = next(i for i, l in enumerate(lines) if 'name="image_coordinates"' in l)
    eo, pts = {}, {}
    for l in lines[a:b]:
        t = l.split()
        if len(t) == 9 and t[0].isdigit() and t[1].isdigit():
            eo[int(t[0])] = [float(x) for x in t[5:8]]
    for l in lines[b:c]:
        t = l.split()
        if len(t) == 9 and re.match(r"^\w+$", t[0]):
            try:
                pts[t[0]] = [float(x) for x in t[4:7]]
            except ValueError:
                pass
    return pts, eo


def _check_point_and_station_precision(H, ba, cam, sigma_of, report):
    pts, eo = _report_point_and_station_precision(report)
    assert len(pts) == 150 and len(eo) == 115
    PT = H.ParameterType
    worst = 0.0
    for p in ba.getObjectCoordinates():
        s = sigma_of([p.getX().getColumn(), p.getY().getColumn(), p.getZ().getColumn()])
        worst = max(worst, np.abs(s - np.array(pts[p.getName()])).max())
    for im in cam.images():
        e = im.getExteriorOrientation()
        s = sigma_of([e.get(t).getColumn() for t in (PT.CAMERA_COORDINATE_X, PT.CAMERA_COORDINATE_Y, PT.CAMERA_COORDINATE_Z)])
        worst = max(worst, np.abs(s - np.array(eo[im.getId()])).max())
    assert worst < 5.1e-5, worst          # the report prints 4 decimals (mm)


def _report_adjustment_in_aicons_datum(H, example_report):
    """The report reader marks every point as datum point (AICONReportFileReader.java:262): the free-network datum over
    all object points, which is the one AICON's own adjustment used (example.htm:55-73) -- no ExampleReport re-selection."""
    pr = H.read_aicon_report(example_report)
    cam = pr.cameras()[0]
    ba = H.BundleAdjustment()
    ba.add(cam)
    for sb in pr.scaleBars():
        ba.add(sb)
    return pr, cam, ba


def test_oracle_reproduces_the_reports_point_and_station_precision(H, example_report, oracle_mod):
    """Second third-party pin of the oracle: in AICON's own datum (inner constraints over all 150 points, b = 6) the
    a-posteriori standard deviations of all object coordinates (450 values) and projection centres (345 values) agree with
    the report to its printed precision -- datum rows (BA:493-635), solve + inverse (MX:338-366), sigma0 (BA:1090-1093)."""
    from bundle_adjustment_amd.host_api import flat_problem
    pr, cam, ba = _report_adjustment_in_aicons_datum(H, example_report)
    ba.useCentroidedCoordinates(False)
    ba.prepareUnknownParameters(); ba.flatten()
    fp = flat_problem(ba).validate()
    assert fp.rank_defect == 6
    v, Q, res = oracle_mod.Oracle(fp).estimate()
    assert res.state == 1
    s2 = res.omega / fp.degree_of_freedom
    d = np.diag(packed_to_full(np.asarray(Q), fp.n_unknowns))
    _check_point_and_station_precision(H, ba, cam, lambda cols: np.sqrt(s2 * d[cols]), example_report)


@pytest.mark.gpu
def test_engine_reproduces_the_reports_point_and_station_precision(H, example_report):
    pr, cam, ba = _report_adjustment_in_aicons_datum(H, example_report)
    ba.setInvertNormalEquation(H.MatrixInversion.FULL)
    assert ba.estimateModel() == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
    s2 = ba.getVarianceFactorAposteriori()
    _check_point_and_station_precision(H, ba, cam, lambda cols: np.sqrt(np.diag(ba.cofactorSub(cols, s2))), example_report)


def _report_residuals(path):
    """AICON's corrections vx, vy of every image coordinate that took part in its adjustment (example.htm:1766ff, 6 decimals)."""
    lines = open(path, encoding="latin-1").read().split("\n")
    a = next(i for i, l in enumerate(lines) if 'name="image_coordinates"' in l)
    b = next(i for i, l in enumerate(lines) if 'name="distances"' in l)
    rep = {}
    for l in lines[a:b]:
        t = l.split()
        if len(t) == 12 and not l.rstrip().endswith("***"):
            try:
                rep[(t[0], int(t[1]))] = (float(t[4]), float(t[5]))
            except ValueError:
                pass
    return rep


def _check_residuals(cam, w_of, report):
    """w = observed - computed at the adjusted parameters must be minus AICON's corrections (computed - observed): the
    residuals do not depend on the datum, so all 19 944 of them pin the functional model -- collinearity, radial (A1..A3
    with R0), tangential, affinity -- on third-party values."""
    rep = _report_residuals(report)
    assert len(rep) == 9972
    k, worst = 0, 0.0
    for im in cam.images():
        for ic in im.coordinates():
            vx, vy = rep[(ic.getObjectCoordinate().getName(), im.getId())]
            w = w_of(k)
            worst = max(worst, abs(w[0] + vx), abs(w[1] + vy))
            k += 1
    assert k == 9972 and worst < 1.5e-6, worst        # mm; the report prints 6 decimals


def test_oracle_reproduces_the_reports_residuals(H, example_report, oracle_mod):
    from bundle_adjustment_amd.host_api import flat_problem
    pr, cam, ba = _report_adjustment_in_aicons_datum(H, example_report)
    ba.useCentroidedCoordinates(False)
    ba.prepareUnknownParameters(); ba.flatten()
    fp = flat_problem(ba).validate()
    O = oracle_mod.Oracle(fp)
    v, Q, res = O.estimate(invert=False)
    assert res.state == 1
    _check_residuals(cam, lambda k: O.rows(v, k)[0], example_report)


@pytest.mark.gpu
def test_engine_reproduces_the_reports_residuals(H, example_report):
    from bundle_adjustment_amd import engine
    from bundle_adjustment_amd.host_api import flat_problem
    pr, cam, ba = _report_adjustment_in_aicons_datum(H, example_report)
    ba.useCentroidedCoordinates(False)
    ba.setInvertNormalEquation(H.MatrixInversion.NONE)
    assert ba.estimateModel() == H.EstimationStateType.ERROR_FREE_ESTIMATION, ba.lastError()
    ba.flatten()                                   # the adjusted values, flattened again
    fp = flat_problem(ba).validate()
    eng = engine.Engine(fp)
    try:
        eng.set_parameters(fp.values)
        w, A = eng.get_rows(0, fp.n_image_points)
    finally:
        eng.close()
    _check_residuals(cam, lambda k: w[k], example_report)


def _control_point_block(H, base, n_ctrl=5):
    """The bundled block with a DirectlyObservedParameterGroup on the first n_ctrl object points (X, Y, Z each)."""
    pr = H.read_aicon_flat(base)
    ba = H.BundleAdjustment()
    for cam in pr.cameras():
        ba.add(cam)
    obs = []
    for p in pr.points()[:n_ctrl]:
        for up in (p.getX(), p.getY(), p.getZ()):
            o = H.ObservationParameter(up); o.setVariance(0.05 ** 2); obs.append(o)
    grp = H.DirectlyObservedParameterGroup(obs)
    ba.add(grp)
    return pr, ba, obs


def test_centroid_coordinates_match_the_oracle_restatement(H, example_base, oracle_mod):
    """BundleAdjustment.centroidCoordinates (BundleAdjustment.java:115-201): host mirror vs the oracle's restatement on the same
    flat problem, forward and back, to 1e-12 of the coordinate scale (north_star: 1e-9) -- parameters AND the observed values
    of the directly observed coordinates (BA:179-200)."""
    from bundle_adjustment_amd.host_api import flat_problem
    pr, ba, obs = _control_point_block(H, example_base)
    ba.prepareUnknownParameters(); ba.flatten()
    fp = flat_problem(ba).validate()
    assert fp.n_direct_rows == 15
    o = oracle_mod.Oracle(fp)
    vo, dgo, c = o.centroid(fp.values, fp.dg_obs)
    ba.centroidCoordinates(False); ba.flatten()
    f1 = flat_problem(ba)
    scale = np.abs(fp.values[:3 * fp.n_points]).max()
    assert np.abs(f1.values - vo).max() <= 1e-12 * scale
    assert np.abs(f1.dg_obs - dgo).max() <= 1e-12 * scale
    # the centroid is the mean over object AND camera coordinates that own a column (BA:119-139)
    cols = fp.slot_columns()
    s_eo = fp.slot_eo(0)
    xs = np.concatenate([fp.values[0:3 * fp.n_points:3], fp.values[s_eo::6]])
    assert np.all(cols[:3 * fp.n_points] >= 0) and abs(c[0] - xs.mean()) <= 1e-12 * scale
    assert np.abs(f1.values[0:3 * fp.n_points:3].sum() + f1.values[s_eo::6].sum()) <= 1e-9 * scale   # centred
    # interior orientation, angles and distortion are not shifted
    np.testing.assert_array_equal(f1.values[3 * fp.n_points:s_eo], fp.values[3 * fp.n_points:s_eo])
    np.testing.assert_array_equal(f1.values[s_eo + 3::6], fp.values[s_eo + 3::6])
    # and back (BA:357-358)
    vb, dgb, _ = o.centroid(vo, dgo, True, c)
    ba.centroidCoordinates(True); ba.flatten()
    f2 = flat_problem(ba)
    assert np.abs(f2.values - vb).max() <= 1e-12 * scale and np.abs(f2.dg_obs - dgb).max() <= 1e-12 * scale
    assert np.abs(f2.values - fp.values).max() <= 1e-12 * scale


def test_centroid_needs_equal_component_counts(H, example_base, oracle_mod):
    """BA:142-151: a fixed coordinate component (it is not an unknown parameter, BA:645-650) makes the counts unequal ->
    UnsupportedOperationException in the reference, an error in both restatements."""
    from bundle_adjustment_amd.host_api import flat_problem
    pr, ba, obs = _control_point_block(H, example_base)
    pr.points()[20].getY().setColumn(H.COLUMN_FIXED)
    ba.prepareUnknownParameters(); ba.flatten()
    fp = flat_problem(ba).validate()
    with pytest.raises(ArithmeticError):
        oracle_mod.Oracle(fp).centroid(fp.values, fp.dg_obs)
    with pytest.raises(Exception):
        ba.centroidCoordinates(False)
