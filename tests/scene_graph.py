"""The synthetic scenes of ``bundle_adjustment_amd.scene`` as an OBJECT GRAPH of the host mirror (Camera / Image / ObjectCoordinate /
DirectlyObservedParameterGroup / ScaleBar, the reference's API names): what a JAICOV user would build by hand.  Test infrastructure:
lets ``BundleAdjustment.estimateModel()`` run at BASELINE's config sizes and be compared with the flat-descriptor path."""
import numpy as np

from bundle_adjustment_amd import problem as P


def object_graph(H, fp):
    """Returns (ba, camera, points, images, groups).  One camera (the scenes have one); images, image coordinates and points are
    added in the order of the flat arrays, so BundleAdjustment.prepareUnknownParameters() numbers them as scene.py did."""
    T, PT = H.DistortionModelType, H.ParameterType
    assert fp.io_col.size == 3, "one camera"
    kinds = [int(k) for k in fp.dist_kind]
    types = []
    for k in kinds:
        t = (T.AFFINITY_AND_SHEAR if k in (P.DIST_AFFINITY_CX, P.DIST_AFFINITY_CY) else
             T.TANGENTIAL_DISTORTION if k in (P.DIST_TANGENTIAL_BX, P.DIST_TANGENTIAL_BY, P.DIST_TANGENTIAL_BI) else
             T.RADIAL_DISTORTION if k == P.DIST_RADIAL_AI else T.DISTANCE_DISTORTION if k == P.DIST_DISTANCE_DI else
             T.ZERNIKE_X if k == P.DIST_ZERNIKE_X else T.ZERNIKE_Y if k == P.DIST_ZERNIKE_Y else T.ZERNIKE_GRADIENT)
        if t not in types:
            types.append(t)
    cam = H.Camera(1, float(fp.cam_r0[0]), types)
    n_pts, n_img = fp.n_points, fp.n_images
    s_io = 3 * n_pts; s_dist = s_io + 3; s_eo = s_dist + len(kinds)
    v = fp.values
    io = cam.getInteriorOrientation()
    for up, val, col in zip((io.getPrinciplePointX(), io.getPrinciplePointY(), io.getPrincipleDistance()), v[s_io:s_io + 3], fp.io_col.ravel()):
        up.setValue(float(val))
        if col < 0:
            up.setColumn(H.COLUMN_FIXED)
    for j, k in enumerate(kinds):
        order = int(fp.dist_order[j])
        if k == P.DIST_AFFINITY_CX: up = cam.getDistortionModel(T.AFFINITY_AND_SHEAR).getCx()
        elif k == P.DIST_AFFINITY_CY: up = cam.getDistortionModel(T.AFFINITY_AND_SHEAR).getCy()
        elif k == P.DIST_TANGENTIAL_BX: up = cam.getDistortionModel(T.TANGENTIAL_DISTORTION).getBx()
        elif k == P.DIST_TANGENTIAL_BY: up = cam.getDistortionModel(T.TANGENTIAL_DISTORTION).getBy()
        elif k == P.DIST_TANGENTIAL_BI: up = cam.getDistortionModel(T.TANGENTIAL_DISTORTION).add(order)
        elif k == P.DIST_RADIAL_AI: up = cam.getDistortionModel(T.RADIAL_DISTORTION).add(order)
        elif k == P.DIST_DISTANCE_DI: up = cam.getDistortionModel(T.DISTANCE_DISTORTION).add(order)
        elif k == P.DIST_ZERNIKE_X: up = cam.getDistortionModel(T.ZERNIKE_X).add(order)
        elif k == P.DIST_ZERNIKE_Y: up = cam.getDistortionModel(T.ZERNIKE_Y).add(order)
        else: up = cam.getDistortionModel(T.ZERNIKE_GRADIENT).add(order)
        up.setValue(float(v[s_dist + j]))
        up.setColumn(H.COLUMN_FIXED if fp.dist_col[j] < 0 else H.COLUMN_NOT_SET)     # Cx, Cy, Bx, By are fixed by default (ASDM:34-40)
    pts = []
    for p in range(n_pts):
        oc = H.ObjectCoordinate(str(p), float(v[3 * p]), float(v[3 * p + 1]), float(v[3 * p + 2]))
        oc.setDatum(bool(fp.point_datum[p]))
        pts.append(oc)
    eo_t = (PT.CAMERA_COORDINATE_X, PT.CAMERA_COORDINATE_Y, PT.CAMERA_COORDINATE_Z, PT.CAMERA_OMEGA, PT.CAMERA_PHI, PT.CAMERA_KAPPA)
    images = []
    first = np.searchsorted(fp.ip_image, np.arange(n_img + 1))
    blk_of_ip0 = {int(fp.blk_ip_begin[b]): b for b in range(fp.n_image_blocks)}
    for i in range(n_img):
        im = cam.add(i)
        eo = im.getExteriorOrientation()
        for k, t in enumerate(eo_t):
            eo.get(t).setValue(float(v[s_eo + 6 * i + k]))
        lo, hi = int(first[i]), int(first[i + 1])
        for ip in range(lo, hi):
            im.add(pts[int(fp.ip_point[ip])], float(fp.ip_x[ip]), float(fp.ip_y[ip]), float(np.sqrt(fp.ip_var_x[ip])),
                   float(np.sqrt(fp.ip_var_y[ip])), float(fp.ip_rho[ip]))
        if lo in blk_of_ip0:      # joint dispersion of the whole image (Image.setDispersion: the API addition of SURVEY 8(d))
            b = blk_of_ip0[lo]
            m = 2 * (hi - lo)
            off = int(fp.blk_disp_offset[b])
            im.setDispersion(fp.blk_disp[off:off + m * m].reshape(m, m))
        images.append(im)
    ba = H.BundleAdjustment()
    ba.add(cam)
    groups = []
    coord = lambda slot: (pts[slot // 3].getX, pts[slot // 3].getY, pts[slot // 3].getZ)[slot % 3]()
    for g in range(len(fp.dg_row_begin) - 1):
        lo, hi = int(fp.dg_row_begin[g]), int(fp.dg_row_begin[g + 1])
        obs = []
        for r in range(lo, hi):
            o = H.ObservationParameter(coord(int(fp.dg_slot[r])))
            o.setValue(float(fp.dg_obs[r])); o.setVariance(float(fp.dg_var[r]))
            obs.append(o)
        off = int(fp.dg_disp_offset[g])
        grp = (H.DirectlyObservedParameterGroup(fp.dg_disp[off:off + (hi - lo) ** 2].reshape(hi - lo, hi - lo), obs) if off >= 0
               else H.DirectlyObservedParameterGroup(obs))
        ba.add(grp); groups.append((grp, obs))
    bars = []
    for a, b, ln, var in zip(fp.sb_point_a, fp.sb_point_b, fp.sb_length, fp.sb_var):
        sb = H.ScaleBar(pts[int(a)], pts[int(b)], float(ln), float(np.sqrt(var)))
        ba.add(sb); bars.append(sb)
    return ba, cam, pts, images, groups + bars


def adjusted_values(H, fp, cam, pts, images):
    """the slot vector [3P | x0 y0 c | distortion | 6I] read back from the object graph"""
    T, PT = H.DistortionModelType, H.ParameterType
    out = [c().getValue() for p in pts for c in (p.getX, p.getY, p.getZ)]
    io = cam.getInteriorOrientation()
    out += [io.getPrinciplePointX().getValue(), io.getPrinciplePointY().getValue(), io.getPrincipleDistance().getValue()]
    per_type = {}
    for j, k in enumerate(int(x) for x in fp.dist_kind):
        order = int(fp.dist_order[j])
        if k == P.DIST_AFFINITY_CX: up = cam.getDistortionModel(T.AFFINITY_AND_SHEAR).getCx()
        elif k == P.DIST_AFFINITY_CY: up = cam.getDistortionModel(T.AFFINITY_AND_SHEAR).getCy()
        elif k == P.DIST_TANGENTIAL_BX: up = cam.getDistortionModel(T.TANGENTIAL_DISTORTION).getBx()
        elif k == P.DIST_TANGENTIAL_BY: up = cam.getDistortionModel(T.TANGENTIAL_DISTORTION).getBy()
        elif k == P.DIST_TANGENTIAL_BI: up = cam.getDistortionModel(T.TANGENTIAL_DISTORTION).get(order)
        elif k == P.DIST_RADIAL_AI: up = cam.getDistortionModel(T.RADIAL_DISTORTION).get(order)
        elif k == P.DIST_DISTANCE_DI: up = cam.getDistortionModel(T.DISTANCE_DISTORTION).get(order)
        elif k == P.DIST_ZERNIKE_X: up = cam.getDistortionModel(T.ZERNIKE_X).get(order)
        elif k == P.DIST_ZERNIKE_Y: up = cam.getDistortionModel(T.ZERNIKE_Y).get(order)
        else: up = cam.getDistortionModel(T.ZERNIKE_GRADIENT).get(order)
        out.append(up.getValue())
    eo_t = (PT.CAMERA_COORDINATE_X, PT.CAMERA_COORDINATE_Y, PT.CAMERA_COORDINATE_Z, PT.CAMERA_OMEGA, PT.CAMERA_PHI, PT.CAMERA_KAPPA)
    for im in images:
        eo = im.getExteriorOrientation()
        out += [eo.get(t).getValue() for t in eo_t]
    return np.array(out)
