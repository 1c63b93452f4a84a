"""Deterministic synthetic close-range scenes for the BASELINE.json configs.

One camera with the interior orientation of the bundled example (JAICOV/example/example.ior:1,5), object points in
a 2000 x 300 x 2000 mm box, stations on a hemisphere aimed at the centroid, observations = collinearity model +
distortion + N(0, sigma^2) noise.  Seed 20260515, counter-based Philox generator, fp64.
"""
from __future__ import annotations

import numpy as np

from . import numbering
from .problem import (DIST_AFFINITY_CX, DIST_AFFINITY_CY, DIST_DISTANCE_DI, DIST_RADIAL_AI, DIST_TANGENTIAL_BI,
                      DIST_TANGENTIAL_BX, DIST_TANGENTIAL_BY, FlatProblem)

SEED = 20260515
C_EX, X0_EX, Y0_EX, R0_EX = 28.78507, 0.01735, 0.05669, 13.488
SENSOR_W, SENSOR_H = 35.968, 23.979
SIGMA_IMG = 0.0005

# (kind, order, true value) in application order
DIST_RADIAL = [(DIST_RADIAL_AI, 1, -1.09607e-4), (DIST_RADIAL_AI, 2, 1.49566e-7), (DIST_RADIAL_AI, 3, -2.0e-11)]
DIST_FULL = [(DIST_AFFINITY_CX, 0, -7.00801e-5), (DIST_AFFINITY_CY, 0, -3.12627e-5),
             (DIST_TANGENTIAL_BX, 0, 5.79843e-6), (DIST_TANGENTIAL_BY, 0, -8.64454e-6), (DIST_TANGENTIAL_BI, 1, 1.0e-5),
             ] + DIST_RADIAL + [(DIST_DISTANCE_DI, 1, 1.0e-3), (DIST_DISTANCE_DI, 2, -2.0e-6), (DIST_DISTANCE_DI, 3, 1.0e-9)]


def rotation(omega, phi, kappa):
    """R(omega,phi,kappa), Luhmann Eq. 2.31 as coded in PartialDerivativeFactory.java:125-135 (vectorised)."""
    co, so, cp, sp, ck, sk = np.cos(omega), np.sin(omega), np.cos(phi), np.sin(phi), np.cos(kappa), np.sin(kappa)
    R = np.empty(np.shape(omega) + (3, 3))
    R[..., 0, 0] = cp * ck; R[..., 0, 1] = -cp * sk; R[..., 0, 2] = sp
    R[..., 1, 0] = co * sk + so * sp * ck; R[..., 1, 1] = co * ck - so * sp * sk; R[..., 1, 2] = -so * cp
    R[..., 2, 0] = so * sk - co * sp * ck; R[..., 2, 1] = so * ck + co * sp * sk; R[..., 2, 2] = co * cp
    return R


def angles_from_rotation(R):
    phi = np.arcsin(np.clip(R[..., 0, 2], -1, 1))
    omega = np.arctan2(-R[..., 1, 2], R[..., 2, 2])
    kappa = np.arctan2(-R[..., 0, 1], R[..., 0, 0])
    return omega, phi, kappa


def project(c, x0, y0, eo, xyz, r0, dist):
    """Model function x = x0 + xs + sum(dx), y = y0 + ys + sum(dy), vectorised over points.
    eo: (6,), xyz: (n,3), dist: list of (kind, order, value).  Returns x, y, N."""
    R = rotation(eo[3], eo[4], eo[5])
    dX = xyz - eo[:3]
    k = dX @ R                      # columns: kx, ky, N   (kx = r11 dX + r21 dY + r31 dZ)
    kx, ky, N = k[:, 0], k[:, 1], k[:, 2]
    xs, ys = -c * kx / N, -c * ky / N
    r2 = xs * xs + ys * ys
    dx = np.zeros_like(xs); dy = np.zeros_like(ys)
    val = {(kd, o): v for kd, o, v in dist}
    cx, cy = val.get((DIST_AFFINITY_CX, 0), 0.0), val.get((DIST_AFFINITY_CY, 0), 0.0)
    dx += cx * xs + cy * ys
    bx, by = val.get((DIST_TANGENTIAL_BX, 0), 0.0), val.get((DIST_TANGENTIAL_BY, 0), 0.0)
    S = np.ones_like(xs)
    for kd, o, v in dist:
        if kd == DIST_TANGENTIAL_BI:
            S = S + v * r2 ** o
    dx += (bx * (r2 + 2 * xs * xs) + by * 2 * xs * ys) * S
    dy += (by * (r2 + 2 * ys * ys) + bx * 2 * xs * ys) * S
    for kd, o, v in dist:
        if kd == DIST_RADIAL_AI:
            Ri = r2 ** o - (r0 * r0) ** o
            dx += xs * v * Ri; dy += ys * v * Ri
        elif kd == DIST_DISTANCE_DI:
            Ri = r2 ** o - (r0 * r0) ** o
            dx += xs * v * Ri / N; dy += ys * v * Ri / N
    return x0 + xs + dx, y0 + ys + dy, N


def make_scene(n_images, n_points, obs_per_image, *, dist=DIST_RADIAL, weights="diag", n_control=4,
               control_dense=False, scale_bar=False, seed=SEED, start_noise=1.0, datum_all=True, min_rays=3, layout="sphere"):
    """Builds a FlatProblem.

    weights: 'diag' (sigma 0.0005), '2x2' (rho ~ U(-0.5,0.5), PartialDerivativeFactory.java:313-319) or 'block'
             (one dense SPD dispersion per image, D = L L', L = diag(sigma) + 0.1 sigma N(0,1) strictly lower).
    layout: 'sphere' (stations on a hemisphere aimed at the centroid, every image sees most of the object and keeps a RANDOM
            subset of obs_per_image points: two points share ~10 % of their images) or 'strips' (a block flown in strips, as real
            photogrammetric blocks are: stations 600-900 mm above the object looking down at a grid of targets in boustrophedon order, an
            image keeps the obs_per_image visible points NEAREST to its principal point: neighbouring points share most of their images,
            and the first-seen numbering of BA:667-782 makes neighbouring points neighbouring columns).
    n_control control points as a DirectlyObservedParameterGroup (d = 0 when >= 3 points are observed in X,Y,Z);
    control_dense gives that group a dense dispersion.  scale_bar adds one ScaleBar (free network: d = 6 with
    n_control = 0).
    """
    rng = np.random.Generator(np.random.Philox(seed))
    P, I = n_points, n_images
    pts = np.stack([rng.uniform(-1000, 1000, P), rng.uniform(-150, 150, P), rng.uniform(-1000, 1000, P)], 1)
    centre = pts.mean(0)
    n_dist = len(dist)
    need_all = obs_per_image >= P
    eo = np.zeros((I, 6))
    ip_image, ip_point, ip_x, ip_y = [], [], [], []
    targets = None
    if layout == "strips":
        ns = max(1, int(round(np.sqrt(I * 0.8))))                 # strips along Z, I / ns stations each, boustrophedon
        per = int(np.ceil(I / ns))
        tx = -1000 + 2000 * (np.arange(ns) + 0.5) / ns
        targets = []
        for a in range(ns):
            tz = -1000 + 2000 * (np.arange(per) + 0.5) / per
            if a & 1:
                tz = tz[::-1]
            targets += [(tx[a] + rng.uniform(-20, 20), z + rng.uniform(-20, 20)) for z in tz]
        targets = np.array(targets[:I])
    elif layout != "sphere":
        raise ValueError(layout)
    for i in range(I):
        radius = rng.uniform(1500, 2500) * (2.2 if need_all else 1.0)
        aim = centre
        if targets is not None:
            # footprint at distance d: 1.25 d x 0.83 d; obs_per_image of P points uniform over 2000 x 2000 need d ~ sqrt(4e6 obs / (1.04 P))
            radius = 1.15 * np.sqrt(4e6 * obs_per_image / (1.04 * P))
            aim = np.array([targets[i, 0], centre[1], targets[i, 1]])
        for _attempt in range(40):
            # station on the hemisphere Y < 0 (as the example block), uniform roll
            az = rng.uniform(0, 2 * np.pi); el = rng.uniform(np.deg2rad(25), np.deg2rad(85))
            if targets is not None:
                el = rng.uniform(np.deg2rad(65), np.deg2rad(88))
            st = aim + radius * np.array([np.cos(el) * np.cos(az), -np.sin(el), np.cos(el) * np.sin(az)])
            r3 = (st - aim) / np.linalg.norm(st - aim)             # camera z axis points away from the scene
            up = np.array([0.0, 0.0, 1.0]) if abs(r3[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
            r1 = np.cross(up, r3); r1 /= np.linalg.norm(r1)
            r2 = np.cross(r3, r1)
            roll = rng.uniform(0, 2 * np.pi)
            a1 = np.cos(roll) * r1 + np.sin(roll) * r2
            a2 = -np.sin(roll) * r1 + np.cos(roll) * r2
            R = np.stack([a1, a2, r3], 1)
            om, ph, ka = angles_from_rotation(R)
            e = np.array([st[0], st[1], st[2], om, ph, ka])
            x, y, N = project(C_EX, X0_EX, Y0_EX, e, pts, R0_EX, dist)
            vis = np.flatnonzero((N < 0) & (np.abs(x) < SENSOR_W / 2) & (np.abs(y) < SENSOR_H / 2))
            if vis.size >= min(obs_per_image, P):
                break
            radius *= 1.1
        else:
            raise RuntimeError("could not place a station seeing enough points")
        eo[i] = e
        if targets is None:
            sel = np.sort(rng.choice(vis, size=min(obs_per_image, vis.size), replace=False))
        else:                                                     # the compact patch around the principal point
            near = np.argsort((x[vis] - X0_EX) ** 2 + (y[vis] - Y0_EX) ** 2)[:min(obs_per_image, vis.size)]
            sel = np.sort(vis[near])
        ip_image.append(np.full(sel.size, i)); ip_point.append(sel)
        ip_x.append(x[sel]); ip_y.append(y[sel])
    ip_image = np.concatenate(ip_image); ip_point = np.concatenate(ip_point)
    ip_x = np.concatenate(ip_x); ip_y = np.concatenate(ip_y)
    n_ip = ip_image.size

    # a point needs >= min_rays rays to be determinable: drop the observations of weaker points
    rays = np.bincount(ip_point, minlength=P)
    keep = rays[ip_point] >= min_rays
    ip_image, ip_point, ip_x, ip_y = ip_image[keep], ip_point[keep], ip_x[keep], ip_y[keep]
    n_ip = ip_image.size
    # keep only observed points (BA numbers points from observations); re-index compactly, keeping ids stable
    seen = np.zeros(P, bool); seen[ip_point] = True
    if not seen.all():
        remap = np.cumsum(seen) - 1
        pts = pts[seen]; ip_point = remap[ip_point]; P = pts.shape[0]

    # stochastic model + noise
    blk_ip_begin = np.zeros(1, np.int32); blk_disp_offset = np.zeros(0, np.int64); blk_disp = np.zeros(0)
    var_x = np.full(n_ip, SIGMA_IMG ** 2); var_y = np.full(n_ip, SIGMA_IMG ** 2); rho = np.zeros(n_ip)
    if weights == "diag":
        ip_x = ip_x + rng.normal(0, SIGMA_IMG, n_ip); ip_y = ip_y + rng.normal(0, SIGMA_IMG, n_ip)
    elif weights == "2x2":
        rho = rng.uniform(-0.5, 0.5, n_ip)
        e1 = rng.normal(0, 1, n_ip); e2 = rng.normal(0, 1, n_ip)
        ip_x = ip_x + SIGMA_IMG * e1
        ip_y = ip_y + SIGMA_IMG * (rho * e1 + np.sqrt(1 - rho * rho) * e2)
    elif weights == "block":
        counts = np.bincount(ip_image, minlength=I)
        blk_ip_begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        sizes = (2 * counts.astype(np.int64)) ** 2
        blk_disp_offset = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        blk_disp = np.empty(int(sizes.sum()))
        for i in range(I):
            m = 2 * int(counts[i])
            L = np.tril(rng.normal(0, 0.1 * SIGMA_IMG, (m, m)), -1)
            L[np.diag_indices(m)] = SIGMA_IMG
            D = L @ L.T
            blk_disp[blk_disp_offset[i]:blk_disp_offset[i] + m * m] = D.ravel()
            noise = L @ rng.normal(0, 1, m)
            b = blk_ip_begin[i]
            ip_x[b:b + m // 2] += noise[0::2]; ip_y[b:b + m // 2] += noise[1::2]
            dg = np.diag(D)
            var_x[b:b + m // 2] = dg[0::2]; var_y[b:b + m // 2] = dg[1::2]
    else:
        raise ValueError(weights)

    # truth + start values (slot vector)
    truth = np.concatenate([pts.ravel(), [X0_EX, Y0_EX, C_EX], [v for _, _, v in dist], eo.ravel()])
    start = truth.copy()
    s_io = 3 * P; s_dist = s_io + 3; s_eo = s_dist + n_dist
    start[:s_io] += rng.normal(0, 1.0 * start_noise, 3 * P)
    start[s_dist:s_eo] *= 0.9
    eo_noise = np.concatenate([rng.normal(0, 5.0 * start_noise, (I, 3)), rng.normal(0, 1e-3 * start_noise, (I, 3))], 1)
    start[s_eo:] += eo_noise.ravel()

    # control points -> directly observed group (ExampleFlatFiles.java:105-150 pattern)
    dg_row_begin = np.zeros(1, np.int32); dg_slot = np.zeros(0, np.int32); dg_obs = np.zeros(0); dg_var = np.zeros(0)
    dg_disp_offset = np.zeros(0, np.int64); dg_disp = np.zeros(0)
    if n_control > 0:
        cp = np.sort(rng.choice(P, size=n_control, replace=False))
        dg_slot = (3 * cp[:, None] + np.arange(3)[None, :]).ravel().astype(np.int32)
        m = dg_slot.size
        sig_cp = 0.05
        if control_dense:
            L = np.tril(rng.normal(0, 0.1 * sig_cp, (m, m)), -1); L[np.diag_indices(m)] = sig_cp
            D = L @ L.T
            dg_obs = truth[dg_slot] + L @ rng.normal(0, 1, m)
            dg_var = np.diag(D).copy(); dg_disp = D.ravel().copy(); dg_disp_offset = np.zeros(1, np.int64)
        else:
            dg_obs = truth[dg_slot] + rng.normal(0, sig_cp, m)
            dg_var = np.full(m, sig_cp ** 2); dg_disp_offset = np.full(1, -1, np.int64)
        dg_row_begin = np.array([0, m], np.int32)
    sb_a = np.zeros(0, np.int32); sb_b = np.zeros(0, np.int32); sb_len = np.zeros(0); sb_var = np.zeros(0)
    if scale_bar:
        a, b = 0, P - 1
        sb_a = np.array([a], np.int32); sb_b = np.array([b], np.int32)
        sb_len = np.array([np.linalg.norm(pts[a] - pts[b]) + rng.normal(0, 0.01)]); sb_var = np.array([0.01 ** 2])

    cam_dist_begin = np.array([0, n_dist], np.int32)
    num = numbering.number_unknowns(P, 1, np.zeros(I, np.int32), ip_point, cam_dist_begin,
                                    sb_point_a=sb_a, sb_point_b=sb_b, dg_slot=dg_slot)
    sigma2 = numbering.sigma2_apriori(var_x, var_y, sb_var, dg_var)
    fp = FlatProblem(
        n_unknowns=num["n_unknowns"], rank_defect=num["rank_defect"], datum_flags=num["datum_flags"],
        point_col=num["point_col"], point_datum=np.full(P, 1 if datum_all else 0, np.uint8),
        io_col=num["io_col"], cam_r0=np.array([R0_EX]), cam_dist_begin=cam_dist_begin,
        dist_kind=np.array([k for k, _, _ in dist], np.int32), dist_order=np.array([o for _, o, _ in dist], np.int32),
        dist_col=num["dist_col"], image_camera=np.zeros(I, np.int32), eo_col=num["eo_col"],
        ip_image=ip_image, ip_point=ip_point, ip_x=ip_x, ip_y=ip_y, ip_var_x=var_x, ip_var_y=var_y, ip_rho=rho,
        values=start, sigma2apriori=sigma2, blk_ip_begin=blk_ip_begin, blk_disp_offset=blk_disp_offset,
        blk_disp=blk_disp, sb_point_a=sb_a, sb_point_b=sb_b, sb_length=sb_len, sb_var=sb_var,
        dg_row_begin=dg_row_begin, dg_slot=dg_slot, dg_obs=dg_obs, dg_var=dg_var, dg_disp_offset=dg_disp_offset,
        dg_disp=dg_disp, truth=truth)
    return fp.validate()


# BASELINE.json configs ----------------------------------------------------------------------
def config(name: str, **kw) -> FlatProblem:
    if name == "cfg2":      # 20 images x 200 points, pinhole + radial A1-A3, diagonal W
        return make_scene(20, 200, 200, dist=DIST_RADIAL, weights="diag", n_control=4, **kw)
    if name == "cfg3":      # 100 x 1000, full interior set, 2x2 blocks
        return make_scene(100, 1000, 400, dist=DIST_FULL, weights="2x2", n_control=6, **kw)
    if name == "cfg3_block":   # config 3's size with config 4's dense per-image dispersions (U = 3 614): the EO-reduced dataflow path at a mid size
        return make_scene(100, 1000, 400, dist=DIST_FULL, weights="block", n_control=15, control_dense=True, **kw)
    if name in ("cfg4", "cfg5"):   # 500 x 5000, dense dispersion per image + dense 45x45 control block
        return make_scene(500, 5000, 500, dist=DIST_FULL, weights="block", n_control=15, control_dense=True, **kw)
    if name == "cfg4_local":   # config 4's size and stochastic model on a block flown in strips: neighbouring points share most of their images
        return make_scene(500, 5000, 500, dist=DIST_FULL, weights="block", n_control=15, control_dense=True, layout="strips", **kw)
    if name == "tiny":      # parity-test size: oracle finishes in milliseconds
        return make_scene(6, 40, 24, dist=DIST_FULL, weights="2x2", n_control=4, **kw)
    if name == "tiny_block":
        return make_scene(5, 36, 20, dist=DIST_FULL, weights="block", n_control=5, control_dense=True, **kw)
    if name == "tiny_free":  # free network with scale bar: d = 6 (datum border, indefinite system)
        return make_scene(6, 40, 24, dist=DIST_RADIAL, weights="diag", n_control=0, scale_bar=True, **kw)
    raise KeyError(name)
