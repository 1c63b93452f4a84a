"""Index contract of the adjustment (integer work, bit-exact with the reference).

Host-side restatement of ``BundleAdjustment.prepareUnknownParameters`` (BundleAdjustment.java:667-782),
``addUnknownParameter`` (BA:645-650) and ``detectRankDefect`` (BA:836-1042) for flat inputs, used by the synthetic scene
generator and the file loaders.  The C++ object-model mirror (``host/jaicov.hpp``) carries the literal, loop-by-loop
version of the same contract; ``tests/test_numbering.py`` checks the two against each other and against the
known answers of the bundled example.
"""
from __future__ import annotations

import numpy as np

from .problem import (COL_FIXED, DATUM_RX, DATUM_RY, DATUM_RZ, DATUM_SCALE, DATUM_TX, DATUM_TY, DATUM_TZ)


def detect_rank_defect(has_scale_bars: bool, direct_kinds, n_fixed_xyz, any_angle_fixed):
    """Closed form of BA:836-1042.

    direct_kinds : iterable of strings in {'X','Y','Z','omega','phi','kappa',None}, one per directly observed
                   parameter (object AND camera coordinates count alike, BA:887-898).
    n_fixed_xyz  : (nx, ny, nz) number of FIXED object + camera-station coordinate components (BA:946-950, 1001-1003).
    any_angle_fixed : (omega, phi, kappa) booleans: some image has that angle fixed (BA:994-999).

    All conditions in the reference are monotone in the running counts and flags only move FREE -> FIXED, so the
    staged loops with their early exits reduce to the conditions evaluated on the final counts.
    """
    kx = ky = kz = 0
    rx = ry = rz = False  # True = FIXED
    for k in direct_kinds:
        if k == 'X':
            kx += 1
        elif k == 'Y':
            ky += 1
        elif k == 'Z':
            kz += 1
        elif k == 'omega':
            rx = True
        elif k == 'phi':
            ry = True
        elif k == 'kappa':
            rz = True
    kx += int(n_fixed_xyz[0]); ky += int(n_fixed_xyz[1]); kz += int(n_fixed_xyz[2])
    rx = rx or bool(any_angle_fixed[0])
    ry = ry or bool(any_angle_fixed[1])
    rz = rz or bool(any_angle_fixed[2])
    tx, ty, tz = kx > 0, ky > 0, kz > 0
    scale = has_scale_bars or (kx >= 2 or ky >= 2 or kz >= 2)
    rx = rx or (ky >= 2 and kz >= 2)
    ry = ry or (kx >= 2 and kz >= 2)
    rz = rz or (kx >= 2 and ky >= 2)
    if kx > 0 and ky > 0 and kz > 0 and (kx + ky + kz >= (6 if has_scale_bars else 7)):
        rx = ry = rz = True
    flags = 0
    for fixed, bit in ((tx, DATUM_TX), (ty, DATUM_TY), (tz, DATUM_TZ), (rx, DATUM_RX), (ry, DATUM_RY),
                       (rz, DATUM_RZ), (scale, DATUM_SCALE)):
        if not fixed:
            flags |= bit
    return flags


def number_unknowns(n_points, n_cameras, image_camera, ip_point, cam_dist_begin, *, point_fixed=None,
                    io_fixed=None, dist_fixed=None, eo_fixed=None, sb_point_a=(), sb_point_b=(), dg_slot=()):
    """Assigns columns exactly as BA:667-782 does.

    Order: object points in first-seen order over the image points (image-major), then per camera x0,y0,c and its
    distortion coefficients, then the six EO parameters per image, then points that occur only in scale bars, then
    parameters referenced only by directly observed groups; fixed parameters own no column; every column += d.
    Images must be grouped camera-major (``image_camera`` non-decreasing), image points image-major.
    Returns dict(point_col, io_col, dist_col, eo_col, n_unknowns, rank_defect, datum_flags).
    """
    image_camera = np.asarray(image_camera, np.int64)
    n_images = image_camera.shape[0]
    assert np.all(np.diff(image_camera) >= 0), "images must be grouped by camera"
    ip_point = np.asarray(ip_point, np.int64)
    n_dist = int(cam_dist_begin[-1])
    point_fixed = np.zeros((n_points, 3), bool) if point_fixed is None else np.asarray(point_fixed, bool).reshape(n_points, 3)
    io_fixed = np.zeros((n_cameras, 3), bool) if io_fixed is None else np.asarray(io_fixed, bool).reshape(n_cameras, 3)
    dist_fixed = np.zeros(n_dist, bool) if dist_fixed is None else np.asarray(dist_fixed, bool)
    eo_fixed = np.zeros((n_images, 6), bool) if eo_fixed is None else np.asarray(eo_fixed, bool).reshape(n_images, 6)

    point_col = np.full((n_points, 3), -2, np.int64)   # -2 = unseen (reference: -1 "not set")
    nxt = 0

    def assign_points(order):
        nonlocal nxt
        # first occurrence order
        _, first = np.unique(order, return_index=True)
        for p in order[np.sort(first)]:
            if point_col[p, 0] != -2:
                continue
            for a in range(3):
                if point_fixed[p, a]:
                    point_col[p, a] = COL_FIXED
                else:
                    point_col[p, a] = nxt
                    nxt += 1

    if ip_point.size:
        assign_points(ip_point)
    io_col = np.full((n_cameras, 3), COL_FIXED, np.int64)
    dist_col = np.full(n_dist, COL_FIXED, np.int64)
    for c in range(n_cameras):
        for a in range(3):
            if not io_fixed[c, a]:
                io_col[c, a] = nxt
                nxt += 1
        for j in range(int(cam_dist_begin[c]), int(cam_dist_begin[c + 1])):
            if not dist_fixed[j]:
                dist_col[j] = nxt
                nxt += 1
    eo_col = np.full((n_images, 6), COL_FIXED, np.int64)
    free = ~eo_fixed
    cnt = int(free.sum())
    eo_col[free] = nxt + np.arange(cnt)       # row-major == image-major, X0,Y0,Z0,omega,phi,kappa
    nxt += cnt
    sb = np.stack([np.asarray(sb_point_a, np.int64), np.asarray(sb_point_b, np.int64)], 1).reshape(-1) \
        if len(sb_point_a) else np.zeros(0, np.int64)
    if sb.size:
        assign_points(sb)
    # directly observed references (BA:747-771): only object points can still be un-numbered here
    dg_slot = np.asarray(dg_slot, np.int64)
    if dg_slot.size:
        pts = dg_slot[dg_slot < 3 * n_points] // 3
        if pts.size:
            assign_points(pts)
    assert np.all(point_col != -2), "object point without any observation"

    # rank defect (BA:773)
    kinds = []
    s_io = 3 * n_points
    s_eo = s_io + 3 * n_cameras + n_dist
    for s in dg_slot:
        if s < s_io:
            kinds.append('XYZ'[s % 3])
        elif s >= s_eo:
            kinds.append(('X', 'Y', 'Z', 'omega', 'phi', 'kappa')[(s - s_eo) % 6])
        else:
            kinds.append(None)
    n_fixed = point_fixed.sum(0) + eo_fixed[:, :3].sum(0)
    flags = detect_rank_defect(len(sb_point_a) > 0, kinds, n_fixed, eo_fixed[:, 3:].any(0))
    d = bin(flags).count("1")
    for arr in (point_col, io_col, dist_col, eo_col):
        arr[arr >= 0] += d                     # BA:776-781
    return dict(point_col=point_col.astype(np.int32), io_col=io_col.astype(np.int32),
                dist_col=dist_col.astype(np.int32), eo_col=eo_col.astype(np.int32),
                n_unknowns=nxt + d, rank_defect=d, datum_flags=flags)


def sigma2_apriori(*variance_arrays) -> float:
    """BA:98,641: sigma0^2 = min(1, min over all observation variances)."""
    s = 1.0
    for v in variance_arrays:
        v = np.asarray(v)
        if v.size:
            s = min(s, float(v.min()))
    return s
