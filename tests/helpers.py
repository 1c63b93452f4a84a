"""Shared test helpers (host side only)."""
import json
import os

import numpy as np

from bundle_adjustment_amd.problem import FlatProblem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden_rows(name="jacobian_rows.json"):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)["sets"]


def problem_from_cases(dist, cases, sigma=5e-4, rho=0.0):
    """One camera, one image and one object point PER case, everything free, d = 0 -- a carrier for row parity."""
    n = len(cases)
    nd = len(dist)
    P, I = n, n
    col = 0
    point_col = np.arange(3 * P).reshape(P, 3); col += 3 * P
    io_col = np.zeros((n, 3), np.int32); dist_col = np.zeros(n * nd, np.int32)
    for c in range(n):
        io_col[c] = col + np.arange(3); col += 3
        dist_col[c * nd:(c + 1) * nd] = col + np.arange(nd); col += nd
    eo_col = (col + np.arange(6 * I)).reshape(I, 6); col += 6 * I
    values = np.concatenate([np.array([c["point"] for c in cases]).ravel(),
                             np.array([c["io"] for c in cases]).ravel(),
                             np.array([c["dist_values"] for c in cases]).ravel() if nd else np.zeros(0),
                             np.array([c["eo"] for c in cases]).ravel()])
    return FlatProblem(
        n_unknowns=col, rank_defect=0, datum_flags=0, point_col=point_col, point_datum=np.ones(P, np.uint8),
        io_col=io_col, cam_r0=np.array([c["r0"] for c in cases]), cam_dist_begin=np.arange(n + 1) * nd,
        dist_kind=np.tile(np.array([k for k, _ in dist], np.int32), n),
        dist_order=np.tile(np.array([o for _, o in dist], np.int32), n), dist_col=dist_col,
        image_camera=np.arange(n), eo_col=eo_col, ip_image=np.arange(n), ip_point=np.arange(n),
        ip_x=np.array([c["obs"][0] for c in cases]), ip_y=np.array([c["obs"][1] for c in cases]),
        ip_var_x=np.full(n, sigma ** 2), ip_var_y=np.full(n, sigma ** 2), ip_rho=np.full(n, rho), values=values,
        sigma2apriori=sigma ** 2).validate()


def rel_err(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
