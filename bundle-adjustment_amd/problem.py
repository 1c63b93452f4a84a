"""Flat (struct-of-arrays) description of one adjustment and its ctypes image of ``jaicov_problem_desc``.

This is host-side plumbing for the C ABI in ``include/jaicov_neq.h``: what a JNI shim would assemble from the Java
object graph after ``BundleAdjustment.prepareUnknownParameters()`` (BundleAdjustment.java:667-782) has run.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

COL_FIXED = -1

# jaicov_dist_kind (DistortionModel.Type application order, DistortionModel.java:29-37)
DIST_AFFINITY_CX, DIST_AFFINITY_CY, DIST_TANGENTIAL_BX, DIST_TANGENTIAL_BY, DIST_TANGENTIAL_BI, DIST_RADIAL_AI, \
    DIST_DISTANCE_DI = range(7)
DIST_ZERNIKE_X, DIST_ZERNIKE_Y, DIST_ZERNIKE_Z = 7, 8, 9   # ZernikeDistortionModel.X / .Y / .Gradient; order = Zernike index >= 1

DATUM_TX, DATUM_TY, DATUM_TZ, DATUM_RX, DATUM_RY, DATUM_RZ, DATUM_SCALE = 1, 2, 4, 8, 16, 32, 64

_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)
_pu8 = C.POINTER(C.c_uint8)
_pf64 = C.POINTER(C.c_double)


class ProblemDesc(C.Structure):
    """ctypes mirror of ``jaicov_problem_desc`` (field order is the ABI)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_unknowns", C.c_int32), ("rank_defect", C.c_int32),
        ("datum_flags", C.c_int32),
        ("n_points", C.c_int32), ("n_cameras", C.c_int32), ("n_images", C.c_int32), ("n_dist", C.c_int32),
        ("n_image_points", C.c_int32), ("n_image_blocks", C.c_int32), ("n_scale_bars", C.c_int32),
        ("n_direct_groups", C.c_int32), ("n_direct_rows", C.c_int32),
        ("point_col", _pi32), ("point_datum", _pu8),
        ("io_col", _pi32), ("cam_r0", _pf64), ("cam_dist_begin", _pi32),
        ("dist_kind", _pi32), ("dist_order", _pi32), ("dist_col", _pi32),
        ("image_camera", _pi32), ("eo_col", _pi32),
        ("ip_image", _pi32), ("ip_point", _pi32), ("ip_x", _pf64), ("ip_y", _pf64),
        ("ip_var_x", _pf64), ("ip_var_y", _pf64), ("ip_rho", _pf64),
        ("blk_ip_begin", _pi32), ("blk_disp_offset", _pi64), ("blk_disp", _pf64),
        ("sb_point_a", _pi32), ("sb_point_b", _pi32), ("sb_length", _pf64), ("sb_var", _pf64),
        ("dg_row_begin", _pi32), ("dg_slot", _pi32), ("dg_obs", _pf64), ("dg_var", _pf64),
        ("dg_disp_offset", _pi64), ("dg_disp", _pf64),
    ]


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
    if shape is not None:
        a = a.reshape(shape)
    return a


@dataclass(repr=False)
class FlatProblem:
    """All arrays are numpy, C-contiguous; see ``include/jaicov_neq.h`` for the meaning of every field."""
    n_unknowns: int
    rank_defect: int
    datum_flags: int
    point_col: np.ndarray            # (P,3) int32
    point_datum: np.ndarray          # (P,) uint8
    io_col: np.ndarray               # (C,3) int32
    cam_r0: np.ndarray               # (C,) f64
    cam_dist_begin: np.ndarray       # (C+1,) int32
    dist_kind: np.ndarray            # (nd,) int32
    dist_order: np.ndarray           # (nd,) int32
    dist_col: np.ndarray             # (nd,) int32
    image_camera: np.ndarray         # (I,) int32
    eo_col: np.ndarray               # (I,6) int32
    ip_image: np.ndarray
    ip_point: np.ndarray
    ip_x: np.ndarray
    ip_y: np.ndarray
    ip_var_x: np.ndarray
    ip_var_y: np.ndarray
    ip_rho: np.ndarray
    values: np.ndarray               # slot vector (initial parameter values)
    sigma2apriori: float = 1.0
    blk_ip_begin: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    blk_disp_offset: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    blk_disp: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    sb_point_a: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    sb_point_b: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    sb_length: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    sb_var: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    dg_row_begin: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    dg_slot: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    dg_obs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    dg_var: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    dg_disp_offset: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    dg_disp: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    n_observations: int = 0          # rows (BA:1056)
    names: Optional[List[str]] = None  # point names (ObjectCoordinate.getName)
    truth: Optional[np.ndarray] = None  # synthetic scenes: true slot vector

    def __post_init__(self):
        i32, f64 = np.int32, np.float64
        self.point_col = _arr(self.point_col, i32, (-1, 3))
        self.point_datum = _arr(self.point_datum, np.uint8)
        self.io_col = _arr(self.io_col, i32, (-1, 3))
        self.cam_r0 = _arr(self.cam_r0, f64)
        self.cam_dist_begin = _arr(self.cam_dist_begin, i32)
        self.dist_kind = _arr(self.dist_kind, i32)
        self.dist_order = _arr(self.dist_order, i32)
        self.dist_col = _arr(self.dist_col, i32)
        self.image_camera = _arr(self.image_camera, i32)
        self.eo_col = _arr(self.eo_col, i32, (-1, 6))
        for n in ("ip_image", "ip_point", "blk_ip_begin", "sb_point_a", "sb_point_b", "dg_row_begin", "dg_slot"):
            setattr(self, n, _arr(getattr(self, n), i32))
        for n in ("ip_x", "ip_y", "ip_var_x", "ip_var_y", "ip_rho", "values", "blk_disp", "sb_length", "sb_var",
                  "dg_obs", "dg_var", "dg_disp"):
            setattr(self, n, _arr(getattr(self, n), f64))
        self.blk_disp_offset = _arr(self.blk_disp_offset, np.int64)
        self.dg_disp_offset = _arr(self.dg_disp_offset, np.int64)
        if self.n_observations == 0:
            self.n_observations = 2 * self.n_image_points + self.n_scale_bars + self.n_direct_rows

    # sizes -------------------------------------------------------------------------------------------------
    n_points = property(lambda s: s.point_col.shape[0])
    n_cameras = property(lambda s: s.io_col.shape[0])
    n_images = property(lambda s: s.image_camera.shape[0])
    n_dist = property(lambda s: s.dist_kind.shape[0])
    n_image_points = property(lambda s: s.ip_image.shape[0])
    n_image_blocks = property(lambda s: s.blk_ip_begin.shape[0] - 1)
    n_scale_bars = property(lambda s: s.sb_point_a.shape[0])
    n_direct_groups = property(lambda s: s.dg_row_begin.shape[0] - 1)
    n_direct_rows = property(lambda s: s.dg_slot.shape[0])
    n_slots = property(lambda s: 3 * s.n_points + 3 * s.n_cameras + s.n_dist + 6 * s.n_images)
    packed_length = property(lambda s: s.n_unknowns * (s.n_unknowns + 1) // 2)
    degree_of_freedom = property(lambda s: s.n_observations - (s.n_unknowns - s.rank_defect) + s.rank_defect)

    # slot layout --------------------------------------------------------------------------------------------
    def slot_point(self, p): return 3 * p
    def slot_io(self, c): return 3 * self.n_points + 3 * c
    def slot_dist(self, j): return 3 * self.n_points + 3 * self.n_cameras + j
    def slot_eo(self, i): return 3 * self.n_points + 3 * self.n_cameras + self.n_dist + 6 * i

    def slot_columns(self) -> np.ndarray:
        return np.concatenate([self.point_col.ravel(), self.io_col.ravel(), self.dist_col, self.eo_col.ravel()]
                              ).astype(np.int32)

    def validate(self):
        U = self.n_unknowns
        assert self.values.shape[0] == self.n_slots, (self.values.shape, self.n_slots)
        cols = self.slot_columns()
        free = cols[cols >= 0]
        assert free.size == U - self.rank_defect, (free.size, U, self.rank_defect)
        assert np.array_equal(np.sort(free), np.arange(self.rank_defect, U)), "columns must be a permutation"
        assert bin(self.datum_flags).count("1") == self.rank_defect
        assert self.cam_dist_begin[0] == 0 and self.cam_dist_begin[-1] == self.n_dist
        if self.n_image_points:
            assert np.all(np.diff(self.ip_image) >= 0), "image points must be image-major"
        assert np.all(np.abs(self.ip_rho) < 1)
        return self

    def as_desc(self):
        """Returns (ProblemDesc, keepalive) -- keep ``keepalive`` referenced while the desc is in use."""
        d = ProblemDesc()
        d.struct_size = C.sizeof(ProblemDesc)
        for n in ("n_unknowns", "rank_defect", "datum_flags", "n_points", "n_cameras", "n_images", "n_dist",
                  "n_image_points", "n_image_blocks", "n_scale_bars", "n_direct_groups", "n_direct_rows"):
            setattr(d, n, int(getattr(self, n)))
        keep = []
        for name, ptype in ProblemDesc._fields_[13:]:
            a = getattr(self, name)
            if name in ("point_col", "io_col", "eo_col"):
                a = a.reshape(-1)
            keep.append(a)
            setattr(d, name, a.ctypes.data_as(ptype) if a.size else C.cast(None, ptype))
        return d, keep


def packed_to_full(ap: np.ndarray, n: int) -> np.ndarray:
    """UPLO='U' column-major packed (MTJ UpperSymmPackMatrix) -> full symmetric (n,n)."""
    full = np.zeros((n, n))
    iu = np.triu_indices(n)
    # column-major packed upper == row-major packed lower: fill lower by rows
    il = np.tril_indices(n)
    full[il] = ap
    full = full + full.T - np.diag(np.diag(full))
    del iu
    return full


def full_to_packed(full: np.ndarray) -> np.ndarray:
    n = full.shape[0]
    return np.ascontiguousarray(full[np.tril_indices(n)])
