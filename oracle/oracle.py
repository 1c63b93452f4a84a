"""ctypes front-end of the CPU oracle (``ba_oracle.c``).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; nothing
under ``bundle-adjustment_amd/`` does.  See the header of ``ba_oracle.c`` for the pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from bundle_adjustment_amd import problem as _problem  # noqa: E402  (struct layout only)

_LIB = None
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


class OracleResult(C.Structure):
    _fields_ = [("state", C.c_int32), ("iterations", C.c_int32), ("omega", C.c_double), ("max_abs_dx", C.c_double),
                ("final_lambda", C.c_double), ("seconds_total", C.c_double), ("seconds_last_pass", C.c_double)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libba_oracle.so")
    src = os.path.join(_HERE, "ba_oracle.c")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "ba_exact.c")))
    if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libba_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        P = C.POINTER(_problem.ProblemDesc)
        L.oracle_eps.restype = C.c_double
        L.oracle_num_slots.argtypes = [P]
        L.oracle_rows.argtypes = [P, _pd, C.c_double, C.c_int, _pd, _pd, C.c_int, _pd]
        L.oracle_accumulate.argtypes = [P, _pd, C.c_double, C.c_int, C.c_int, C.c_int, _pd, _pd]
        L.oracle_finalize.argtypes = [P, _pd, C.c_double, C.c_int, _pd, _pd, _pd]
        L.oracle_build.argtypes = [P, _pd, C.c_double, C.c_double, C.c_int, _pd, _pd, _pd]
        L.oracle_precondition.argtypes = [C.c_int, _pd, _pd, _pd]
        L.oracle_precondition.restype = None
        L.oracle_solve.argtypes = [C.c_int, _pd, _pd, C.c_int]
        L.oracle_dsptrf.argtypes = [C.c_int, _pd, _pi]
        L.oracle_dsptrs.argtypes = [C.c_int, _pd, _pi, _pd]
        L.oracle_dsptrs.restype = None
        L.oracle_dsptri.argtypes = [C.c_int, _pd, _pi, _pd]
        L.oracle_dpptrf.argtypes = [C.c_int, _pd]
        L.oracle_dpptri.argtypes = [C.c_int, _pd]
        L.oracle_dispersion_to_weight.argtypes = [C.c_int, _pd, C.c_double, _pd]
        L.oracle_omega.argtypes = [P, _pd, C.c_double, _pd, _pd]
        L.oracle_update.argtypes = [P, _pd, _pd]
        L.oracle_update.restype = C.c_double
        L.oracle_estimate.argtypes = [P, _pd, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _pd,
                                      C.POINTER(OracleResult)]
        L.oracle_residual_ld.argtypes = [C.c_int, _pd, _pd, _pd, _pd]
        L.oracle_residual_ld.restype = None
        L.oracle_centroid.argtypes = [P, _pd, _pd, C.c_int, _pd]
        L.oracle_reduce.argtypes = [P, _pd, _pd, C.c_int]
        L.oracle_extract_reduced.argtypes = [P, _pd, _pd]
        L.oracle_extract_reduced.restype = None
        L.oracle_reduced_rows.argtypes = [P]
        L.oracle_faithful_image_points.argtypes = [P, _pd, C.c_double, C.c_int, C.c_int, _pd, _pd]
        L.oracle_faithful_image_points.restype = C.c_double
        L.oracle_block_weight.argtypes = [P, C.c_double, C.c_int, _pd]
        L.oracle_block_fair.argtypes = [P, _pd, C.c_double, C.c_int, _pd, _pd, _pd]
        # ba_exact.c: extended-precision ground truth (NOT the reference's arithmetic; accuracy study only)
        L.oracle_exact_accumulate.argtypes = [P, _pd, C.c_double, _pd, _pd, _pd, _pd]
        L.oracle_exact_block_weight.argtypes = [P, C.c_double, C.c_int, _pd, _pd]
        L.oracle_inverse_residual_q.argtypes = [C.c_int, _pd, _pd, _pd, C.c_double, C.c_int, C.c_int]
        L.oracle_inverse_residual_q.restype = C.c_double
        L.oracle_inverse_residual_probe_q.argtypes = [C.c_int, _pd, _pd, _pd, C.c_double, C.c_int, C.c_ulonglong]
        L.oracle_inverse_residual_probe_q.restype = C.c_double
        L.oracle_residual_ld2.argtypes = [C.c_int, _pd, _pd, C.c_int, _pd, _pd, _pd, _pd]
        L.oracle_residual_ld2.restype = None
        L.oracle_matvec_ld2.argtypes = [C.c_int, _pd, _pd, _pd, _pd]
        L.oracle_matvec_ld2.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_pd) if a is not None else C.cast(None, _pd)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


KLOC = 12 + 20  # local row layout shared with jaicov_neq_get_rows (12 + JAICOV_MAX_DIST_PER_CAMERA)


class Oracle:
    """Oracle bound to one FlatProblem."""

    def __init__(self, fp):
        self.fp = fp
        self.desc, self._keep = fp.as_desc()
        self.L = lib()
        self.U = fp.n_unknowns

    def rows(self, values, ip, sigma2=None):
        v = _f(values); w = np.zeros(2); A = np.zeros((2, KLOC)); P = np.zeros(4)
        diag = self.L.oracle_rows(C.byref(self.desc), _p(v), self.fp.sigma2apriori if sigma2 is None else sigma2,
                                  int(ip), _p(w), _p(A), KLOC, _p(P))
        return w, A, P.reshape(2, 2), bool(diag)

    def accumulate(self, values, sigma2, image_begin=0, image_end=None, shared=True):
        v = _f(values)
        N = np.zeros(self.fp.packed_length); n = np.zeros(self.U)
        ie = self.fp.n_images if image_end is None else image_end
        info = self.L.oracle_accumulate(C.byref(self.desc), _p(v), sigma2, image_begin, ie, int(shared), _p(N), _p(n))
        if info:
            raise ArithmeticError(f"oracle_accumulate info={info}")
        return N, n

    def finalize(self, values, N, n, lam=0.0, simulation=False):
        v = _f(values); V = np.zeros(self.U)
        info = self.L.oracle_finalize(C.byref(self.desc), _p(v), lam, int(simulation), _p(N), _p(n), _p(V))
        if info:
            raise ArithmeticError(f"oracle_finalize info={info}")
        return V

    def build(self, values, sigma2, lam=0.0, simulation=False):
        N, n = self.accumulate(values, sigma2)
        V = self.finalize(values, N, n, lam, simulation)
        return N, n, V

    def precondition(self, V, M, m):
        self.L.oracle_precondition(self.U, _p(V), _p(M), _p(m))

    def solve(self, N, n, invert):
        """In place: n <- x, N <- factor or inverse (MathExtension.java:338-366).  Returns LAPACK info."""
        return self.L.oracle_solve(self.U, _p(N), _p(n), int(invert))

    def step(self, values, sigma2, lam=0.0, invert=False):
        """One loop body: build, precondition, solve, un-precondition.  Returns dx, Qxx (packed or None), N, n."""
        N, n, V = self.build(values, sigma2, lam)
        N0, n0 = N.copy(), n.copy()
        self.precondition(V, N, n)
        info = self.solve(N, n, invert)
        if info:
            raise ArithmeticError(f"singular, info={info}")
        self.precondition(V, N if invert else None, n)
        return n, (N if invert else None), N0, n0

    def omega(self, values, sigma2, dx):
        v = _f(values); dx = _f(dx); out = C.c_double(0)
        om = np.zeros(1)
        info = self.L.oracle_omega(C.byref(self.desc), _p(v), sigma2, _p(dx), _p(om))
        del out
        if info:
            raise ArithmeticError(f"oracle_omega info={info}")
        return float(om[0])

    def update(self, values, dx):
        v = _f(values).copy(); dx = _f(dx)
        mx = self.L.oracle_update(C.byref(self.desc), _p(v), _p(dx))
        return v, float(mx)

    def centroid(self, values, dg_obs, invert=False, centroid=None):
        """BundleAdjustment.centroidCoordinates (BA:115-201).  Returns (values, dg_obs, centroid) -- copies."""
        v = _f(values).copy(); o = _f(dg_obs).copy()
        c = np.zeros(3) if centroid is None else _f(centroid).copy()
        rc = self.L.oracle_centroid(C.byref(self.desc), _p(v), _p(o) if o.size else C.cast(None, _pd), int(invert), _p(c))
        if rc:
            raise ArithmeticError(f"oracle_centroid: the numbers of coordinate components are un-equal or zero (rc={rc})")
        return v, o, c

    def reduce(self, N, n, pre_elimination=False):
        """reduceNormalEquationSystem (BA:1197-1342), in place on the preconditioned system."""
        info = self.L.oracle_reduce(C.byref(self.desc), _p(N), _p(n), int(pre_elimination))
        if info:
            raise ArithmeticError(f"oracle_reduce info={info}")

    def extract_reduced(self, N, n):
        self.L.oracle_extract_reduced(C.byref(self.desc), _p(N), _p(n))

    def reduced_rows(self):
        return int(self.L.oracle_reduced_rows(C.byref(self.desc)))

    def estimate(self, values=None, sigma2=None, lam0=0.0, max_iter=5000, invert=True, simulation=False):
        v = _f(self.fp.values if values is None else values).copy()
        Q = np.zeros(self.fp.packed_length) if invert else None
        res = OracleResult()
        self.L.oracle_estimate(C.byref(self.desc), _p(v), self.fp.sigma2apriori if sigma2 is None else sigma2, lam0,
                               max_iter, int(invert), int(simulation), _p(Q), C.byref(res))
        return v, Q, res

    def block_weight(self, sigma2, blk):
        m = 2 * int(self.fp.blk_ip_begin[blk + 1] - self.fp.blk_ip_begin[blk])
        Pm = np.zeros((m, m))
        info = self.L.oracle_block_weight(C.byref(self.desc), sigma2, blk, _p(Pm))
        if info:
            raise ArithmeticError(f"oracle_block_weight info={info}")
        return Pm

    def block_fair(self, values, sigma2, blk, Pm, N, n):
        v = _f(values)
        return self.L.oracle_block_fair(C.byref(self.desc), _p(v), sigma2, blk, _p(Pm), _p(N), _p(n))

    # ---- extended-precision ground truth (ba_exact.c) ----------------------------------------------------------
    def exact_accumulate(self, values, sigma2):
        """N, n of all observation groups with P = sigma0^2 inv(D) and every sum in x87 extended precision, each rounded once
        to a (hi, lo) pair: returns N_hi, N_lo, n_hi, n_lo (packed 'U')."""
        v = _f(values)
        Nh = np.zeros(self.fp.packed_length); Nl = np.zeros(self.fp.packed_length)
        nh = np.zeros(self.U); nl = np.zeros(self.U)
        info = self.L.oracle_exact_accumulate(C.byref(self.desc), _p(v), sigma2, _p(Nh), _p(Nl), _p(nh), _p(nl))
        if info:
            raise ArithmeticError(f"oracle_exact_accumulate info={info}")
        return Nh, Nl, nh, nl

    def exact_block_weight(self, sigma2, blk):
        m = 2 * int(self.fp.blk_ip_begin[blk + 1] - self.fp.blk_ip_begin[blk])
        Ph = np.zeros((m, m)); Pl = np.zeros((m, m))
        info = self.L.oracle_exact_block_weight(C.byref(self.desc), sigma2, blk, _p(Ph), _p(Pl))
        if info:
            raise ArithmeticError(f"oracle_exact_block_weight info={info}")
        return Ph, Pl

    def faithful_image_points(self, values, sigma2, ip_begin, ip_end, N, n):
        v = _f(values)
        return self.L.oracle_faithful_image_points(C.byref(self.desc), _p(v), sigma2, ip_begin, ip_end, _p(N), _p(n))
