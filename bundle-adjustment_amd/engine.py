"""ctypes binding of the C ABI in ``include/jaicov_neq.h`` / ``include/jaicov_dense.h`` (``csrc/libjaicov_neq.so``).

This is the Python image of the stub a JNI shim would hold.  There is no CPU path: loading fails loudly when the HIP
library has not been built, and every call fails with ``EngineError`` when no gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .problem import FlatProblem, ProblemDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libjaicov_neq.so")

STATUS = {0: "OK", -1: "BAD_ARGUMENT", -2: "BAD_STATE", -3: "UNSUPPORTED", -4: "OUT_OF_MEMORY", -5: "DEVICE",
          -6: "NO_DEVICE", 1: "SINGULAR", 2: "NOT_FINITE"}

EXPORTS = [
    "jaicov_neq_create", "jaicov_neq_destroy", "jaicov_neq_last_error", "jaicov_neq_abi_version",
    "jaicov_neq_num_slots", "jaicov_neq_packed_length", "jaicov_neq_set_parameters", "jaicov_neq_get_parameters",
    "jaicov_neq_build", "jaicov_neq_accumulate", "jaicov_neq_accumulate2", "jaicov_neq_prepare_inverse", "jaicov_neq_reduced_order", "jaicov_neq_cofactor_order",
    "jaicov_neq_finalize", "jaicov_neq_reduce_buffer", "jaicov_neq_reduce_buffer_async",
    "jaicov_neq_solve", "jaicov_neq_omega", "jaicov_neq_update", "jaicov_neq_get_normal", "jaicov_neq_get_cofactor",
    "jaicov_neq_get_cofactor_sub", "jaicov_neq_get_dispersion_sub", "jaicov_neq_get_rows", "jaicov_neq_estimate", "jaicov_neq_last_timings",
    "jaicov_neq_set_profiling", "jaicov_neq_kernel_stats", "jaicov_neq_cancel",
    "jaicov_dense_spd_solve_packed", "jaicov_dense_gemm", "jaicov_neq_eo_step_buffer",
    "jaicov_neq_create_timings", "jaicov_neq_get_block_weight", "jaicov_neq_expansion_buffer",
]

KROW = 32  # 12 + JAICOV_MAX_DIST_PER_CAMERA
INVERT_NONE, INVERT_FULL, INVERT_REDUCED = 0, 1, 2   # MatrixInversion (BundleAdjustment.java:65-70)
INVERT_FULL_EXPANDED = 3     # all of Qxx like FULL, computed from the EO-reduced system (jaicov_neq.h)

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


class EngineError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"jaicov status {code} ({STATUS.get(code, '?')}): {msg}")


class EngineOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("image_begin", C.c_int32),
                ("image_end", C.c_int32), ("apply_shared", C.c_int32), ("assembly_mode", C.c_int32),
                ("block_size", C.c_int32), ("reduced_reference_quirk", C.c_int32), ("deterministic", C.c_int32), ("refinement", C.c_int32),
                ("ordinary_group_elimination", C.c_int32), ("dispersion_refinement", C.c_int32),
                ("expansion_exchange", C.c_int32), ("inverse_refinement", C.c_int32), ("reserved", C.c_int32 * 1)]


class EstimateOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_iterations", C.c_int32), ("invert", C.c_int32),
                ("simulation", C.c_int32), ("lambda0", C.c_double), ("sigma2apriori", C.c_double)]


class EstimateResult(C.Structure):
    _fields_ = [("state", C.c_int32), ("iterations", C.c_int32), ("omega", C.c_double), ("max_abs_dx", C.c_double),
                ("final_lambda", C.c_double), ("seconds_total", C.c_double), ("seconds_last_pass", C.c_double)]


_LIB = None


def build_library(force: bool = False) -> str:
    """Compiles csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                          "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.jaicov_neq_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(EngineOptions), C.POINTER(vp)]
    L.jaicov_neq_destroy.argtypes = [vp]
    L.jaicov_neq_destroy.restype = None
    L.jaicov_neq_last_error.argtypes = [vp]
    L.jaicov_neq_last_error.restype = C.c_char_p
    L.jaicov_neq_num_slots.argtypes = [vp]
    L.jaicov_neq_num_slots.restype = C.c_size_t
    L.jaicov_neq_packed_length.argtypes = [vp]
    L.jaicov_neq_packed_length.restype = C.c_size_t
    L.jaicov_neq_set_parameters.argtypes = [vp, _pd, C.c_size_t]
    L.jaicov_neq_get_parameters.argtypes = [vp, _pd, C.c_size_t]
    L.jaicov_neq_build.argtypes = [vp, C.c_double, C.c_double, C.c_int]
    L.jaicov_neq_accumulate.argtypes = [vp, C.c_double]
    L.jaicov_neq_accumulate2.argtypes = [vp, C.c_double, C.c_double]
    L.jaicov_neq_prepare_inverse.argtypes = [vp, C.c_int]
    L.jaicov_neq_reduced_order.argtypes = [vp]
    L.jaicov_neq_cofactor_order.argtypes = [vp]
    L.jaicov_neq_finalize.argtypes = [vp, C.c_double, C.c_double, C.c_int]
    L.jaicov_neq_reduce_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.jaicov_neq_reduce_buffer_async.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]
    L.jaicov_neq_solve.argtypes = [vp, C.c_int, _pd]
    L.jaicov_neq_eo_step_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.jaicov_neq_expansion_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.jaicov_neq_omega.argtypes = [vp, C.c_double, _pd, _pd]
    L.jaicov_neq_update.argtypes = [vp, _pd, _pd]
    L.jaicov_neq_get_normal.argtypes = [vp, _pd, C.c_size_t, _pd, C.c_size_t]
    L.jaicov_neq_get_cofactor.argtypes = [vp, _pd, C.c_size_t]
    L.jaicov_neq_get_cofactor_sub.argtypes = [vp, _pi, C.c_int32, _pd]
    L.jaicov_neq_get_dispersion_sub.argtypes = [vp, C.c_double, _pi, C.c_int32, _pd]
    L.jaicov_neq_get_rows.argtypes = [vp, C.c_int32, C.c_int32, _pd, _pd]
    L.jaicov_neq_estimate.argtypes = [vp, C.POINTER(EstimateOptions), C.POINTER(EstimateResult)]
    L.jaicov_neq_last_timings.argtypes = [vp, _pd, C.c_int32]
    L.jaicov_neq_set_profiling.argtypes = [vp, C.c_int]
    L.jaicov_neq_kernel_stats.argtypes = [vp, _pd, C.c_int32, C.c_int]
    L.jaicov_neq_cancel.argtypes = [vp]
    L.jaicov_neq_create_timings.argtypes = [vp, _pd, C.c_int32]
    L.jaicov_neq_get_block_weight.argtypes = [vp, C.c_int32, _pd, C.c_size_t]
    L.jaicov_dense_spd_solve_packed.argtypes = [C.c_int32, _pd, _pd, C.c_int32, C.c_int32, _pd]
    L.jaicov_dense_gemm.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, _pd, C.c_int64,
                                    _pd, C.c_int64, C.c_double, _pd, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _pd]
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(_pd)


class Engine:
    """One engine per adjustment (``BundleAdjustment`` is single-shot: BundleAdjustment.java:203)."""

    def __init__(self, fp: FlatProblem, device: int = 0, image_range=None, apply_shared: bool = True, assembly_mode: int = 0,
                 reduced_reference_quirk: bool = False, deterministic=None, refinement: int = 0,
                 ordinary_group_elimination: int = 0, dispersion_refinement: int = 0, expansion_exchange: bool = False,
                 inverse_refinement: int = 0):
        self.L = load_library()
        self.fp = fp
        self.U = fp.n_unknowns
        self._desc, self._keep = fp.as_desc()
        opts = EngineOptions()
        opts.struct_size = C.sizeof(EngineOptions)
        opts.device = device
        opts.image_begin, opts.image_end = image_range if image_range is not None else (-1, -1)
        opts.apply_shared = int(apply_shared)
        opts.assembly_mode = int(assembly_mode)
        opts.reduced_reference_quirk = int(reduced_reference_quirk)
        # None: the engine's default (deterministic since round 4); True / False: on / off (arrival-order atomics, 0.3 ms faster at config 4)
        opts.deterministic = 0 if deterministic is None else (1 if deterministic else -1)
        opts.ordinary_group_elimination = int(ordinary_group_elimination)   # < 0: ordinary image groups stay outside the EO pre-elimination
        opts.dispersion_refinement = int(dispersion_refinement)             # < 0: inv(D) as the blocked Cholesky leaves it
        opts.expansion_exchange = int(expansion_exchange)                   # sharded engines: the caller all-reduces expansion_buffer()
        self.expansion_exchange = bool(expansion_exchange)
        opts.inverse_refinement = int(inverse_refinement)                   # < 0: no Newton-Schulz step on the inverse of orders <= 8192
        opts.refinement = int(refinement)      # 0 = default (one step of iterative refinement per solve), < 0 = none, k = k steps
        self._h = C.c_void_p()
        rc = self.L.jaicov_neq_create(C.byref(self._desc), C.byref(opts), C.byref(self._h))
        if rc != 0:
            msg = self.L.jaicov_neq_last_error(self._h).decode() if self._h else ""
            if self._h:
                self.L.jaicov_neq_destroy(self._h)
                self._h = C.c_void_p()
            raise EngineError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            self.L.jaicov_neq_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(rc, self.L.jaicov_neq_last_error(self._h).decode())

    # parameters ---------------------------------------------------------------------------------------------
    def set_parameters(self, values):
        v = np.ascontiguousarray(values, np.float64)
        self._chk(self.L.jaicov_neq_set_parameters(self._h, _p(v), v.size))

    def get_parameters(self):
        v = np.zeros(self.fp.n_slots)
        self._chk(self.L.jaicov_neq_get_parameters(self._h, _p(v), v.size))
        return v

    # loop body ----------------------------------------------------------------------------------------------
    def build(self, sigma2, lam=0.0, simulation=False):
        self._chk(self.L.jaicov_neq_build(self._h, sigma2, lam, int(simulation)))

    def accumulate(self, sigma2, lam=0.0):
        self._chk(self.L.jaicov_neq_accumulate2(self._h, sigma2, lam))

    def reduced_order(self):
        """Order of the system assembled by the last accumulate (U, or the first EO column with EO pre-elimination)."""
        return int(self.L.jaicov_neq_reduced_order(self._h))

    def prepare_inverse(self, inverse_follows=True):
        """Announce the `invert` value of the solve after the next build (final pass): INVERT_FULL (True) makes that
        build assemble the full system, INVERT_REDUCED keeps the EO pre-elimination."""
        self._chk(self.L.jaicov_neq_prepare_inverse(self._h, int(inverse_follows)))

    def cofactor_order(self):
        return int(self.L.jaicov_neq_cofactor_order(self._h))

    def finalize(self, sigma2, lam=0.0, simulation=False):
        self._chk(self.L.jaicov_neq_finalize(self._h, sigma2, lam, int(simulation)))

    def reduce_buffer(self):
        ptr = C.c_void_p(); cnt = C.c_size_t()
        self._chk(self.L.jaicov_neq_reduce_buffer(self._h, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def reduce_buffer_async(self):
        """(device pointer, count, hipStream_t) -- no host wait; the buffer is complete in the order of that stream."""
        ptr = C.c_void_p(); cnt = C.c_size_t(); st = C.c_void_p()
        self._chk(self.L.jaicov_neq_reduce_buffer_async(self._h, C.byref(ptr), C.byref(cnt), C.byref(st)))
        return ptr.value, cnt.value, st.value

    def eo_step_buffer(self):
        """(device pointer, count) of the EO steps this engine back-substituted in the last solve (6 per image, zeros elsewhere)."""
        ptr = C.c_void_p(); cnt = C.c_size_t()
        self._chk(self.L.jaicov_neq_eo_step_buffer(self._h, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def expansion_buffer(self):
        """(device pointer, count) of [F | L_E^-1] for a sharded FULL_EXPANDED final pass: sum over the ranks, then solve."""
        ptr = C.c_void_p(); cnt = C.c_size_t()
        self._chk(self.L.jaicov_neq_expansion_buffer(self._h, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def solve(self, invert=False):
        dx = np.zeros(max(self.U, 1))
        self._chk(self.L.jaicov_neq_solve(self._h, int(invert), _p(dx)))
        return dx[:self.U]

    def omega(self, sigma2, dx):
        dx = np.ascontiguousarray(dx, np.float64)
        om = np.zeros(1)
        self._chk(self.L.jaicov_neq_omega(self._h, sigma2, _p(dx), _p(om)))
        return float(om[0])

    def update(self, dx):
        dx = np.ascontiguousarray(dx, np.float64)
        mx = np.zeros(1)
        self._chk(self.L.jaicov_neq_update(self._h, _p(dx), _p(mx)))
        return float(mx[0])

    # results ------------------------------------------------------------------------------------------------
    def get_normal(self):
        N = np.zeros(self.fp.packed_length); n = np.zeros(self.U)
        self._chk(self.L.jaicov_neq_get_normal(self._h, _p(N), N.size, _p(n), n.size))
        return N, n

    def get_cofactor(self):
        """Packed 'U' cofactor matrix of order cofactor_order() (U after INVERT_FULL, reduced_order() after INVERT_REDUCED)."""
        k = self.cofactor_order()
        if k < 0:
            raise EngineError(-2, "no cofactor matrix: solve with invert != 0 first")
        Q = np.zeros(k * (k + 1) // 2)
        self._chk(self.L.jaicov_neq_get_cofactor(self._h, _p(Q), Q.size))
        return Q

    def get_cofactor_sub(self, idx):
        idx = np.ascontiguousarray(idx, np.int32)
        out = np.zeros((idx.size, idx.size))
        self._chk(self.L.jaicov_neq_get_cofactor_sub(self._h, idx.ctypes.data_as(_pi), idx.size, _p(out)))
        return out

    def get_dispersion_sub(self, sigma2_aposteriori, idx):
        """sigma2 * Qxx[idx, idx] gathered and scaled on the device (what the result writers print)."""
        idx = np.ascontiguousarray(idx, np.int32)
        out = np.zeros((idx.size, idx.size))
        self._chk(self.L.jaicov_neq_get_dispersion_sub(self._h, float(sigma2_aposteriori), idx.ctypes.data_as(_pi), idx.size, _p(out)))
        return out

    def get_rows(self, ip_begin, ip_count):
        w = np.zeros((ip_count, 2)); A = np.zeros((ip_count, 2, KROW))
        self._chk(self.L.jaicov_neq_get_rows(self._h, ip_begin, ip_count, _p(w), _p(A)))
        return w, A

    def timings(self):
        ms = np.zeros(8)
        self._chk(self.L.jaicov_neq_last_timings(self._h, _p(ms), 8))
        return dict(zip(("rows", "assembly", "finalize", "factor", "solve", "inverse", "omega", "total"), ms))

    def create_timings(self):
        """What jaicov_neq_create spent, ms of host wall clock."""
        ms = np.zeros(8)
        self._chk(self.L.jaicov_neq_create_timings(self._h, _p(ms), 8))
        return {"create_ms": float(ms[0]), "dispersion_upload_host_ms": float(ms[1]), "dispersions_to_weights_ms": float(ms[2]),
                "tables_and_structure_upload_ms": float(ms[3]), "work_buffers_and_solver_ms": float(ms[4]), "elimination_buffers_and_reduced_solver_ms": float(ms[5])}

    def get_block_weight(self, block):
        """inv(D) of image block `block` as cached at create (DOPG:82-86 caches sigma0^2 times it), caller's observation order."""
        m = 2 * int(self.fp.blk_ip_begin[block + 1] - self.fp.blk_ip_begin[block])
        out = np.zeros((m, m))
        self._chk(self.L.jaicov_neq_get_block_weight(self._h, int(block), _p(out), out.size))
        return out

    def set_profiling(self, on=True):
        self._chk(self.L.jaicov_neq_set_profiling(self._h, int(on)))

    def kernel_stats(self, reset=False):
        st = np.zeros(13)
        self._chk(self.L.jaicov_neq_kernel_stats(self._h, _p(st), 13, int(reset)))
        return {"launches": st[0], "ms": st[1], "flops": st[2], "dense_passes": st[3], "dense_gemm_ms": st[4], "dense_flops": st[5],
                "flow_retries": int(st[6]), "flow_stale_events": int(st[7]), "flow_stale_confirmed": int(st[8]),
                "flow_rescued": int(st[9]), "last_refinement_correction": float(st[10]), "refine_steps": int(st[11]), "gather_strip_columns": int(st[12])}

    def cancel(self):
        """``BundleAdjustment.interrupt()`` (BundleAdjustment.java:1455): the running / next ``estimate`` ends with state -1."""
        self._chk(self.L.jaicov_neq_cancel(self._h))

    def estimate(self, values=None, sigma2=None, lam0=0.0, max_iter=5000, invert=True, simulation=False):
        """``BundleAdjustment.estimateModel()`` (BundleAdjustment.java:203-387) run natively on the engine."""
        self.set_parameters(self.fp.values if values is None else values)
        o = EstimateOptions()
        o.struct_size = C.sizeof(EstimateOptions)
        o.max_iterations = max_iter; o.invert = int(invert); o.simulation = int(simulation)
        o.lambda0 = lam0; o.sigma2apriori = self.fp.sigma2apriori if sigma2 is None else sigma2
        res = EstimateResult()
        self._chk(self.L.jaicov_neq_estimate(self._h, C.byref(o), C.byref(res)))
        return self.get_parameters(), res


def dense_spd_solve_packed(ap, b=None, invert=False):
    """MathExtension.solve(UpperSPDPackMatrix ...) / MathExtension.inv on the device.  Returns (x, ap_out, ms)."""
    L = load_library()
    ap = np.ascontiguousarray(ap, np.float64).copy()
    n = int(round((np.sqrt(8 * ap.size + 1) - 1) / 2))
    nrhs = 0
    if b is not None:
        b = np.ascontiguousarray(np.atleast_2d(b), np.float64).copy()
        nrhs = b.shape[0]
    ms = np.zeros(1)
    rc = L.jaicov_dense_spd_solve_packed(n, _p(ap), _p(b) if b is not None else C.cast(None, _pd), nrhs, int(invert), _p(ms))
    if rc != 0:
        raise EngineError(rc, "dense SPD solve")
    return b, ap, float(ms[0])


def dense_gemm(alay, blay, A, B, C_in, M, N, K, alpha=1.0, beta=0.0, lower_only=False, kmode=0, repeats=0):
    L = load_library()
    A = np.ascontiguousarray(A, np.float64); B = np.ascontiguousarray(B, np.float64)
    Cm = np.ascontiguousarray(C_in, np.float64).copy()
    ms = np.zeros(1)
    rc = L.jaicov_dense_gemm(alay, blay, M, N, K, alpha, _p(A), A.shape[1], _p(B), B.shape[1], beta, _p(Cm), Cm.shape[1],
                             int(lower_only), kmode, repeats, _p(ms))
    if rc != 0:
        raise EngineError(rc, "dense gemm")
    return Cm, float(ms[0])
