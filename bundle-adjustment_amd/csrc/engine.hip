// C ABI of the normal-equation engine (include/jaicov_neq.h): host control + small glue kernels.  gfx950 only.
// There is NO CPU fallback here: every entry point fails with JAICOV_ERR_NO_DEVICE / JAICOV_ERR_DEVICE when HIP is
// unusable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/jaicov_dense.h"
#include "../../include/jaicov_neq.h"
#include "ba_kernels.h"
#include "batchinv.h"
#include "dense.h"
#include "gemm_f64.h"

namespace jaicov {
hipError_t launch_rows(hipStream_t, const DevProblem &, const double *, int, int, double *, double *);
hipError_t launch_assemble_small(hipStream_t, const DevProblem &, const int32_t *, const int32_t *, int, const double *,
                                 const double *, double, double *, double *);
hipError_t launch_assemble_blocks(hipStream_t, const DevProblem &, const int32_t *, int, int, const int32_t *, int,
                                  const double *, const double *, double *, double, double *, double *, const PPGather &,
                                  const SchurBufs &, double *, hipStream_t, hipEvent_t, hipEvent_t);
hipError_t launch_schur_backsub(hipStream_t, const DevProblem &, const int32_t *, int, const double *, const double *,
                                const double *, double *);
hipError_t launch_shared_groups(hipStream_t, const DevProblem &, const double *, double, double *, double *,
                                const double *, double *);
hipError_t launch_schur_expand_f(hipStream_t, const DevProblem &, const int32_t *, int, const double *, const double *, const double *,
                                 const double *, double *, long);
hipError_t launch_omega(hipStream_t, const DevProblem &, const uint8_t *, int, int, const int32_t *, int, int,
                        const double *, const double *, const double *, double, double *, double *);
}  // namespace jaicov

namespace jaicov {
struct RefineBorder { double kappa[7]; double rk[7]; };
hipError_t launch_residual_dd(hipStream_t, const double *, long, int, int, int, const double *, const double *, const double *,
                              const double *, const double *, long, const RefineBorder &, double *, double *);
}
using namespace jaicov;

static const double EPS53 = 1.1102230246251565e-16;   // Constant.EPS = 2^-53 (Constant.java:68-75)

// ---------------------------------------------------------------------------------------------------------------
// glue kernels
// ---------------------------------------------------------------------------------------------------------------
// BA:814-828: damping on unknown columns, then V = 1/sqrt(diag) where diag > EPS (V = 1 on the border, whose
// diagonal is zero); padded rows get V = 1
__global__ void damp_and_precond_kernel(double *N, long ld, int U, int Upad, int d, double lambda, double *V,
                                        const double *diagcorr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Upad) return;
    double v = 1.0;
    if (c >= d && c < U) {
        double diag = N[(long)c * ld + c];
        if (lambda > 0.0) {
            // the reference damps the diagonal of the UNREDUCED normal matrix; diagcorr = what the EO elimination took
            diag += lambda * (diag + (diagcorr ? diagcorr[c] : 0.0));
            N[(long)c * ld + c] = diag;
        }
        v = diag > 1.1102230246251565e-16 ? 1.0 / sqrt(diag) : 1.0;
    }
    V[c] = v;
}

// zero the lower triangle (row r: columns 0 .. r, rounded up to 4) of the leading `rows` rows; grid (ceil(rows/1024), rows)
// rows skip_lo..skip_hi (the point rows the gather stores itself, PPGather::plain): only the columns before skip_lo
__global__ __launch_bounds__(256) void zero_lower_kernel(double *__restrict__ N, long ld, int rows, int skip_lo, int skip_hi, int row0) {
    const int r = row0 + blockIdx.y, c = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (c > r || r >= rows) return;
    if (r >= skip_lo && r <= skip_hi) {
        if (c >= skip_lo) return;
        if (c + 4 > skip_lo) {           // the quad straddles the first stored column
            for (int k = c; k < skip_lo; k++) N[(long)r * ld + k] = 0.0;
            return;
        }
    }
    *reinterpret_cast<d4_t *>(N + (long)r * ld + c) = (d4_t){0.0, 0.0, 0.0, 0.0};
}

// M = V N V + Bh' Bh on the unknown block (lower part), identity on border and padding.  Bh: [d][bstride]
__global__ __launch_bounds__(256) void scale_copy_kernel(const double *__restrict__ N, long ldN, double *__restrict__ M,
                                                         long ld, int U, int Upad, int d, const double *__restrict__ V,
                                                         const double *__restrict__ Bh, int bstride, int c_begin, int c_end) {
    const int j = c_begin + blockIdx.x * 256 + threadIdx.x;   // columns [c_begin, c_end)
    const int i = blockIdx.y;
    if (j > i || j >= c_end) return;
    double v;
    if (i < d || i >= U || j < d) v = (i == j) ? 1.0 : 0.0;
    else {
        v = V[i] * N[(long)i * ldN + j] * V[j];
        for (int a = 0; a < d; a++) v += Bh[(long)a * bstride + i] * Bh[(long)a * bstride + j];
    }
    M[(long)i * ld + j] = v;
}

__global__ void scale_vec_kernel(const double *n, const double *V, double *out, int U, int Upad, int d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Upad) return;
    out[c] = (c >= d && c < U) ? V[c] * n[c] : 0.0;
}

// final cofactor: Qxx[i][j] = V_i V_j (Q[i][j] - sum_a G[a][i] H[a][j]) on the unknown block, border from F, E.
// G: [d][gs]; H, F: [d][Upad]; E: [d][d].  Lower part (j <= i) of the leading `order` rows only.
__global__ __launch_bounds__(256) void qfix_kernel(double *__restrict__ Q, long ld, int U, int Upad, int d,
                                                   const double *__restrict__ V, const double *__restrict__ G, long gs,
                                                   const double *__restrict__ H, const double *__restrict__ F,
                                                   const double *__restrict__ E, int order) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j > i || j >= order) return;
    double v;
    if (i >= U) v = 0.0;
    else if (i < d) v = E[i * d + j];
    else if (j < d) v = F[(long)j * Upad + i];
    else {
        v = Q[(long)i * ld + j];
        for (int a = 0; a < d; a++) v -= G[(long)a * gs + i] * H[(long)a * Upad + j];
        v *= V[i] * V[j];
    }
    Q[(long)i * ld + j] = v;
}

// Full cofactor matrix from the inverse of the EO-reduced system (JAICOV_INVERT_FULL_EXPANDED):
//   rows / columns < e0        Qr (the reduced inverse, final: unscaled, border included)
//   EO row e0 + r, column < e0   T1[r][c] = -(F Q_RR)[r][c]
//   EO row, EO column            T2 = -(T1 F') (lower) + N_EE^-1 on the 6 x 6 diagonal blocks, N_EE^-1 = L_E^-T L_E^-1
// lower part (j <= i) of the leading U rows of Qf.
__global__ __launch_bounds__(256) void expand_cofactor_kernel(double *__restrict__ Qf, long ldf, int U, int e0,
                                                              const double *__restrict__ Qr, long ldr,
                                                              const double *__restrict__ T1, long ld1,
                                                              const double *__restrict__ T2, long ld2,
                                                              const double *__restrict__ Linv) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j > i || i >= U) return;
    double v;
    if (i < e0) v = Qr[(long)i * ldr + j];
    else if (j < e0) v = T1[(long)(i - e0) * ld1 + j];
    else {
        const int ri = i - e0, rj = j - e0;
        v = T2[(long)ri * ld2 + rj];
        if (ri / 6 == rj / 6) {
            const double *Li = Linv + (long)(ri / 6) * 36;
            const int a = ri % 6, b = rj % 6;
            for (int m = a > b ? a : b; m < 6; m++) v += Li[6 * m + a] * Li[6 * m + b];
        }
    }
    Qf[(long)i * ldf + j] = v;
}

// row-major lower square <-> packed ('U' column-major == row-major lower packed)
__global__ __launch_bounds__(256) void pack_kernel(const double *__restrict__ M, long ld, int U, double *__restrict__ ap) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j > i || i >= U) return;
    ap[(size_t)i * (i + 1) / 2 + j] = M[(long)i * ld + j];
}
__global__ __launch_bounds__(256) void unpack_kernel(const double *__restrict__ ap, long ld, int U, double *__restrict__ M) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j > i || i >= U) return;
    M[(long)i * ld + j] = ap[(size_t)i * (i + 1) / 2 + j];
}
__global__ void gather_sub_kernel(const double *__restrict__ Q, long ld, const int32_t *idx, int k, double *out, double scale) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * k) return;
    const int a = t / k, b = t - a * k;
    const int i = idx[a], j = idx[b];
    out[t] = scale * (i >= j ? Q[(long)i * ld + j] : Q[(long)j * ld + i]);
}
// dense dispersion (row-major m x m) -> padded lower square with identity padding
// perm (optional): engine point position -> caller's point position inside the block (rows 2q, 2q+1 move together)
__global__ void load_disp_kernel(const double *__restrict__ D, int m, double *__restrict__ L, long ld, int mp,
                                 const int32_t *__restrict__ perm) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= mp) return;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < m && j < m) {
        const int si = perm ? 2 * perm[i >> 1] + (i & 1) : i, sj = perm ? 2 * perm[j >> 1] + (j & 1) : j;
        v = D[(long)si * m + sj];
    }
    L[(long)i * ld + j] = v;
}
__global__ void store_inv_kernel(const double *__restrict__ Q, long ld, int m, double *__restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= m) return;
    out[(long)i * m + j] = i >= j ? Q[(long)i * ld + j] : Q[(long)j * ld + i];
}

// ---------------------------------------------------------------------------------------------------------------
struct jaicov_engine {
    std::string err = "";
    int device = 0;
    int flow_retries = 0;      // dataflow factorisations that were abandoned and repeated (solve)
    double *ns_work = nullptr; // three squares of the inverse's Newton-Schulz step (orders <= 8192, option inverse_refinement)
    size_t ns_work_len = 0;
    int refine_steps = 1;      // iterative refinement of the step (refine.hip): engine option `refinement` / JAICOV_REFINE
    double *d_Braw = nullptr, *d_refP = nullptr, *d_ref = nullptr;   // unscaled datum rows; partial-sum table; rhs | tmp | delta
    size_t refP_len = 0;
    double last_refine_correction = 0.0;   // max |correction| / max |dx| of the last refinement step (diagnostics)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // assembly: camera-side kernels on a side stream (assemble.hip)
    hipStream_t stream = nullptr;
    DevProblem p{};
    std::vector<void *> allocs;
    // host copies
    jaicov_engine_options opts{};
    int U = 0, Upad = 0, d = 0, datum_flags = 0, n_slots = 0;
    std::vector<int32_t> h_slot_col, h_point_col;
    std::vector<uint8_t> h_point_datum;
    std::vector<double> h_vals, h_V;
    int n_points = 0;
    // image range handled by this engine
    int ip0 = 0, ip_count = 0;
    // assembly tables
    int n_seg = 0, n_blk_list = 0, max_m = 0, n_blk_ip = 0;
    int32_t *d_seg_begin = nullptr, *d_seg_end = nullptr, *d_blk_list = nullptr, *d_blk_ip_list = nullptr;
    uint8_t *d_in_block = nullptr;
    PPGather pp;
    // EO pre-elimination (schur.hip)
    bool schur_ok = false, schur_active = false;
    DenseMode dm;                    // assembly_mode = 1
    bool dense_mode = false;
    double dm_flops_per_pass = 0.0, dm_stat_passes = 0.0, dm_stat_ms = 0.0, dm_stat_flops = 0.0;
    int inverse_mode_next = 0;       // JAICOV_INVERT_* announced for the solve after the next build
    bool all_images = true;          // this engine accumulates every image (not a shard): FULL_EXPANDED needs every image's F band and L_E ...
    double *d_expF = nullptr;        // ... a shard's caller sums them over the ranks: [F (6 images padded x reduced order padded) | Linv (36 per image)]
    size_t expF_len = 0;
    bool exp_ready = false;          // d_expF belongs to the system accumulated last and has been handed out (jaicov_neq_expansion_buffer)
    bool solver_has_Q = false;       // solver.Q alone is allocated (target of the expansion)
    int q_order = 0;                 // order of the cofactor matrix on the device (U, or e0 for the reduced one)
    bool q_reduced = false, solverS_has_inverse = false;
    std::vector<int> h_blk_images;   // image of every block handled by this engine
    std::vector<int32_t> h_caller_block; // internal image block -> the caller's block index (-1: an ordinary image served as a block); empty: identity
    bool synthesized_blocks = false;     // ordinary image groups are served as image blocks with block-diagonal weights (create_impl)
    bool compact_blocks = false;         // ... and their weights are kept as 2 x 2 blocks (DevProblem::ip_w3), not as m x m matrices
    std::vector<int32_t> h_perm_local;   // per image point of a dense block: engine position -> caller's position inside the block (empty: identity)
    std::vector<int64_t> h_blk_w_off;    // offset of every image block's weight in p.blk_w
    std::vector<int32_t> h_blk_ip_begin; // copy of blk_ip_begin
    std::vector<uint8_t> h_blk_mine;     // this engine holds the block's weight
    std::vector<int32_t> ip_old2new; // empty, or: engine position of the caller's observation (dense blocks are column-sorted)
    int e0 = 0;                   // first EO column == order of the reduced system
    SchurBufs sb;
    double *d_xE = nullptr;
    DenseSolver solverS;
    bool solverS_ready = false;
    double lambda_acc = 0.0;
    // device state
    double *d_vals = nullptr, *d_rowsA = nullptr, *d_rowsW = nullptr, *d_T = nullptr, *d_vbuf = nullptr;
    double *d_N = nullptr, *d_n = nullptr;          // one allocation: N (Upad x Upad) followed by n (Upad)
    double *d_packed = nullptr;                     // reduce buffer: packed N (U(U+1)/2) + n (U)
    double *d_V = nullptr, *d_B = nullptr, *d_dx = nullptr;
    double *d_cc_partial = nullptr;                 // [blocks of this engine][16 parts][KC_MAX (KC_MAX + 1)] partial camera blocks (assemble.hip)
    double *d_omega = nullptr, *d_G = nullptr, *d_H = nullptr, *d_F = nullptr, *d_E = nullptr;
    int32_t *d_idx = nullptr;
    DenseSolver solver;
    bool solver_has_inverse = false;
    enum { ST_NEW, ST_PARAMS, ST_ACCUMULATED, ST_BUILT, ST_SOLVED } state = ST_NEW;
    bool have_Q = false, rows_valid = false, reduced = false;
    bool deterministic = true;   // engine option `deterministic` (default on): fixed summation order in the assembly of the image groups
    std::atomic<int> cancel{0};  // BundleAdjustment.interrupt() (BA:1455): polled by estimate() where the reference polls (BA:240, 320)
    bool sim_built = false;      // the system at hand was built with simulation != 0: the right-hand side is zero for ALL unknowns (BA:830-831)
    double lambda_used = 0.0;
    std::vector<double> hB;      // [d][Upad] datum rows (unscaled), host
    double timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double create_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // jaicov_neq_create: [0] whole call, [1] host time in the dispersion uploads, [2] dispersions -> weights (wall),
                                                      // [3] validation + tables + structure upload, [4] work buffers + full-order solver, [5] EO pre-elimination buffers + reduced solver
    hipEvent_t ev[10];
    bool pp_plain_ok = false;   // the point x point gather may store its strips (see PPGather::plain)
    hipEvent_t ev_first = nullptr, ev_all = nullptr;   // solve(): first panel's columns / whole matrix copied into the solver
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;       // solve(): device part of a refinement step
};

#define FAIL(e, code, msg)                 \
    do {                                   \
        (e)->err = (msg);                  \
        return (code);                     \
    } while (0)
#define HIPE(e, x)                                                                        \
    do {                                                                                  \
        hipError_t _err = (x);                                                            \
        if (_err != hipSuccess) {                                                         \
            (e)->err = std::string(#x) + ": " + hipGetErrorString(_err);                  \
            return _err == hipErrorOutOfMemory ? JAICOV_ERR_OUT_OF_MEMORY : JAICOV_ERR_DEVICE; \
        }                                                                                 \
    } while (0)

template <typename T>
static int upload(jaicov_engine *e, const T *src, size_t count, const T **dst) {
    *dst = nullptr;
    if (count == 0) return JAICOV_OK;
    void *ptr = nullptr;
    HIPE(e, hipMalloc(&ptr, count * sizeof(T)));
    e->allocs.push_back(ptr);
    HIPE(e, hipMemcpy(ptr, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T *)ptr;
    return JAICOV_OK;
}
template <typename T>
static int dalloc(jaicov_engine *e, size_t count, T **dst, bool zero = false) {
    *dst = nullptr;
    if (count == 0) count = 1;
    void *ptr = nullptr;
    HIPE(e, hipMalloc(&ptr, count * sizeof(T)));
    e->allocs.push_back(ptr);
    if (zero) HIPE(e, hipMemset(ptr, 0, count * sizeof(T)));
    *dst = (T *)ptr;
    return JAICOV_OK;
}

static int check_device(std::string &err) {
    int count = 0;
    hipError_t r = hipGetDeviceCount(&count);
    if (r != hipSuccess || count <= 0) {
        err = "no HIP device available (this engine has no CPU fallback)";
        return JAICOV_ERR_NO_DEVICE;
    }
    return JAICOV_OK;
}

// ---- dense dispersions -> weights, all groups of one padded order together (batchinv.hip) ------------------------------------
struct DispItem { const double *host; double *dst; const int32_t *perm; int m; };   // dst, perm: device pointers
struct DispDesc { const double *src; double *dst; const int32_t *perm; int m; int pad; };
// dense dispersion (row-major m x m, in the staging buffer) -> padded square with identity padding, in the engine's point order
// perm (optional): engine point position -> caller's point position inside the block (rows 2q, 2q+1 move together)
__global__ void load_disp_batched_kernel(const DispDesc *__restrict__ desc, double *__restrict__ Lb, double *__restrict__ Db, long ld, long msz, int mp) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= mp) return;
    const DispDesc dd = desc[blockIdx.z];
    double v = (i == j) ? 1.0 : 0.0;
    if (i < dd.m && j < dd.m) {
        const int si = dd.perm ? 2 * dd.perm[i >> 1] + (i & 1) : i, sj = dd.perm ? 2 * dd.perm[j >> 1] + (j & 1) : j;
        // the LOWER triangle of the caller's array, mirrored (dpptrf / the per-matrix path of rounds 1-3 read nothing else: a caller that fills
        // one triangle only, or a slightly asymmetric matrix, must not reach the Newton-Schulz step as a non-symmetric D)
        v = si >= sj ? dd.src[(long)si * dd.m + sj] : dd.src[(long)sj * dd.m + si];
    }
    Lb[(long)blockIdx.z * msz + (long)i * ld + j] = v;
    if (Db) Db[(long)blockIdx.z * msz + (long)i * ld + j] = v;      // the refinement of the inverse reads the matrix once more
}
__global__ void store_inv_batched_kernel(const DispDesc *__restrict__ desc, const double *__restrict__ Qb, long ld, long msz) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    const DispDesc dd = desc[blockIdx.z];
    if (i >= dd.m || j >= dd.m) return;
    dd.dst[(long)i * dd.m + j] = Qb[(long)blockIdx.z * msz + (long)i * ld + j];
}

// The same weights in compact form (DevProblem::ip_w3): (w00, w01, w11) per image point, nothing else stored -- an ordinary image with m
// rows then costs 1.5 m doubles instead of m^2, and T = inv(D) [A_c | w] two multiplications per entry instead of a dense product.
__global__ void fill_ip_w3_kernel(const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ rho, int n_ip,
                                  double *__restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_ip) return;
    const double x = vx[q], y = vy[q], c = rho[q];
    double w00, w01, w11;
    if (c == 0.0) { w00 = 1.0 / x; w01 = 0.0; w11 = 1.0 / y; }
    else {
        const double inv_det = 1.0 / ((1.0 - c * c) * x * y);
        w00 = inv_det * y; w01 = -inv_det * c * sqrt(x * y); w11 = inv_det * x;
    }
    out[3 * (long)q] = w00; out[3 * (long)q + 1] = w01; out[3 * (long)q + 2] = w11;
}

// inv(D) of an ordinary image served as a block: 2 x 2 blocks [[vx, rho s], [rho s, vy]]^-1, s = sqrt(vx vy), on the diagonal
// (PDF:313-319 divided by sigma0^2; rho == 0: 1 / vx, 1 / vy as PDF:308-312), zeros elsewhere.  One workgroup per row.
__global__ void fill_diag_weight_kernel(const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ rho, int ipb, int m,
                                        double *__restrict__ out) {
    const int i = blockIdx.x, q = i >> 1, r = i & 1;
    const double x = vx[ipb + q], y = vy[ipb + q], c = rho[ipb + q];
    double w0, w1;   // entries (i, 2q), (i, 2q + 1)
    if (c == 0.0) { w0 = r == 0 ? 1.0 / x : 0.0; w1 = r == 0 ? 0.0 : 1.0 / y; }
    else {
        const double inv_det = 1.0 / ((1.0 - c * c) * x * y), off = -inv_det * c * sqrt(x * y);
        w0 = r == 0 ? inv_det * y : off;
        w1 = r == 0 ? off : inv_det * x;
    }
    double *o = out + (long)i * m;
    for (int j = threadIdx.x; j < m; j += blockDim.x) o[j] = j == 2 * q ? w0 : (j == 2 * q + 1 ? w1 : 0.0);
}

// DOPG:82-86 for every jointly dispersed group of the problem: items of one padded order are inverted in chunks (one set of
// batched launches per chunk); the upload of chunk c + 1 runs on a copy stream beside the inversion of chunk c.
// Timings (ms) are left in e->create_ms: [1] host time spent in the uploads, [2] everything (wall).
static int invert_dispersions(jaicov_engine *e, std::vector<DispItem> &items) {
    if (items.empty()) return JAICOV_OK;
    const auto t_all = std::chrono::steady_clock::now();
    double up_ms = 0.0;
    std::stable_sort(items.begin(), items.end(), [](const DispItem &a, const DispItem &b) { return (a.m + 127) / 128 < (b.m + 127) / 128; });
    hipStream_t cstream = nullptr;
    cstream = jaicov::stream_acquire(jaicov::STREAM_PLAIN);
    if (!cstream) FAIL(e, JAICOV_ERR_DEVICE, "no stream for the dispersion uploads");
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    int status = JAICOV_OK;
    std::string msg;
    for (int b = 0; b < 2 && status == JAICOV_OK; b++)      // (no early return from here on: the stream goes back to the pool and the events are destroyed below)
        if (hipEventCreateWithFlags(&ev_up[b], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_free[b], hipEventDisableTiming) != hipSuccess) {
            status = JAICOV_ERR_DEVICE; msg = "inversion of the dispersion matrices: no events";
        }
    for (size_t g0 = 0; g0 < items.size() && status == JAICOV_OK;) {
        const int mp = ((items[g0].m + 127) / 128) * 128;
        size_t g1 = g0;
        int mmax = 0;
        while (g1 < items.size() && ((items[g1].m + 127) / 128) * 128 == mp) { mmax = std::max(mmax, items[g1].m); g1++; }
        const int count = (int)(g1 - g0);
        const size_t msz = (size_t)mp * mp, mm = (size_t)mmax * mmax;
        // chunk: <= 64 matrices and <= ~4 GB of workspace (3 squares per matrix, 6 with the refinement) -- 8 chunks at config 4, the first upload (0.5 GB)
        // is the only one that nothing overlaps
        const bool refine = e->opts.dispersion_refinement >= 0;
        int cap = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)4 << 30) / ((refine ? 6 : 3) * msz * sizeof(double))));
        cap = std::min(cap, count);
        BatchedSpdInverse bi;
        double *d_stage[2] = {nullptr, nullptr};
        DispDesc *d_desc = nullptr;
        std::vector<DispDesc> desc(count);
        hipError_t he = bi.init(e->stream, mp, cap, refine);
        for (int b = 0; b < 2 && he == hipSuccess; b++) he = hipMalloc(&d_stage[b], (size_t)cap * mm * sizeof(double));
        if (he == hipSuccess) he = hipMalloc(&d_desc, (size_t)count * sizeof(DispDesc));
        if (he == hipSuccess) {
            for (int t = 0; t < count; t++) {
                const int c = t / cap, b = c & 1;
                desc[t] = DispDesc{d_stage[b] + (size_t)(t - c * cap) * mm, items[g0 + t].dst, items[g0 + t].perm, items[g0 + t].m, 0};
            }
            he = hipMemcpyAsync(d_desc, desc.data(), (size_t)count * sizeof(DispDesc), hipMemcpyHostToDevice, e->stream);
        }
        for (int c = 0; he == hipSuccess && c * cap < count; c++) {
            const int b = c & 1, first = c * cap, cnt = std::min(cap, count - first);
            if (c >= 2) he = hipStreamWaitEvent(cstream, ev_free[b], 0);      // the load kernel of chunk c - 2 has consumed this buffer
            const auto t_up = std::chrono::steady_clock::now();
            for (int t = 0; t < cnt && he == hipSuccess; t++) {
                const DispItem &it = items[g0 + first + t];
                he = hipMemcpyAsync(d_stage[b] + (size_t)t * mm, it.host, (size_t)it.m * it.m * sizeof(double), hipMemcpyHostToDevice, cstream);
            }
            up_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up).count();
            if (he == hipSuccess) he = hipEventRecord(ev_up[b], cstream);
            if (he == hipSuccess) he = hipStreamWaitEvent(e->stream, ev_up[b], 0);
            if (he != hipSuccess) break;
            hipLaunchKernelGGL(load_disp_batched_kernel, dim3((mp + 255) / 256, mp, cnt), dim3(256), 0, e->stream, d_desc + first, bi.Lb, bi.refine ? bi.Db : (double *)nullptr, bi.ld, bi.msz, mp);
            he = hipEventRecord(ev_free[b], e->stream);
            if (he == hipSuccess) he = bi.run(cnt);
            if (he != hipSuccess) break;
            hipLaunchKernelGGL(store_inv_batched_kernel, dim3((mmax + 255) / 256, mmax, cnt), dim3(256), 0, e->stream, d_desc + first, bi.Qb, bi.ld, bi.msz);
        }
        int info = 0;
        if (he == hipSuccess) he = hipMemcpyAsync(&info, bi.d_info, sizeof(int), hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        hipStreamSynchronize(cstream);
        bi.release();
        hipFree(d_stage[0]); hipFree(d_stage[1]); hipFree(d_desc);
        if (he != hipSuccess) { status = he == hipErrorOutOfMemory ? JAICOV_ERR_OUT_OF_MEMORY : JAICOV_ERR_DEVICE; msg = std::string("inversion of the dispersion matrices: ") + hipGetErrorString(he); }
        else if (info != 0) { status = JAICOV_ERR_SINGULAR; msg = "dispersion matrix is not positive definite (MatrixNotSPDException, DOPG:85-86)"; }
        g0 = g1;
    }
    for (int b = 0; b < 2; b++) { if (ev_up[b]) hipEventDestroy(ev_up[b]); if (ev_free[b]) hipEventDestroy(ev_free[b]); }
    jaicov::stream_release(jaicov::STREAM_PLAIN, cstream);
    e->create_ms[1] = up_ms;
    e->create_ms[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count();
    if (status != JAICOV_OK) FAIL(e, status, msg);
    return JAICOV_OK;
}

extern "C" int jaicov_neq_abi_version(void) { return JAICOV_NEQ_ABI_VERSION; }

extern "C" const char *jaicov_neq_last_error(const jaicov_engine *e) { return e ? e->err.c_str() : "null engine"; }

extern "C" void jaicov_neq_destroy(jaicov_engine *e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    e->solver.release();
    e->solverS.release();
    e->dm.release();
    for (void *ptr : e->allocs) hipFree(ptr);
    for (auto &evt : e->ev)
        if (evt) hipEventDestroy(evt);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    if (e->ev_first) hipEventDestroy(e->ev_first);
    if (e->ev_all) hipEventDestroy(e->ev_all);
    if (e->ev_r0) hipEventDestroy(e->ev_r0);
    if (e->ev_r1) hipEventDestroy(e->ev_r1);
    if (e->d_refP) hipFree(e->d_refP);
    if (e->d_expF) hipFree(e->d_expF);
    if (e->ns_work) hipFree(e->ns_work);
    jaicov::stream_release(jaicov::STREAM_PLAIN, e->stream);
    delete e;
}

static int create_impl(jaicov_engine *e, const jaicov_problem_desc *D_in, const jaicov_engine_options *opts) {
    const jaicov_problem_desc *D = D_in;
    jaicov_problem_desc Dperm;   // the description with the observations of dense image blocks in column order
    std::vector<int32_t> pv_image, pv_point, perm_local;
    std::vector<double> pv_x, pv_y, pv_vx, pv_vy, pv_rho;
    HIPE(e, hipSetDevice(e->device));
    e->deterministic = e->opts.deterministic >= 0;      // 0 = default = ON since round 4 (costs 0.3 ms per pass at config 4); < 0: arrival-order sums
    e->refine_steps = e->opts.refinement == 0 ? 1 : (e->opts.refinement < 0 ? 0 : std::min(e->opts.refinement, 4));
    hipDeviceProp_t prop;
    HIPE(e, hipGetDeviceProperties(&prop, e->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        FAIL(e, JAICOV_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    e->stream = jaicov::stream_acquire(jaicov::STREAM_PLAIN);
    if (!e->stream) FAIL(e, JAICOV_ERR_DEVICE, "no stream");
    for (auto &evt : e->ev) HIPE(e, hipEventCreate(&evt));
    HIPE(e, hipEventCreateWithFlags(&e->ev_first, hipEventDisableTiming));
    HIPE(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIPE(e, hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    HIPE(e, hipEventCreateWithFlags(&e->ev_all, hipEventDisableTiming));
    HIPE(e, hipEventCreate(&e->ev_r0));
    HIPE(e, hipEventCreate(&e->ev_r1));

    const int U = D->n_unknowns, d = D->rank_defect;
    e->U = U; e->d = d; e->datum_flags = D->datum_flags;
    e->Upad = ((U + 127) / 128) * 128;
    if (e->Upad == 0) e->Upad = 128;
    e->n_points = D->n_points;
    e->n_slots = 3 * D->n_points + 3 * D->n_cameras + D->n_dist + 6 * D->n_images;
    // ---- validate --------------------------------------------------------------------------------------------
    if (__builtin_popcount((unsigned)D->datum_flags) != d || d < 0 || d > 7) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "datum_flags / rank_defect mismatch");
    for (int c = 0; c < D->n_cameras; c++) {
        const int jb = D->cam_dist_begin[c], je = D->cam_dist_begin[c + 1];
        if (je - jb > JAICOV_MAX_DIST_PER_CAMERA) FAIL(e, JAICOV_ERR_UNSUPPORTED, "too many distortion coefficients for one camera");
        for (int j = jb + 1; j < je; j++)
            if (D->dist_kind[j] < D->dist_kind[j - 1]) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "distortion coefficients must be in DistortionModel.Type order");
        // AffinityShearDistortionModel always owns Cx and Cy, TangentialDistortionModel Bx and By (+ optional Bi): a lone
        // member would silently drop out of the model (ASF:37-81, TDF:39-134 read both)
        int cnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = jb; j < je; j++) {
            if (D->dist_kind[j] < 0 || D->dist_kind[j] > JAICOV_DIST_ZERNIKE_Z) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "unknown distortion coefficient kind");
            if (D->dist_kind[j] >= JAICOV_DIST_ZERNIKE_X && (D->dist_order[j] < 1 || D->dist_order[j] > 119))
                FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "Zernike coefficient order must be 1..119 (ZernikeDistortionModel.java:67-68)");
            cnt[D->dist_kind[j]]++;
        }
        if (cnt[0] != cnt[1] || cnt[0] > 1 || cnt[2] != cnt[3] || cnt[2] > 1 || (cnt[4] > 0 && cnt[2] == 0))
            FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "affinity (Cx, Cy) and tangential (Bx, By [, Bi]) coefficients come as complete models");
    }
    for (int i = 1; i < D->n_image_points; i++)
        if (D->ip_image[i] < D->ip_image[i - 1]) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "image points must be image-major");
    e->h_slot_col.resize(e->n_slots);
    {
        int s = 0;
        for (int i = 0; i < 3 * D->n_points; i++) e->h_slot_col[s++] = D->point_col[i];
        for (int i = 0; i < 3 * D->n_cameras; i++) e->h_slot_col[s++] = D->io_col[i];
        for (int i = 0; i < D->n_dist; i++) e->h_slot_col[s++] = D->dist_col[i];
        for (int i = 0; i < 6 * D->n_images; i++) e->h_slot_col[s++] = D->eo_col[i];
        std::vector<char> seen(U > 0 ? U : 1, 0);
        for (int c : e->h_slot_col) {
            if (c == JAICOV_COL_FIXED) continue;
            if (c < d || c >= U || seen[c]) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "columns must be a permutation of d..U-1");
            seen[c] = 1;
        }
    }
    e->h_point_col.assign(D->point_col, D->point_col + 3 * D->n_points);
    e->h_point_datum.assign(D->point_datum, D->point_datum + D->n_points);

    // ---- image range of this engine ----------------------------------------------------------------------------
    int ib = 0, ie = D->n_images;
    if (opts && opts->image_begin >= 0 && opts->image_end >= 0) { ib = opts->image_begin; ie = opts->image_end; }
    if (ib < 0 || ie > D->n_images || ib > ie) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "bad image range");
    e->all_images = ib == 0 && ie == D->n_images;
    {
        const int32_t *b = std::lower_bound(D->ip_image, D->ip_image + D->n_image_points, ib);
        const int32_t *en = std::lower_bound(D->ip_image, D->ip_image + D->n_image_points, ie);
        e->ip0 = (int)(b - D->ip_image);
        e->ip_count = (int)(en - b);
    }
    // ---- ordinary image groups as jointly dispersed groups with a block-diagonal weight ----------------------------------
    // reduceNormalEquationSystem (BA:1197-1342) eliminates the exterior orientation of EVERY image, whatever its stochastic
    // model; the device path of that elimination (schur.hip) is written for image groups with a joint weight matrix.  An image
    // whose points are ordinary ImageCoordinate groups (diagonal / 2 x 2 weights, PDF:296-319) is the special case of a
    // block-diagonal joint weight: when the whole problem qualifies, every such image becomes an internal image block whose
    // inv(D) is filled from (var_x, var_y, rho) in closed form -- no dispersion is uploaded or inverted -- and the
    // elimination, the point x point gather and the reduced solve serve it like any other block.  All or nothing, decided on
    // the WHOLE problem (every rank of a sharded run must assemble a system of the same order).
    jaicov_problem_desc Dsyn;
    std::vector<int32_t> syn_begin;
    std::vector<int64_t> syn_off;
    e->h_caller_block.clear();
    {
        bool ok = e->opts.ordinary_group_elimination >= 0 && e->opts.assembly_mode == 0 && D->n_images > 0;
        const int e0 = D->n_images > 0 ? D->eo_col[0] : -1;
        ok = ok && e0 >= d && e0 + 6 * D->n_images == U;
        for (int i = 0; ok && i < 6 * D->n_images; i++) ok = D->eo_col[i] == e0 + i;
        const int s_eo = 3 * D->n_points + 3 * D->n_cameras + D->n_dist;
        for (int r = 0; ok && r < D->n_direct_rows; r++) ok = D->dg_slot[r] < s_eo;
        std::vector<int32_t> img_b(D->n_images + 1, 0), img_blk(std::max(1, D->n_images), -1);
        if (ok) {
            for (int ip = 0; ip < D->n_image_points; ip++) {
                if (D->ip_image[ip] < 0 || D->ip_image[ip] >= D->n_images) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "image index out of range");
                img_b[D->ip_image[ip] + 1]++;
            }
            for (int i = 0; i < D->n_images; i++) img_b[i + 1] += img_b[i];
            for (int g = 0; ok && g < D->n_image_blocks; g++) {       // a caller's block must be ALL observations of its image
                const int b = D->blk_ip_begin[g], en = D->blk_ip_begin[g + 1];
                if (en <= b || b < 0 || en > D->n_image_points) { ok = false; break; }
                const int img = D->ip_image[b];
                ok = img_b[img] == b && img_b[img + 1] == en && img_blk[img] < 0;
                if (ok) img_blk[img] = g;
            }
            int64_t syn_bytes = 0;
            bool any = false;
            for (int i = 0; ok && i < D->n_images; i++) {
                const int64_t cnt = img_b[i + 1] - img_b[i];
                if (img_blk[i] >= 0) continue;
                any = true;
                ok = cnt >= 3 && 2 * cnt <= 4096;
                syn_bytes += 4 * cnt * cnt * (int64_t)sizeof(double);
            }
            // (the dense form of the block-diagonal weights -- the alternative assembly forms only -- needs m^2 doubles per image)
            ok = ok && any && (assembly_form() == ASSEMBLY_DEFAULT || assembly_form() == ASSEMBLY_NO_FORK || syn_bytes <= ((int64_t)16 << 30));
            // Size rule (round 5; option 0 = default): serving ordinary images as blocks costs six block-kernel launches where the ordinary assembly
            // is one (+0.10 ms per pass at BASELINE config 2), and pays through the factorisation's block columns: eliminate when the 6 I exterior-
            // orientation columns are at least two 128-blocks of it (config 2: order 726 -> 606, six block columns -> five: not worth it, 0.79 vs
            // 0.86 ms per pass; config 3: 29 -> 24, the bundled example 10 -> 4: yes).  A property of the WHOLE problem, so every rank of a
            // sharded run decides alike.  > 0 forces the elimination at any size (tests), < 0 switches it off.
            // (A problem that ALSO has jointly dispersed images runs the block kernels anyway, and only with every image served as a block
            // can any exterior orientation be eliminated: no size rule there.)
            if (ok && e->opts.ordinary_group_elimination == 0 && D->n_image_blocks == 0)
                ok = (U + 127) / 128 - (U - 6 * D->n_images + 127) / 128 >= 2;
        }
        if (ok) {
            syn_begin.push_back(0);
            for (int i = 0; i < D->n_images; i++) {
                syn_begin.push_back(img_b[i + 1]);
                syn_off.push_back(img_blk[i] >= 0 ? D->blk_disp_offset[img_blk[i]] : (int64_t)-1);
                e->h_caller_block.push_back(img_blk[i]);
            }
            Dsyn = *D;
            Dsyn.n_image_blocks = D->n_images;
            Dsyn.blk_ip_begin = syn_begin.data();
            Dsyn.blk_disp_offset = syn_off.data();
            D = &Dsyn;
            e->synthesized_blocks = true;
        }
    }
    // ---- blocks / segments -------------------------------------------------------------------------------------
    std::vector<uint8_t> in_block(D->n_image_points + 1, 0);
    std::vector<int32_t> blk_list, blk_ip_list, seg_b, seg_e;
    std::vector<int64_t> blk_w_off(D->n_image_blocks + 1, 0);
    int64_t w_total = 0, w_total_saved = 0;
    for (int g = 0; g < D->n_image_blocks; g++) {
        const int b = D->blk_ip_begin[g], en = D->blk_ip_begin[g + 1];
        if (en < b || b < 0 || en > D->n_image_points) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "image blocks must be ascending ranges");
        for (int ip = b; ip < en; ip++) {
            if (D->ip_image[ip] != D->ip_image[b]) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "an image block must not span images");
            in_block[ip] = 1;
        }
        // an ordinary image served as a block (no dispersion of its own): block-diagonal weights in compact form (DevProblem::ip_w3),
        // dense only under the alternative assembly forms, whose kernels read m x m weights
        const bool compact = e->synthesized_blocks && D->blk_disp_offset[g] < 0 && (assembly_form() == ASSEMBLY_DEFAULT || assembly_form() == ASSEMBLY_NO_FORK);
        blk_w_off[g] = compact ? -1 : w_total;
        const bool mine = en > b && D->ip_image[b] >= ib && D->ip_image[b] < ie;
        if (mine) {
            const int64_t m = 2 * (int64_t)(en - b);
            if (!compact) w_total += m * m;
            else e->compact_blocks = true;
            blk_list.push_back(g);
            for (int ip = b; ip < en; ip++) blk_ip_list.push_back(ip);
            e->max_m = std::max(e->max_m, (int)m);
        }
    }
    for (int ip = e->ip0; ip < e->ip0 + e->ip_count;) {
        if (in_block[ip]) { ip++; continue; }
        int en = ip;
        while (en < e->ip0 + e->ip_count && !in_block[en] && D->ip_image[en] == D->ip_image[ip] && en - ip < SEG) en++;
        seg_b.push_back(ip); seg_e.push_back(en);
        ip = en;
    }
    w_total_saved = w_total;
    // ---- column-sorted storage inside dense image blocks -------------------------------------------------------
    // The point x point gather (assemble.hip) streams, for one object point and one range of columns, the weights of
    // the partner points of an image; with the image's observations stored in the order of their points' columns
    // those partners are contiguous.  The engine therefore keeps the observations of every dense block (and the
    // block's dispersion) in column order; e->ip_old2new maps the caller's observation index for get_rows().
    {
        const int NOCOL = 1 << 30;
        auto key = [&](int ip) {
            int k = NOCOL;
            for (int a = 0; a < 3; a++) {
                const int c = D->point_col[3 * D->ip_point[ip] + a];
                if (c >= 0) k = std::min(k, c);
            }
            return k;
        };
        std::vector<int32_t> new2old(D->n_image_points);
        for (int ip = 0; ip < D->n_image_points; ip++) new2old[ip] = ip;
        bool permuted = false;
        for (int g : blk_list) {
            const int b = D->blk_ip_begin[g], en = D->blk_ip_begin[g + 1];
            std::vector<int> k(en - b);
            for (int ip = b; ip < en; ip++) k[ip - b] = key(ip);
            std::stable_sort(new2old.begin() + b, new2old.begin() + en, [&](int x, int y) { return k[x - b] < k[y - b]; });
            for (int ip = b; ip < en && !permuted; ip++) permuted = new2old[ip] != ip;
        }
        if (permuted) {
            const size_t n = (size_t)D->n_image_points;
            pv_image.resize(n); pv_point.resize(n); pv_x.resize(n); pv_y.resize(n); pv_vx.resize(n); pv_vy.resize(n); pv_rho.resize(n);
            perm_local.resize(n);
            e->ip_old2new.resize(n);
            for (size_t i = 0; i < n; i++) {
                const int o = new2old[i];
                pv_image[i] = D->ip_image[o]; pv_point[i] = D->ip_point[o]; pv_x[i] = D->ip_x[o]; pv_y[i] = D->ip_y[o];
                pv_vx[i] = D->ip_var_x[o]; pv_vy[i] = D->ip_var_y[o]; pv_rho[i] = D->ip_rho[o];
                e->ip_old2new[o] = (int32_t)i;
            }
            for (int g : blk_list)
                for (int ip = D->blk_ip_begin[g]; ip < D->blk_ip_begin[g + 1]; ip++) perm_local[ip] = new2old[ip] - D->blk_ip_begin[g];
            Dperm = *D;
            Dperm.ip_image = pv_image.data(); Dperm.ip_point = pv_point.data(); Dperm.ip_x = pv_x.data(); Dperm.ip_y = pv_y.data();
            Dperm.ip_var_x = pv_vx.data(); Dperm.ip_var_y = pv_vy.data(); Dperm.ip_rho = pv_rho.data();
            D = &Dperm;
        }
    }
    for (int g : blk_list) e->h_blk_images.push_back(D->ip_image[D->blk_ip_begin[g]]);
    e->n_seg = (int)seg_b.size();
    e->n_blk_list = (int)blk_list.size();
    e->n_blk_ip = (int)blk_ip_list.size();

    // ---- upload structure --------------------------------------------------------------------------------------
    const auto t_phase0 = std::chrono::steady_clock::now();
    DevProblem &p = e->p;
    p.U = U; p.Upad = e->Upad; p.d = d; p.ld = e->Upad;
    p.n_points = D->n_points; p.n_cameras = D->n_cameras; p.n_images = D->n_images; p.n_dist = D->n_dist;
    p.n_ip = D->n_image_points; p.n_blocks = D->n_image_blocks; p.n_sb = D->n_scale_bars; p.n_dg = D->n_direct_groups;
    p.n_dg_rows = D->n_direct_rows; p.n_slots = e->n_slots;
    int rc;
#define UP(field, src, cnt) if ((rc = upload(e, src, (size_t)(cnt), &p.field)) != JAICOV_OK) return rc
    UP(point_col, D->point_col, 3 * D->n_points);
    UP(io_col, D->io_col, 3 * D->n_cameras);
    UP(cam_dist_begin, D->cam_dist_begin, D->n_cameras + 1);
    UP(dist_kind, D->dist_kind, D->n_dist);
    UP(dist_order, D->dist_order, D->n_dist);
    UP(dist_col, D->dist_col, D->n_dist);
    UP(image_camera, D->image_camera, D->n_images);
    UP(eo_col, D->eo_col, 6 * D->n_images);
    UP(cam_r0, D->cam_r0, D->n_cameras);
    UP(ip_image, D->ip_image, D->n_image_points);
    UP(ip_point, D->ip_point, D->n_image_points);
    UP(ip_x, D->ip_x, D->n_image_points);
    UP(ip_y, D->ip_y, D->n_image_points);
    UP(ip_var_x, D->ip_var_x, D->n_image_points);
    UP(ip_var_y, D->ip_var_y, D->n_image_points);
    UP(ip_rho, D->ip_rho, D->n_image_points);
    UP(blk_ip_begin, D->blk_ip_begin, D->n_image_blocks + 1);
    UP(blk_w_offset, blk_w_off.data(), D->n_image_blocks);
    UP(sb_a, D->sb_point_a, D->n_scale_bars);
    UP(sb_b, D->sb_point_b, D->n_scale_bars);
    UP(sb_len, D->sb_length, D->n_scale_bars);
    UP(sb_var, D->sb_var, D->n_scale_bars);
    UP(dg_row_begin, D->dg_row_begin, D->n_direct_groups + 1);
    UP(dg_slot, D->dg_slot, D->n_direct_rows);
    UP(dg_obs, D->dg_obs, D->n_direct_rows);
    UP(dg_var, D->dg_var, D->n_direct_rows);
    UP(slot_col, e->h_slot_col.data(), e->n_slots);
#undef UP
    const int32_t *tmp32; const uint8_t *tmp8;
    if ((rc = upload(e, seg_b.data(), seg_b.size(), &tmp32))) return rc; e->d_seg_begin = (int32_t *)tmp32;
    if ((rc = upload(e, seg_e.data(), seg_e.size(), &tmp32))) return rc; e->d_seg_end = (int32_t *)tmp32;
    if ((rc = upload(e, blk_list.data(), blk_list.size(), &tmp32))) return rc; e->d_blk_list = (int32_t *)tmp32;
    if ((rc = upload(e, blk_ip_list.data(), blk_ip_list.size(), &tmp32))) return rc; e->d_blk_ip_list = (int32_t *)tmp32;
    if ((rc = upload(e, in_block.data(), in_block.size(), &tmp8))) return rc; e->d_in_block = (uint8_t *)tmp8;
    if (!blk_ip_list.empty()) {
        // point -> block image points (CSR, image order) for the atomics-free point x point gather
        std::vector<int32_t> cnt(D->n_points + 1, 0), blk_of_ip(D->n_image_points, -1);
        for (size_t t = 0; t < blk_list.size(); t++)
            for (int ip = D->blk_ip_begin[blk_list[t]]; ip < D->blk_ip_begin[blk_list[t] + 1]; ip++) blk_of_ip[ip] = blk_list[t];
        for (int ip : blk_ip_list) cnt[D->ip_point[ip] + 1]++;
        for (int i = 0; i < D->n_points; i++) cnt[i + 1] += cnt[i];
        std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1), list(blk_ip_list.size());
        int cmin = 1 << 30, cmax = -1;
        for (int ip : blk_ip_list) {
            const int pt = D->ip_point[ip];
            list[fill[pt]++] = ip;
            for (int a = 0; a < 3; a++) {
                const int c = D->point_col[3 * pt + a];
                if (c >= 0) { cmin = std::min(cmin, c); cmax = std::max(cmax, c); }
            }
        }
        if (cmax >= cmin) {
            std::vector<PPRecord> recs(list.size());
            for (size_t t = 0; t < list.size(); t++) {
                const int ip = list[t], g = blk_of_ip[ip];
                PPRecord r{};
                r.ipb = D->blk_ip_begin[g]; r.mp = D->blk_ip_begin[g + 1] - r.ipb; r.lp = ip - r.ipb; r.poff = blk_w_off[g];
                recs[t] = r;
            }
            std::vector<int32_t> ipcol((size_t)3 * D->n_image_points, -1);
            for (int ip : blk_ip_list)
                for (int a = 0; a < 3; a++) ipcol[(size_t)3 * ip + a] = D->point_col[3 * D->ip_point[ip] + a];
            // per record and column chunk: the range of partner positions (the block is stored in column order)
            int cw_rt = std::max(64, std::min(PP_CW, 6400)) / 4 * 4;  // 3 * cw doubles of LDS (<= 150 KB), quarters for the DET form
            // XCD-partitioned block order (assemble.hip): measured no faster (3.24 vs 3.26 ms at 960 columns, slower where the chunks
            // do not divide evenly over eight XCDs) and FETCH_SIZE fell by 6 % only: the partner records are not what the kernel waits for
            e->pp.xcd_map = 0;
            const int NOCOL = 1 << 30;
            std::vector<int32_t> lo_col(D->n_image_points, NOCOL), hi_col(D->n_image_points, -1);
            for (int ip : blk_ip_list)
                for (int a = 0; a < 3; a++) {
                    const int c = ipcol[(size_t)3 * ip + a];
                    if (c >= 0) { lo_col[ip] = std::min(lo_col[ip], c); hi_col[ip] = std::max(hi_col[ip], c); }
                }
            // The strip width follows the scene: a wave takes the partners of ONE image inside the strip's columns, 64 at a time.  With
            // random visibility the default width holds 55 +- 7 of an image's 500 points; on a block flown in strips the
            // points of an image are neighbours in column order and sit in two or three strips, hundreds in each -- eight segments of one
            // image inside one turn of the deterministic form.  Then the strip is narrowed until an image's partners in a strip are about
            // one wave again (never below 256 columns: the range table has one entry per image point and strip).
            {
                std::vector<int32_t> per_block;                      // columns spanned by the partners of a block, and how many there are
                double span = 0.0, cnt = 0.0;
                for (size_t t = 0; t < blk_list.size(); t++) {
                    const int g = blk_list[t], ipb = D->blk_ip_begin[g], mp = D->blk_ip_begin[g + 1] - ipb;
                    // distinct default-width strips touched by this block's points
                    int touched = 0, last = -1;
                    for (int j = 0; j < mp; j++) {
                        if (lo_col[ipb + j] == NOCOL) continue;
                        const int c = (lo_col[ipb + j] - cmin) / cw_rt;
                        if (c != last) { ++touched; last = c; }
                    }
                    if (touched > 0) { span += touched; cnt += mp; }
                }
                const double per_strip = span > 0 ? cnt / span : 0.0;   // mean partners of an image per touched strip
                if (per_strip > 96.0) {
                    int cw2 = (int)(cw_rt * 56.0 / per_strip) / 64 * 64;
                    cw_rt = std::max(256, std::min(cw_rt, cw2));
                }
            }
            e->pp.cw = cw_rt;
            const int n_chunks = (cmax - cmin + cw_rt) / cw_rt;
            std::vector<int32_t> chunk_lo((size_t)blk_list.size() * n_chunks), chunk_hi((size_t)blk_list.size() * n_chunks);
            std::vector<int32_t> blk_pos(D->n_image_blocks, -1);
            for (size_t t = 0; t < blk_list.size(); t++) {
                const int g = blk_list[t], ipb = D->blk_ip_begin[g], mp = D->blk_ip_begin[g + 1] - ipb;
                blk_pos[g] = (int)t;
                for (int j = 1; j < mp; j++)
                    if (lo_col[ipb + j] < lo_col[ipb + j - 1]) FAIL(e, JAICOV_ERR_DEVICE, "internal: dense block not in column order");
                for (int c = 0; c < n_chunks; c++) {
                    const int c0 = cmin + c * cw_rt, c1 = c0 + cw_rt;
                    int lo = mp, hi = 0;
                    for (int j = 0; j < mp; j++)
                        if (hi_col[ipb + j] >= c0 && lo_col[ipb + j] < c1) { lo = std::min(lo, j); hi = j + 1; }
                    chunk_lo[t * n_chunks + c] = lo; chunk_hi[t * n_chunks + c] = std::max(hi, lo);
                }
            }
            std::vector<int32_t> range((size_t)2 * list.size() * n_chunks);
            for (size_t o = 0; o < list.size(); o++) {
                const int ip = list[o], g = blk_of_ip[ip], ipb = D->blk_ip_begin[g], mp = D->blk_ip_begin[g + 1] - ipb;
                const int t = blk_pos[g];
                // partners whose first column is <= the largest row column of this point: a prefix of the block
                const int qend = (int)(std::upper_bound(lo_col.begin() + ipb, lo_col.begin() + ipb + mp, hi_col[ip]) - (lo_col.begin() + ipb));
                for (int c = 0; c < n_chunks; c++) {
                    const int lo = chunk_lo[(size_t)t * n_chunks + c], hi = std::min(chunk_hi[(size_t)t * n_chunks + c], qend);
                    range[2 * (o * n_chunks + c)] = lo;
                    range[2 * (o * n_chunks + c) + 1] = std::max(hi, lo);
                }
            }
            if ((rc = upload(e, cnt.data(), cnt.size(), &e->pp.pt_ip_begin))) return rc;
            if ((rc = upload(e, recs.data(), recs.size(), &e->pp.recs))) return rc;
            {   // the gather reads the columns as three arrays over the image points (coalesced like A_q and U_q)
                std::vector<int32_t> soa(ipcol.size());
                const size_t S = (size_t)D->n_image_points;
                for (size_t ip = 0; ip < S; ip++)
                    for (int a = 0; a < 3; a++) soa[(size_t)a * S + ip] = ipcol[3 * ip + a];
                if ((rc = upload(e, soa.data(), soa.size(), &e->pp.ipcol))) return rc;
            }
            if ((rc = upload(e, range.data(), range.size(), &e->pp.range))) return rc;
            e->pp.det = e->deterministic ? 1 : 0;
            e->pp.cmin = cmin;
            e->pp.n_chunks = n_chunks;
            e->pp.cmax = cmax;
            {   // rows cmin..cmax all point rows?  (points are numbered first and contiguously, BA:667-782)
                std::vector<char> is_pt(cmax - cmin + 1, 0);
                for (int i = 0; i < 3 * D->n_points; i++) {
                    const int c = D->point_col[i];
                    if (c >= cmin && c <= cmax) is_pt[c - cmin] = 1;
                }
                bool all = true;
                for (char f : is_pt) all = all && f;
                e->pp_plain_ok = all;
            }
        }
    }

    e->create_ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_phase0).count();
    // ---- dense dispersions -> D^-1 on the device (DOPG:82-86: dpptrf + dpptri once, cached) -----------------------
    {
        double *d_w = nullptr;
        if ((rc = dalloc(e, (size_t)w_total, &d_w))) return rc;
        p.blk_w = d_w;
        std::vector<int64_t> dg_w_off(D->n_direct_groups + 1, -1);
        int64_t dg_total = 0;
        int max_dm = 0;
        if (e->opts.apply_shared)
            for (int g = 0; g < D->n_direct_groups; g++) {
                const int m = D->dg_row_begin[g + 1] - D->dg_row_begin[g];
                if (D->dg_disp_offset && D->dg_disp_offset[g] >= 0 && m > 0) {
                    dg_w_off[g] = dg_total;
                    dg_total += (int64_t)m * m;
                    max_dm = std::max(max_dm, m);
                }
            }
        double *d_dgw = nullptr;
        if ((rc = dalloc(e, (size_t)dg_total, &d_dgw))) return rc;
        p.dg_w = d_dgw;
        if ((rc = upload(e, dg_w_off.data(), (size_t)D->n_direct_groups, &p.dg_w_offset))) return rc;
        {
            std::vector<DispItem> items;
            const int32_t *d_perm_local = nullptr;
            if (!perm_local.empty() && (rc = upload(e, perm_local.data(), perm_local.size(), &d_perm_local))) return rc;
            for (int g : blk_list) {
                const int m = 2 * (D->blk_ip_begin[g + 1] - D->blk_ip_begin[g]);
                if (D->blk_disp_offset[g] < 0 && blk_w_off[g] < 0) continue;      // compact form: filled below for all image points at once
                if (D->blk_disp_offset[g] < 0) {      // an ordinary image: inv(D) = diag of 2 x 2 blocks in closed form, engine order (PDF:296-319)
                    hipLaunchKernelGGL(fill_diag_weight_kernel, dim3(m), dim3(256), 0, e->stream, p.ip_var_x, p.ip_var_y, p.ip_rho,
                                       D->blk_ip_begin[g], m, d_w + blk_w_off[g]);
                    continue;
                }
                items.push_back(DispItem{D->blk_disp + D->blk_disp_offset[g], d_w + blk_w_off[g], d_perm_local ? d_perm_local + D->blk_ip_begin[g] : nullptr, m});
            }
            for (int g = 0; g < D->n_direct_groups; g++) {
                if (dg_w_off[g] < 0) continue;
                items.push_back(DispItem{D->dg_disp + D->dg_disp_offset[g], d_dgw + dg_w_off[g], nullptr, D->dg_row_begin[g + 1] - D->dg_row_begin[g]});
            }
            (void)max_dm;
            e->h_perm_local = perm_local;
            e->h_blk_w_off.assign(blk_w_off.begin(), blk_w_off.end());
            e->h_blk_ip_begin.assign(D->blk_ip_begin, D->blk_ip_begin + D->n_image_blocks + 1);
            e->h_blk_mine.assign(D->n_image_blocks, 0);
            for (int g : blk_list) e->h_blk_mine[g] = 1;
            if (e->compact_blocks) {
                double *d_w3 = nullptr;
                if ((rc = dalloc(e, (size_t)3 * D->n_image_points, &d_w3))) return rc;
                hipLaunchKernelGGL(fill_ip_w3_kernel, dim3((D->n_image_points + 255) / 256), dim3(256), 0, e->stream, p.ip_var_x, p.ip_var_y, p.ip_rho,
                                   D->n_image_points, d_w3);
                p.ip_w3 = d_w3;
            }
            if ((rc = invert_dispersions(e, items))) return rc;
        }
    }

    // ---- work buffers ------------------------------------------------------------------------------------------
    const auto t_phase1 = std::chrono::steady_clock::now();
    const size_t sq = (size_t)e->Upad * e->Upad;
    if ((rc = dalloc(e, (size_t)e->n_slots, &e->d_vals))) return rc;
    if ((rc = dalloc(e, (size_t)2 * KROW * std::max(1, D->n_image_points), &e->d_rowsA, true))) return rc;
    if ((rc = dalloc(e, (size_t)2 * std::max(1, D->n_image_points), &e->d_rowsW, true))) return rc;
    if ((rc = dalloc(e, (size_t)2 * std::max(1, D->n_image_points) * KC_LD, &e->d_T))) return rc;
    if ((rc = dalloc(e, (size_t)2 * std::max(1, D->n_image_points), &e->d_vbuf))) return rc;
    if ((rc = dalloc(e, sq + e->Upad, &e->d_N))) return rc;
    e->d_n = e->d_N + sq;
    if ((rc = dalloc(e, (size_t)e->Upad, &e->d_V))) return rc;
    if ((rc = dalloc(e, (size_t)8 * e->Upad, &e->d_B, true))) return rc;
    if ((rc = dalloc(e, (size_t)e->Upad, &e->d_dx, true))) return rc;
    if ((rc = dalloc(e, (size_t)8 * e->Upad, &e->d_G, true))) return rc;
    if ((rc = dalloc(e, (size_t)8 * e->Upad, &e->d_Braw, true))) return rc;
    if ((rc = dalloc(e, (size_t)3 * e->Upad, &e->d_ref, true))) return rc;
    if ((rc = dalloc(e, (size_t)8 * e->Upad, &e->d_H, true))) return rc;
    if ((rc = dalloc(e, (size_t)8 * e->Upad, &e->d_F, true))) return rc;
    if ((rc = dalloc(e, (size_t)64, &e->d_E, true))) return rc;
    if ((rc = dalloc(e, (size_t)1, &e->d_omega, true))) return rc;
    if ((rc = dalloc(e, (size_t)std::max(1, e->n_blk_list) * 16 * KC_MAX * (KC_MAX + 1), &e->d_cc_partial))) return rc;
    HIPE(e, e->solver.init(e->stream, e->Upad, false, true));
    e->create_ms[4] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_phase1).count();
    const auto t_phase2 = std::chrono::steady_clock::now();
    // ---- EO pre-elimination is possible when every image point sits in an image block, the EO columns are the
    //      trailing columns e0 + 6*image + k, and no directly observed parameter is an EO parameter ------------------
    {
        // Eligibility is a property of the WHOLE problem, not of this engine's image range: every rank of a sharded run must
        // reach the same decision, or their reduce buffers differ in length (and only some ranks enter the EO exchange).
        size_t n_in_block = 0;
        for (int ip = 0; ip < D->n_image_points; ip++) n_in_block += in_block[ip] ? 1 : 0;
        bool ok = D->n_images > 0 && D->n_image_blocks > 0 && n_in_block == (size_t)D->n_image_points;
        if ((e->opts.assembly_mode == 1 || e->opts.assembly_mode == 2) && e->n_blk_list > 0) {   // densified MFMA contraction of the image groups (densemode.hip)
            int max_k1 = 0;
            for (int g : blk_list) {
                const int mp = D->blk_ip_begin[g + 1] - D->blk_ip_begin[g], cam = D->image_camera[D->ip_image[D->blk_ip_begin[g]]];
                const int k1 = 3 * mp + 9 + D->cam_dist_begin[cam + 1] - D->cam_dist_begin[cam] + 1;
                max_k1 = std::max(max_k1, k1);
                e->dm_flops_per_pass += 2.0 * (2.0 * mp) * (2.0 * mp) * k1 + (2.0 * mp) * k1 * (k1 + 1.0);
            }
            HIPE(e, e->dm.init(e->max_m, max_k1, e->n_blk_list, e->opts.assembly_mode == 2));
            e->dense_mode = true;
            ok = false;
        } else if (e->opts.assembly_mode < 0 || e->opts.assembly_mode > 2)
            FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "assembly_mode must be 0 (structure-aware), 1 (densified fp64 MFMA contraction) or 2 (the same with fp32 operands and accumulation)");
        const int e0 = D->n_images > 0 ? D->eo_col[0] : -1;
        ok = ok && e0 >= d && e0 + 6 * D->n_images == U;
        for (int i = 0; ok && i < 6 * D->n_images; i++) ok = D->eo_col[i] == e0 + i;
        const int s_eo = 3 * D->n_points + 3 * D->n_cameras + D->n_dist;
        for (int r = 0; ok && r < D->n_direct_rows; r++) ok = D->dg_slot[r] < s_eo;
        for (int g = 0; ok && g < D->n_image_blocks; g++) ok = D->blk_ip_begin[g + 1] - D->blk_ip_begin[g] >= 3;
        if (ok) {
            if ((rc = dalloc(e, (size_t)16 * std::max(1, D->n_image_points), &e->sb.U, true))) return rc;
            if ((rc = dalloc(e, (size_t)12 * std::max(1, D->n_image_points), &e->sb.Ug, true))) return rc;   // U again, in the gather's layout
            if ((rc = dalloc(e, (size_t)36 * D->n_images, &e->sb.Linv, true))) return rc;
            if ((rc = dalloc(e, (size_t)6 * SCHUR_GLD * D->n_images, &e->sb.G, true))) return rc;
            // P' = sigma2 Dinv - U U' is formed inside the point x point gather; a copy in memory (4 GB at config 4) only on request
            e->sb.materialise = assembly_form() == ASSEMBLY_MATERIALISE ? 1 : 0;     // test hook (JAICOV_ASSEMBLY_FORM, assemble.hip)
            if (e->sb.materialise && (rc = dalloc(e, (size_t)std::max<int64_t>(w_total_saved, 1), &e->sb.Pp))) return rc;
            if ((rc = dalloc(e, (size_t)6 * D->n_images, &e->d_xE, true))) return rc;
            if (e->opts.reduced_reference_quirk && (rc = dalloc(e, (size_t)6 * D->n_images, &e->sb.xq, true))) return rc;
            if ((rc = dalloc(e, (size_t)1, &e->sb.info, true))) return rc;
            if ((rc = dalloc(e, (size_t)e->Upad, &e->sb.diagcorr, true))) return rc;
            e->schur_ok = true;
            e->e0 = e0;
            // the solver of the reduced system is created here, with its streams, not at the first solve: stream creation
            // order matters when the host also runs RCCL (bench.py: communicator after the engine)
            HIPE(e, e->solverS.init(e->stream, ((e0 + 127) / 128) * 128, false, true, &e->solver));      // never at work beside the full-order solver: shares its side streams
            e->solverS_ready = true;
        }
    }
    e->create_ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_phase2).count();
    e->h_vals.assign(e->n_slots, 0.0);
    e->h_V.assign(e->Upad, 1.0);
    e->hB.assign((size_t)8 * e->Upad, 0.0);
    HIPE(e, hipStreamSynchronize(e->stream));
    return JAICOV_OK;
}

extern "C" int jaicov_neq_create(const jaicov_problem_desc *desc, const jaicov_engine_options *opts, jaicov_engine **out) {
    if (!out) return JAICOV_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (!desc || desc->struct_size != sizeof(jaicov_problem_desc)) return JAICOV_ERR_BAD_ARGUMENT;
    std::string err;
    int rc = check_device(err);
    if (rc != JAICOV_OK) return rc;
    jaicov_engine *e = new jaicov_engine();
    for (auto &evt : e->ev) evt = nullptr;
    if (opts) {
        if (opts->struct_size != sizeof(jaicov_engine_options)) { delete e; return JAICOV_ERR_BAD_ARGUMENT; }
        e->opts = *opts;
        e->device = opts->device;
    } else {
        e->opts.image_begin = e->opts.image_end = -1;
        e->opts.apply_shared = 1;
    }
    *out = e;     // returned even on failure so that the caller can read jaicov_neq_last_error(); destroy() frees it
    const auto t0 = std::chrono::steady_clock::now();
    rc = create_impl(e, desc, opts);
    e->create_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" size_t jaicov_neq_num_slots(const jaicov_engine *e) { return e ? (size_t)e->n_slots : 0; }
extern "C" size_t jaicov_neq_packed_length(const jaicov_engine *e) { return e ? (size_t)e->U * (e->U + 1) / 2 : 0; }

extern "C" int jaicov_neq_set_parameters(jaicov_engine *e, const double *slots, size_t n) {
    if (!e || !slots || n != (size_t)e->n_slots) return JAICOV_ERR_BAD_ARGUMENT;
    HIPE(e, hipSetDevice(e->device));
    memcpy(e->h_vals.data(), slots, n * sizeof(double));
    HIPE(e, hipMemcpyAsync(e->d_vals, e->h_vals.data(), n * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPE(e, hipStreamSynchronize(e->stream));
    e->rows_valid = false;
    if (e->state == jaicov_engine::ST_NEW) e->state = jaicov_engine::ST_PARAMS;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_get_parameters(jaicov_engine *e, double *slots, size_t n) {
    if (!e || !slots || n != (size_t)e->n_slots) return JAICOV_ERR_BAD_ARGUMENT;
    memcpy(slots, e->h_vals.data(), n * sizeof(double));
    return JAICOV_OK;
}

static int ensure_rows(jaicov_engine *e) {
    if (e->rows_valid) return JAICOV_OK;
    HIPE(e, launch_rows(e->stream, e->p, e->d_vals, e->ip0, e->ip_count, e->d_rowsA, e->d_rowsW));
    e->rows_valid = true;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_accumulate(jaicov_engine *e, double sigma2) {
    if (!e) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state == jaicov_engine::ST_NEW) FAIL(e, JAICOV_ERR_BAD_STATE, "set_parameters first");
    if (!(sigma2 > 0)) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "variance of unit weight must be positive (DOPG:68-69)");
    HIPE(e, hipSetDevice(e->device));
    e->reduced = false;          // a packed buffer of an earlier accumulate must not be unpacked over this one
    const size_t sq = (size_t)e->Upad * e->Upad;
    HIPE(e, hipEventRecord(e->ev[0], e->stream));
    int rc = ensure_rows(e);
    if (rc) return rc;
    HIPE(e, hipEventRecord(e->ev[1], e->stream));
    e->schur_active = e->schur_ok && e->inverse_mode_next != JAICOV_INVERT_FULL;
    // The point x point gather owns every entry of rows cmin..cmax from column cmin on: when it runs before everything
    // else that adds into that region it stores its strips, and the region needs neither zeroing nor reading back.
    const bool plain = e->pp_plain_ok && !e->dense_mode && e->pp.pt_ip_begin != nullptr && e->n_blk_list > 0;
    {   // only the lower triangle of N is ever written or read (nadd, pack, scale_copy); the reduced system has e0 rows
        const int rows = e->schur_active ? std::min(e->Upad, ((e->e0 + 127) / 128) * 128) : e->Upad;
        // three launches, each over the rows and columns that need zeroing only (one launch over the whole square spent 60 us at
        // config 4 dispatching 226 000 workgroups of which 99.9 % returned at once): rows above the stored strips, the stored rows'
        // columns before the first stored column (nothing when the points come first and there is no datum border), rows below
        const int slo = plain ? e->pp.cmin : rows, shi = plain ? e->pp.cmax : -1;
        auto zero_rows = [&](int r0, int r1, int cols) {
            if (r1 > r0 && cols > 0)
                hipLaunchKernelGGL(zero_lower_kernel, dim3((cols + 1023) / 1024, r1 - r0), dim3(256), 0, e->stream, e->d_N, (long)e->Upad, rows, slo, shi, r0);
        };
        zero_rows(0, std::min(slo, rows), std::min(slo, rows));
        if (plain) {
            zero_rows(slo, std::min(shi + 1, rows), slo);
            zero_rows(std::min(shi + 1, rows), rows, rows);
        }
        HIPE(e, hipMemsetAsync(e->d_N + sq, 0, (size_t)e->Upad * sizeof(double), e->stream));
    }
    if (!plain)
        HIPE(e, launch_assemble_small(e->stream, e->p, e->d_seg_begin, e->d_seg_end, e->n_seg, e->d_rowsA, e->d_rowsW, sigma2, e->d_N, e->d_n));
    SchurBufs sb = e->sb;
    sb.active = e->schur_active;
    sb.lambda = e->lambda_acc;
    if (e->schur_active) {
        HIPE(e, hipMemsetAsync(e->sb.info, 0, sizeof(int), e->stream));
        HIPE(e, hipMemsetAsync(e->sb.diagcorr, 0, (size_t)e->Upad * sizeof(double), e->stream));
    }
    if (e->dense_mode) {
        float ms = 0.f;
        HIPE(e, e->dm.assemble(e->stream, e->p, e->d_blk_list, e->n_blk_list, e->d_rowsA, e->d_rowsW, sigma2, e->d_N, e->d_n,
                               e->solver.profile ? &ms : nullptr));
        if (e->solver.profile) { e->dm_stat_passes += 1.0; e->dm_stat_ms += ms; e->dm_stat_flops += e->dm_flops_per_pass; }
    } else {
        PPGather ppg = e->pp;
        ppg.plain = plain ? 1 : 0;
        ppg.ug = sb.active ? sb.Ug : nullptr;
        HIPE(e, launch_assemble_blocks(e->stream, e->p, e->d_blk_list, e->n_blk_list, e->max_m, e->d_blk_ip_list, e->n_blk_ip,
                                       e->d_rowsA, e->d_rowsW, e->d_T, sigma2, e->d_N, e->d_n, ppg, sb, e->d_cc_partial,
                                       e->solver.pstream, e->ev_fork, e->ev_join));   // side stream: the solver's (idle during the assembly)
    }
    if (plain)   // the images outside the dense blocks come after the stores
        HIPE(e, launch_assemble_small(e->stream, e->p, e->d_seg_begin, e->d_seg_end, e->n_seg, e->d_rowsA, e->d_rowsW, sigma2, e->d_N, e->d_n));
    if (e->opts.apply_shared) HIPE(e, launch_shared_groups(e->stream, e->p, e->d_vals, sigma2, e->d_N, e->d_n, nullptr, nullptr));
    HIPE(e, hipEventRecord(e->ev[2], e->stream));
    e->exp_ready = false;            // the expansion buffer of an earlier pass is stale now
    e->state = jaicov_engine::ST_ACCUMULATED;
    e->have_Q = false;
    return JAICOV_OK;
}

// datum rows on the host (BA:493-635) from the host copy of the parameter values
static int datum_rows_host(jaicov_engine *e) {
    const int d = e->d, Upad = e->Upad;
    std::fill(e->hB.begin(), e->hB.end(), 0.0);
    if (d == 0) return JAICOV_OK;
    double x0 = 0, y0 = 0, z0 = 0;
    int count = 0;
    for (int pt = 0; pt < e->n_points; pt++) {
        const int32_t *c = &e->h_point_col[3 * pt];
        if (!e->h_point_datum[pt] || c[0] < 0 || c[1] < 0 || c[2] < 0) continue;
        x0 += e->h_vals[3 * pt]; y0 += e->h_vals[3 * pt + 1]; z0 += e->h_vals[3 * pt + 2];
        count++;
    }
    if (count < 3) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "not enough object points to realise the frame datum (BA:515-516)");
    x0 /= (double)count; y0 /= (double)count; z0 /= (double)count;
    int row = 0;
    const int f = e->datum_flags;
    const int tx = (f & JAICOV_DATUM_TX) ? row++ : -1, ty = (f & JAICOV_DATUM_TY) ? row++ : -1, tz = (f & JAICOV_DATUM_TZ) ? row++ : -1;
    const int rx = (f & JAICOV_DATUM_RX) ? row++ : -1, ry = (f & JAICOV_DATUM_RY) ? row++ : -1, rz = (f & JAICOV_DATUM_RZ) ? row++ : -1;
    const int ms = (f & JAICOV_DATUM_SCALE) ? row++ : -1;
    double norm[7] = {0, 0, 0, 0, 0, 0, 0};
    auto B = [&](int r, int c) -> double & { return e->hB[(size_t)r * Upad + c]; };
    for (int pt = 0; pt < e->n_points; pt++) {
        const int32_t *c = &e->h_point_col[3 * pt];
        if (!e->h_point_datum[pt] || c[0] < 0 || c[1] < 0 || c[2] < 0) continue;
        const double x = e->h_vals[3 * pt] - x0, y = e->h_vals[3 * pt + 1] - y0, z = e->h_vals[3 * pt + 2] - z0;
        if (tx >= 0) { B(tx, c[0]) = 1.0; norm[tx] += 1.0; }
        if (ty >= 0) { B(ty, c[1]) = 1.0; norm[ty] += 1.0; }
        if (tz >= 0) { B(tz, c[2]) = 1.0; norm[tz] += 1.0; }
        if (rx >= 0) { B(rx, c[1]) = z; B(rx, c[2]) = -y; norm[rx] += z * z + y * y; }
        if (ry >= 0) { B(ry, c[0]) = -z; B(ry, c[2]) = x; norm[ry] += z * z + x * x; }
        if (rz >= 0) { B(rz, c[0]) = y; B(rz, c[1]) = -x; norm[rz] += x * x + y * y; }
        if (ms >= 0) { B(ms, c[0]) = x; B(ms, c[1]) = y; B(ms, c[2]) = z; norm[ms] += x * x + y * y + z * z; }
    }
    for (int r = 0; r < d; r++) {
        const double s = sqrt(norm[r]);
        for (int c = 0; c < e->U; c++)
            if (e->hB[(size_t)r * Upad + c] != 0.0) e->hB[(size_t)r * Upad + c] = e->hB[(size_t)r * Upad + c] / s;
    }
    return JAICOV_OK;
}

extern "C" int jaicov_neq_finalize(jaicov_engine *e, double sigma2, double lambda, int simulation) {
    if (!e) return JAICOV_ERR_BAD_ARGUMENT;
    (void)sigma2;
    if (e->state != jaicov_engine::ST_ACCUMULATED) FAIL(e, JAICOV_ERR_BAD_STATE, "accumulate first");
    HIPE(e, hipSetDevice(e->device));
    int rc = datum_rows_host(e);
    if (rc) return rc;
    if (e->reduced) {   // the host has summed the packed buffer over ranks: bring it back into the square
        const int Ua = e->schur_active ? e->e0 : e->U;
        const size_t len = (size_t)Ua * (Ua + 1) / 2;
        hipLaunchKernelGGL(unpack_kernel, dim3((Ua + 255) / 256, std::max(Ua, 1)), dim3(256), 0, e->stream, e->d_packed, (long)e->Upad, Ua, e->d_N);
        HIPE(e, hipMemcpyAsync(e->d_n, e->d_packed + len, (size_t)Ua * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        if (e->schur_active)   // what the EO elimination took from the diagonal, summed over the ranks like N itself (LM damping)
            HIPE(e, hipMemcpyAsync(e->sb.diagcorr, e->d_packed + len + Ua, (size_t)Ua * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        e->reduced = false;
    }
    if (e->schur_active && lambda != e->lambda_acc && (lambda > 0 || e->lambda_acc > 0))
        FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "EO pre-elimination: pass the damping value to accumulate2()/build() as well");
    hipLaunchKernelGGL(damp_and_precond_kernel, dim3((e->Upad + 255) / 256), dim3(256), 0, e->stream, e->d_N, (long)e->Upad,
                       e->schur_active ? e->e0 : e->U, e->Upad, e->d, lambda > 0 ? lambda : 0.0, e->d_V,
                       e->schur_active ? e->sb.diagcorr : nullptr);
    if (simulation) HIPE(e, hipMemsetAsync(e->d_n, 0, e->Upad * sizeof(double), e->stream));   // BA:830-831
    e->sim_built = simulation != 0;
    HIPE(e, hipEventRecord(e->ev[3], e->stream));
    e->lambda_used = lambda;
    e->state = jaicov_engine::ST_BUILT;
    return JAICOV_OK;
}

// buffers of the inverse (W = L^-1 and Q, two squares of the solver's order) of the solver that the announced mode will use
static int ensure_inverse_buffers(jaicov_engine *e, bool reduced_system) {
    DenseSolver &slv = reduced_system ? e->solverS : e->solver;
    bool &has_inv = reduced_system ? e->solverS_has_inverse : e->solver_has_inverse;
    if (has_inv) return JAICOV_OK;
    const size_t sq = (size_t)slv.n * slv.ld * sizeof(double);
    HIPE(e, hipMalloc(&slv.W, sq));
    if (!slv.Q) HIPE(e, hipMalloc(&slv.Q, sq));
    has_inv = true;
    return JAICOV_OK;
}

// FULL_EXPANDED is FULL wherever the expansion cannot be done: no EO pre-elimination, or an engine that sees a shard of the images
// and whose caller has not promised to sum the expansion's inputs over the ranks (engine option expansion_exchange)
static int effective_invert(const jaicov_engine *e, int invert) {
    if (invert != JAICOV_INVERT_FULL_EXPANDED) return invert;
    if (!(e->schur_ok && (e->all_images || e->opts.expansion_exchange != 0) && e->solverS_ready)) return JAICOV_INVERT_FULL;
    // the expansion's workspace is the reduced solver's W square: F and T1 ([6 images, padded] x order) and T2 must fit (they do
    // unless the exterior orientations outnumber the other unknowns several times over: then the literal route is taken)
    const size_t I6p = ((size_t)6 * e->p.n_images + 127) / 128 * 128, Up = (size_t)e->solverS.nfact;
    if (I6p * (2 * Up + I6p) > (size_t)e->solverS.n * e->solverS.ld) return JAICOV_INVERT_FULL;
    return invert;
}

extern "C" int jaicov_neq_prepare_inverse(jaicov_engine *e, int inverse_follows) {
    if (!e) return JAICOV_ERR_BAD_ARGUMENT;
    if (inverse_follows < 0 || inverse_follows > JAICOV_INVERT_FULL_EXPANDED) return JAICOV_ERR_BAD_ARGUMENT;
    inverse_follows = effective_invert(e, inverse_follows);
    e->inverse_mode_next = inverse_follows;
    // the inverse's buffers are allocated HERE, when the final pass is announced (BA:250: known before the build), not inside
    // the solve: hipMalloc of 2 x 1.8 .. 2.6 GB cost the first inverting solve 45 ms at config 4
    if (inverse_follows != JAICOV_INVERT_NONE) {
        HIPE(e, hipSetDevice(e->device));
        const bool reduced = e->schur_ok && (inverse_follows == JAICOV_INVERT_REDUCED || inverse_follows == JAICOV_INVERT_FULL_EXPANDED) && e->solverS_ready;
        int rc = ensure_inverse_buffers(e, reduced);
        if (rc) return rc;
        if (inverse_follows == JAICOV_INVERT_FULL_EXPANDED && !e->solver.Q) {     // the expanded matrix lives in the full-order solver's Q
            HIPE(e, hipMalloc(&e->solver.Q, (size_t)e->solver.n * e->solver.ld * sizeof(double)));
            e->solver_has_Q = true;
        }
    }
    return JAICOV_OK;
}

extern "C" int jaicov_neq_cofactor_order(const jaicov_engine *e) {
    if (!e || !e->have_Q) return -1;
    return e->q_order;
}

extern "C" int jaicov_neq_reduced_order(const jaicov_engine *e) {
    if (!e) return -1;
    return e->schur_active ? e->e0 : e->U;
}

extern "C" int jaicov_neq_accumulate2(jaicov_engine *e, double sigma2, double lambda) {
    if (!e) return JAICOV_ERR_BAD_ARGUMENT;
    e->lambda_acc = lambda > 0 ? lambda : 0.0;
    return jaicov_neq_accumulate(e, sigma2);
}

extern "C" int jaicov_neq_build(jaicov_engine *e, double sigma2, double lambda, int simulation) {
    if (!e) return JAICOV_ERR_BAD_ARGUMENT;
    e->lambda_acc = lambda > 0 ? lambda : 0.0;
    int rc = jaicov_neq_accumulate(e, sigma2);
    if (rc) return rc;
    return jaicov_neq_finalize(e, sigma2, lambda, simulation);
}

static int reduce_buffer_impl(jaicov_engine *e, void **device_ptr, size_t *count, bool sync) {
    if (!e || !device_ptr || !count) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state != jaicov_engine::ST_ACCUMULATED) FAIL(e, JAICOV_ERR_BAD_STATE, "accumulate first");
    HIPE(e, hipSetDevice(e->device));
    // packed lower triangle (== UPLO='U' packed) followed by n: one contiguous array the collective sums as it is.
    // With the EO pre-elimination only the reduced system (order e0) exists.
    const int Ua = e->schur_active ? e->e0 : e->U;
    const size_t len = (size_t)Ua * (Ua + 1) / 2;
    if (!e->d_packed) {
        int rc = dalloc(e, (size_t)e->U * (e->U + 1) / 2 + 2 * (size_t)e->U, &e->d_packed);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(pack_kernel, dim3((Ua + 255) / 256, std::max(Ua, 1)), dim3(256), 0, e->stream, e->d_N, (long)e->Upad, Ua, e->d_packed);
    HIPE(e, hipMemcpyAsync(e->d_packed + len, e->d_n, (size_t)Ua * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    // with the EO pre-elimination the LM damping needs diag(N) - diag(N_reduced), also a sum over the ranks' images
    if (e->schur_active)
        HIPE(e, hipMemcpyAsync(e->d_packed + len + Ua, e->sb.diagcorr, (size_t)Ua * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    if (sync) HIPE(e, hipStreamSynchronize(e->stream));   // the caller's collective runs on its own stream
    e->reduced = true;
    *device_ptr = e->d_packed;
    *count = len + Ua + (e->schur_active ? (size_t)Ua : 0);
    return JAICOV_OK;
}

extern "C" int jaicov_neq_reduce_buffer(jaicov_engine *e, void **device_ptr, size_t *count) {
    return reduce_buffer_impl(e, device_ptr, count, true);
}

extern "C" int jaicov_neq_reduce_buffer_async(jaicov_engine *e, void **device_ptr, size_t *count, void **stream) {
    if (!stream) return JAICOV_ERR_BAD_ARGUMENT;
    const int rc = reduce_buffer_impl(e, device_ptr, count, false);
    if (rc == JAICOV_OK) *stream = (void *)e->stream;
    return rc;
}

// small dense helpers on the host (d <= 7)
static bool small_inverse(int d, const double *S, double *Sinv) {
    double a[7][14];
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) { a[i][j] = S[i * d + j]; a[i][d + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < d; c++) {
        int piv = c;
        for (int r = c + 1; r < d; r++)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 2 * d; j++) std::swap(a[c][j], a[piv][j]);
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 2 * d; j++) a[c][j] *= inv;
        for (int r = 0; r < d; r++) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.0)
                for (int j = 0; j < 2 * d; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) Sinv[i * d + j] = a[i][d + j];
    return true;
}

extern "C" int jaicov_neq_solve(jaicov_engine *e, int invert, double *dx_out) {
    if (!e || !dx_out) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state != jaicov_engine::ST_BUILT) FAIL(e, JAICOV_ERR_BAD_STATE, "build first");
    HIPE(e, hipSetDevice(e->device));
    const bool schur = e->schur_active;
    if (invert < 0 || invert > JAICOV_INVERT_FULL_EXPANDED) return JAICOV_ERR_BAD_ARGUMENT;
    invert = effective_invert(e, invert);
    if (invert == JAICOV_INVERT_FULL_EXPANDED && !schur) invert = JAICOV_INVERT_FULL;     // the system at hand was assembled unreduced
    const bool expand = invert == JAICOV_INVERT_FULL_EXPANDED;
    if (schur && invert == JAICOV_INVERT_FULL)
        FAIL(e, JAICOV_ERR_BAD_STATE, "the normal equations were assembled with the EO blocks pre-eliminated: call "
                                      "jaicov_neq_prepare_inverse(e, JAICOV_INVERT_FULL) before the build whose solve shall invert");
    // order of the system that is factorised: the reduced one (points, IO, distortion) or the full one
    const int U = schur ? e->e0 : e->U, d = e->d, nrhs = d + 1;
    const int Upad = e->Upad;                      // leading dimension of N, V, B and the rhs/solution vectors
    if (schur && !e->solverS_ready) {
        HIPE(e, e->solverS.init(e->stream, ((e->e0 + 127) / 128) * 128, false, true, &e->solver));
        e->solverS_ready = true;
    }
    DenseSolver &slv = schur ? e->solverS : e->solver;
    slv.profile = e->solver.profile;
    slv.flops_order = (double)(schur ? e->e0 : e->U);
    const int Up = slv.nfact;                      // padded order of the factorised system
    const long ld = slv.ld;
    if (invert) {   // normally done by prepare_inverse(); a host that did not announce the final pass pays the allocation here
        const int rc_inv = ensure_inverse_buffers(e, schur);
        if (rc_inv) return rc_inv;
        if (expand && !e->solver.Q) {
            HIPE(e, hipMalloc(&e->solver.Q, (size_t)e->solver.n * e->solver.ld * sizeof(double)));
            e->solver_has_Q = true;
        }
    }
    // V to the host, scaled + row-normalised datum rows Bh = R B V (NES:82-91 scaling of the border)
    // This round trip stays even where nothing of it is needed before the factorisation (d = 0): with the host running ahead of
    // the assembly -- enqueuing the factorisation and spinning in its residency handshake while the assembly kernels still run --
    // every pass was 0.6 ms SLOWER (assembly +0.19, factorisation +0.4; measured A/B in round 3).
    // What d = 0 does save is the round trip between the substitution and the first refinement step (`fast` below).
    const bool fast = d == 0;
    int hinfo = 0;   // status of the per-image EO eliminations, fetched in the same round trip
    {
        HIPE(e, hipMemcpyAsync(e->h_V.data(), e->d_V, Upad * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        if (schur) HIPE(e, hipMemcpyAsync(&hinfo, e->sb.info, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPE(e, hipStreamSynchronize(e->stream));
        if (hinfo != 0) FAIL(e, JAICOV_ERR_SINGULAR, "exterior orientation block of image " + std::to_string(hinfo - 1000000) + " is not positive definite");
    }
    std::vector<double> Bh((size_t)8 * Upad, 0.0);
    double R[7] = {1, 1, 1, 1, 1, 1, 1};
    for (int a = 0; a < d; a++) {
        double s = 0.0;
        for (int c = d; c < U; c++) {
            const double v = e->hB[(size_t)a * Upad + c] * e->h_V[c];
            Bh[(size_t)a * Upad + c] = v;
            s += v * v;
        }
        if (!(s > 0.0)) FAIL(e, JAICOV_ERR_SINGULAR, "empty datum condition row");
        R[a] = 1.0 / sqrt(s);
        for (int c = d; c < U; c++) Bh[(size_t)a * Upad + c] *= R[a];
    }
    if (d > 0) HIPE(e, hipMemcpyAsync(e->d_B, Bh.data(), (size_t)d * Upad * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPE(e, hipEventRecord(e->ev[4], e->stream));
    // right-hand sides: row 0 = V n, rows 1..d = Bh.  They are the extra rows below the matrix, so the factorisation
    // itself carries out the forward substitution (dense.hip).
    const int vs = Up;   // stride between the solution vectors (the solver's order)
    const bool fused = slv.flow_ready;
    const int c1 = std::min(slv.first_panel_cols(), Up);
    // dataflow factorisation: the tile kernel reads N itself and scales on the fly (no copy of the matrix at all);
    // stream-scheduled one: the columns of the first panel go first, together with the right-hand sides; the panel then
    // factors on its stream while the rest of the matrix is still being scaled and copied (0.4 ms at config 4).
    // (in a loop: a dataflow factorisation that was abandoned -- cholflow.hip, "Visibility": a wait ran into its time limit, seen
    // about once in 1 000 factorisations at config 4 -- is repeated; N, V and the datum rows are untouched by it)
    std::vector<double> X, delta0;
    bool fast_refined = false;
    int info = 0;
    for (int attempt = 0;; attempt++) {
    HIPE(e, slv.begin_refactor());      // the side stream's premultiply of the last pass may still read the old factor: nothing writes L before this
    if (!fused)
        hipLaunchKernelGGL(scale_copy_kernel, dim3((c1 + 255) / 256, Up), dim3(256), 0, e->stream, e->d_N, (long)Upad, slv.L, ld, U,
                           Up, d, e->d_V, e->d_B, Upad, 0, c1);
    HIPE(e, hipMemsetAsync(slv.rhs_row(0), 0, (size_t)128 * ld * sizeof(double), e->stream));
    hipLaunchKernelGGL(scale_vec_kernel, dim3((Up + 255) / 256), dim3(256), 0, e->stream, e->d_n, e->d_V, slv.rhs_row(0), U, Up, d);
    if (d > 0)
        HIPE(e, hipMemcpy2DAsync(slv.rhs_row(1), (size_t)ld * sizeof(double), e->d_B, (size_t)Upad * sizeof(double),
                                 (size_t)Up * sizeof(double), (size_t)d, hipMemcpyDeviceToDevice, e->stream));
    HIPE(e, hipEventRecord(e->ev_first, e->stream));
    if (!fused && c1 < Up)
        hipLaunchKernelGGL(scale_copy_kernel, dim3((Up - c1 + 255) / 256, Up), dim3(256), 0, e->stream, e->d_N, (long)Upad, slv.L,
                           ld, U, Up, d, e->d_V, e->d_B, Upad, c1, Up);
    HIPE(e, hipEventRecord(e->ev_all, e->stream));
    if (fused) slv.flow_set_source(e->d_N, (long)Upad, e->d_V, e->d_B, Upad, d, U);
    HIPE(e, slv.potrf(e->ev_first, e->ev_all));
    HIPE(e, hipEventRecord(e->ev[5], e->stream));
    HIPE(e, slv.backsolve_aug(e->d_G, vs, nrhs));           // G <- L^-T (L^-1 Y)   (row 0: y~, rows 1..d: G^)
    HIPE(e, hipEventRecord(e->ev[6], e->stream));
    X.assign((size_t)nrhs * vs, 0.0);
    HIPE(e, hipMemcpyAsync(X.data(), e->d_G, X.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    fast_refined = false;
    if (fast && e->refine_steps >= 1 && !e->sim_built) {
        // first refinement step without the host: dx = V y on the device, residual of the unscaled system in two-fold precision,
        // forward + backward substitution (refine.hip, dense.hip); the correction comes back together with y
        const int nbk = Up / 128;
        const size_t need = (size_t)nbk * nbk * 256;
        if (e->refP_len < need) {
            if (e->d_refP) hipFree(e->d_refP);
            e->d_refP = nullptr; e->refP_len = 0;
            HIPE(e, hipMalloc(&e->d_refP, need * sizeof(double)));
            e->refP_len = need;
        }
        double *d_rhs = e->d_ref, *d_tmp = e->d_ref + Upad, *d_delta = e->d_ref + 2 * (size_t)Upad;
        HIPE(e, hipEventRecord(e->ev_r0, e->stream));
        hipLaunchKernelGGL(scale_vec_kernel, dim3((Up + 255) / 256), dim3(256), 0, e->stream, e->d_G, e->d_V, e->d_dx, U, Up, 0);   // dx = V y
        HIPE(e, launch_residual_dd(e->stream, e->d_N, (long)Upad, U, 0, Up, e->d_dx, e->d_n, e->d_V, e->d_Braw, e->d_B, (long)Upad,
                                   RefineBorder{}, e->d_refP, d_rhs));
        HIPE(e, slv.solve_rhs(d_rhs, d_tmp, d_delta));
        HIPE(e, hipEventRecord(e->ev_r1, e->stream));
        delta0.assign((size_t)Up, 0.0);
        HIPE(e, hipMemcpyAsync(delta0.data(), d_delta, delta0.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        fast_refined = true;
    }
    info = slv.fetch_info();
    if (info == -9 && attempt < 2) {
        ++e->flow_retries;
        fprintf(stderr, "jaicov: factorisation abandoned on the device (a wait ran into its time limit); repeating it (%d)\n", attempt + 1);
        continue;
    }
    break;
    }
    if (info < 0) FAIL(e, JAICOV_ERR_DEVICE, "factorisation did not complete on the device (code " + std::to_string(info) + ")");
    if (info != 0) FAIL(e, JAICOV_ERR_SINGULAR, "normal-equation matrix is singular / not positive definite at pivot " + std::to_string(info));
    // ---- rank-d border algebra on the host ---------------------------------------------------------------------
    double Sm[49], Sinv[49], kh[7];
    std::vector<double> y(X.begin(), X.begin() + vs);
    if (d > 0) {
        for (int a = 0; a < d; a++)
            for (int b = 0; b < d; b++) {
                double s = 0.0;
                for (int c = d; c < U; c++) s += Bh[(size_t)a * Upad + c] * X[(size_t)(1 + b) * vs + c];
                Sm[a * d + b] = s;
            }
        if (!small_inverse(d, Sm, Sinv)) FAIL(e, JAICOV_ERR_SINGULAR, "datum conditions are linearly dependent");
        double by[7];
        for (int a = 0; a < d; a++) {
            double s = 0.0;
            for (int c = d; c < U; c++) s += Bh[(size_t)a * Upad + c] * X[c];
            by[a] = s;
        }
        for (int a = 0; a < d; a++) {
            double s = 0.0;
            for (int b = 0; b < d; b++) s += Sinv[a * d + b] * by[b];
            kh[a] = s;
        }
        for (int c = d; c < U; c++) {
            double s = 0.0;
            for (int a = 0; a < d; a++) s += X[(size_t)(1 + a) * vs + c] * kh[a];
            y[c] -= s;
        }
    }
    for (int c = 0; c < U; c++) dx_out[c] = c < d ? R[c] * kh[c] : e->h_V[c] * y[c];
    for (int c = 0; c < U; c++)
        if (!std::isfinite(dx_out[c])) FAIL(e, JAICOV_ERR_NOT_FINITE, "non-finite step");
    // ---- iterative refinement (refine.hip): residual of the unscaled bordered system in two-fold precision, one forward and
    // one backward substitution with the factor at hand per step.  The correction of the border follows the same rank-d
    // algebra as above with r_y = V rho_x, r_k = R rho_kappa:  delta0 = M^-1 (r_y + Bh' r_k),
    // dk = S^-1 (Bh delta0 - r_k),  dy = delta0 - G^ dk.
    float refine_ms = 0.f;
    e->last_refine_correction = 0.0;
    if (fast_refined) {      // the step the device took on its own (d = 0: no border algebra): dx += V delta
        hipEventElapsedTime(&refine_ms, e->ev_r0, e->ev_r1);
        double cmax = 0.0, xmax = 0.0;
        for (int c = 0; c < U; c++) {
            const double corr = e->h_V[c] * delta0[c];
            if (!std::isfinite(corr)) FAIL(e, JAICOV_ERR_NOT_FINITE, "non-finite refinement step");
            dx_out[c] += corr;
            cmax = std::max(cmax, fabs(corr)); xmax = std::max(xmax, fabs(dx_out[c]));
        }
        e->last_refine_correction = xmax > 0.0 ? cmax / xmax : 0.0;
    }
    for (int step = fast_refined ? 1 : 0; step < e->refine_steps && !e->sim_built; step++) {
        const int nbk = Up / 128;
        const size_t need = (size_t)nbk * nbk * 256;
        if (e->refP_len < need) {
            if (e->d_refP) hipFree(e->d_refP);
            e->d_refP = nullptr; e->refP_len = 0;
            HIPE(e, hipMalloc(&e->d_refP, need * sizeof(double)));
            e->refP_len = need;
        }
        RefineBorder bd{};
        for (int a = 0; a < d; a++) {
            long double sacc = 0.0L;
            for (int c = d; c < U; c++) sacc += (long double)e->hB[(size_t)a * Upad + c] * (long double)dx_out[c];
            bd.kappa[a] = dx_out[a];
            bd.rk[a] = (double)(-(long double)R[a] * sacc);
        }
        double *d_rhs = e->d_ref, *d_tmp = e->d_ref + Upad, *d_delta = e->d_ref + 2 * (size_t)Upad;
        HIPE(e, hipEventRecord(e->ev_r0, e->stream));
        if (d > 0 && step == 0)
            HIPE(e, hipMemcpyAsync(e->d_Braw, e->hB.data(), (size_t)d * Upad * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPE(e, hipMemcpyAsync(e->d_dx, dx_out, (size_t)U * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPE(e, launch_residual_dd(e->stream, e->d_N, (long)Upad, U, d, Up, e->d_dx, e->d_n, e->d_V, e->d_Braw, e->d_B, (long)Upad,
                                   bd, e->d_refP, d_rhs));
        HIPE(e, slv.solve_rhs(d_rhs, d_tmp, d_delta));
        HIPE(e, hipEventRecord(e->ev_r1, e->stream));
        std::vector<double> dl((size_t)Up);
        HIPE(e, hipMemcpyAsync(dl.data(), d_delta, dl.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPE(e, hipStreamSynchronize(e->stream));
        float ms1 = 0.f;
        hipEventElapsedTime(&ms1, e->ev_r0, e->ev_r1);
        refine_ms += ms1;
        double dk[7] = {0, 0, 0, 0, 0, 0, 0};
        if (d > 0) {
            double tt[7];
            for (int a = 0; a < d; a++) {
                double sacc = 0.0;
                for (int c = d; c < U; c++) sacc += Bh[(size_t)a * Upad + c] * dl[c];
                tt[a] = sacc - bd.rk[a];
            }
            for (int a = 0; a < d; a++) {
                double sacc = 0.0;
                for (int b = 0; b < d; b++) sacc += Sinv[a * d + b] * tt[b];
                dk[a] = sacc;
            }
        }
        double cmax = 0.0, xmax = 0.0;
        for (int c = d; c < U; c++) {
            double dy = dl[c];
            for (int a = 0; a < d; a++) dy -= X[(size_t)(1 + a) * vs + c] * dk[a];
            const double corr = e->h_V[c] * dy;
            if (!std::isfinite(corr)) FAIL(e, JAICOV_ERR_NOT_FINITE, "non-finite refinement step");
            dx_out[c] += corr;
            cmax = std::max(cmax, fabs(corr)); xmax = std::max(xmax, fabs(dx_out[c]));
        }
        for (int a = 0; a < d; a++) dx_out[a] += R[a] * dk[a];
        e->last_refine_correction = xmax > 0.0 ? cmax / xmax : 0.0;
    }
    if (schur && e->sim_built) {
        // SIMULATION zeroes the right-hand side of the WHOLE system (BA:830-831 `n.zero()`): the eliminated exterior
        // orientations get no step either (their back substitution would use the real misclosures)
        for (int c = U; c < e->U; c++) dx_out[c] = 0.0;
    } else if (schur && e->opts.reduced_reference_quirk && invert == JAICOV_INVERT_REDUCED) {
        // what the reference's last pass leaves in the EO entries of dx under MatrixInversion.REDUCED (quirk Q1): V_c^2 n_c
        std::vector<double> xq((size_t)6 * e->p.n_images);
        HIPE(e, hipMemcpyAsync(xq.data(), e->sb.xq, xq.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPE(e, hipStreamSynchronize(e->stream));
        for (int c = U; c < e->U; c++) dx_out[c] = 0.0;
        for (int img : e->h_blk_images)
            for (int k = 0; k < 6; k++) dx_out[e->e0 + 6 * img + k] = xq[(size_t)6 * img + k];
    } else if (schur) {
        // EO step of every image of this engine: dx_E = L_E^-T U' (w - A_r dx_R)   (schur.hip)
        for (int c = U; c < e->U; c++) dx_out[c] = 0.0;
        HIPE(e, hipMemcpyAsync(e->d_dx, dx_out, (size_t)e->U * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPE(e, launch_omega(e->stream, e->p, e->d_in_block, e->ip0, e->ip_count, e->d_blk_list, 0, e->max_m, e->d_rowsA,
                             e->d_rowsW, e->d_dx, 1.0, e->d_vbuf, e->d_omega));
        HIPE(e, launch_schur_backsub(e->stream, e->p, e->d_blk_list, e->n_blk_list, e->sb.U, e->sb.Linv, e->d_vbuf, e->d_xE));
        std::vector<double> xE((size_t)6 * e->p.n_images);
        HIPE(e, hipMemcpyAsync(xE.data(), e->d_xE, xE.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPE(e, hipStreamSynchronize(e->stream));
        {
            // Round 5: once in ~10 runs of the GPU test-suite a solve returned an EO step of EXACT zeros for every image (N, n and the reduced
            // step correct; 1 200 solves of the same scene alone: never).  The back-substitution kernel cannot produce that from finite
            // operands, so the copy is checked: all zeros -> read again with a blocking copy after a device-wide wait, and say so.
            bool all_zero = !e->h_blk_images.empty();
            for (int img : e->h_blk_images)
                for (int k = 0; k < 6 && all_zero; k++) all_zero = xE[(size_t)6 * img + k] == 0.0;
            if (all_zero) {
                HIPE(e, hipDeviceSynchronize());
                HIPE(e, hipMemcpy(xE.data(), e->d_xE, xE.size() * sizeof(double), hipMemcpyDeviceToHost));
                bool still = true;
                for (int img : e->h_blk_images)
                    for (int k = 0; k < 6 && still; k++) still = xE[(size_t)6 * img + k] == 0.0;
                if (still) {      // the kernel itself left zeros: run the two kernels once more
                    HIPE(e, launch_omega(e->stream, e->p, e->d_in_block, e->ip0, e->ip_count, e->d_blk_list, 0, e->max_m, e->d_rowsA,
                                         e->d_rowsW, e->d_dx, 1.0, e->d_vbuf, e->d_omega));
                    HIPE(e, launch_schur_backsub(e->stream, e->p, e->d_blk_list, e->n_blk_list, e->sb.U, e->sb.Linv, e->d_vbuf, e->d_xE));
                    HIPE(e, hipStreamSynchronize(e->stream));
                    HIPE(e, hipMemcpy(xE.data(), e->d_xE, xE.size() * sizeof(double), hipMemcpyDeviceToHost));
                }
                fprintf(stderr, "jaicov: the exterior-orientation step came back as exact zeros; %s\n",
                        still ? "the back-substitution was run again" : "a second, blocking copy had the values (the first copy returned before the data)");
            }
        }
        for (int img : e->h_blk_images)
            for (int k = 0; k < 6; k++) {
                dx_out[e->e0 + 6 * img + k] = xE[(size_t)6 * img + k];
                if (!std::isfinite(xE[(size_t)6 * img + k])) FAIL(e, JAICOV_ERR_NOT_FINITE, "non-finite step");
            }
    }
    if (!schur && e->opts.reduced_reference_quirk && invert == JAICOV_INVERT_REDUCED) {
        // full-order path (no EO pre-elimination possible): the same quirk from the assembled right-hand side, V_c^2 n_c
        std::vector<double> hn((size_t)e->U);
        HIPE(e, hipMemcpyAsync(hn.data(), e->d_n, hn.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPE(e, hipStreamSynchronize(e->stream));
        const int ns = e->n_slots, s_eo = ns - 6 * e->p.n_images;
        for (int sl = s_eo; sl < ns; sl++) {
            const int c = e->h_slot_col[sl];
            if (c >= 0) dx_out[c] = e->h_V[c] * e->h_V[c] * hn[c];
        }
    }
    if (invert) {
        // FULL: Qxx of all unknowns.  REDUCED on a pre-eliminated system: the inverse of the reduced normal equations,
        // which IS the block of Qxx that belongs to the datum border, points, interior orientation and distortion
        // (MatrixInversion.REDUCED / PRE_ELIMINATION, BA:261-267: solve(N, n, numRows, true) on the reduced system).
        HIPE(e, slv.trtri());
        HIPE(e, slv.lauum());
        if (e->opts.inverse_refinement >= 0 && Up <= 8192) {
            // One Newton-Schulz step on Q = inv(M) with the residual I - M Q formed exactly (batchinv.hip): M is materialised once more
            // (scaled, with the datum border's term; the factor in slv.L is not needed again before the next factorisation), W is free
            // after lauum(), three more squares are kept with the engine.
            const size_t sq = (size_t)Up * ld;
            if (e->ns_work_len < 3 * sq) {
                if (e->ns_work) hipFree(e->ns_work);
                e->ns_work = nullptr; e->ns_work_len = 0;
                HIPE(e, hipMalloc(&e->ns_work, 3 * sq * sizeof(double)));
                e->ns_work_len = 3 * sq;
            }
            HIPE(e, slv.begin_refactor());
            HIPE(e, slv.symmetrize(slv.Q));
            hipLaunchKernelGGL(scale_copy_kernel, dim3((Up + 255) / 256, Up), dim3(256), 0, e->stream, e->d_N, (long)Upad, slv.L, ld, U,
                               Up, d, e->d_V, e->d_B, Upad, 0, Up);
            HIPE(e, slv.symmetrize(slv.L));
            HIPE(e, newton_schulz_exact(e->stream, Up, ld, slv.L, slv.Q, slv.W, e->ns_work, e->ns_work + sq, e->ns_work + 2 * sq));
        }
        // H = Sinv G^ ; F = R (Sinv G^) V ; E = R (I - Sinv) R
        std::vector<double> H((size_t)8 * Upad, 0.0), F((size_t)8 * Upad, 0.0);
        double E[49];
        for (int a = 0; a < d; a++) {
            for (int c = d; c < U; c++) {
                double s = 0.0;
                for (int b = 0; b < d; b++) s += Sinv[a * d + b] * X[(size_t)(1 + b) * vs + c];
                H[(size_t)a * Upad + c] = s;
                F[(size_t)a * Upad + c] = R[a] * s * e->h_V[c];
            }
            for (int b = 0; b < d; b++) E[a * d + b] = R[a] * ((a == b ? 1.0 : 0.0) - Sinv[a * d + b]) * R[b];
        }
        if (d > 0) {
            HIPE(e, hipMemcpyAsync(e->d_H, H.data(), (size_t)d * Upad * sizeof(double), hipMemcpyHostToDevice, e->stream));
            HIPE(e, hipMemcpyAsync(e->d_F, F.data(), (size_t)d * Upad * sizeof(double), hipMemcpyHostToDevice, e->stream));
            HIPE(e, hipMemcpyAsync(e->d_E, E, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, e->stream));
        }
        hipLaunchKernelGGL(qfix_kernel, dim3((Up + 255) / 256, Up), dim3(256), 0, e->stream, slv.Q, ld, U, Upad, d,
                           e->d_V, e->d_G + vs, (long)vs, e->d_H, e->d_F, e->d_E, Up);
        if (expand) {
            // ---- the full cofactor matrix from the reduced one (schur.hip, blk_expand_f_kernel): workspace = W of the reduced
            //      solver, free again after lauum():  F [6I pad][Up] | T1 = -F Q_RR [6I pad][Up] | T2 = -T1 F' [6I pad][6I pad]
            const int I6 = 6 * e->p.n_images, I6p = ((I6 + 127) / 128) * 128;
            if ((size_t)I6p * ((size_t)2 * Up + I6p) > (size_t)slv.n * slv.ld)
                FAIL(e, JAICOV_ERR_UNSUPPORTED, "FULL_EXPANDED: more exterior orientations than the workspace holds; use JAICOV_INVERT_FULL");
            double *Fm = slv.W, *T1 = Fm + (size_t)I6p * Up, *T2 = T1 + (size_t)I6p * Up;
            const double *LinvAll = e->sb.Linv;
            if (e->all_images) {
                HIPE(e, hipMemsetAsync(Fm, 0, (size_t)I6p * Up * sizeof(double), e->stream));
                HIPE(e, launch_schur_expand_f(e->stream, e->p, e->d_blk_list, e->n_blk_list, e->d_rowsA, e->sb.U, e->sb.Linv, e->sb.G, Fm, (long)Up));
            } else {      // a shard: F and L_E^-1 of EVERY image, summed over the ranks by the caller (jaicov_neq_expansion_buffer)
                if (!e->exp_ready || e->expF_len != (size_t)I6p * Up + (size_t)36 * e->p.n_images)
                    FAIL(e, JAICOV_ERR_BAD_STATE, "FULL_EXPANDED on a sharded engine: all-reduce jaicov_neq_expansion_buffer() between accumulate and solve");
                HIPE(e, hipMemcpyAsync(Fm, e->d_expF, (size_t)I6p * Up * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
                // the summed L_E^-1 records are READ from the buffer, never copied over sb.Linv: that array must keep zeros for the
                // foreign images, or the next pass's expansion buffer would carry their stale records into the sum (ADVICE r4, high:
                // a second sharded FULL_EXPANDED pass gave R times the EO cofactors)
                LinvAll = e->d_expF + (size_t)I6p * Up;
            }
            HIPE(e, slv.symmetrize(slv.Q));
            GemmArgs g1{};
            g1.A = Fm; g1.lda = Up; g1.B = slv.Q; g1.ldb = ld; g1.C = T1; g1.ldc = Up;
            g1.M = I6p; g1.N = Up; g1.K = Up; g1.alpha = -1.0; g1.beta = 0.0; g1.kmode = KMODE_FULL;
            HIPE(e, gemm_f64(e->stream, LAY_KC, LAY_KC, g1));             // Q_RR symmetric: F Q_RR = F Q_RR'
            GemmArgs g2{};
            g2.A = T1; g2.lda = Up; g2.B = Fm; g2.ldb = Up; g2.C = T2; g2.ldc = I6p;
            g2.M = I6p; g2.N = I6p; g2.K = Up; g2.alpha = -1.0; g2.beta = 0.0; g2.kmode = KMODE_FULL; g2.lower_only = 1;
            HIPE(e, gemm_f64(e->stream, LAY_KC, LAY_KC, g2));
            hipLaunchKernelGGL(expand_cofactor_kernel, dim3((e->U + 255) / 256, e->U), dim3(256), 0, e->stream, e->solver.Q, e->solver.ld,
                               e->U, e->e0, slv.Q, ld, T1, (long)Up, T2, (long)I6p, LinvAll);
        }
        HIPE(e, hipEventRecord(e->ev[7], e->stream));
        HIPE(e, hipStreamSynchronize(e->stream));
        e->have_Q = true;
        e->q_reduced = schur && !expand;
        e->q_order = expand ? e->U : U;
    } else {
        HIPE(e, hipEventRecord(e->ev[7], e->stream));
        HIPE(e, hipStreamSynchronize(e->stream));
    }
    slv.prof_collect();
    if (schur) {
        e->solver.stat_launches += slv.stat_launches; e->solver.stat_ms += slv.stat_ms; e->solver.stat_flops += slv.stat_flops;
        slv.stat_launches = slv.stat_ms = slv.stat_flops = 0.0;
    }
    float ms;
    hipEventElapsedTime(&ms, e->ev[0], e->ev[1]); e->timings[0] = ms;
    hipEventElapsedTime(&ms, e->ev[1], e->ev[2]); e->timings[1] = ms;
    hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->timings[2] = ms;
    hipEventElapsedTime(&ms, e->ev[4], e->ev[5]); e->timings[3] = ms;
    hipEventElapsedTime(&ms, e->ev[5], e->ev[6]); e->timings[4] = ms + refine_ms;   // substitution + refinement steps
    hipEventElapsedTime(&ms, e->ev[6], e->ev[7]); e->timings[5] = std::max(0.f, ms - refine_ms);   // the refinement sits between these events: counted under [4]
    hipEventElapsedTime(&ms, e->ev[0], e->ev[7]); e->timings[7] = ms;
    e->state = jaicov_engine::ST_SOLVED;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_expansion_buffer(jaicov_engine *e, void **device_ptr, size_t *count) {
    if (!e || !device_ptr || !count) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state != jaicov_engine::ST_ACCUMULATED || !e->schur_active || !e->solverS_ready)
        FAIL(e, JAICOV_ERR_BAD_STATE, "expansion buffer: accumulate a pre-eliminated system first (prepare_inverse(JAICOV_INVERT_FULL_EXPANDED))");
    HIPE(e, hipSetDevice(e->device));
    const size_t I6p = ((size_t)6 * e->p.n_images + 127) / 128 * 128, Up = (size_t)e->solverS.nfact;
    const size_t len = I6p * Up + (size_t)36 * e->p.n_images;
    if (!e->d_expF || e->expF_len != len) {
        if (e->d_expF) hipFree(e->d_expF);
    if (e->ns_work) hipFree(e->ns_work);
        e->d_expF = nullptr;
        HIPE(e, hipMalloc(&e->d_expF, len * sizeof(double)));
        e->expF_len = len;
    }
    { int rr = ensure_rows(e); if (rr) return rr; }
    HIPE(e, hipMemsetAsync(e->d_expF, 0, I6p * Up * sizeof(double), e->stream));       // foreign images' bands: zero
    HIPE(e, launch_schur_expand_f(e->stream, e->p, e->d_blk_list, e->n_blk_list, e->d_rowsA, e->sb.U, e->sb.Linv, e->sb.G, e->d_expF, (long)Up));
    // L_E^-1: written for this engine's images only (blk_elim_kernel), zero elsewhere since create
    HIPE(e, hipMemcpyAsync(e->d_expF + I6p * Up, e->sb.Linv, (size_t)36 * e->p.n_images * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIPE(e, hipStreamSynchronize(e->stream));
    e->exp_ready = true;
    *device_ptr = e->d_expF;
    *count = len;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_eo_step_buffer(jaicov_engine *e, void **device_ptr, size_t *count) {
    if (!e || !device_ptr || !count) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state != jaicov_engine::ST_SOLVED || !e->schur_active || !e->d_xE) FAIL(e, JAICOV_ERR_BAD_STATE, "no pre-eliminated exterior orientations: solve a reduced system first");
    *device_ptr = e->d_xE;
    *count = (size_t)6 * e->p.n_images;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_omega(jaicov_engine *e, double sigma2, const double *dx, double *omega) {
    if (!e || !dx || !omega) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state == jaicov_engine::ST_NEW) FAIL(e, JAICOV_ERR_BAD_STATE, "set_parameters first");
    HIPE(e, hipSetDevice(e->device));
    int rc = ensure_rows(e);
    if (rc) return rc;
    hipEvent_t t0 = e->ev[8], t1 = e->ev[9];
    HIPE(e, hipEventRecord(t0, e->stream));
    HIPE(e, hipMemcpyAsync(e->d_dx, dx, (size_t)e->U * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPE(e, hipMemsetAsync(e->d_omega, 0, sizeof(double), e->stream));
    HIPE(e, launch_omega(e->stream, e->p, e->d_in_block, e->ip0, e->ip_count, e->d_blk_list, e->n_blk_list, e->max_m,
                         e->d_rowsA, e->d_rowsW, e->d_dx, sigma2, e->d_vbuf, e->d_omega));
    if (e->opts.apply_shared) HIPE(e, launch_shared_groups(e->stream, e->p, e->d_vals, sigma2, nullptr, nullptr, e->d_dx, e->d_omega));
    HIPE(e, hipEventRecord(t1, e->stream));
    HIPE(e, hipMemcpyAsync(omega, e->d_omega, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPE(e, hipStreamSynchronize(e->stream));
    float ms;
    hipEventElapsedTime(&ms, t0, t1);
    e->timings[6] = ms;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_update(jaicov_engine *e, const double *dx, double *max_abs_dx) {
    if (!e || !dx) return JAICOV_ERR_BAD_ARGUMENT;
    HIPE(e, hipSetDevice(e->device));
    double mx = 0.0;
    for (int s = 0; s < e->n_slots; s++) {
        const int c = e->h_slot_col[s];
        if (c >= 0) {
            const double dv = dx[c];
            mx = std::max(mx, fabs(dv));
            e->h_vals[s] = e->h_vals[s] + dv;
        }
    }
    if (max_abs_dx) *max_abs_dx = mx;
    HIPE(e, hipMemcpyAsync(e->d_vals, e->h_vals.data(), (size_t)e->n_slots * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPE(e, hipStreamSynchronize(e->stream));
    e->rows_valid = false;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_get_normal(jaicov_engine *e, double *N_packed, size_t len, double *n, size_t Ulen) {
    if (!e || !N_packed || !n) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state != jaicov_engine::ST_BUILT && e->state != jaicov_engine::ST_SOLVED && e->state != jaicov_engine::ST_ACCUMULATED)
        FAIL(e, JAICOV_ERR_BAD_STATE, "build first");
    const int U = e->U;
    if (len != (size_t)U * (U + 1) / 2 || Ulen != (size_t)U) return JAICOV_ERR_BAD_ARGUMENT;
    HIPE(e, hipSetDevice(e->device));
    // with the exterior orientations pre-eliminated only the reduced system (order e0) was assembled: the rows from e0 on
    // of the square are neither zeroed nor written in such a pass.  The header's contract: reduced system in the leading
    // e0 rows / columns, zeros elsewhere.
    const int Ua = e->schur_active ? e->e0 : U;
    const size_t lenA = (size_t)Ua * (Ua + 1) / 2;
    double *d_ap = nullptr;
    HIPE(e, hipMalloc(&d_ap, std::max<size_t>(lenA, 1) * sizeof(double)));
    hipLaunchKernelGGL(pack_kernel, dim3((Ua + 255) / 256, std::max(Ua, 1)), dim3(256), 0, e->stream, e->d_N, (long)e->Upad, Ua, d_ap);
    hipError_t he = hipMemcpyAsync(N_packed, d_ap, lenA * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(n, e->d_n, Ua * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    hipFree(d_ap);
    HIPE(e, he);
    if (lenA < len) memset(N_packed + lenA, 0, (len - lenA) * sizeof(double));
    for (int c = Ua; c < U; c++) n[c] = 0.0;
    // the datum border lives on the host (rows 0..d-1): K[r][c] = B[r][c]  (BA:548-593), packed index r + c(c+1)/2
    if (e->state != jaicov_engine::ST_ACCUMULATED)
        for (int r = 0; r < e->d; r++)
            for (int c = e->d; c < Ua; c++) N_packed[(size_t)r + (size_t)c * (c + 1) / 2] = e->hB[(size_t)r * e->Upad + c];
    return JAICOV_OK;
}

extern "C" int jaicov_neq_get_cofactor(jaicov_engine *e, double *Q_packed, size_t len) {
    if (!e || !Q_packed) return JAICOV_ERR_BAD_ARGUMENT;
    if (!e->have_Q) FAIL(e, JAICOV_ERR_BAD_STATE, "no cofactor matrix: solve with invert != 0 first (MatrixInversion.NONE, BA:1177)");
    const int U = e->q_order;
    if (len != (size_t)U * (U + 1) / 2) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "cofactor buffer length must be order(order+1)/2 with order = jaicov_neq_cofactor_order()");
    HIPE(e, hipSetDevice(e->device));
    const DenseSolver &qs = e->q_reduced ? e->solverS : e->solver;
    double *d_ap = nullptr;
    HIPE(e, hipMalloc(&d_ap, std::max<size_t>(len, 1) * sizeof(double)));
    hipLaunchKernelGGL(pack_kernel, dim3((U + 255) / 256, std::max(U, 1)), dim3(256), 0, e->stream, qs.Q, qs.ld, U, d_ap);
    hipError_t he = hipMemcpyAsync(Q_packed, d_ap, len * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    hipFree(d_ap);
    HIPE(e, he);
    return JAICOV_OK;
}

static int cofactor_sub_impl(jaicov_engine *e, const int32_t *idx, int32_t k, double scale, double *out);

extern "C" int jaicov_neq_get_cofactor_sub(jaicov_engine *e, const int32_t *idx, int32_t k, double *out) {
    return cofactor_sub_impl(e, idx, k, 1.0, out);
}

extern "C" int jaicov_neq_get_dispersion_sub(jaicov_engine *e, double sigma2_aposteriori, const int32_t *idx, int32_t k, double *out) {
    return cofactor_sub_impl(e, idx, k, sigma2_aposteriori, out);
}

static int cofactor_sub_impl(jaicov_engine *e, const int32_t *idx, int32_t k, double scale, double *out) {
    if (!e || !idx || !out || k <= 0) return JAICOV_ERR_BAD_ARGUMENT;
    if (!e->have_Q) FAIL(e, JAICOV_ERR_BAD_STATE, "no cofactor matrix: solve with invert != 0 first");
    for (int i = 0; i < k; i++)
        if (idx[i] < 0 || idx[i] >= e->q_order) return JAICOV_ERR_BAD_ARGUMENT;
    const DenseSolver &qs = e->q_reduced ? e->solverS : e->solver;
    HIPE(e, hipSetDevice(e->device));
    int32_t *d_idx = nullptr;
    double *d_out = nullptr;
    HIPE(e, hipMalloc(&d_idx, (size_t)k * sizeof(int32_t)));
    hipError_t he = hipMalloc(&d_out, (size_t)k * k * sizeof(double));
    if (he == hipSuccess) he = hipMemcpyAsync(d_idx, idx, (size_t)k * sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(gather_sub_kernel, dim3(((size_t)k * k + 255) / 256), dim3(256), 0, e->stream, qs.Q, qs.ld, d_idx, k, d_out, scale);
        he = hipMemcpyAsync(out, d_out, (size_t)k * k * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    hipFree(d_idx); hipFree(d_out);
    HIPE(e, he);
    return JAICOV_OK;
}

extern "C" int jaicov_neq_get_rows(jaicov_engine *e, int32_t ip_begin, int32_t ip_count, double *w, double *A) {
    if (!e || !w || !A || ip_begin < e->ip0 || ip_count < 0 || ip_begin + ip_count > e->ip0 + e->ip_count) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state == jaicov_engine::ST_NEW) FAIL(e, JAICOV_ERR_BAD_STATE, "set_parameters first");
    HIPE(e, hipSetDevice(e->device));
    int rc = ensure_rows(e);
    if (rc) return rc;
    const size_t S = (size_t)e->p.n_ip;
    std::vector<double> hA((size_t)2 * KROW * S), hW(2 * S);
    HIPE(e, hipMemcpyAsync(hA.data(), e->d_rowsA, hA.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPE(e, hipMemcpyAsync(hW.data(), e->d_rowsW, hW.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPE(e, hipStreamSynchronize(e->stream));
    for (int i = 0; i < ip_count; i++) {
        const size_t ip = e->ip_old2new.empty() ? (size_t)ip_begin + i : (size_t)e->ip_old2new[(size_t)ip_begin + i];
        w[2 * i] = hW[ip]; w[2 * i + 1] = hW[S + ip];
        for (int r = 0; r < 2; r++)
            for (int l = 0; l < KROW; l++) A[((size_t)2 * i + r) * KROW + l] = hA[(size_t)(2 * l + r) * S + ip];
    }
    return JAICOV_OK;
}

extern "C" int jaicov_neq_set_profiling(jaicov_engine *e, int enable) {
    if (!e) return JAICOV_ERR_BAD_ARGUMENT;
    e->solver.profile = enable != 0;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_kernel_stats(jaicov_engine *e, double *stats, int32_t n, int reset) {
    if (!e || !stats || n < 3) return JAICOV_ERR_BAD_ARGUMENT;
    stats[0] = e->solver.stat_launches; stats[1] = e->solver.stat_ms; stats[2] = e->solver.stat_flops;
    if (n >= 6) { stats[3] = e->dm_stat_passes; stats[4] = e->dm_stat_ms; stats[5] = e->dm_stat_flops; }
    if (n >= 10) {   // health of the dataflow factorisation since create (never reset): see jaicov_neq.h
        stats[6] = (double)e->flow_retries;
        stats[7] = (double)(e->solver.flow_stale_events + e->solverS.flow_stale_events);
        stats[8] = (double)(e->solver.flow_stale_confirmed + e->solverS.flow_stale_confirmed);
        stats[9] = (double)(e->solver.flow_rescued + e->solverS.flow_rescued);
    }
    if (n >= 11) stats[10] = e->last_refine_correction;
    if (n >= 12) stats[11] = (double)e->refine_steps;      // refinement steps per solve the engine actually runs (option `refinement`, clamped)
    if (n >= 13) stats[12] = (double)e->pp.cw;             // columns of one LDS strip of the point x point gather (chosen at create from the scene)
    if (reset) e->solver.stat_launches = e->solver.stat_ms = e->solver.stat_flops = e->dm_stat_passes = e->dm_stat_ms = e->dm_stat_flops = 0.0;
    return JAICOV_OK;
}

extern "C" int jaicov_neq_create_timings(jaicov_engine *e, double *ms, int32_t n) {
    if (!e || !ms) return JAICOV_ERR_BAD_ARGUMENT;
    for (int i = 0; i < n && i < 8; i++) ms[i] = e->create_ms[i];
    return JAICOV_OK;
}

// diagnostic / parity hook: the cached inverse dispersion inv(D) of image block `block` (DOPG:82-86 caches sigma0^2 inv(D / sigma0^2),
// which is sigma0^2 times this), m x m row-major in the CALLER's order of the block's observations
extern "C" int jaicov_neq_get_block_weight(jaicov_engine *e, int32_t block, double *out, size_t len) {
    if (!e || !out) return JAICOV_ERR_BAD_ARGUMENT;
    if (!e->h_caller_block.empty()) {      // the engine's internal blocks are the images; find the caller's
        int internal = -1;
        for (size_t g = 0; g < e->h_caller_block.size(); g++)
            if (e->h_caller_block[g] == block) internal = (int)g;
        block = internal;
    }
    if (block < 0 || block >= (int)e->h_blk_mine.size() || !e->h_blk_mine[block]) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "no such image block on this engine");
    const int b = e->h_blk_ip_begin[block], mpts = e->h_blk_ip_begin[block + 1] - b, m = 2 * mpts;
    if (len != (size_t)m * m) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "len must be (2 * points of the block)^2");
    HIPE(e, hipSetDevice(e->device));
    if (e->h_blk_w_off[block] < 0) FAIL(e, JAICOV_ERR_BAD_ARGUMENT, "this block has no dispersion of its own");
    std::vector<double> w((size_t)m * m);
    HIPE(e, hipMemcpy(w.data(), e->p.blk_w + e->h_blk_w_off[block], w.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int32_t *perm = e->h_perm_local.empty() ? nullptr : e->h_perm_local.data() + b;
    for (int i = 0; i < m; i++) {
        const int si = perm ? 2 * perm[i >> 1] + (i & 1) : i;
        for (int j = 0; j < m; j++) {
            const int sj = perm ? 2 * perm[j >> 1] + (j & 1) : j;
            out[(size_t)si * m + sj] = w[(size_t)i * m + j];
        }
    }
    return JAICOV_OK;
}

extern "C" int jaicov_neq_last_timings(jaicov_engine *e, double *ms, int32_t n) {
    if (!e || !ms) return JAICOV_ERR_BAD_ARGUMENT;
    for (int i = 0; i < n && i < 8; i++) ms[i] = e->timings[i];
    return JAICOV_OK;
}

// BA.estimateModel (BA:203-387) + updateModel (BA:389-442); inversion modes NONE / FULL / REDUCED (PRE_ELIMINATION = REDUCED:
// the engine pre-eliminates the exterior orientations in every pass whenever the problem allows it)
extern "C" int jaicov_neq_estimate(jaicov_engine *e, const jaicov_estimate_options *o, jaicov_estimate_result *res) {
    if (!e || !o || !res || o->struct_size != sizeof(jaicov_estimate_options)) return JAICOV_ERR_BAD_ARGUMENT;
    if (e->state == jaicov_engine::ST_NEW) FAIL(e, JAICOV_ERR_BAD_STATE, "set_parameters first");
    const double SQRT_EPS = sqrt(EPS53);
    if (o->invert < 0 || o->invert > JAICOV_INVERT_FULL_EXPANDED) return JAICOV_ERR_BAD_ARGUMENT;
    // MatrixInversion.FULL: all of Qxx, computed from the EO-reduced system where the engine can (jaicov_neq.h, FULL_EXPANDED)
    const int inv_mode = o->invert == JAICOV_INVERT_FULL ? JAICOV_INVERT_FULL_EXPANDED : o->invert;
    const int max_iter = o->max_iterations;
    auto t0 = std::chrono::steady_clock::now();
    bool deriveFirst = o->lambda0 > 0;
    double adapted = 0.0;
    const double damping = fabs(o->lambda0);
    double maxAbsDx = 0.0, lastValid = 0.0, omega = 0.0;
    int runs = max_iter - 1;
    bool isEstimated = false, complete = false, isConverge = true;
    if (max_iter == 0) { complete = isEstimated = true; adapted = 0; }
    const double sigma2 = o->sigma2apriori > 0 ? o->sigma2apriori : 1.0;
    std::vector<double> dx(e->U > 0 ? e->U : 1);
    int state = 0, iter = 0, rc = JAICOV_OK;
    do {
        auto tp = std::chrono::steady_clock::now();
        maxAbsDx = 0.0;
        iter = max_iter - runs;
        if (deriveFirst) { adapted = damping; deriveFirst = false; }
        jaicov_neq_prepare_inverse(e, isEstimated ? inv_mode : JAICOV_INVERT_NONE);
        rc = jaicov_neq_build(e, sigma2, adapted, o->simulation);
        if (rc == JAICOV_ERR_OUT_OF_MEMORY) { state = -7; break; }
        if (rc) { state = rc == JAICOV_ERR_BAD_ARGUMENT || rc > 0 ? -2 : -1; break; }
        if (e->cancel.exchange(0)) { state = -1; rc = JAICOV_OK; break; }       // BA:240-245: INTERRUPT, flag cleared
        complete = isEstimated;
        rc = jaicov_neq_solve(e, complete ? inv_mode : JAICOV_INVERT_NONE, dx.data());
        if (rc == JAICOV_ERR_OUT_OF_MEMORY) { state = -7; break; }
        if (rc) { state = (rc > 0 || rc == JAICOV_ERR_BAD_ARGUMENT) ? -2 : -1; break; }
        bool rejected = false;
        if (adapted > 0) {
            double alpha = 0.25 * pow(adapted, -0.05);
            alpha = std::min(alpha, 0.75);
            for (auto &v : dx) v *= alpha;
            double prevOmega = omega, curOmega = 0.0;
            if ((rc = jaicov_neq_omega(e, sigma2, dx.data(), &curOmega))) { state = -1; break; }
            prevOmega = prevOmega <= 0 ? 1.7976931348623157e308 : prevOmega;
            const bool lmaConverge = prevOmega >= curOmega;
            omega = curOmega;
            if (lmaConverge) adapted *= 0.2;
            else {
                adapted *= 5.0;
                if (adapted > 1.0 / SQRT_EPS) { adapted = 1.0 / SQRT_EPS; omega = 0.0; }
            }
            if (!lmaConverge) { maxAbsDx = lastValid; rejected = true; }
        }
        if (!rejected) {
            if (complete) {
                if (o->simulation) omega = 0.0;
                else if ((rc = jaicov_neq_omega(e, sigma2, dx.data(), &omega))) { state = -1; break; }
            }
            if ((rc = jaicov_neq_update(e, dx.data(), &maxAbsDx))) { state = -1; break; }
            lastValid = maxAbsDx;
        }
        res->seconds_last_pass = std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
        if (e->cancel.exchange(0)) { state = -1; rc = JAICOV_OK; break; }       // BA:320-325
        if (std::isinf(maxAbsDx) || std::isnan(maxAbsDx)) { state = -2; break; }
        else if (maxAbsDx <= SQRT_EPS && runs > 0 && adapted == 0) isEstimated = true;
        else if (runs-- <= 1) {
            if (complete) isConverge = false;
            isEstimated = true;
        }
        if (isEstimated || adapted <= SQRT_EPS || runs < max_iter * 0.5 + 1) adapted = 0.0;
    } while (!complete);
    if (state == 0) state = isConverge ? 1 : -4;
    res->state = state;
    res->iterations = iter;
    res->omega = omega;
    res->max_abs_dx = maxAbsDx;
    res->final_lambda = adapted;
    res->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return (state == 1 || state == -4 || state == -2 || state == -1) ? (state == -1 && rc ? rc : JAICOV_OK) : rc;
}

// BundleAdjustment.interrupt() (BA:1455-1457): may be called from another thread while estimate() runs
extern "C" int jaicov_neq_cancel(jaicov_engine *e) {
    if (!e) return JAICOV_ERR_BAD_ARGUMENT;
    e->cancel.store(1);
    return JAICOV_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// jaicov_dense.h: stand-alone dense SPD solve / inverse (MX.solve / MX.inv on UpperSPDPackMatrix, MX:239-264,304-324)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int jaicov_dense_spd_solve_packed(int32_t n, double *ap, double *b, int32_t nrhs, int32_t invert, double *ms_out) {
    if (n <= 0 || !ap || nrhs < 0 || nrhs > DENSE_MAX_RHS || (nrhs > 0 && !b)) return JAICOV_ERR_BAD_ARGUMENT;
    std::string err;
    int rc = check_device(err);
    if (rc) return rc;
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return JAICOV_ERR_DEVICE;
    const int np = ((n + 127) / 128) * 128;
    DenseSolver ds;
    int status = JAICOV_OK;
    double *d_ap = nullptr, *d_Y = nullptr;
    const size_t len = (size_t)n * (n + 1) / 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    do {
        if (ds.init(s, np, invert != 0, true) != hipSuccess) { status = JAICOV_ERR_OUT_OF_MEMORY; break; }
        if (hipMalloc(&d_ap, len * sizeof(double)) != hipSuccess) { status = JAICOV_ERR_OUT_OF_MEMORY; break; }
        hipMalloc(&d_Y, (size_t)DENSE_MAX_RHS * np * sizeof(double));
        hipMemcpyAsync(d_ap, ap, len * sizeof(double), hipMemcpyHostToDevice, s);
        int info = 0;
        for (int attempt = 0;; attempt++) {      // an abandoned dataflow factorisation (-9) is repeated: the packed input is still there
            // identity padding, then unpack the lower triangle
            hipLaunchKernelGGL(load_disp_kernel, dim3((np + 255) / 256, np), dim3(256), 0, s, (const double *)nullptr, 0, ds.L, ds.ld, np, (const int32_t *)nullptr);
            hipLaunchKernelGGL(unpack_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, d_ap, ds.ld, n, ds.L);
            hipMemsetAsync(ds.rhs_row(0), 0, (size_t)128 * ds.ld * sizeof(double), s);
            for (int q = 0; q < nrhs; q++) hipMemcpyAsync(ds.rhs_row(q), b + (size_t)q * n, n * sizeof(double), hipMemcpyHostToDevice, s);
            hipEventRecord(e0, s);
            if (ds.potrf() != hipSuccess) { info = -1; break; }
            if (nrhs > 0) ds.backsolve_aug(d_Y, np, nrhs);
            if (invert) {
                ds.trtri();
                ds.lauum();
            }
            hipEventRecord(e1, s);
            info = ds.fetch_info();
            if (info == -9 && attempt < 2) {
                fprintf(stderr, "jaicov: factorisation abandoned on the device (a wait ran into its time limit); repeating it (%d)\n", attempt + 1);
                continue;
            }
            break;
        }
        if (info < 0) { status = JAICOV_ERR_DEVICE; break; }
        if (info != 0) { status = JAICOV_ERR_SINGULAR; break; }
        for (int q = 0; q < nrhs; q++) hipMemcpyAsync(b + (size_t)q * n, d_Y + (size_t)q * np, n * sizeof(double), hipMemcpyDeviceToHost, s);
        if (invert) {
            hipLaunchKernelGGL(pack_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, ds.Q, ds.ld, n, d_ap);
            hipMemcpyAsync(ap, d_ap, len * sizeof(double), hipMemcpyDeviceToHost, s);
        }
        if (hipStreamSynchronize(s) != hipSuccess) status = JAICOV_ERR_DEVICE;
        if (ms_out) { float ms = 0; hipEventElapsedTime(&ms, e0, e1); *ms_out = ms; }
    } while (0);
    hipStreamSynchronize(s);
    hipFree(d_ap); hipFree(d_Y);
    ds.release();
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipStreamDestroy(s);
    return status;
}

namespace jaicov { hipError_t diag_kernel_bench(int dbg, int iters, float *ms_out); hipError_t mfma_peak_bench(int, int, float *, double *); }
namespace jaicov { hipError_t cumask_bench(const uint32_t *, int, int, float *, double *); }
// Per-workgroup timeline of one trailing-update shaped GEMM (C -= A A', K columns): out[8 * tiles]
extern "C" int jaicov_debug_gemm_trace(int M, int K, int lower_only, long long *out) {
    std::string err;
    if (check_device(err)) return JAICOV_ERR_NO_DEVICE;
    const int tm = M / 128, tiles = lower_only ? tm * (tm + 1) / 2 : tm * tm;
    double *dA = nullptr, *dC = nullptr; long long *dT = nullptr;
    hipMalloc(&dA, (size_t)M * K * 8); hipMalloc(&dC, (size_t)M * M * 8); hipMalloc(&dT, (size_t)tiles * 64);
    hipMemset(dC, 0, (size_t)M * M * 8); hipMemset(dT, 0, (size_t)tiles * 64);
    {   // random operands (the matrix pipe draws more power on them than on zeros) and a warm chip
        std::vector<double> h((size_t)M * K);
        uint64_t x = 88172645463325252ull;
        for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
        hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    GemmArgs g{};
    g.A = dA; g.B = dA; g.C = dC; g.lda = K; g.ldb = K; g.ldc = M; g.M = M; g.N = M; g.K = K;
    g.alpha = -1e-3; g.beta = 1.0; g.lower_only = lower_only; g.kmode = KMODE_FULL;
    int warm = 40;
    if (const char *e = getenv("JAICOV_TRACE_WARM")) warm = atoi(e);
    const int tag = getenv("JAICOV_TRACE_TAG") ? atoi(getenv("JAICOV_TRACE_TAG")) : 0;
    for (int i = 0; i < warm; i++) gemm_f64(nullptr, LAY_KC, LAY_KC, g, 1, 0, tag);
    g.trace = dT;
    hipError_t he = gemm_f64(nullptr, LAY_KC, LAY_KC, g, 1, 0, tag);
    he = he == hipSuccess ? hipDeviceSynchronize() : he;
    hipMemcpy(out, dT, (size_t)tiles * 64, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dC); hipFree(dT);
    return he == hipSuccess ? JAICOV_OK : JAICOV_ERR_DEVICE;
}

extern "C" int jaicov_debug_cumask(const uint32_t *mask, int blocks, int iters, double *ms_out, double *tflops) {
    float ms = 0;
    hipError_t he = jaicov::cumask_bench(mask, blocks, iters, &ms, tflops);
    *ms_out = ms;
    return he == hipSuccess ? 0 : (int)he;
}
extern "C" int jaicov_debug_mfma_peak(int blocks, int iters, double *ms_out, double *tflops) {
    float ms = 0;
    hipError_t he = jaicov::mfma_peak_bench(blocks, iters, &ms, tflops);
    *ms_out = ms;
    return he == hipSuccess ? 0 : -5;
}
extern "C" int jaicov_debug_diag_bench(int dbg, int iters, double *ms_out) {
    float ms = 0;
    hipError_t he = jaicov::diag_kernel_bench(dbg, iters, &ms);
    *ms_out = ms;
    return he == hipSuccess ? 0 : -5;
}

// Stand-alone timing of the factorisation on a synthetic SPD matrix of order n (multiple of 128): M = R + n I, R uniform in
// (-0.5, 0.5), right-hand-side rows random.  ms_out[r] = device time of repetition r (HIP events on the solver's stream).
// trace_out (optional, dataflow factorisation only): [tasks][8] of the LAST repetition, see cholflow.hip; tasks_out = task count.
__global__ void fill_spd_kernel(double *L, long ld, int n, int nfact) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j > i || i >= n) return;
    unsigned long long x = 0x9E3779B97F4A7C15ull * ((unsigned long long)i * 65537ull + (unsigned long long)j + 1ull);
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    double v = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    if (i == j && i < nfact) v += (double)nfact;
    L[(long)i * ld + j] = v;
}
extern "C" int jaicov_debug_potrf_bench(int n, int reps, double *ms_out, long long *trace_out, long long trace_cap, int *tasks_out) {
    std::string err;
    if (check_device(err)) return JAICOV_ERR_NO_DEVICE;
    if (n <= 0 || n % 128 || reps < 1 || !ms_out) return JAICOV_ERR_BAD_ARGUMENT;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return JAICOV_ERR_DEVICE;
    DenseSolver ds;
    int status = JAICOV_OK;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    do {
        if (ds.init(s, n, false, true) != hipSuccess) { status = JAICOV_ERR_OUT_OF_MEMORY; break; }
        if (trace_out && ds.flow_ready) ds.flow_enable_trace(true);
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(fill_spd_kernel, dim3((ds.n + 255) / 256, ds.n), dim3(256), 0, s, ds.L, ds.ld, ds.n, n);
            hipEventRecord(e0, s);
            const hipError_t pe = ds.potrf();
            if (pe != hipSuccess) { fprintf(stderr, "potrf: %s\n", hipGetErrorString(pe)); status = JAICOV_ERR_DEVICE; break; }
            hipEventRecord(e1, s);
            const int info = ds.fetch_info();
            if (info < 0) fprintf(stderr, "potrf: info %d (flow_ready %d)\n", info, (int)ds.flow_ready);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms_out[r] = ms;
            if (info < 0) { status = JAICOV_ERR_DEVICE; break; }
            if (info != 0) { status = JAICOV_ERR_SINGULAR; break; }
        }
        if (status == JAICOV_OK && trace_out && ds.flow_trace) {
            const long long cnt = std::min<long long>(trace_cap, ((long long)ds.flow_tasks + n / 128) * 8);   // per task, then per block column (chain kernel)
            hipMemcpy(trace_out, ds.flow_trace, (size_t)cnt * sizeof(long long), hipMemcpyDeviceToHost);
        }
        if (tasks_out) *tasks_out = ds.flow_ready ? ds.flow_tasks : 0;
    } while (0);
    hipStreamSynchronize(s);
    ds.release();
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipStreamDestroy(s);
    return status;
}

// debug: factor a host matrix (n x n row-major, lower part used, n a multiple of 128) with DenseSolver::potrf and return the
// factor (lower part, row-major n x n) -- scripts/flow_small_check.py compares it tile by tile with LAPACK
extern "C" int jaicov_debug_potrf_factor(int n, const double *A, double *L_out) {
    std::string err;
    if (check_device(err)) return JAICOV_ERR_NO_DEVICE;
    if (n <= 0 || n % 128 || !A || !L_out) return JAICOV_ERR_BAD_ARGUMENT;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return JAICOV_ERR_DEVICE;
    DenseSolver ds;
    int status = JAICOV_OK;
    do {
        if (ds.init(s, n, false, true) != hipSuccess) { status = JAICOV_ERR_OUT_OF_MEMORY; break; }
        hipLaunchKernelGGL(fill_spd_kernel, dim3((ds.n + 255) / 256, ds.n), dim3(256), 0, s, ds.L, ds.ld, ds.n, n);   // incl. the right-hand-side rows
        if (hipMemcpy2DAsync(ds.L, ds.ld * sizeof(double), A, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n, hipMemcpyHostToDevice, s) != hipSuccess) { status = JAICOV_ERR_DEVICE; break; }
        if (ds.potrf() != hipSuccess) { status = JAICOV_ERR_DEVICE; break; }
        const int info = ds.fetch_info();
        if (info < 0) { status = JAICOV_ERR_DEVICE; break; }
        if (info > 0) { status = JAICOV_ERR_SINGULAR; break; }
        if (hipMemcpy2DAsync(L_out, (size_t)n * sizeof(double), ds.L, ds.ld * sizeof(double), (size_t)n * sizeof(double), n, hipMemcpyDeviceToHost, s) != hipSuccess) status = JAICOV_ERR_DEVICE;
    } while (0);
    hipStreamSynchronize(s);
    ds.release();
    hipStreamDestroy(s);
    return status;
}

// C (M x N row-major) = alpha * op(A) op(B) + beta * C on the device, host buffers in/out (kernel parity + timing)
extern "C" int jaicov_dense_gemm(int32_t alay, int32_t blay, int32_t M, int32_t N, int32_t K, double alpha, const double *A,
                                 int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc,
                                 int32_t lower_only, int32_t kmode, int32_t repeats, double *ms_out) {

    if (M % 128 || N % 128 || K % 16 || M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return JAICOV_ERR_BAD_ARGUMENT;
    std::string err;
    int rc = check_device(err);
    if (rc) return rc;
    const size_t sa = (size_t)(alay == LAY_KC ? M : K) * lda, sb = (size_t)(blay == LAY_KC ? N : K) * ldb, sc = (size_t)M * ldc;
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    if (hipMalloc(&dA, sa * 8) != hipSuccess || hipMalloc(&dB, sb * 8) != hipSuccess || hipMalloc(&dC, sc * 8) != hipSuccess) {
        hipFree(dA); hipFree(dB); hipFree(dC);
        return JAICOV_ERR_OUT_OF_MEMORY;
    }
    hipMemcpy(dA, A, sa * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B, sb * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C, sc * 8, hipMemcpyHostToDevice);
    GemmArgs g{};
    g.A = dA; g.B = dB; g.C = dC; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta; g.lower_only = lower_only; g.kmode = kmode;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipError_t he = gemm_f64(nullptr, alay, blay, g);
    hipMemcpy(C, dC, sc * 8, hipMemcpyDeviceToHost);
    if (repeats > 0) {
        hipEventRecord(e0, nullptr);
        for (int r = 0; r < repeats; r++) gemm_f64(nullptr, alay, blay, g);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms_out) *ms_out = ms / repeats;
    }
    he = he == hipSuccess ? hipDeviceSynchronize() : he;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dA); hipFree(dB); hipFree(dC);
    return he == hipSuccess ? JAICOV_OK : JAICOV_ERR_DEVICE;
}
