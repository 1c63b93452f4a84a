// Batched inverse of dense SPD matrices: see batchinv.h.  gfx950 only.
#include "batchinv.h"

#include "gemm_f64.h"
#include "potrf_diag.h"

namespace jaicov {

#define HIPCHK(x)                                  \
    do {                                           \
        hipError_t _e = (x);                       \
        if (_e != hipSuccess) return _e;           \
    } while (0)

// diagonal block k of every matrix of the chunk: Cholesky + inverse in LDS (potrf_diag.h), one workgroup per matrix
__global__ __launch_bounds__(256) void potrf_diag_batched_kernel(double *A, long ld, long strideA, double *inv_out, long stride_inv,
                                                                 int *info, int blk) {
    __shared__ double S[128 * DP];
    __shared__ double Wd[8 * 16 * WDP];
    potrf_diag_body(A + (long)blockIdx.x * strideA, ld, inv_out + (long)blockIdx.x * stride_inv, info, blk, 0, S, Wd);
}

// W: zero everywhere except the inverted diagonal blocks (the triangular inverse starts from them)
__global__ __launch_bounds__(256) void trtri_seed_batched_kernel(const double *__restrict__ invd, int nb, double *__restrict__ W, long ld, long msz) {
    const int row = blockIdx.x, b = blockIdx.y;          // one workgroup per matrix row
    const int k = row >> 7, r = row & 127;
    double *w = W + (long)b * msz + (long)row * ld;
    const double *src = invd + ((long)b * nb + k) * 16384 + r * 128;
    for (int c = threadIdx.x; c < ld; c += 256) {
        const int ck = c >> 7;
        w[c] = ck == k ? src[c & 127] : 0.0;
    }
}

// M[c][r] = M[r][c] for c < r, tile by tile through LDS; blockIdx.y = matrix
__global__ __launch_bounds__(256) void symmetrize_batched_kernel(double *M, long ld, long msz) {
    __shared__ double tile[32][33];
    const int t = blockIdx.x;
    int tr = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(tr + 1) * (tr + 2) / 2 <= t) ++tr;
    while ((long)tr * (tr + 1) / 2 > t) --tr;
    const int tc = t - tr * (tr + 1) / 2;
    double *Mb = M + (long)blockIdx.y * msz;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8) tile[y][tx] = Mb[(long)(tr * 32 + y) * ld + tc * 32 + tx];
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int r = tc * 32 + y, c = tr * 32 + tx;
        if (c > r) Mb[(long)r * ld + c] = tile[tx][y];
    }
}


// ---- refinement of the inverses: one Newton-Schulz step X <- X + X (I - D X) with an ACCURATE residual ----------------------------
// An fp64 inverse of a dispersion with cond(D) ~ 2e7 (BASELINE config 4) is good to ~3e-11 of its largest entry whatever the
// algorithm (the reference's dpptrf + dpptri as much as the blocked form above), and N = A' inv(D) A inherits that.  The residual
// R = I - D X (~5e-11) cannot be formed in fp64 either: its terms are ~1e7 times larger than the result.  Split instead (Ozaki's
// error-free scheme, one level): D = D1 + D2, X = X1 + X2 with D1 = the leading 20 bits of every entry relative to its ROW's largest,
// X1 likewise relative to its COLUMN's largest.  Then every product D1[i][k] X1[k][j] is an integer multiple of one unit u_ij and the
// sum of 1024 of them stays below 2^53 u_ij: the fp64 matrix cores compute D1 X1 EXACTLY, whatever their summation order.  The rest,
// D1 X2 + D2 X, is 2^-20 of the size, so its fp64 rounding (1e-16 relative to ITS terms) is ~1e-15 absolute: R is known to 3-4
// digits of itself, and X + X R has the forward error of R^2 plus fp64 rounding of X (measured on the host: 2.9e-11 -> 1.2e-14).
// Three exact-or-small GEMMs + one for X R per matrix; the slices are relative to powers of two, so slicing itself is exact.
constexpr int SLICE_BITS = 20;

// hi = leading SLICE_BITS bits of every entry relative to the row's largest entry, lo = src - hi (exact).  One workgroup per row;
// hi and lo may alias src (each thread reads its entries before any is written: two passes, barrier between).
__global__ __launch_bounds__(256) void slice_rows_kernel(const double *src, double *hi, double *lo, long ld, long msz, int mp) {
    __shared__ double red[256];
    const long base = (long)blockIdx.y * msz + (long)blockIdx.x * ld;
    double v[8];                       // mp <= 2048: the row stays in registers; longer rows (the solver's inverse, below) are read twice
    const bool cached = mp <= 2048;
    double mx = 0.0;
    if (cached) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int j = threadIdx.x + 256 * t;
            v[t] = j < mp ? src[base + j] : 0.0;
            mx = fmax(mx, fabs(v[t]));
        }
    } else {
        for (int j = threadIdx.x; j < mp; j += 256) mx = fmax(mx, fabs(src[base + j]));
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    mx = red[0];
    int ex = 0;
    (void)frexp(mx, &ex);              // mx = f 2^ex, f in [0.5, 1): every |entry| <= mx < 2^ex
    const double unit = ldexp(1.0, ex - SLICE_BITS), inv_unit = ldexp(1.0, SLICE_BITS - ex);
    if (cached) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int j = threadIdx.x + 256 * t;
            if (j < mp) {
                const double h = mx > 0.0 ? rint(v[t] * inv_unit) * unit : 0.0;
                hi[base + j] = h;
                lo[base + j] = v[t] - h;
            }
        }
    } else {                           // (every thread reads and writes its own entries only: hi / lo may alias src here too)
        for (int j = threadIdx.x; j < mp; j += 256) {
            const double x = src[base + j];
            const double h = mx > 0.0 ? rint(x * inv_unit) * unit : 0.0;
            hi[base + j] = h;
            lo[base + j] = x - h;
        }
    }
}

// R = I - T, in place
__global__ __launch_bounds__(256) void eye_minus_kernel(double *T, long ld, long msz, int mp) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= mp) return;
    double *t = T + (long)blockIdx.z * msz + (long)i * ld + j;
    *t = (i == j ? 1.0 : 0.0) - *t;
}

// X <- X + (Y + Y') / 2, full square
__global__ __launch_bounds__(256) void newton_update_kernel(double *X, const double *__restrict__ Y, long ld, long msz, int mp) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= mp) return;
    const long b = (long)blockIdx.z * msz;
    X[b + (long)i * ld + j] += 0.5 * (Y[b + (long)i * ld + j] + Y[b + (long)j * ld + i]);
}

hipError_t BatchedSpdInverse::init(hipStream_t s, int padded_order, int matrices_per_chunk, bool with_refinement) {
    stream = s;
    mp = padded_order; nb = mp / 128; cap = matrices_per_chunk;
    ld = mp; msz = (long)mp * mp;
    const size_t sq = (size_t)cap * msz * sizeof(double);
    HIPCHK(hipMalloc(&Lb, sq));
    HIPCHK(hipMalloc(&Wb, sq));
    HIPCHK(hipMalloc(&Qb, sq));
    refine = with_refinement && mp <= 2048;
    if (refine) {
        HIPCHK(hipMalloc(&Db, sq));
        HIPCHK(hipMalloc(&S1, sq));
        HIPCHK(hipMalloc(&S2, sq));
    }
    HIPCHK(hipMalloc(&invd, (size_t)cap * nb * 16384 * sizeof(double)));
    HIPCHK(hipMemsetAsync(invd, 0, (size_t)cap * nb * 16384 * sizeof(double), s));   // the diagonal kernel writes the lower parts only
    HIPCHK(hipMalloc(&d_info, sizeof(int)));
    HIPCHK(hipMemsetAsync(d_info, 0, sizeof(int), s));
    return hipSuccess;
}

void BatchedSpdInverse::release() {
    hipFree(Lb); hipFree(Wb); hipFree(Qb); hipFree(invd); hipFree(d_info); hipFree(Db); hipFree(S1); hipFree(S2);
    Lb = Wb = Qb = invd = Db = S1 = S2 = nullptr; d_info = nullptr;
}

hipError_t BatchedSpdInverse::run(int count) {
    if (count <= 0) return hipSuccess;
    if (count > cap) return hipErrorInvalidValue;
    // ---- right-looking blocked Cholesky, block 128, all matrices of the chunk in every launch (gridDim.z) ----------------
    for (int k = 0; k < nb; k++) {
        double *Akk = Lb + (long)(k * 128) * ld + k * 128;
        hipLaunchKernelGGL(potrf_diag_batched_kernel, dim3(count), dim3(256), 0, stream, Akk, ld, msz, invd + (long)k * 16384,
                           (long)nb * 16384, d_info, k);
        const int rows = (nb - k - 1) * 128;
        if (rows <= 0) break;
        double *A21 = Lb + (long)((k + 1) * 128) * ld + k * 128;
        GemmArgs g{};              // L21 = A21 inv(L11)' in place: one column tile, every workgroup reads and writes its own rows only
        g.A = A21; g.lda = ld; g.B = invd + (long)k * 16384; g.ldb = 128; g.C = A21; g.ldc = ld;
        g.M = rows; g.N = 128; g.K = 128; g.alpha = 1.0; g.beta = 0.0; g.kmode = KMODE_FULL;
        g.strideA2 = msz; g.strideC2 = msz; g.strideB2 = (long)nb * 16384;
        HIPCHK(gemm_f64(stream, LAY_KC, LAY_KC, g, 1, 1, 0, count));
        GemmArgs u{};              // trailing matrix -= L21 L21' (lower tiles)
        u.A = A21; u.lda = ld; u.B = A21; u.ldb = ld; u.C = Lb + (long)((k + 1) * 128) * (ld + 1); u.ldc = ld;
        u.M = rows; u.N = rows; u.K = 128; u.alpha = -1.0; u.beta = 1.0; u.lower_only = 1; u.kmode = KMODE_FULL;
        u.strideA2 = u.strideB2 = u.strideC2 = msz;
        HIPCHK(gemm_f64(stream, LAY_KC, LAY_KC, u, 1, 0, 0, count));
    }
    // ---- W = L^-1, level by level (dense.hip DenseSolver::trtri, with the matrix index as second batch dimension) -------
    hipLaunchKernelGGL(trtri_seed_batched_kernel, dim3(mp, count), dim3(256), 0, stream, invd, nb, Wb, ld, msz);
    for (int h = 1; h < nb; h *= 2) {
        const int full = nb / (2 * h);
        const long pair_stride = (long)(2 * h) * 128 * (ld + 1);
        auto merge = [&](int lo, int mid, int hi, int batch) -> hipError_t {
            const int M = (hi - mid) * 128, N = (mid - lo) * 128;
            GemmArgs t{};          // T = L21 W11 into the W21 position of Qb (free until the last product fills it)
            t.A = Lb + (long)(mid * 128) * ld + lo * 128; t.lda = ld;
            t.B = Wb + (long)(lo * 128) * ld + lo * 128; t.ldb = ld;
            t.C = Qb + (long)(mid * 128) * ld + lo * 128; t.ldc = ld;
            t.M = M; t.N = N; t.K = N; t.alpha = 1.0; t.beta = 0.0; t.kmode = KMODE_GE_COL;
            t.strideA = t.strideB = t.strideC = pair_stride;
            t.strideA2 = t.strideB2 = t.strideC2 = msz;
            HIPCHK(gemm_f64(stream, LAY_KC, LAY_XC, t, batch, -1, 0, count));
            GemmArgs w{};          // W21 = -W22 T
            w.A = Wb + (long)(mid * 128) * ld + mid * 128; w.lda = ld;
            w.B = t.C; w.ldb = ld;
            w.C = Wb + (long)(mid * 128) * ld + lo * 128; w.ldc = ld;
            w.M = M; w.N = N; w.K = M; w.alpha = -1.0; w.beta = 0.0; w.kmode = KMODE_LE_ROW;
            w.strideA = w.strideB = w.strideC = pair_stride;
            w.strideA2 = w.strideB2 = w.strideC2 = msz;
            return gemm_f64(stream, LAY_KC, LAY_XC, w, batch, -1, 0, count);
        };
        if (full > 0) HIPCHK(merge(0, h, 2 * h, full));
        const int lo = full * 2 * h, mid = lo + h;
        if (mid < nb) HIPCHK(merge(lo, mid, nb, 1));
    }
    // ---- Q = W' W (lower tiles), then the upper triangle by symmetry ---------------------------------------------------
    GemmArgs q{};
    q.A = Wb; q.lda = ld; q.B = Wb; q.ldb = ld; q.C = Qb; q.ldc = ld;
    q.M = mp; q.N = mp; q.K = mp; q.alpha = 1.0; q.beta = 0.0; q.lower_only = 1; q.kmode = KMODE_GE_ROW;
    q.strideA2 = q.strideB2 = q.strideC2 = msz;
    HIPCHK(gemm_f64(stream, LAY_XC, LAY_XC, q, 1, -1, 0, count));
    const int nt = mp / 32;
    hipLaunchKernelGGL(symmetrize_batched_kernel, dim3(nt * (nt + 1) / 2, count), dim3(256), 0, stream, Qb, ld, msz);
    if (!refine) return hipGetLastError();
    // ---- X <- X + sym(X (I - D X)), the residual by the split scheme described above -----------------------------------------------
    // D1 -> Lb, D2 -> Db (in place); row slices of X (X is symmetric: they are the column slices, read transposed): X1' -> Wb, X2' -> S1
    const dim3 gr(mp, count), ge((mp + 255) / 256, mp, count);
    hipLaunchKernelGGL(slice_rows_kernel, gr, dim3(256), 0, stream, Db, Lb, Db, ld, msz, mp);
    hipLaunchKernelGGL(slice_rows_kernel, gr, dim3(256), 0, stream, Qb, Wb, S1, ld, msz, mp);
    auto prod = [&](const double *A, const double *B, int blay, double *C, double alpha, double beta) -> hipError_t {
        GemmArgs g{};
        g.A = A; g.lda = ld; g.B = B; g.ldb = ld; g.C = C; g.ldc = ld;
        g.M = mp; g.N = mp; g.K = mp; g.alpha = alpha; g.beta = beta; g.kmode = KMODE_FULL;
        g.strideA2 = g.strideB2 = g.strideC2 = msz;
        return gemm_f64(stream, LAY_KC, blay, g, 1, 0, 0, count);
    };
    HIPCHK(prod(Lb, Wb, LAY_KC, S2, 1.0, 0.0));            // T = D1 X1, exact (B(k, j) = X1'[j][k])
    hipLaunchKernelGGL(eye_minus_kernel, ge, dim3(256), 0, stream, S2, ld, msz, mp);
    HIPCHK(prod(Lb, S1, LAY_KC, S2, -1.0, 1.0));           // R -= D1 X2
    HIPCHK(prod(Db, Qb, LAY_XC, S2, -1.0, 1.0));           // R -= D2 X
    HIPCHK(prod(Qb, S2, LAY_XC, Lb, 1.0, 0.0));            // Y = X R
    hipLaunchKernelGGL(newton_update_kernel, ge, dim3(256), 0, stream, Qb, Lb, ld, msz, mp);
    return hipGetLastError();
}


// The same step for ONE matrix of order n <= 8192 (20 + 20 bits + log2(n) <= 53: the product of the leading slices is still exact): the inverse
// Q of the solver's scaled matrix M (engine.hip, solve with an inverse; round 5).  An inverse formed from a Cholesky factor at cond(M) ~ 4e8
// (BASELINE config 3) is 2-3e-9 from the exact inverse, correlation-scaled, where the reference's dspsv + dsptri reaches 5e-10
// (tests/golden/cfg3/cfg3_exactN.json): north_star's 1e-9 on the covariances was missed by that factor.  One step with the exact residual
// takes it to the rounding of Q's entries.
//   M, Q: FULL symmetric squares (leading dimension ld); M is destroyed (becomes its low slice), W, T1, T2, T3: n x ld work squares.
hipError_t newton_schulz_exact(hipStream_t stream, int n, long ld, double *M, double *Q, double *W, double *T1, double *T2, double *T3) {
    if (n <= 0 || n % 128 != 0 || n > 8192) return hipErrorInvalidValue;
    const long msz = (long)n * ld;
    const dim3 gr(n, 1), ge((n + 255) / 256, n, 1);
    hipLaunchKernelGGL(slice_rows_kernel, gr, dim3(256), 0, stream, M, T1, M, ld, msz, n);        // D1 -> T1, D2 -> M (in place)
    hipLaunchKernelGGL(slice_rows_kernel, gr, dim3(256), 0, stream, Q, W, T2, ld, msz, n);        // X1' -> W, X2' -> T2 (rows of the symmetric X)
    auto prod = [&](const double *A, const double *B, int blay, double *C, double alpha, double beta) -> hipError_t {
        GemmArgs g{};
        g.A = A; g.lda = ld; g.B = B; g.ldb = ld; g.C = C; g.ldc = ld;
        g.M = n; g.N = n; g.K = n; g.alpha = alpha; g.beta = beta; g.kmode = KMODE_FULL;
        return gemm_f64(stream, LAY_KC, blay, g);
    };
    HIPCHK(prod(T1, W, LAY_KC, T3, 1.0, 0.0));             // T = D1 X1, exact
    hipLaunchKernelGGL(eye_minus_kernel, ge, dim3(256), 0, stream, T3, ld, msz, n);
    HIPCHK(prod(T1, T2, LAY_KC, T3, -1.0, 1.0));           // R -= D1 X2
    HIPCHK(prod(M, Q, LAY_XC, T3, -1.0, 1.0));             // R -= D2 X
    HIPCHK(prod(Q, T3, LAY_XC, T1, 1.0, 0.0));             // Y = X R
    hipLaunchKernelGGL(newton_update_kernel, ge, dim3(256), 0, stream, Q, T1, ld, msz, n);        // X += (Y + Y') / 2
    return hipGetLastError();
}

}  // namespace jaicov
