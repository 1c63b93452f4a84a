// assembly_mode = 1: J'WJ of the jointly dispersed image groups as a DENSE contraction on the fp64 matrix cores, the literal
// form of PartialDerivativeFactory.stackNormalEquationSystem's full-weight branch (PDF:486-498) and of SURVEY 8(d)'s
// "dense group" row:   N_g = A_g' (P_g A_g),  n_g = A_g' (P_g w_g),  A_g the m x k Jacobian of the group over the k columns
// it touches (3 per object point seen + interior orientation + distortion + the image's 6 exterior-orientation columns),
// w_g appended as column k.  Per batch of images:
//   pack      P_g -> zero-padded mpad x mpad, [A_g | w_g] -> zero-padded mpad x kpad (dense, row-major), column map
//   GEMM 1    B = P A            gemm_f64_kernel<KC,XC>   2 m^2 (k+1) flop
//   GEMM 2    S = A' B (lower)   gemm_f64_kernel<XC,XC>   m (k+1)(k+2) flop
//   scatter   N[col_i][col_j] += sigma0^2 S_ij, n[col_j] += sigma0^2 S_kj   (fp64 atomics: images share points)
// The structure-aware path (assemble.hip, schur.hip) computes the same sums with ~2 % of the arithmetic because A_g has
// 3 + kc non-zeros per row; this mode exists to measure "J'WJ MFMA utilisation" (BASELINE.json) and as an independent
// second path for the parity tests.  The EO pre-elimination is off in this mode (the full system is assembled).
#include <cstdlib>
#include "ba_kernels.h"
#include "gemm_f32.h"
#include "gemm_f64.h"

namespace jaicov {

__device__ __forceinline__ int dm_shared_local(int c) { return c < 3 ? 3 + c : (c < 9 ? 6 + (c - 3) : 12 + (c - 9)); }
__device__ __forceinline__ int dm_shared_col(const DevProblem &p, int img, int cam, int jb, int c) {
    return c < 3 ? p.io_col[3 * cam + c] : (c < 9 ? p.eo_col[6 * img + (c - 3)] : p.dist_col[jb + (c - 9)]);
}

// P_g (m x m) -> Ppad (mpad x mpad, zero padded).  grid (mpad/256, mpad, batch).  T = double, or float for the
// fp32-accumulate variant of BASELINE config 5 (assembly_mode = 2): operands rounded to fp32, fp32 MFMA accumulation.
template <typename T>
__global__ __launch_bounds__(256) void dm_pack_weight_kernel(DevProblem p, const int32_t *__restrict__ blk_list, int first,
                                                             int n_list, T *__restrict__ Ppad, int mpad) {
    const int b = blockIdx.z;
    if (first + b >= n_list) return;
    const int g = blk_list[first + b];
    const int m = 2 * (p.blk_ip_begin[g + 1] - p.blk_ip_begin[g]);
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= mpad) return;
    const double *P = p.blk_w + p.blk_w_offset[g];
    Ppad[(long)b * mpad * mpad + (long)i * mpad + j] = (T)((i < m && j < m) ? P[(long)i * m + j] : 0.0);
}

// [A_g | w_g] -> Apad (mpad x kpad, zero padded; Apad must be zeroed before), column map cmap[b][kpad] (-1 = no column,
// entry k = -2 marks the misclosure column).  grid (ceil(m/256), batch), thread = row of the group
template <typename T>
__global__ __launch_bounds__(256) void dm_pack_rows_kernel(DevProblem p, const int32_t *__restrict__ blk_list, int first,
                                                           int n_list, const double *__restrict__ rowsA,
                                                           const double *__restrict__ rowsW, T *__restrict__ Apad,
                                                           int32_t *__restrict__ cmap, int mpad, int kpad) {
    const int b = blockIdx.y;
    if (first + b >= n_list) return;
    const int g = blk_list[first + b];
    const int ipb = p.blk_ip_begin[g], mp = p.blk_ip_begin[g + 1] - ipb, m = 2 * mp;
    const long S = p.n_ip;
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    const int jb = p.cam_dist_begin[cam], kc = 9 + p.cam_dist_begin[cam + 1] - jb;
    const int k = 3 * mp + kc;
    const int row = blockIdx.x * 256 + threadIdx.x;
    int32_t *cm = cmap + (long)b * kpad;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < kpad; c += 256) {
            int col = -1;
            if (c < 3 * mp) col = p.point_col[3 * p.ip_point[ipb + c / 3] + c % 3];
            else if (c < k) col = dm_shared_col(p, img, cam, jb, c - 3 * mp);
            else if (c == k) col = -2;
            cm[c] = col;
        }
    }
    if (row >= m) return;
    const int q = row >> 1, r = row & 1;
    T *out = Apad + (long)b * mpad * kpad + (long)row * kpad;
#pragma unroll
    for (int a = 0; a < 3; a++) out[3 * q + a] = (T)rowsA[(long)(2 * a + r) * S + ipb + q];
    for (int c = 0; c < kc; c++) out[3 * mp + c] = (T)rowsA[(long)(2 * dm_shared_local(c) + r) * S + ipb + q];
    out[k] = (T)rowsW[(long)r * S + ipb + q];
}

// scatter of the lower triangle of S (kpad x kpad) through the column map.  grid (kpad/256, kpad, batch)
template <typename T>
__global__ __launch_bounds__(256) void dm_scatter_kernel(const T *__restrict__ Sbuf, const int32_t *__restrict__ cmap,
                                                         int first, int n_list, int kpad, double sigma2,
                                                         double *__restrict__ N, long ld, double *__restrict__ n) {
    const int b = blockIdx.z;
    if (first + b >= n_list) return;
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j > i || j >= kpad) return;
    const int32_t *cm = cmap + (long)b * kpad;
    const int ci = cm[i], cj = cm[j];
    if (ci == -1 || cj < 0) return;
    const double v = sigma2 * (double)Sbuf[(long)b * kpad * kpad + (long)i * kpad + j];
    if (ci == -2) unsafeAtomicAdd(n + cj, v);
    else nadd(N, ld, ci, cj, v);
}

hipError_t DenseMode::init(int max_m, int max_k1, int n_blocks, bool single) {
    fp32 = single;
    mpad = ((max_m + 127) / 128) * 128;
    kpad = ((max_k1 + 127) / 128) * 128;
    // images per launch: the lower-triangular S launch has 78 tiles per image at config 4, so a small batch leaves
    // the last of its ceil(78*batch/512) rounds of workgroups partly empty (batch 16: 2.44 rounds -> 3)
    // and a large one amortises it: 55.4 TFLOP/s at 16, 60.7 at 32, 62.8 at 500 (round 2's sweep).
    // Default: as many images as fit 32 GB of workspace (all 500 of config 4: 26 GB of the 288 GB).
    const size_t es = fp32 ? sizeof(float) : sizeof(double);
    const size_t per_image = ((size_t)mpad * mpad + 2 * (size_t)mpad * kpad + (size_t)kpad * kpad) * es;
    long want = (long)(((size_t)32 << 30) / per_image);
    if (want < 1) want = 1;
    batch = n_blocks < want ? (n_blocks > 0 ? n_blocks : 1) : (int)want;
    hipError_t he;
    if ((he = hipMalloc(&Ppad, (size_t)batch * mpad * mpad * es)) != hipSuccess) return he;
    if ((he = hipMalloc(&Apad, (size_t)batch * mpad * kpad * es)) != hipSuccess) return he;
    if ((he = hipMalloc(&Bbuf, (size_t)batch * mpad * kpad * es)) != hipSuccess) return he;
    if ((he = hipMalloc(&Sbuf, (size_t)batch * kpad * kpad * es)) != hipSuccess) return he;
    if ((he = hipMalloc(&cmap, (size_t)batch * kpad * sizeof(int32_t))) != hipSuccess) return he;
    if ((he = hipEventCreate(&ev0)) != hipSuccess) return he;
    return hipEventCreate(&ev1);
}

void DenseMode::release() {
    hipFree(Ppad); hipFree(Apad); hipFree(Bbuf); hipFree(Sbuf); hipFree(cmap);
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
    Ppad = Apad = Bbuf = Sbuf = nullptr; cmap = nullptr; ev0 = ev1 = nullptr;
}

template <typename T>
static hipError_t dm_batch(DenseMode &d, hipStream_t s, const DevProblem &p, const int32_t *blk_list, int first, int nb,
                           int n_list, const double *rowsA, const double *rowsW, double sigma2, double *N, double *n) {
    const int mpad = d.mpad, kpad = d.kpad;
    T *Ppad = (T *)d.Ppad, *Apad = (T *)d.Apad, *Bbuf = (T *)d.Bbuf, *Sbuf = (T *)d.Sbuf;
    hipError_t he = hipMemsetAsync(Apad, 0, (size_t)nb * mpad * kpad * sizeof(T), s);
    if (he != hipSuccess) return he;
    hipLaunchKernelGGL(dm_pack_weight_kernel<T>, dim3((mpad + 255) / 256, mpad, nb), dim3(256), 0, s, p, blk_list, first, n_list,
                       Ppad, mpad);
    hipLaunchKernelGGL(dm_pack_rows_kernel<T>, dim3((mpad + 255) / 256, nb), dim3(256), 0, s, p, blk_list, first, n_list, rowsA,
                       rowsW, Apad, d.cmap, mpad, kpad);
    if ((he = hipEventRecord(d.ev0, s)) != hipSuccess) return he;
    if (sizeof(T) == sizeof(double)) {
        GemmArgs g1{};   // B = P A
        g1.A = (const double *)Ppad; g1.lda = mpad; g1.B = (const double *)Apad; g1.ldb = kpad; g1.C = (double *)Bbuf; g1.ldc = kpad;
        g1.M = mpad; g1.N = kpad; g1.K = mpad; g1.alpha = 1.0; g1.beta = 0.0; g1.kmode = KMODE_FULL;
        g1.strideA = (long)mpad * mpad; g1.strideB = (long)mpad * kpad; g1.strideC = (long)mpad * kpad;
        if ((he = gemm_f64(s, LAY_KC, LAY_XC, g1, nb)) != hipSuccess) return he;
        GemmArgs g2{};   // S = A' B, lower tiles
        g2.A = (const double *)Apad; g2.lda = kpad; g2.B = (const double *)Bbuf; g2.ldb = kpad; g2.C = (double *)Sbuf; g2.ldc = kpad;
        g2.M = kpad; g2.N = kpad; g2.K = mpad; g2.alpha = 1.0; g2.beta = 0.0; g2.lower_only = 1; g2.kmode = KMODE_FULL;
        g2.strideA = (long)mpad * kpad; g2.strideB = (long)mpad * kpad; g2.strideC = (long)kpad * kpad;
        if ((he = gemm_f64(s, LAY_XC, LAY_XC, g2, nb)) != hipSuccess) return he;
    } else {
        GemmArgsF g1{};
        g1.A = (const float *)Ppad; g1.lda = mpad; g1.B = (const float *)Apad; g1.ldb = kpad; g1.C = (float *)Bbuf; g1.ldc = kpad;
        g1.M = mpad; g1.N = kpad; g1.K = mpad;
        g1.strideA = (long)mpad * mpad; g1.strideB = (long)mpad * kpad; g1.strideC = (long)mpad * kpad;
        if ((he = gemm_f32(s, LAY_KC, LAY_XC, g1, nb)) != hipSuccess) return he;
        GemmArgsF g2{};
        g2.A = (const float *)Apad; g2.lda = kpad; g2.B = (const float *)Bbuf; g2.ldb = kpad; g2.C = (float *)Sbuf; g2.ldc = kpad;
        g2.M = kpad; g2.N = kpad; g2.K = mpad; g2.lower_only = 1;
        g2.strideA = (long)mpad * kpad; g2.strideB = (long)mpad * kpad; g2.strideC = (long)kpad * kpad;
        if ((he = gemm_f32(s, LAY_XC, LAY_XC, g2, nb)) != hipSuccess) return he;
    }
    if ((he = hipEventRecord(d.ev1, s)) != hipSuccess) return he;
    hipLaunchKernelGGL(dm_scatter_kernel<T>, dim3((kpad + 255) / 256, kpad, nb), dim3(256), 0, s, (const T *)Sbuf, d.cmap, first,
                       n_list, kpad, sigma2, N, p.ld, n);
    return hipGetLastError();
}

// returns the summed duration of the GEMM launches through *gemm_ms (HIP events on `s`)
hipError_t DenseMode::assemble(hipStream_t s, const DevProblem &p, const int32_t *blk_list, int n_list, const double *rowsA,
                               const double *rowsW, double sigma2, double *N, double *n, float *gemm_ms) {
    float total = 0.f;
    for (int first = 0; first < n_list; first += batch) {
        const int nb = n_list - first < batch ? n_list - first : batch;
        hipError_t he = fp32 ? dm_batch<float>(*this, s, p, blk_list, first, nb, n_list, rowsA, rowsW, sigma2, N, n)
                             : dm_batch<double>(*this, s, p, blk_list, first, nb, n_list, rowsA, rowsW, sigma2, N, n);
        if (he != hipSuccess) return he;
        if (gemm_ms) {
            if ((he = hipEventSynchronize(ev1)) != hipSuccess) return he;
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev0, ev1);
            total += ms;
        }
    }
    if (gemm_ms) *gemm_ms = total;
    return hipGetLastError();
}

}  // namespace jaicov
