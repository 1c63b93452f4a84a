// Normal-equation assembly N += A'PA, n += A'Pw (PDF:475-505 stackNormalEquationSystem) from the compact rows,
// structure-aware: the shared camera/EO blocks of an image are reduced on chip (LDS) before they touch HBM, only the
// point-indexed blocks go out as fp64 atomics.  N is row-major LOWER (== UPLO='U' column-major).  gfx950 only.
#include <algorithm>
#include <cstdlib>
#include <string.h>

#include "ba_kernels.h"
#include "gemm_f64.h"

namespace jaicov {

// shared local column c (0..kc) -> local row id l in the rows buffer, and its global column
//   c: 0..2 = x0,y0,c ; 3..8 = X0,Y0,Z0,omega,phi,kappa ; 9.. = distortion coefficients
__device__ __forceinline__ int shared_local(int c) { return c < 3 ? 3 + c : (c < 9 ? 6 + (c - 3) : 12 + (c - 9)); }
__device__ __forceinline__ int shared_col(const DevProblem &p, int img, int cam, int jb, int c) {
    return c < 3 ? p.io_col[3 * cam + c] : (c < 9 ? p.eo_col[6 * img + (c - 3)] : p.dist_col[jb + (c - 9)]);
}

// 2x2 weight of an ordinary image point (PDF:296-319): returns p00, p01, p11
__device__ __forceinline__ void weight2x2(const DevProblem &p, int ip, double sigma2, double &p00, double &p01,
                                          double &p11) {
    const double vx = p.ip_var_x[ip], vy = p.ip_var_y[ip], rho = p.ip_rho[ip];
    if (rho == 0) {
        p00 = sigma2 / vx; p11 = sigma2 / vy; p01 = 0.0;
    } else {
        const double invDet = sigma2 / ((1.0 - rho * rho) * vx * vy);
        p00 = invDet * vy;
        p11 = invDet * vx;
        p01 = -invDet * rho * sqrt(vx * vy);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ordinary (2x2 / diagonal) image points: one workgroup per segment of <= 256 points of one image
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_small_kernel(DevProblem p, const int32_t *__restrict__ seg_begin,
                                                             const int32_t *__restrict__ seg_end,
                                                             const double *__restrict__ rowsA,
                                                             const double *__restrict__ rowsW, double sigma2,
                                                             double *__restrict__ N, double *__restrict__ n) {
    __shared__ double As[2 * SEG * KC_LD];   // [obs row k][c], c = kc holds w
    __shared__ double Ps[SEG * 3];
    __shared__ int cols[KC_MAX];
    const int tid = threadIdx.x;
    const int b = seg_begin[blockIdx.x], e = seg_end[blockIdx.x], cnt = e - b;
    const long S = p.n_ip;
    const int img = p.ip_image[b], cam = p.image_camera[img];
    const int jb = p.cam_dist_begin[cam], nd = p.cam_dist_begin[cam + 1] - jb;
    const int kc = 9 + nd;
    if (tid < kc) cols[tid] = shared_col(p, img, cam, jb, tid);
    // stage the shared sub-rows
    for (int idx = tid; idx < 2 * cnt * (kc + 1); idx += 256) {
        const int c = idx / (2 * cnt), k = idx - c * (2 * cnt);      // k = 2*o + r ... coalesce over o below
        const int r = k / cnt, o = k - r * cnt;
        const double v = (c < kc) ? rowsA[(long)(2 * shared_local(c) + r) * S + b + o] : rowsW[(long)r * S + b + o];
        As[(2 * o + r) * KC_LD + c] = v;
    }
    if (tid < cnt) {
        double p00, p01, p11;
        weight2x2(p, b + tid, sigma2, p00, p01, p11);
        Ps[3 * tid] = p00; Ps[3 * tid + 1] = p01; Ps[3 * tid + 2] = p11;
    }
    __syncthreads();
    // ---- per observation: point blocks ---------------------------------------------------------------------
    if (tid < cnt) {
        const int ip = b + tid, pt = p.ip_point[ip];
        const double p00 = Ps[3 * tid], p01 = Ps[3 * tid + 1], p11 = Ps[3 * tid + 2];
        double ap[2][3], g[2][3];      // g = P * A_p
        int pc[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            ap[0][a] = rowsA[(long)(2 * a) * S + ip];
            ap[1][a] = rowsA[(long)(2 * a + 1) * S + ip];
            g[0][a] = p00 * ap[0][a] + p01 * ap[1][a];
            g[1][a] = p01 * ap[0][a] + p11 * ap[1][a];
            pc[a] = p.point_col[3 * pt + a];
        }
        const double w0 = As[(2 * tid) * KC_LD + kc], w1 = As[(2 * tid + 1) * KC_LD + kc];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (pc[a] < 0) continue;
            unsafeAtomicAdd(n + pc[a], g[0][a] * w0 + g[1][a] * w1);
#pragma unroll
            for (int c2 = 0; c2 <= a; c2++)
                if (pc[c2] >= 0) nadd(N, p.ld, pc[a], pc[c2], g[0][a] * ap[0][c2] + g[1][a] * ap[1][c2]);
        }
        for (int c = 0; c < kc; c++) {
            const int gc = cols[c];
            if (gc < 0) continue;
            const double a0 = As[(2 * tid) * KC_LD + c], a1 = As[(2 * tid + 1) * KC_LD + c];
#pragma unroll
            for (int a = 0; a < 3; a++)
                if (pc[a] >= 0) nadd(N, p.ld, gc, pc[a], g[0][a] * a0 + g[1][a] * a1);
        }
    }
    // ---- shared block: S[a][b] = sum_o A_o[:,a]' P_o A_o[:,b] (b <= a), and n_c ----------------------------
    const int nent = kc * (kc + 1) / 2 + kc;
    for (int ent = tid; ent < nent; ent += 256) {
        int a, bb;
        if (ent < kc) { a = ent; bb = kc; }                      // n entries: column kc = w
        else {
            const int t = ent - kc;
            a = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while ((a + 1) * (a + 2) / 2 <= t) ++a;
            while (a * (a + 1) / 2 > t) --a;
            bb = t - a * (a + 1) / 2;
        }
        const int ga = cols[a];
        const int gb = bb < kc ? cols[bb] : 0;
        if (ga < 0 || gb < 0) continue;
        double s = 0.0;
        for (int o = 0; o < cnt; o++) {
            const double a0 = As[(2 * o) * KC_LD + a], a1 = As[(2 * o + 1) * KC_LD + a];
            const double b0 = As[(2 * o) * KC_LD + bb], b1 = As[(2 * o + 1) * KC_LD + bb];
            const double p00 = Ps[3 * o], p01 = Ps[3 * o + 1], p11 = Ps[3 * o + 2];
            s += a0 * (p00 * b0 + p01 * b1) + a1 * (p01 * b0 + p11 * b1);
        }
        if (bb == kc) unsafeAtomicAdd(n + ga, s);
        else nadd(N, p.ld, ga, gb, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// image blocks (jointly dispersed image points of one image; weight = sigma2 * Dinv, Dinv dense m x m)
// ---------------------------------------------------------------------------------------------------------------
// B1: T = Dinv * [A_c | w]   (m x (kc+1)); P streamed once (symmetric: column r == row r, coalesced).  A thread owns
// TR rows: the [A_c | w] values of a row k are the same for every lane and come from LDS as 16-byte broadcast reads, so
// the LDS instruction count per byte of P (what bounds this kernel) falls with TR.
constexpr int T_TR = 2, T_NT = 128, T_KB = 8;
__global__ __launch_bounds__(T_NT) void blk_T_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                     const double *__restrict__ rowsA, const double *__restrict__ rowsW,
                                                     double *__restrict__ T) {
    __shared__ __attribute__((aligned(16))) double Ac[64 * KC_LD];
    const int g = blk_list[blockIdx.y];
    const int ipb = p.blk_ip_begin[g], m = 2 * (p.blk_ip_begin[g + 1] - ipb);
    const int rbase = blockIdx.x * (T_NT * T_TR) + threadIdx.x;
    if ((int)blockIdx.x * (T_NT * T_TR) >= m || p.blk_w_offset[g] < 0) return;      // (block-diagonal weights: blk_T_diag_kernel)
    const long S = p.n_ip;
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    const int kc = 9 + p.cam_dist_begin[cam + 1] - p.cam_dist_begin[cam];
    const double *P = p.blk_w + p.blk_w_offset[g];
    double acc[T_TR][KC_LD];
    int rr_[T_TR];
    double live[T_TR];
#pragma unroll
    for (int t = 0; t < T_TR; t++) {
        const int r = rbase + T_NT * t;
        rr_[t] = min(r, m - 1);
        live[t] = r < m ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < KC_LD; c++) acc[t][c] = 0.0;
    }
    for (int k0 = 0; k0 < m; k0 += 64) {
        const int kn = min(64, m - k0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < kn * KC_LD; idx += T_NT) {
            const int c = idx / kn, kk = idx - c * kn;
            const int k = k0 + kk, o = k >> 1, rr = k & 1;
            double v = 0.0;
            if (c < kc) v = rowsA[(long)(2 * shared_local(c) + rr) * S + ipb + o];
            else if (c == kc) v = rowsW[(long)rr * S + ipb + o];
            Ac[kk * KC_LD + c] = v;
        }
        __syncthreads();
        // P values of T_KB rows k are fetched ahead of the FMAs that use them (two batches in flight)
        double pk[T_TR][T_KB], pn[T_TR][T_KB];
#pragma unroll
        for (int u = 0; u < T_KB; u++)
#pragma unroll
            for (int t = 0; t < T_TR; t++) pk[t][u] = P[(long)min(k0 + u, m - 1) * m + rr_[t]];
        for (int kb = 0; kb < kn; kb += T_KB) {
#pragma unroll
            for (int u = 0; u < T_KB; u++)
#pragma unroll
                for (int t = 0; t < T_TR; t++) pn[t][u] = P[(long)min(k0 + kb + T_KB + u, m - 1) * m + rr_[t]];
#pragma unroll
            for (int u = 0; u < T_KB; u++) {
                if (kb + u >= kn) break;
#pragma unroll
                for (int c = 0; c < KC_LD; c += 2) {
                    const d2_t a = *reinterpret_cast<const d2_t *>(&Ac[(kb + u) * KC_LD + c]);
#pragma unroll
                    for (int t = 0; t < T_TR; t++) {
                        acc[t][c] += pk[t][u] * a.x;
                        acc[t][c + 1] += pk[t][u] * a.y;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < T_KB; u++)
#pragma unroll
                for (int t = 0; t < T_TR; t++) pk[t][u] = pn[t][u];
        }
    }
#pragma unroll
    for (int t = 0; t < T_TR; t++) {
        const int r = rbase + T_NT * t;
        if (r < m && live[t] != 0.0) {
            double *out = T + ((long)2 * ipb + r) * KC_LD;
#pragma unroll
            for (int c = 0; c < KC_LD; c++) out[c] = acc[t][c];
        }
    }
}

// The same product on the matrix cores (default; JAICOV_T_VECTOR=1 selects the kernel above).  T = Dinv [A_c | w] is a skinny
// GEMM (m x m times m x 30): as vector FMAs it needs 30 v_fma_f64 per 8 bytes of Dinv and ran at 36 % of the fp64 vector rate
// (LDS broadcast reads and FMA issue beside the streaming loads: 1.07 ms for the 4 GB at config 4); one v_mfma_f64_16x16x4
// does the work of 16 of those instructions.  A workgroup = 256 rows of one image, a wave 64 rows x 32 columns (4 x 2 MFMA
// tiles).  The A operand is read through the symmetry of Dinv as its transpose -- lane (l15, l4) of row tile t takes
// Dinv[k + l4][i0 + 16 t + l15]: 16 consecutive doubles per k, whole lines -- 32 k ahead of the MFMAs that use it; the B
// operand [A_c | w] of 64 k comes from LDS.
constexpr int TM_LD = 40;      // LDS row stride of the staged [A_c | w] (32 columns + padding)
__global__ __launch_bounds__(256, 2) void blk_T_mfma_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                            const double *__restrict__ rowsA, const double *__restrict__ rowsW,
                                                            double *__restrict__ T) {
    __shared__ double Ac[64 * TM_LD];
    const int g = blk_list[blockIdx.y];
    const int ipb = p.blk_ip_begin[g], m = 2 * (p.blk_ip_begin[g + 1] - ipb);
    if ((int)blockIdx.x * 256 >= m || p.blk_w_offset[g] < 0) return;      // (block-diagonal weights: blk_T_diag_kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const long S = p.n_ip;
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    const int kc = 9 + p.cam_dist_begin[cam + 1] - p.cam_dist_begin[cam];
    const double *P = p.blk_w + p.blk_w_offset[g];
    const int i0 = blockIdx.x * 256 + 64 * wave;
    d4_t acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int u = 0; u < 2; u++) acc[t][u] = (d4_t){0.0, 0.0, 0.0, 0.0};
    // the lane's four columns of Dinv (clamped: rows beyond m are not stored)
    const double *pc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) pc[t] = P + min(i0 + 16 * t + l15, m - 1);
    double cur[8][4], nxt[8][4];
    auto fetch = [&](double (&v)[8][4], int kbase) __attribute__((always_inline)) {     // 32 k from kbase on (clamped rows: their B rows are zero)
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const long row = (long)min(kbase + 4 * ks + l4, m - 1) * m;
#pragma unroll
            for (int t = 0; t < 4; t++) v[ks][t] = pc[t][row];
        }
    };
    // [A_c | w] of 64 rows k: 8 values per thread, fetched one block ahead as well (zeros beyond m and beyond kc)
    double sv[8];
    auto stage_fetch = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int idx = tid + 256 * j, c = idx >> 6, kk = idx & 63;
            const int k = kb + kk, o = k >> 1, rr = k & 1;
            double v = 0.0;
            if (k < m) {
                if (c < kc) v = rowsA[(long)(2 * shared_local(c) + rr) * S + ipb + o];
                else if (c == kc) v = rowsW[(long)rr * S + ipb + o];
            }
            sv[j] = v;
        }
    };
    stage_fetch(0);
    fetch(cur, 0);
    for (int k0 = 0; k0 < m; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int idx = tid + 256 * j;
            Ac[(idx & 63) * TM_LD + (idx >> 6)] = sv[j];
        }
        __syncthreads();
        stage_fetch(k0 + 64);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            fetch(nxt, k0 + 32 * half + 32);                   // past the end: clamped reads of the last row, never used
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {
                const double *br = Ac + (32 * half + 4 * ks + l4) * TM_LD + l15;
                const double b0 = br[0], b1 = br[16];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[ks][t], b0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[ks][t], b1, acc[t][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 8; ks++)
#pragma unroll
                for (int t = 0; t < 4; t++) cur[ks][t] = nxt[ks][t];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = i0 + 16 * t + l4 + 4 * r;
            if (row >= m) continue;
            double *out = T + ((long)2 * ipb + row) * KC_LD;
            out[l15] = acc[t][0][r];
            if (16 + l15 < KC_LD) out[16 + l15] = acc[t][1][r];
        }
}

// T = inv(D) [A_c | w] of an image whose weights are BLOCK-DIAGONAL (an ordinary image served as an image block, DevProblem::ip_w3): the two
// rows of an image point are its 2 x 2 weight times its own two rows of [A_c | w] -- one thread per image point and column, no sum over the
// image.  grid (ceil(points / 8), n_list), block (32 columns, 8 points).
__global__ __launch_bounds__(256) void blk_T_diag_kernel(DevProblem p, const int32_t *__restrict__ blk_list, const double *__restrict__ rowsA,
                                                         const double *__restrict__ rowsW, double *__restrict__ T) {
    const int g = blk_list[blockIdx.y];
    if (p.blk_w_offset[g] >= 0) return;
    const int ipb = p.blk_ip_begin[g], mp = p.blk_ip_begin[g + 1] - ipb;
    const int o = blockIdx.x * 8 + threadIdx.y, c = threadIdx.x;
    if (o >= mp || c >= KC_LD) return;
    const long S = p.n_ip;
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    const int kc = 9 + p.cam_dist_begin[cam + 1] - p.cam_dist_begin[cam];
    double x0 = 0.0, x1 = 0.0;
    if (c < kc) { x0 = rowsA[(long)(2 * shared_local(c)) * S + ipb + o]; x1 = rowsA[(long)(2 * shared_local(c) + 1) * S + ipb + o]; }
    else if (c == kc) { x0 = rowsW[ipb + o]; x1 = rowsW[S + ipb + o]; }
    const double *w = p.ip_w3 + 3 * (long)(ipb + o);
    double *out = T + ((long)2 * (ipb + o)) * KC_LD + c;
    out[0] = w[0] * x0 + w[1] * x1;
    out[KC_LD] = w[1] * x0 + w[2] * x1;
}

// B2: shared block S_cc = A_c' T_c, n_c = A_c' T_w.  Two launches, and no atomics: every image block deals its rows to
// gridDim.y workgroups (one workgroup per image left 500 serial iterations on a quarter-filled chip) which store their partial
// sums; blk_cc_reduce_kernel then adds them up in a FIXED order (block list order, then part order), so that the camera block of
// N -- the entries every image contributes to -- is the same bit pattern in every run (memory-side fp64 atomics summed it in
// arrival order).  partial: [block in list][part][CC_ENT] doubles.
constexpr int CC_ENT = KC_MAX * (KC_MAX + 1);
__global__ __launch_bounds__(256) void blk_cc_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                     const double *__restrict__ rowsA, const double *__restrict__ T,
                                                     double sigma2, double *__restrict__ partial, int schur) {
    const int g = blk_list[blockIdx.x];
    const int ipb = p.blk_ip_begin[g], m = 2 * (p.blk_ip_begin[g + 1] - ipb);
    const long S = p.n_ip;
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    const int kc = 9 + p.cam_dist_begin[cam + 1] - p.cam_dist_begin[cam];
    const int nent = kc * (kc + 1);   // a in [0,kc), b in [0,kc] (b == kc -> n)
    double *out = partial + ((long)blockIdx.x * gridDim.y + blockIdx.y) * CC_ENT;
    for (int ent = threadIdx.x; ent < nent; ent += 256) {
        const int a = ent / (kc + 1), bb = ent - a * (kc + 1);
        // schur: the EO columns (shared 3..8) are pre-eliminated (schur.hip) and do not enter the reduced system
        const bool skip = (bb < kc && bb > a) || (schur && ((a >= 3 && a < 9) || (bb >= 3 && bb < 9)));
        double s = 0.0;
        if (!skip) {
            const double *ra = rowsA + (long)(2 * shared_local(a)) * S + ipb;
            const double *tb = T + (long)2 * ipb * KC_LD + bb;
            const int o0 = (int)((long)(m / 2) * blockIdx.y / gridDim.y), o1 = (int)((long)(m / 2) * (blockIdx.y + 1) / gridDim.y);
            for (int o = o0; o < o1; o++)
                s += ra[o] * tb[(long)(2 * o) * KC_LD] + ra[S + o] * tb[(long)(2 * o + 1) * KC_LD];
            s *= sigma2;
        }
        out[ent] = s;
    }
}

// grid (CC_ENT, cameras), 256 threads: entry `ent` of camera `cam`.  Entries that involve an exterior-orientation column belong
// to one image: the thread that visits that image's block adds its parts straight into N (sole owner).  The others are summed
// over all blocks of the camera: thread t takes blocks t, t + 256, ... of the list, then a tree over the 256 partial sums.
__global__ __launch_bounds__(256) void blk_cc_reduce_kernel(DevProblem p, const int32_t *__restrict__ blk_list, int n_list,
                                                            int parts, const double *__restrict__ partial,
                                                            double *__restrict__ N, double *__restrict__ n, int schur) {
    __shared__ double red[256];
    const int ent = blockIdx.x, cam = blockIdx.y, tid = threadIdx.x;
    const int jb = p.cam_dist_begin[cam], kc = 9 + p.cam_dist_begin[cam + 1] - jb;
    if (ent >= kc * (kc + 1)) return;
    const int a = ent / (kc + 1), bb = ent - a * (kc + 1);
    if (bb < kc && bb > a) return;
    const bool eo = (a >= 3 && a < 9) || (bb >= 3 && bb < 9);
    if (eo && schur) return;
    double acc = 0.0;
    for (int t = tid; t < n_list; t += 256) {
        const int g = blk_list[t];
        const int img = p.ip_image[p.blk_ip_begin[g]];
        if (p.image_camera[img] != cam) continue;
        double s = 0.0;
        for (int q = 0; q < parts; q++) s += partial[((long)t * parts + q) * CC_ENT + ent];
        if (eo) {
            const int ga = shared_col(p, img, cam, jb, a), gb = bb < kc ? shared_col(p, img, cam, jb, bb) : 0;
            if (ga >= 0 && gb >= 0) {
                if (bb == kc) n[ga] += s;
                else N[(long)max(ga, gb) * p.ld + min(ga, gb)] += s;
            }
        } else
            acc += s;
    }
    if (eo) return;
    red[tid] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        const int ga = shared_col(p, 0, cam, jb, a), gb = bb < kc ? shared_col(p, 0, cam, jb, bb) : 0;   // no EO column: the image is irrelevant
        if (ga >= 0 && gb >= 0) {
            if (bb == kc) n[ga] += red[0];
            else N[(long)max(ga, gb) * p.ld + min(ga, gb)] += red[0];
        }
    }
}

// B3: point x shared blocks and n[point]: thread per (block image point, shared column c <= kc)
__global__ __launch_bounds__(256) void blk_pc_kernel(DevProblem p, const int32_t *__restrict__ blk_of_ip,
                                                     const int32_t *__restrict__ ip_list, int n_list,
                                                     const double *__restrict__ rowsA, const double *__restrict__ T,
                                                     double sigma2, double *__restrict__ N, double *__restrict__ n,
                                                     int schur) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int li = (int)(gid / KC_LD), c = (int)(gid - (long)li * KC_LD);
    if (li >= n_list || (schur && c >= 3 && c < 9)) return;
    const int ip = ip_list[li];
    (void)blk_of_ip;
    const long S = p.n_ip;
    const int img = p.ip_image[ip], cam = p.image_camera[img], pt = p.ip_point[ip];
    const int jb = p.cam_dist_begin[cam], kc = 9 + p.cam_dist_begin[cam + 1] - jb;
    if (c > kc) return;
    const int gc = c < kc ? shared_col(p, img, cam, jb, c) : 0;
    if (gc < 0) return;
    const double t0 = T[((long)2 * ip) * KC_LD + c], t1 = T[((long)2 * ip + 1) * KC_LD + c];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int pc = p.point_col[3 * pt + a];
        if (pc < 0) continue;
        const double v = sigma2 * (rowsA[(long)(2 * a) * S + ip] * t0 + rowsA[(long)(2 * a + 1) * S + ip] * t1);
        if (c == kc) unsafeAtomicAdd(n + pc, v);
        else nadd(N, p.ld, gc, pc, v);
    }
}

// B3': the same sums without memory-side contention.  A (point, shared column) entry of N collects one term per image
// that sees the point (50 at config 4), so the atomic version serialises on its hottest addresses (0.48 ms).  Here one
// thread owns (point, local shared column c) and walks the point's incidences (the CSR of the point x point gather);
// the target column only changes with the camera (c < 3, c >= 9) or with every image (EO, 3 <= c < 9), so the running
// sums are flushed with an atomic whenever it changes -- once per point and column in the usual one-camera block.
__global__ __launch_bounds__(256) void blk_pc_gather_kernel(DevProblem p, PPGather pp, const double *__restrict__ rowsA,
                                                            const double *__restrict__ T, double sigma2,
                                                            double *__restrict__ N, double *__restrict__ n, int schur) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int pt = (int)(gid / KC_LD), c = (int)(gid - (long)pt * KC_LD);
    if (pt >= p.n_points || (schur && c >= 3 && c < 9)) return;
    const int ob = pp.pt_ip_begin[pt], oe = pp.pt_ip_begin[pt + 1];
    if (ob == oe) return;
    const long S = p.n_ip;
    const int pc0 = p.point_col[3 * pt], pc1 = p.point_col[3 * pt + 1], pc2 = p.point_col[3 * pt + 2];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int cur = -2;                     // target: -2 nothing pending, -1 skipped column, -3 the right-hand side, >= 0 column of N
    auto flush = [&]() {
        if (cur == -3) {
            if (pc0 >= 0) unsafeAtomicAdd(n + pc0, s0);
            if (pc1 >= 0) unsafeAtomicAdd(n + pc1, s1);
            if (pc2 >= 0) unsafeAtomicAdd(n + pc2, s2);
        } else if (cur >= 0) {
            if (pc0 >= 0) nadd(N, p.ld, cur, pc0, s0);
            if (pc1 >= 0) nadd(N, p.ld, cur, pc1, s1);
            if (pc2 >= 0) nadd(N, p.ld, cur, pc2, s2);
        }
        s0 = s1 = s2 = 0.0;
    };
    for (int o = ob; o < oe; o++) {
        const PPRecord r = pp.recs[o];
        const int ip = r.ipb + r.lp;
        const int img = p.ip_image[ip], cam = p.image_camera[img];
        const int jb = p.cam_dist_begin[cam], kc = 9 + p.cam_dist_begin[cam + 1] - jb;
        int tgt;
        if (c > kc) tgt = -1;
        else if (c == kc) tgt = -3;
        else { tgt = shared_col(p, img, cam, jb, c); if (tgt < 0) tgt = -1; }
        if (tgt != cur) { flush(); cur = tgt; }
        if (tgt == -1) continue;
        const double t0 = T[((long)2 * ip) * KC_LD + c], t1 = T[((long)2 * ip + 1) * KC_LD + c];
        s0 += sigma2 * (rowsA[ip] * t0 + rowsA[S + ip] * t1);
        s1 += sigma2 * (rowsA[2 * S + ip] * t0 + rowsA[3 * S + ip] * t1);
        s2 += sigma2 * (rowsA[4 * S + ip] * t0 + rowsA[5 * S + ip] * t1);
    }
    flush();
}

// B4: point x point blocks: N[pt_p, pt_q] += A_p' Dinv[rows p, rows q] A_q * sigma2, q <= p within the block.
// One thread per (q, b): it owns COLUMN b of the 3x3 block and issues one atomic per row a, so the three lanes of a
// point write 24 contiguous bytes of one row of N in a single wave instruction (one memory-side atomic request
// instead of three).  grid (ceil(mp/64), mp, n_list), block 192 = 64 points q x 3 columns.
__global__ __launch_bounds__(192) void blk_pp_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                     const double *__restrict__ rowsA, double sigma2,
                                                     double *__restrict__ N) {
    const int g = blk_list[blockIdx.z];
    const int ipb = p.blk_ip_begin[g], mp = p.blk_ip_begin[g + 1] - ipb, m = 2 * mp;
    const int pp = blockIdx.y;
    if (pp >= mp || (int)blockIdx.x * 64 > pp) return;
    const int qq = threadIdx.x / 3, b = threadIdx.x - 3 * qq;
    const int q = blockIdx.x * 64 + qq;
    if (q > pp) return;
    const long S = p.n_ip;
    d2_t P0 = {0.0, 0.0}, P1 = {0.0, 0.0};
    if (p.blk_w_offset[g] >= 0) {
        const double *P = p.blk_w + p.blk_w_offset[g];
        P0 = *reinterpret_cast<const d2_t *>(P + (long)(2 * pp) * m + 2 * q);
        P1 = *reinterpret_cast<const d2_t *>(P + (long)(2 * pp + 1) * m + 2 * q);
    } else if (q == pp) {      // block-diagonal weights: the pair of a point with itself is all there is
        const double *w = p.ip_w3 + 3 * (long)(ipb + pp);
        P0.x = w[0]; P0.y = w[1]; P1.x = w[1]; P1.y = w[2];
    } else {
        return;
    }
    const int ipp = ipb + pp, ipq = ipb + q;
    const int ptp = p.ip_point[ipp], ptq = p.ip_point[ipq];
    const int cq = p.point_col[3 * ptq + b];
    if (cq < 0) return;
    const double aq0 = rowsA[(long)(2 * b) * S + ipq], aq1 = rowsA[(long)(2 * b + 1) * S + ipq];
    const double g0 = sigma2 * (P0.x * aq0 + P0.y * aq1), g1 = sigma2 * (P1.x * aq0 + P1.y * aq1);   // (Dinv_pq A_q)[:, b]
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (pp == q && b > a) continue;          // diagonal pair: lower triangle of the symmetric 3x3 only
        const int cp = p.point_col[3 * ptp + a];
        if (cp < 0) continue;
        nadd(N, p.ld, cp, cq, rowsA[(long)(2 * a) * S + ipp] * g0 + rowsA[(long)(2 * a + 1) * S + ipp] * g1);
    }
}

// B4': point x point blocks without memory-side atomics.  One workgroup owns the rows of ONE object point p (3 rows
// of N) for a chunk of PP_CW columns and keeps that strip in LDS (39 KB: four workgroups per CU); its waves walk the
// images that observe p (wave w takes every (PP_NT/64)-th image) and add A_p' P_pq A_q for the points q of the image
// whose columns fall into the chunk and do not exceed the row's own (lower triangle).  The engine stores a dense block
// in column order, so these q are a contiguous range (PPGather::range, built at create) and the weight entries P_pq are
// streamed exactly once over the whole launch: 2 x 2 doubles per point pair.  Two images may share a partner point,
// so the strip is updated with LDS atomics (ds_add_f64) and the waves never wait for each other; every wave keeps
// record and range of its image after next and the operands of its next image in flight while it accumulates.
// The strip is added to N once at the end: every entry of the point-point block has exactly one owner workgroup.
// FUSED = the EO pre-elimination's rank-6 downdate applied on the fly: the weights are P' = sigma2 Dinv - U U' (schur.hip)
// and the 2 x 2 block P'_pq is formed in registers from the streamed Dinv_pq and the rows U_p (wave-uniform) and U_q (12 more
// doubles per partner, from the 32 MB U buffer that stays in L2 / the Infinity Cache).  P' is then never written to or read
// from HBM (2 x 2 GB per pass at config 4, and the 4 GB buffer), and D^-1 is read once here instead of once more in a kernel
// of its own.
template <bool FUSED>
struct PPData {           // what one thread needs of one partner point q
    d2_t P0, P1;
    double aq[6];
    int cq[3];
    d2_t uq[FUSED ? 6 : 1];   // rows 2q, 2q+1 of U (6 values each)
};

// MIXED: some image blocks have block-diagonal weights in compact form (r.poff < 0, DevProblem::ip_w3: an ordinary image served as a block):
// nothing is streamed for them, the pair of the row point with itself carries its 2 x 2 weight and every other pair the rank-6 term only.
template <bool FUSED, bool MIXED>
__device__ __forceinline__ void pp_load(PPData<FUSED> &d, const DevProblem &p, const PPRecord &r, const int32_t *__restrict__ ipcol,
                                        const double *__restrict__ rowsA, const double *__restrict__ Ubuf, int q, int qend,
                                        const double *__restrict__ ug) {
    const long S = p.n_ip;
    if (q < qend) {
        if (!MIXED || r.poff >= 0) {
            const int m = 2 * r.mp;
            const double *P = p.blk_w + r.poff + (long)(2 * r.lp) * m + 2 * q;
            d.P0 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(P));      // streamed exactly once over the launch
            d.P1 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(P + m));
        } else {
            const double *w = p.ip_w3 + 3 * (long)(r.ipb + r.lp);                      // the same for every lane
            const bool self = q == r.lp;
            d.P0.x = self ? w[0] : 0.0; d.P0.y = self ? w[1] : 0.0;
            d.P1.x = self ? w[1] : 0.0; d.P1.y = self ? w[2] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < 3; b++) {
            d.cq[b] = ipcol[(long)b * S + r.ipb + q];          // SoA like the other partner operands (round 5: as [ip][3] a wave's load touched 6 lines instead of 2)
            d.aq[2 * b] = rowsA[(long)(2 * b) * S + r.ipb + q];
            d.aq[2 * b + 1] = rowsA[(long)(2 * b + 1) * S + r.ipb + q];
        }
        if (FUSED) {
            // U in the gather's layout (schur.hip, blk_elim_kernel): six coalesced 16-byte loads, 8 cache lines per wave instruction.
            // From the row-major U buffer (a lane per 128-byte record) every one of these instructions touched 64 lines: -0.38 ms.
            (void)Ubuf;
            const d2_t *u = reinterpret_cast<const d2_t *>(ug) + r.ipb + q;
#pragma unroll
            for (int k = 0; k < 6; k++) d.uq[k] = u[(long)k * S];
        }
    } else {
        d.cq[0] = d.cq[1] = d.cq[2] = -1;
    }
}

// The row point's own operands are the same for the 64 lanes of a wave: A_p (6 values) and, FUSED, its two rows of U (12).
// They are fetched ahead as ONE vector load (lane l < 6: A_p[l]; lanes 8..19: U) and spread to scalar registers with
// v_readlane when the wave moves on to that image -- prefetching them as scalars would need 36 more SGPRs than there are.
template <bool FUSED>
__device__ __forceinline__ double pp_fetch_row(const DevProblem &p, const PPRecord &r, const double *__restrict__ rowsA,
                                               const double *__restrict__ Ubuf, int lane) {
    const long S = p.n_ip;
    const int ip = r.ipb + r.lp;
    const int k = lane - 8;
    const double *src = lane < 6 ? rowsA + (long)lane * S + ip
                                 : (FUSED && k >= 0 && k < 12 ? Ubuf + (long)2 * ip * 8 + (k < 6 ? k : k + 2) : nullptr);
    return src ? *src : 0.0;
}
__device__ __forceinline__ double pp_lane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <bool FUSED>
__device__ __forceinline__ void pp_spread_row(double rowvec, double sigma2, double (&ap)[6], double (&up)[12]) {
#pragma unroll
    for (int a = 0; a < 6; a++) ap[a] = (FUSED ? 1.0 : sigma2) * pp_lane(rowvec, a);   // FUSED: the factor goes onto Dinv
    if (FUSED) {
#pragma unroll
        for (int k = 0; k < 12; k++) up[k] = pp_lane(rowvec, 8 + k);
    }
}

// What one partner contributes, formed before anything is added: g[b] = P'_pq A_q[:, b] (2 values per partner column b).
template <bool FUSED>
__device__ __forceinline__ void pp_products(const PPData<FUSED> &d, const double (&up)[12], double sigma2, double (&g)[6]) {
    double p00 = d.P0.x, p01 = d.P0.y, p10 = d.P1.x, p11 = d.P1.y;
    if (FUSED) {
        p00 *= sigma2; p01 *= sigma2; p10 *= sigma2; p11 *= sigma2;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            p00 -= up[2 * k] * d.uq[k].x + up[2 * k + 1] * d.uq[k].y;
            p01 -= up[2 * k] * d.uq[3 + k].x + up[2 * k + 1] * d.uq[3 + k].y;
            p10 -= up[6 + 2 * k] * d.uq[k].x + up[6 + 2 * k + 1] * d.uq[k].y;
            p11 -= up[6 + 2 * k] * d.uq[3 + k].x + up[6 + 2 * k + 1] * d.uq[3 + k].y;
        }
    }
#pragma unroll
    for (int b = 0; b < 3; b++) {
        g[2 * b] = p00 * d.aq[2 * b] + p01 * d.aq[2 * b + 1];
        g[2 * b + 1] = p10 * d.aq[2 * b] + p11 * d.aq[2 * b + 1];
    }
}
__device__ __forceinline__ void pp_apply(const double (&g)[6], const int (&cqs)[3], const double (&ap)[6], double *strip, int cw, int c0,
                                         int cp0, int cp1, int cp2, int wlo, int whi) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const int cq = cqs[b];
        if (cq < c0 + wlo || cq >= c0 + whi) continue;      // [wlo, whi): the part of the strip this wave may touch
        const double g0 = g[2 * b], g1 = g[2 * b + 1];
        if (cp0 >= cq) unsafeAtomicAdd(&strip[cq - c0], ap[0] * g0 + ap[1] * g1);
        if (cp1 >= cq) unsafeAtomicAdd(&strip[cw + cq - c0], ap[2] * g0 + ap[3] * g1);
        if (cp2 >= cq) unsafeAtomicAdd(&strip[2 * cw + cq - c0], ap[4] * g0 + ap[5] * g1);
    }
}
template <bool FUSED>
__device__ __forceinline__ void pp_accumulate(const PPData<FUSED> &d, const double (&ap)[6], const double (&up)[12], double sigma2,
                                              double *strip, int cw, int c0, int cp0, int cp1, int cp2, int wlo, int whi) {
    double g[6];
    pp_products<FUSED>(d, up, sigma2, g);
    pp_apply(g, d.cq, ap, strip, cw, c0, cp0, cp1, cp2, wlo, whi);
}

// everything a wave derives from its image record is the same for its 64 lanes: held in scalar registers (the compiler cannot
// see that `wave` is uniform), which is what leaves the vector registers for the U rows at four workgroups per CU
__device__ __forceinline__ PPRecord pp_record(const PPRecord *recs, int o) {
    const int4 *src = reinterpret_cast<const int4 *>(recs + o);
    const int4 a = src[0], b = src[1];
    PPRecord r;
    r.ipb = __builtin_amdgcn_readfirstlane(a.x); r.mp = __builtin_amdgcn_readfirstlane(a.y);
    r.lp = __builtin_amdgcn_readfirstlane(a.z); r.pad = 0;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(b.x), hi = (unsigned)__builtin_amdgcn_readfirstlane(b.y);
    r.poff = (int64_t)(((unsigned long long)hi << 32) | lo);
    r.pad2 = 0;
    return r;
}
__device__ __forceinline__ int2 pp_range(const int2 *range, long idx) {
    const int2 g = range[idx];
    return make_int2(__builtin_amdgcn_readfirstlane(g.x), __builtin_amdgcn_readfirstlane(g.y));
}

template <bool FUSED, bool DET = false, bool MIXED = false, bool PASSES = false>
__global__ __launch_bounds__(PP_NT, 4) void blk_pp_gather_kernel(DevProblem p, PPGather pp, const double *__restrict__ rowsA,
                                                            const double *__restrict__ Ubuf, double sigma2, double *__restrict__ N) {
    extern __shared__ double strip[];              // 3 rows x cw columns, + one word (DET: the turn, see below)
    constexpr int NW = PP_NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, nch = pp.n_chunks, cw = pp.cw;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Which (point, chunk).  Default: grid (points, chunks).  xcd_map: a 1-D grid in which block b belongs to XCD b % 8 (workgroups
    // are dealt round-robin to the XCDs: an observation used for speed only, every (point, chunk) is computed exactly once whatever
    // the placement) and every XCD walks its OWN chunks one after the other, all points of a chunk in order.  The partner records
    // of a chunk -- A_q, U_q, columns: 156 B per image point, ~2.4 MB per chunk of 960 columns at config 4, re-read by every row
    // point of the image -- then stay in that XCD's 4 MB L2 instead of being fetched from the Infinity Cache / HBM by all eight
    // (11 GB of fetches per launch against 2.9 GB algorithmic before).  Chunks are dealt boustrophedon (0..7, 15..8, 16..23, ...)
    // because the work of a chunk falls with its index (lower triangle).
    int pt, chunk;
    if (pp.xcd_map) {
        const int b = blockIdx.x, xcd = b & 7, sl = b >> 3;
        const int kth = sl / p.n_points;
        pt = sl - kth * p.n_points;
        chunk = 8 * kth + ((kth & 1) ? 7 - xcd : xcd);
        if (chunk >= nch) return;
    } else {
        pt = blockIdx.x; chunk = blockIdx.y;
    }
    const int c0 = pp.cmin + chunk * cw;
    const int cp0 = p.point_col[3 * pt], cp1 = p.point_col[3 * pt + 1], cp2 = p.point_col[3 * pt + 2];
    const int rmax = max(cp0, max(cp1, cp2));
    const int ob = pp.pt_ip_begin[pt], oe = pp.pt_ip_begin[pt + 1];
    if (rmax < c0 || (ob == oe && !(pp.plain && rmax <= pp.cmax))) return;   // plain mode: an unobserved point stores zeros
    const int2 *range = reinterpret_cast<const int2 *>(pp.range);
    int *det_turn = reinterpret_cast<int *>(strip + 3 * cw);
    if (DET && tid == 0) *det_turn = 0;
    for (int i = tid; i < 3 * cw; i += PP_NT) strip[i] = 0.0;
    __syncthreads();
    // Wave w takes every NW-th image of the point, with the partner ranges of the whole chunk.
    constexpr int STEP = NW;
    const int2 *rng = range;
    const int nrc = nch, rci = chunk;
    const int wlo = 0, whi = cw;
    // One loop for both forms, over SEGMENTS of 64 partners: a wave holds the operands of the segment at hand (`cur`) and has the
    // next one in flight (`nxt`): the next 64 partners of the same image if its range in this chunk is longer than a wave, else the
    // first 64 of the wave's next image (with that image's row operands).  Round 3 prefetched across images only and walked the rest of
    // a long range without prefetch -- rare on the SURVEY 8(d) scene (55 +- 7 partners per image and chunk), the rule on a block flown
    // in strips, where the partners of a point sit in one or two chunks (cfg4_local: up to 500 per image and chunk).
    // (Measured alternatives of round 3, all slower at config 4: loads without the else branch 3.2-3.6 ms -- the compiler then keeps old
    // and new contents of the operand registers alive and spills --, unconditional loads from clamped indices 3.3-3.7, two operand sets
    // used alternately instead of the copy 3.6.)
    // (Round 5: the instance the default pass runs -- FUSED, deterministic, dense blocks only -- deals SEGMENTS pass by pass instead, see
    // PASSES above; what follows is the image-major loop of the arrival-order form and of the MIXED / unfused deterministic instances.)
    // DETERMINISTIC form (the default): the ADDS into the strip happen in image order.  A wave forms the products of its segment
    // (pp_products: everything up to the 18 multiplications by A_p) while earlier images are still being added, waits for its image's
    // turn -- a sequence word in LDS that counts the images of this point that have been added --, issues its LDS adds (all segments
    // of the image inside the one turn), drains them (s_waitcnt lgkmcnt(0): no vmcnt, the prefetched operands stay in flight) and
    // passes the turn on.  Every entry of the strip is summed in image order whatever the timing; only the adds themselves are
    // serialised inside a workgroup.  (Round 3's form let the waves add in turn with a workgroup barrier after every turn, the products
    // formed inside the turn: +0.7 ms per pass at config 4; round 2's walked all images with every wave: +1.7 ms; this one +0.3.  All
    // three sum in image order: identical bits.)
    if constexpr (PASSES) {
        // PASS-MAJOR distribution (round 5).  The unit of work is a SEGMENT of 64 partners; the segments of this (point, chunk) are ordered
        // (window of 64 images, segment number s inside its image, image) and dealt to the waves in turn, the deterministic form adds them in
        // exactly that order (the turn word counts segments).  An image whose partners fill k segments -- the rule on a block flown in strips --
        // no longer holds the turn for k segments while three waves wait with one product set each: four waves form products side by side
        // in every pass, as on a scene with one segment per image.  Images without a partner in this chunk (most of them, with locality)
        // are never visited: every wave reads the ranges of up to 64 images of the point with ONE vector load and finds its segments by
        // scalar bit scans of a ballot.
        PPData<FUSED> cur, nxt;
        double apc[6], upc[12];
        int turn_base = 0;
        for (int wb = ob; wb < oe; wb += 64) {
            int2 rl = make_int2(0, 0);
            if (wb + lane < oe) rl = rng[(long)(wb + lane) * nrc + rci];
            const int nseg_l = rl.y > rl.x ? (rl.y - rl.x + 63) >> 6 : 0;
            // scalar walk over (pass s, images with a segment s): state of the generator
            int gs = 0, gidx = 0;                        // pass, rank of the NEXT candidate among the images of this pass
            unsigned long long gm = __ballot(nseg_l > 0);
            int gbase = turn_base;                       // turn position of rank 0 of this pass
            // item = (lane b of its image in the window, pass s, turn position); b < 0: none left
            auto next_item = [&](int &b, int &sg, int &pos) __attribute__((always_inline)) {
                for (;;) {
                    if (gm == 0ull) {                    // next pass
                        gbase += gidx;
                        ++gs; gidx = 0;
                        gm = __ballot(nseg_l > gs);
                        if (gm == 0ull) { b = -1; return; }
                    }
                    const int bit = __builtin_ctzll(gm);
                    gm &= gm - 1ull;
                    const int k = gidx++;
                    if ((k & (NW - 1)) == wave) { b = bit; sg = gs; pos = gbase + k; return; }
                }
            };
            static_assert((NW & (NW - 1)) == 0, "waves per workgroup: a power of two");
            int b0, s0, p0, b1, s1, p1;
            next_item(b0, s0, p0);
            if (b0 >= 0) {
                next_item(b1, s1, p1);
                PPRecord r0 = pp_record(pp.recs, wb + b0);
                PPRecord r1 = pp_record(pp.recs, wb + (b1 >= 0 ? b1 : b0));
                {
                    const int gx = __builtin_amdgcn_readlane(rl.x, b0), gy = __builtin_amdgcn_readlane(rl.y, b0);
                    pp_load<FUSED, MIXED>(cur, p, r0, pp.ipcol, rowsA, Ubuf, gx + 64 * s0 + lane, gy, pp.ug);
                    pp_spread_row<FUSED>(pp_fetch_row<FUSED>(p, r0, rowsA, Ubuf, lane), sigma2, apc, upc);
                }
                for (;;) {
                    int b2, s2, p2;
                    if (b1 >= 0) next_item(b2, s2, p2); else b2 = -1;
                    const PPRecord r2 = pp_record(pp.recs, wb + (b2 >= 0 ? b2 : (b1 >= 0 ? b1 : b0)));      // two items ahead, like the image-major loop
                    double rown = 0.0;
                    if (b1 >= 0) {
                        const int gx = __builtin_amdgcn_readlane(rl.x, b1), gy = __builtin_amdgcn_readlane(rl.y, b1);
                        pp_load<FUSED, MIXED>(nxt, p, r1, pp.ipcol, rowsA, Ubuf, gx + 64 * s1 + lane, gy, pp.ug);
                        rown = pp_fetch_row<FUSED>(p, r1, rowsA, Ubuf, lane);
                    }
                    if constexpr (DET) {
                        double gq[6];
                        const int cqs[3] = {cur.cq[0], cur.cq[1], cur.cq[2]};
                        pp_products<FUSED>(cur, upc, sigma2, gq);
                        int spin = 0;
                        while (__hip_atomic_load(det_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != p0) {
                            __builtin_amdgcn_s_sleep(0);
                            if (++spin > (1 << 26)) { gq[0] = __builtin_nan(""); break; }      // (bounded: see the image-major loop)
                        }
                        asm volatile("" ::: "memory");
                        pp_apply(gq, cqs, apc, strip, cw, c0, cp0, cp1, cp2, wlo, whi);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(det_turn, p0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        pp_accumulate<FUSED>(cur, apc, upc, sigma2, strip, cw, c0, cp0, cp1, cp2, wlo, whi);
                    }
                    if (b1 < 0) break;
                    cur = nxt;
                    pp_spread_row<FUSED>(rown, sigma2, apc, upc);
                    b0 = b1; s0 = s1; p0 = p1; r0 = r1;
                    b1 = b2; s1 = s2; p1 = p2; r1 = r2;
                }
            }
            // every wave has walked the same lists: the turn positions of the next window start behind this one's segments
            {
                int total = 0;
                for (int sg = 0;; sg++) {
                    const unsigned long long m = __ballot(nseg_l > sg);
                    if (m == 0ull) break;
                    total += __builtin_popcountll(m);
                }
                turn_base += total;
            }
        }
    } else
    if (ob + wave < oe) {
        int o = ob + wave;
        PPRecord r1 = pp_record(pp.recs, o);
        int2 g1 = pp_range(rng, (long)o * nrc + rci);
        const int o1 = min(o + STEP, oe - 1);
        PPRecord r2 = pp_record(pp.recs, o1);
        int2 g2 = pp_range(rng, (long)o1 * nrc + rci);
        PPData<FUSED> cur, nxt;
        double apc[6], upc[12];
        int js = g1.x;                      // first partner of the segment at hand
        bool first = true;                  // ... which is the first of its image (DET: the turn has not been taken yet)
        pp_load<FUSED, MIXED>(cur, p, r1, pp.ipcol, rowsA, Ubuf, js + lane, g1.y, pp.ug);
        pp_spread_row<FUSED>(pp_fetch_row<FUSED>(p, r1, rowsA, Ubuf, lane), sigma2, apc, upc);
        for (;;) {
            const bool more = js + 64 < g1.y;
            const int o2 = min(o + 2 * STEP, oe - 1);
            const PPRecord r3 = pp_record(pp.recs, o2);
            const int2 g3 = pp_range(rng, (long)o2 * nrc + rci);
            double rown = 0.0;
            const bool adv = !more && o + STEP < oe;
            // ONE load site for `nxt` (two, one per branch, and the compiler keeps both sets of operand registers: 66 spilled VGPRs):
            // which record and which partners is decided in scalar registers
            PPRecord rn;
            rn.ipb = more ? r1.ipb : r2.ipb; rn.mp = more ? r1.mp : r2.mp; rn.lp = more ? r1.lp : r2.lp; rn.pad = 0;
            rn.poff = more ? r1.poff : r2.poff; rn.pad2 = 0;
            const int qn = more ? js + 64 : g2.x, qe = more ? g1.y : g2.y;
            if (more || adv) {
                pp_load<FUSED, MIXED>(nxt, p, rn, pp.ipcol, rowsA, Ubuf, qn + lane, qe, pp.ug);
                if (adv) rown = pp_fetch_row<FUSED>(p, r2, rowsA, Ubuf, lane);
            }
            if constexpr (DET) {
                double gq[6];
                const int cqs[3] = {cur.cq[0], cur.cq[1], cur.cq[2]};
                pp_products<FUSED>(cur, upc, sigma2, gq);
                if (first) {                // my image's turn: o - ob images have been added
                    // bounded: every image of [ob, oe) passes the turn today, but a wave that ever left early would otherwise
                    // hang the GPU; on expiry (2^26 polls, seconds) the products are poisoned -- N gets NaNs, the solve reports NOT_FINITE
                    int spin = 0;
                    while (__hip_atomic_load(det_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != o - ob) {
                        __builtin_amdgcn_s_sleep(0);
                        if (++spin > (1 << 26)) { gq[0] = __builtin_nan(""); break; }
                    }
                    asm volatile("" ::: "memory");
                }
                pp_apply(gq, cqs, apc, strip, cw, c0, cp0, cp1, cp2, wlo, whi);
                if (!more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's adds have been performed (LDS executes a wave's operations in order)
                    if (lane == 0) __hip_atomic_store(det_turn, o - ob + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                pp_accumulate<FUSED>(cur, apc, upc, sigma2, strip, cw, c0, cp0, cp1, cp2, wlo, whi);
            }
            if (more) {
                js += 64;
                cur = nxt;
                first = false;
                continue;
            }
            o += STEP;
            if (o >= oe) break;
            cur = nxt;
            pp_spread_row<FUSED>(rown, sigma2, apc, upc);
            r1 = r2; g1 = g2; r2 = r3; g2 = g3;
            js = g1.x;
            first = true;
        }
    }
    __syncthreads();
    const int cps[3] = {cp0, cp1, cp2};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int r = cps[a];
        if (r < 0) continue;
        const int cend = min(cw, r - c0 + 1);
        double *nrow = N + (long)r * p.ld + c0;
        if (pp.plain) {
            for (int c = tid; c < cend; c += PP_NT) nrow[c] = strip[a * cw + c];
        } else {
            for (int c = tid; c < cend; c += PP_NT) {
                const double v = strip[a * cw + c];
                if (v != 0.0) nrow[c] += v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// scale bars (PDF:210-283): one thread each
// ---------------------------------------------------------------------------------------------------------------
__global__ void scalebar_kernel(DevProblem p, const double *__restrict__ vals, double sigma2, double *N, double *n,
                                const double *dx, double *omega) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.n_sb) return;
    const int pa = p.sb_a[s], pb = p.sb_b[s];
    const double *a = vals + 3 * pa, *b = vals + 3 * pb;
    const double dX = b[0] - a[0], dY = b[1] - a[1], dZ = b[2] - a[2];
    const double len = sqrt(dX * dX + dY * dY + dZ * dZ);
    const double ev[6] = {-dX / len, -dY / len, -dZ / len, dX / len, dY / len, dZ / len};
    const double P = sigma2 / p.sb_var[s];
    const double w = p.sb_len[s] - len;
    int col[6];
    for (int t = 0; t < 6; t++) col[t] = t < 3 ? p.point_col[3 * pa + t] : p.point_col[3 * pb + t - 3];
    if (dx) {   // omega mode (BA:480-488)
        double v = w;
        for (int t = 0; t < 6; t++)
            if (col[t] >= 0) v -= ev[t] * dx[col[t]];
        unsafeAtomicAdd(omega, v * P * v);
        return;
    }
    for (int t = 0; t < 6; t++) {
        if (col[t] < 0) continue;
        unsafeAtomicAdd(n + col[t], ev[t] * P * w);
        for (int u = 0; u <= t; u++)
            if (col[u] >= 0) nadd(N, p.ld, col[t], col[u], ev[t] * P * ev[u]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// directly observed groups (PDF:447-473): A is a selection matrix -> N[cols,cols] += P, n[cols] += P w
// one workgroup per group
// ---------------------------------------------------------------------------------------------------------------
// Per-row data of the group (column, v = observed - value - dx) are staged in LDS by one thread per row FIRST: read inside the
// row x column loops they were 45 dependent chains of five global loads each for a 45-row group -- 60 us of pure latency on the
// critical path of every pass at config 4.  DIRECT_STAGE rows per group in LDS; longer groups read past it from memory.
constexpr int DIRECT_STAGE = 2048;
__global__ __launch_bounds__(256) void direct_kernel(DevProblem p, const double *__restrict__ vals, double sigma2,
                                                     double *N, double *n, const double *dx, double *omega) {
    const int g = blockIdx.x;
    const int b = p.dg_row_begin[g], m = p.dg_row_begin[g + 1] - b;
    const long woff = p.dg_w_offset[g];
    const double *W = woff >= 0 ? p.dg_w + woff : nullptr;
    __shared__ double red[256];
    __shared__ double s_v[DIRECT_STAGE];
    __shared__ int s_col[DIRECT_STAGE];
    auto row_col = [&](int c) { return p.slot_col[p.dg_slot[b + c]]; };
    auto row_v = [&](int c, int cc) {
        double vc = p.dg_obs[b + c] - vals[p.dg_slot[b + c]];
        if (dx && cc >= 0) vc -= dx[cc];
        return vc;
    };
    for (int c = threadIdx.x; c < m && c < DIRECT_STAGE; c += 256) {
        const int cc = row_col(c);
        s_col[c] = cc;
        s_v[c] = row_v(c, cc);
    }
    __syncthreads();
    auto col_of = [&](int c) { return c < DIRECT_STAGE ? s_col[c] : row_col(c); };
    auto v_of = [&](int c) { return c < DIRECT_STAGE ? s_v[c] : row_v(c, row_col(c)); };
    double om = 0.0;
    for (int r = threadIdx.x; r < m; r += 256) {
        const int cr = col_of(r);
        const double vr = v_of(r);
        double pw = 0.0;    // (P v)[r], v = w - A dx
        if (W) {
            for (int c = 0; c < m; c++) pw += sigma2 * W[(long)r * m + c] * v_of(c);
        } else {
            pw = sigma2 / p.dg_var[b + r] * vr;
        }
        if (dx) {
            om += vr * pw;
        } else if (cr >= 0) {
            unsafeAtomicAdd(n + cr, pw);
            if (W) {
                for (int c = 0; c <= r; c++) {
                    const int cc = col_of(c);
                    if (cc >= 0) nadd(N, p.ld, cr, cc, sigma2 * W[(long)r * m + c]);
                }
            } else {
                nadd(N, p.ld, cr, cr, sigma2 / p.dg_var[b + r]);
            }
        }
    }
    if (dx) {
        red[threadIdx.x] = om;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) unsafeAtomicAdd(omega, red[0]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// omega (BA:472-491): v = w - A dx per image point; ordinary points reduce v'Pv directly, block points park v
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void omega_points_kernel(DevProblem p, const uint8_t *__restrict__ in_block,
                                                           int ip0, int count, const double *__restrict__ rowsA,
                                                           const double *__restrict__ rowsW,
                                                           const double *__restrict__ dx, double sigma2,
                                                           double *__restrict__ vbuf, double *omega) {
    __shared__ double red[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double om = 0.0;
    if (i < count) {
        const int ip = ip0 + i;
        const long S = p.n_ip;
        const int img = p.ip_image[ip], cam = p.image_camera[img], pt = p.ip_point[ip];
        const int jb = p.cam_dist_begin[cam], nd = p.cam_dist_begin[cam + 1] - jb;
        double v0 = rowsW[ip], v1 = rowsW[S + ip];
        for (int l = 0; l < 12 + nd; l++) {
            const int col = l < 3 ? p.point_col[3 * pt + l]
                                  : (l < 6 ? p.io_col[3 * cam + l - 3] : (l < 12 ? p.eo_col[6 * img + l - 6] : p.dist_col[jb + l - 12]));
            if (col < 0) continue;
            const double d = dx[col];
            v0 -= rowsA[(long)(2 * l) * S + ip] * d;
            v1 -= rowsA[(long)(2 * l + 1) * S + ip] * d;
        }
        if (in_block[ip]) {
            vbuf[2 * (long)ip] = v0;
            vbuf[2 * (long)ip + 1] = v1;
        } else {
            double p00, p01, p11;
            weight2x2(p, ip, sigma2, p00, p01, p11);
            om = v0 * (p00 * v0 + p01 * v1) + v1 * (p01 * v0 + p11 * v1);
        }
    }
    red[threadIdx.x] = om;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] != 0.0) unsafeAtomicAdd(omega, red[0]);
}

// omega of an image block: sigma2 * v' Dinv v ; grid (ceil(m/256), n_list)
__global__ __launch_bounds__(256) void omega_block_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                          const double *__restrict__ vbuf, double sigma2,
                                                          double *omega) {
    __shared__ double red[256];
    const int g = blk_list[blockIdx.y];
    const int ipb = p.blk_ip_begin[g], m = 2 * (p.blk_ip_begin[g + 1] - ipb);
    const int r = blockIdx.x * 256 + threadIdx.x;
    const double *v = vbuf + 2 * (long)ipb;
    double s = 0.0;
    if (r < m && p.blk_w_offset[g] >= 0) {
        const double *P = p.blk_w + p.blk_w_offset[g];
        for (int k = 0; k < m; k++) s += P[(long)k * m + r] * v[k];
        s *= v[r] * sigma2;
    } else if (r < m) {        // block-diagonal weights (DevProblem::ip_w3): row r of the 2 x 2 block of its image point
        const double *w = p.ip_w3 + 3 * (long)(ipb + (r >> 1));
        const double pr0 = (r & 1) ? w[1] : w[0], pr1 = (r & 1) ? w[2] : w[1];
        s = (pr0 * v[r & ~1] + pr1 * v[r | 1]) * v[r] * sigma2;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int t = 128; t > 0; t >>= 1) {
        if (threadIdx.x < t) red[threadIdx.x] += red[threadIdx.x + t];
        __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(omega, red[0]);
}

// ---------------------------------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_assemble_small(hipStream_t s, const DevProblem &p, const int32_t *seg_begin, const int32_t *seg_end,
                                 int n_seg, const double *rowsA, const double *rowsW, double sigma2, double *N,
                                 double *n) {
    if (n_seg <= 0) return hipSuccess;
    hipLaunchKernelGGL(assemble_small_kernel, dim3(n_seg), dim3(256), 0, s, p, seg_begin, seg_end, rowsA, rowsW, sigma2,
                       N, n);
    return hipGetLastError();
}

hipError_t launch_schur_eliminate(hipStream_t, const DevProblem &, const int32_t *, int, int, const int32_t *, int,
                                  const double *, const double *, double *, double, double, double *, double *, double *,
                                  double *, int *, double *, double *, const PPGather *, double *);

hipError_t launch_schur_tfix(hipStream_t, const DevProblem &, const int32_t *, int, const double *, const double *, double, double *);

// JAICOV_ASSEMBLY_FORM = t_vector | no_fork | materialise: the alternative forms of the dense-block assembly kept as second paths
// for parity (the vector form of T = Dinv [A_c | w]; the camera-side kernels in front of the gather instead of beside it; P' written
// out instead of formed in the gather's registers), each named by tests/test_gpu_parity.py::test_assembly_forms_give_the_same_system.
int assembly_form() {
    const char *e = getenv("JAICOV_ASSEMBLY_FORM");
    if (!e) return ASSEMBLY_DEFAULT;
    if (!strcmp(e, "t_vector")) return ASSEMBLY_T_VECTOR;
    if (!strcmp(e, "no_fork")) return ASSEMBLY_NO_FORK;
    if (!strcmp(e, "materialise")) return ASSEMBLY_MATERIALISE;
    return ASSEMBLY_DEFAULT;
}

hipError_t launch_assemble_blocks(hipStream_t s, const DevProblem &p, const int32_t *blk_list, int n_list, int max_m,
                                  const int32_t *ip_list, int n_ip_list, const double *rowsA, const double *rowsW,
                                  double *T, double sigma2, double *N, double *n, const PPGather &pp, const SchurBufs &sb,
                                  double *cc_partial, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join) {
    if (n_list <= 0) return hipSuccess;
    const int schur = sb.active ? 1 : 0;
    // the fused gather (P' formed in registers) reads U_q from the coalesced copy Ug only: there is no row-major fallback
    if (schur && !sb.materialise && pp.pt_ip_begin && !pp.ug) return hipErrorInvalidValue;
    const int form = assembly_form();
    const bool t_vector = form == ASSEMBLY_T_VECTOR;
    if (t_vector) hipLaunchKernelGGL(blk_T_kernel, dim3((max_m + T_NT * T_TR - 1) / (T_NT * T_TR), n_list), dim3(T_NT), 0, s, p, blk_list, rowsA, rowsW, T);
    else hipLaunchKernelGGL(blk_T_mfma_kernel, dim3((max_m + 255) / 256, n_list), dim3(256), 0, s, p, blk_list, rowsA, rowsW, T);
    if (p.ip_w3)      // images with block-diagonal weights (both T kernels leave them alone)
        hipLaunchKernelGGL(blk_T_diag_kernel, dim3((max_m / 2 + 7) / 8, n_list), dim3(32, 8), 0, s, p, blk_list, rowsA, rowsW, T);
    DevProblem q = p;
    double s2 = sigma2;
    if (schur) {   // EO pre-elimination: weights become P' = sigma2 Dinv - U U', T becomes P' [A_r | w]
        hipError_t he = launch_schur_eliminate(s, p, blk_list, n_list, max_m, ip_list, n_ip_list, rowsA, rowsW, T, sigma2,
                                               sb.lambda, sb.U, sb.Linv, sb.G, sb.materialise ? sb.Pp : nullptr, sb.info, sb.diagcorr, sb.xq,
                                               pp.det && pp.pt_ip_begin ? &pp : nullptr, sb.Ug);
        if (he != hipSuccess) return he;
        if (sb.materialise) q.blk_w = sb.Pp;
        s2 = 1.0;
    }
    // The camera-side blocks (camera x camera, point x camera, their parts of n) and the point x point gather write disjoint
    // parts of N (the gather owns the point rows from the first point column on and does not touch n) and read the same
    // finished T / U: with a side stream the three small kernels -- 0.4 ms of mostly latency -- run beside the gather, which
    // is latency-bound itself, instead of in front of it.  Their own order stays (the sums into n keep their order).
    constexpr int cc_parts = 8;
    const bool no_fork = form == ASSEMBLY_NO_FORK;
    const bool fork = side && ev_fork && ev_join && pp.pt_ip_begin && !no_fork;
    hipStream_t cs = fork ? side : s;
    // T' = sigma2 T - U (U' A_c) is needed by the camera-side kernels only, so it could run on their side stream beside the gather.
    // Measured at config 4 (round 3): assembly 2.84 ms that way against 2.80 ms with T' in front of the fork -- the gather loses
    // more to the company than the critical path gains.
    if (schur) {
        hipError_t he = launch_schur_tfix(s, p, ip_list, n_ip_list, sb.U, sb.G, sigma2, T);
        if (he != hipSuccess) return he;
    }
    if (fork) {
        hipError_t he = hipEventRecord(ev_fork, s);
        if (he == hipSuccess) he = hipStreamWaitEvent(side, ev_fork, 0);
        if (he != hipSuccess) return he;
    }
    hipLaunchKernelGGL(blk_cc_kernel, dim3(n_list, cc_parts), dim3(256), 0, cs, p, blk_list, rowsA, T, s2, cc_partial, schur);
    hipLaunchKernelGGL(blk_cc_reduce_kernel, dim3(CC_ENT, p.n_cameras), dim3(256), 0, cs, p, blk_list, n_list, cc_parts, cc_partial, N, n, schur);
    const long tot = (long)n_ip_list * KC_LD;
    if (pp.pt_ip_begin) {
        const long totp = (long)p.n_points * KC_LD;
        hipLaunchKernelGGL(blk_pc_gather_kernel, dim3((unsigned)((totp + 255) / 256)), dim3(256), 0, cs, p, pp, rowsA, T, s2, N, n, schur);
    } else
        hipLaunchKernelGGL(blk_pc_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, cs, p, (const int32_t *)nullptr,
                           ip_list, n_ip_list, rowsA, T, s2, N, n, schur);
    if (fork) {
        hipError_t he = hipEventRecord(ev_join, side);
        if (he != hipSuccess) return he;
    }
    if (pp.pt_ip_begin) {
        const dim3 gg = pp.xcd_map ? dim3((unsigned)(8 * p.n_points * ((pp.n_chunks + 7) / 8))) : dim3(p.n_points, pp.n_chunks);
        const dim3 gb(PP_NT);
        const size_t lds = (size_t)3 * pp.cw * sizeof(double) + sizeof(double);      // + the DET form's turn word
        const bool det = pp.det != 0;
        if (schur && !sb.materialise) {   // the downdate P' = sigma2 Dinv - U U' on the fly: weights = Dinv, factor sigma2 inside
            const bool mixed = p.ip_w3 != nullptr;       // (a separate instance: the dense-only kernel keeps its code and its registers)
            if (det && mixed) hipLaunchKernelGGL((blk_pp_gather_kernel<true, true, true>), gg, gb, lds, s, p, pp, rowsA, sb.U, sigma2, N);
            else if (mixed) hipLaunchKernelGGL((blk_pp_gather_kernel<true, false, true>), gg, gb, lds, s, p, pp, rowsA, sb.U, sigma2, N);
            else if (det) hipLaunchKernelGGL((blk_pp_gather_kernel<true, true, false, true>), gg, gb, lds, s, p, pp, rowsA, sb.U, sigma2, N);      // pass-major (round 5)
            else hipLaunchKernelGGL((blk_pp_gather_kernel<true, false>), gg, gb, lds, s, p, pp, rowsA, sb.U, sigma2, N);
        } else {
            const bool mixed = q.ip_w3 != nullptr;
            if (det && mixed) hipLaunchKernelGGL((blk_pp_gather_kernel<false, true, true>), gg, gb, lds, s, q, pp, rowsA, (const double *)nullptr, s2, N);
            else if (mixed) hipLaunchKernelGGL((blk_pp_gather_kernel<false, false, true>), gg, gb, lds, s, q, pp, rowsA, (const double *)nullptr, s2, N);
            else if (det) hipLaunchKernelGGL((blk_pp_gather_kernel<false, true>), gg, gb, lds, s, q, pp, rowsA, (const double *)nullptr, s2, N);
            else hipLaunchKernelGGL((blk_pp_gather_kernel<false, false>), gg, gb, lds, s, q, pp, rowsA, (const double *)nullptr, s2, N);
        }
    } else {
        if (schur && !sb.materialise) return hipErrorInvalidValue;   // the per-pair atomic kernel reads a materialised P'
        const int mp = max_m / 2;
        hipLaunchKernelGGL(blk_pp_kernel, dim3((mp + 63) / 64, mp, n_list), dim3(192), 0, s, q, blk_list, rowsA, s2, N);
    }
    if (fork) {
        hipError_t he = hipStreamWaitEvent(s, ev_join, 0);
        if (he != hipSuccess) return he;
    }
    return hipGetLastError();
}

hipError_t launch_shared_groups(hipStream_t s, const DevProblem &p, const double *vals, double sigma2, double *N,
                                double *n, const double *dx, double *omega) {
    if (p.n_sb > 0)
        hipLaunchKernelGGL(scalebar_kernel, dim3((p.n_sb + 63) / 64), dim3(64), 0, s, p, vals, sigma2, N, n, dx, omega);
    if (p.n_dg > 0) hipLaunchKernelGGL(direct_kernel, dim3(p.n_dg), dim3(256), 0, s, p, vals, sigma2, N, n, dx, omega);
    return hipGetLastError();
}

hipError_t launch_omega(hipStream_t s, const DevProblem &p, const uint8_t *in_block, int ip0, int count,
                        const int32_t *blk_list, int n_list, int max_m, const double *rowsA, const double *rowsW,
                        const double *dx, double sigma2, double *vbuf, double *omega) {
    if (count > 0)
        hipLaunchKernelGGL(omega_points_kernel, dim3((count + 255) / 256), dim3(256), 0, s, p, in_block, ip0, count,
                           rowsA, rowsW, dx, sigma2, vbuf, omega);
    if (n_list > 0)
        hipLaunchKernelGGL(omega_block_kernel, dim3((max_m + 255) / 256, n_list), dim3(256), 0, s, p, blk_list, vbuf,
                           sigma2, omega);
    return hipGetLastError();
}

}  // namespace jaicov
