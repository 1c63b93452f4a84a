// Per-image pre-elimination of the exterior-orientation unknowns (the idea of MatrixInversion.PRE_ELIMINATION,
// BundleAdjustment.java:1197-1453, re-derived for the device).
//
// The six EO parameters of an image couple only with that image's observations, so N_EE is block diagonal and the
// Schur complement  S = N_RR - N_RE N_EE^-1 N_ER  of the remaining unknowns R (points, interior orientation,
// distortion) is a sum of per-image terms.  For an image with weight matrix P (= sigma0^2 D^-1) and EO columns A_e:
//     A_r' P A_r - (A_r' P A_e) E^-1 (A_e' P A_r) = A_r' ( P - U U' ) A_r ,   E = A_e' P A_e = L_E L_E' ,  U = P A_e L_E^-T
// i.e. eliminating the EO block IS a rank-6 downdate of the image's weight matrix.  The reduced normal equations are
// therefore assembled by the very same kernels from P' = P - U U' with the EO columns dropped, the dense factorisation
// shrinks from u to u - 6*images (18 014 -> 15 014 at config 4: 42 % fewer flops), and
//     dx_E = L_E^-T U' (w - A_r dx_R)
// recovers the EO step afterwards.  Levenberg-Marquardt damping enters through E_kk (1 + lambda) (BA:814-822).
#include "ba_kernels.h"
#include "gemm_f64.h"

namespace jaicov {

__device__ __forceinline__ int schur_shared_local(int c) { return c < 3 ? 3 + c : (c < 9 ? 6 + (c - 3) : 12 + (c - 9)); }

// one workgroup per image block: E, n_E, L_E^-1, U, and G = U' [A_r,shared | w]
__global__ __launch_bounds__(256) void blk_elim_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                       const double *__restrict__ rowsA, const double *__restrict__ rowsW,
                                                       const double *__restrict__ T, double sigma2, double lambda,
                                                       double *__restrict__ Ubuf, double *__restrict__ Linv_out,
                                                       double *__restrict__ G_out, int *info, double *__restrict__ xq,
                                                       double *__restrict__ Ug) {
    __shared__ double red[42];        // E (36) + nE (6)
    __shared__ double redw[4][42];    // per-wave partial sums: summed in wave order (LDS atomics would sum in arrival order)
    __shared__ double Linv[36];
    __shared__ double Gs[6 * SCHUR_GLD];
    __shared__ double Gp[8][6 * SCHUR_GLD];   // per row-group partial sums of G, summed in group order
    const int tid = threadIdx.x;
    const int g = blk_list[blockIdx.x];
    const int ipb = p.blk_ip_begin[g], mp = p.blk_ip_begin[g + 1] - ipb, m = 2 * mp;
    const long S = p.n_ip;
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    const int kc = 9 + p.cam_dist_begin[cam + 1] - p.cam_dist_begin[cam];
    if (tid < 42) red[tid] = 0.0;
    __syncthreads();
    {
        double acc[42];
#pragma unroll
        for (int i = 0; i < 42; i++) acc[i] = 0.0;
        for (int row = tid; row < m; row += 256) {
            const int o = row >> 1, r = row & 1;
            const double *t = T + ((long)2 * ipb + row) * KC_LD;
            double ae[6];
#pragma unroll
            for (int k = 0; k < 6; k++) ae[k] = rowsA[(long)(2 * (6 + k) + r) * S + ipb + o];
#pragma unroll
            for (int k = 0; k < 6; k++) {
#pragma unroll
                for (int j = 0; j < 6; j++) acc[6 * k + j] += ae[k] * t[3 + j];
                acc[36 + k] += ae[k] * t[kc];
            }
        }
#pragma unroll
        for (int i = 0; i < 42; i++) {
            // wave reduction, then one slot per wave
            double v = acc[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if ((tid & 63) == 0) redw[tid >> 6][i] = v;
        }
    }
    __syncthreads();
    if (tid < 42) red[tid] = (redw[0][tid] + redw[1][tid]) + (redw[2][tid] + redw[3][tid]);
    __syncthreads();
    if (tid == 0) {
        double E[6][6], Lm[6][6], Li[6][6];
        for (int k = 0; k < 6; k++)
            for (int j = 0; j < 6; j++) E[k][j] = sigma2 * 0.5 * (red[6 * k + j] + red[6 * j + k]);
        if (lambda > 0.0)
            for (int k = 0; k < 6; k++) E[k][k] += lambda * E[k][k];
        if (xq)   // V_c^2 n_c with V_c = 1/sqrt(N_cc) where N_cc > EPS, else 1 (BA:825-828, NES:82-91 twice; quirk Q1)
            for (int k = 0; k < 6; k++) {
                const double nE = sigma2 * red[36 + k];
                xq[(long)img * 6 + k] = E[k][k] > 1.1102230246251565e-16 ? nE / E[k][k] : nE;
            }
        bool ok = true;
        for (int j = 0; j < 6; j++) {
            double d = E[j][j];
            for (int k = 0; k < j; k++) d -= Lm[j][k] * Lm[j][k];
            if (!(d > 0.0)) { ok = false; d = 1.0; }
            Lm[j][j] = sqrt(d);
            for (int i = j + 1; i < 6; i++) {
                double s = E[i][j];
                for (int k = 0; k < j; k++) s -= Lm[i][k] * Lm[j][k];
                Lm[i][j] = s / Lm[j][j];
            }
            for (int i = 0; i < j; i++) Lm[i][j] = 0.0;
        }
        if (!ok) atomicCAS(info, 0, 1000000 + img);
        for (int c = 0; c < 6; c++)
            for (int i = 0; i < 6; i++) {
                double s = (i == c) ? 1.0 : 0.0;
                for (int k = 0; k < i; k++) s -= Lm[i][k] * Li[k][c];
                Li[i][c] = s / Lm[i][i];
            }
        for (int i = 0; i < 6; i++)
            for (int c = 0; c < 6; c++) {
                Linv[6 * i + c] = Li[i][c];
                Linv_out[(long)img * 36 + 6 * i + c] = Li[i][c];
            }
    }
    __syncthreads();
    // U[row][k] = sigma2 * sum_j T_e[row][j] Linv[k][j]
    for (int row = tid; row < m; row += 256) {
        const double *t = T + ((long)2 * ipb + row) * KC_LD;
        double *uo = Ubuf + ((long)2 * ipb + row) * 8;
        // second copy for the point x point gather, where lane q of a wave reads the U rows of partner q: slot j of image point
        // ip is the pair Ug[(j * n_ip + ip) * 2 + {0, 1}], j < 3: (U[2 ip][2 j], U[2 ip][2 j + 1]), j >= 3: the same of row 2 ip + 1
        // -- consecutive lanes then read consecutive 16-byte pieces (8 cache lines per wave instruction instead of 64)
        double *ug = Ug ? Ug + (((long)(3 * (row & 1)) * S + ipb + (row >> 1)) * 2) : nullptr;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j <= k; j++) s += t[3 + j] * Linv[6 * k + j];
            uo[k] = sigma2 * s;
            if (ug) ug[(long)(k >> 1) * S * 2 + (k & 1)] = sigma2 * s;
        }
    }
    __threadfence_block();
    __syncthreads();
    // G[k][c] = sum_rows U[row][k] * Ar[row][c]  (c: io 0..2, dist 3.., w last).  Thread = (column c, one of 8 row
    // groups): six private sums, then one LDS add per thread (two lanes of a wave share an address, not sixty-four).
    const int ncr = kc - 6;      // reduced shared columns (io + dist)
    {
        const int c = tid & 31, rs = tid >> 5;
        if (c <= ncr) {
            const int cs = c < 3 ? c : c + 6;     // shared column index of reduced column c
            const double *acol = c < ncr ? rowsA + (long)(2 * schur_shared_local(cs)) * S + ipb : rowsW + ipb;
            double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int row = rs; row < m; row += 8) {
                const double a = acol[(long)(row & 1) * S + (row >> 1)];
                const double *u = Ubuf + ((long)2 * ipb + row) * 8;
#pragma unroll
                for (int k = 0; k < 6; k++) acc[k] += u[k] * a;
            }
#pragma unroll
            for (int k = 0; k < 6; k++) Gp[rs][k * SCHUR_GLD + c] = acc[k];
        } else if (c < SCHUR_GLD) {
#pragma unroll
            for (int k = 0; k < 6; k++) Gp[rs][k * SCHUR_GLD + c] = 0.0;
        }
    }
    __syncthreads();
    for (int i = tid; i < 6 * SCHUR_GLD; i += 256) {
        double v = 0.0;
#pragma unroll
        for (int g8 = 0; g8 < 8; g8++) v += Gp[g8][i];
        Gs[i] = v;
        G_out[(long)img * 6 * SCHUR_GLD + i] = v;
    }
}

// P' = sigma2 * Dinv - U U'   (row-major m x m; only the lower triangle including the 2 x 2 diagonal blocks is written:
// the point x point assembly, its only reader, pairs a point with the partners stored before it).
// grid (ceil(m/64), ceil(m/64), n_list), block (64,4): a 64 x 64 tile, thread = one column x 16 rows; the U rows of the
// tile sit in LDS (the same for all lanes of a wave).
__global__ __launch_bounds__(256) void blk_pprime_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                         const double *__restrict__ Ubuf, double sigma2,
                                                         double *__restrict__ Pp) {
    __shared__ __attribute__((aligned(16))) double Us[64 * 6];
    const int g = blk_list[blockIdx.z];
    const int ipb = p.blk_ip_begin[g], m = 2 * (p.blk_ip_begin[g + 1] - ipb);
    if ((int)blockIdx.x * 64 >= m || (int)blockIdx.y * 64 >= m || blockIdx.x > blockIdx.y) return;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int r0 = blockIdx.y * 64;
    for (int i = tid; i < 64 * 6; i += 256) {
        const int rr = i / 6, k = i - 6 * rr;
        Us[i] = r0 + rr < m ? Ubuf[((long)2 * ipb + r0 + rr) * 8 + k] : 0.0;
    }
    __syncthreads();
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= m) return;
    const double *P = p.blk_w + p.blk_w_offset[g];
    double *Po = Pp + p.blk_w_offset[g];
    double uc[6];
    {
        const double *u = Ubuf + ((long)2 * ipb + c) * 8;
#pragma unroll
        for (int k = 0; k < 6; k++) uc[k] = u[k];
    }
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = r0 + threadIdx.y * 16 + i;
        v[i] = (r < m && c <= (r | 1)) ? P[(long)r * m + c] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int rl = threadIdx.y * 16 + i, r = r0 + rl;
        if (r >= m || c > (r | 1)) continue;
        const d2_t u01 = *reinterpret_cast<const d2_t *>(&Us[rl * 6]), u23 = *reinterpret_cast<const d2_t *>(&Us[rl * 6 + 2]),
                   u45 = *reinterpret_cast<const d2_t *>(&Us[rl * 6 + 4]);
        Po[(long)r * m + c] = sigma2 * v[i] - (u01.x * uc[0] + u01.y * uc[1] + u23.x * uc[2] + u23.y * uc[3] + u45.x * uc[4] + u45.y * uc[5]);
    }
}

// T'[row][c] = sigma2 T[row][c] - sum_k U[row][k] G[k][c'] for the reduced shared columns and w (in place; the EO
// columns of T are left alone and ignored downstream)
__global__ __launch_bounds__(256) void blk_tfix_kernel(DevProblem p, const int32_t *__restrict__ ip_list, int n_list,
                                                       const double *__restrict__ Ubuf, const double *__restrict__ G,
                                                       double sigma2, double *__restrict__ T) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int li = (int)(gid >> 1), r = (int)(gid & 1);
    if (li >= n_list) return;
    const int ip = ip_list[li];
    const int img = p.ip_image[ip], cam = p.image_camera[img];
    const int kc = 9 + p.cam_dist_begin[cam + 1] - p.cam_dist_begin[cam];
    const int ncr = kc - 6;
    const double *u = Ubuf + ((long)2 * ip + r) * 8;
    const double *Gi = G + (long)img * 6 * SCHUR_GLD;
    double *t = T + ((long)2 * ip + r) * KC_LD;
    for (int c = 0; c <= ncr; c++) {
        const int cs = c < 3 ? c : (c < ncr ? c + 6 : kc);
        double s = sigma2 * t[cs];
#pragma unroll
        for (int k = 0; k < 6; k++) s -= u[k] * Gi[k * SCHUR_GLD + c];
        t[cs] = s;
    }
}

// dx_E = L_E^-T U' v per image block, v = w - A_r dx_R (vbuf).  One workgroup per block.
__global__ __launch_bounds__(256) void blk_backsub_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                          const double *__restrict__ Ubuf, const double *__restrict__ Linv,
                                                          const double *__restrict__ vbuf, double *__restrict__ xE) {
    __shared__ double t[6];
    __shared__ double tw[4][6];       // per-wave partial sums, added in wave order (an LDS atomic would add in arrival order)
    const int tid = threadIdx.x;
    const int g = blk_list[blockIdx.x];
    const int ipb = p.blk_ip_begin[g], m = 2 * (p.blk_ip_begin[g + 1] - ipb);
    const int img = p.ip_image[ipb];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int row = tid; row < m; row += 256) {
        const double v = vbuf[(long)2 * ipb + row];
        const double *u = Ubuf + ((long)2 * ipb + row) * 8;
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] += u[k] * v;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((tid & 63) == 0) tw[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 6) t[tid] = (tw[0][tid] + tw[1][tid]) + (tw[2][tid] + tw[3][tid]);
    __syncthreads();
    if (tid < 6) {
        const double *Li = Linv + (long)img * 36;
        double s = 0.0;
        for (int j = tid; j < 6; j++) s += Li[6 * j + tid] * t[j];      // (L^-T t)[k] = sum_j Linv[j][k] t[j]
        xE[(long)img * 6 + tid] = s;
    }
}

// diag(N_RR) - diag(S) = diag(Y'Y): what the elimination removed from the diagonal.  The LM damping of the reference
// scales the UNREDUCED diagonal (BA:814-822), so finalize needs it.  thread per (block image point, coordinate b) plus
// one thread per (image, reduced shared column)
__global__ __launch_bounds__(256) void blk_diagcorr_kernel(DevProblem p, const int32_t *__restrict__ ip_list, int n_list,
                                                           const int32_t *__restrict__ blk_list, int n_blk,
                                                           const double *__restrict__ rowsA, const double *__restrict__ Ubuf,
                                                           const double *__restrict__ G, double *__restrict__ diagcorr) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long S = p.n_ip;
    if (gid < (long)3 * n_list) {
        const int li = (int)(gid / 3), b = (int)(gid - 3L * li);
        const int ip = ip_list[li];
        const int col = p.point_col[3 * p.ip_point[ip] + b];
        if (col < 0) return;
        const double a0 = rowsA[(long)(2 * b) * S + ip], a1 = rowsA[(long)(2 * b + 1) * S + ip];
        const double *u0 = Ubuf + ((long)2 * ip) * 8, *u1 = u0 + 8;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double y = u0[k] * a0 + u1[k] * a1;
            s += y * y;
        }
        unsafeAtomicAdd(diagcorr + col, s);
        return;
    }
    const long h = gid - (long)3 * n_list;
    const int bi = (int)(h / SCHUR_GLD), c = (int)(h - (long)bi * SCHUR_GLD);
    if (bi >= n_blk) return;
    const int ipb = p.blk_ip_begin[blk_list[bi]];
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    const int jb = p.cam_dist_begin[cam], ncr = 3 + p.cam_dist_begin[cam + 1] - jb;
    if (c >= ncr) return;
    const int col = c < 3 ? p.io_col[3 * cam + c] : p.dist_col[jb + c - 3];
    if (col < 0) return;
    const double *Gi = G + (long)img * 6 * SCHUR_GLD;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) s += Gi[k * SCHUR_GLD + c] * Gi[k * SCHUR_GLD + c];
    unsafeAtomicAdd(diagcorr + col, s);
}

// the same without atomics (engine option `deterministic`): one thread owns one column.  Point columns: thread (point, b) walks
// the point's incidences in the order of the gather's CSR; camera columns: thread (camera, c) walks the camera's blocks in list
// order.  grid covers 3 * n_points + n_cameras * SCHUR_GLD threads.
__global__ __launch_bounds__(256) void blk_diagcorr_det_kernel(DevProblem p, PPGather pp, const int32_t *__restrict__ blk_list, int n_blk,
                                                               const double *__restrict__ rowsA, const double *__restrict__ Ubuf,
                                                               const double *__restrict__ G, double *__restrict__ diagcorr) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long S = p.n_ip;
    if (gid < (long)3 * p.n_points) {
        const int pt = (int)(gid / 3), b = (int)(gid - 3L * pt);
        const int col = p.point_col[3 * pt + b];
        if (col < 0) return;
        double s = 0.0;
        for (int o = pp.pt_ip_begin[pt]; o < pp.pt_ip_begin[pt + 1]; o++) {
            const int ip = pp.recs[o].ipb + pp.recs[o].lp;
            const double a0 = rowsA[(long)(2 * b) * S + ip], a1 = rowsA[(long)(2 * b + 1) * S + ip];
            const double *u0 = Ubuf + ((long)2 * ip) * 8, *u1 = u0 + 8;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const double y = u0[k] * a0 + u1[k] * a1;
                s += y * y;
            }
        }
        if (pp.pt_ip_begin[pt + 1] > pp.pt_ip_begin[pt]) diagcorr[col] += s;
        return;
    }
    const long h = gid - (long)3 * p.n_points;
    const int cam = (int)(h / SCHUR_GLD), c = (int)(h - (long)cam * SCHUR_GLD);
    if (cam >= p.n_cameras) return;
    const int jb = p.cam_dist_begin[cam], ncr = 3 + p.cam_dist_begin[cam + 1] - jb;
    if (c >= ncr) return;
    const int col = c < 3 ? p.io_col[3 * cam + c] : p.dist_col[jb + c - 3];
    if (col < 0) return;
    double s = 0.0;
    for (int bi = 0; bi < n_blk; bi++) {
        const int img = p.ip_image[p.blk_ip_begin[blk_list[bi]]];
        if (p.image_camera[img] != cam) continue;
        const double *Gi = G + (long)img * 6 * SCHUR_GLD;
#pragma unroll
        for (int k = 0; k < 6; k++) s += Gi[k * SCHUR_GLD + c] * Gi[k * SCHUR_GLD + c];
    }
    diagcorr[col] += s;
}

// F = N_EE^-1 N_ER, one 6-row band per image, dense [6 * images (padded)][ldf]:  F_img = L_E^-T (U' A_r)  (U' = L_E^-1 A_e' P).
// With Q_RR = the inverse of the reduced system the rest of the full cofactor matrix follows from it (engine.hip, solve with
// JAICOV_INVERT_FULL_EXPANDED):  Q_ER = -F Q_RR,  Q_EE = N_EE^-1 - Q_ER F'.  One workgroup per image block; a point is seen
// once per image, so every entry of F has one writer.  F must be zero beforehand.
__global__ __launch_bounds__(256) void blk_expand_f_kernel(DevProblem p, const int32_t *__restrict__ blk_list,
                                                           const double *__restrict__ rowsA, const double *__restrict__ Ubuf,
                                                           const double *__restrict__ Linv, const double *__restrict__ G,
                                                           double *__restrict__ F, long ldf) {
    __shared__ double Li[36];
    const int tid = threadIdx.x;
    const int g = blk_list[blockIdx.x];
    const int ipb = p.blk_ip_begin[g], mp = p.blk_ip_begin[g + 1] - ipb;
    const long S = p.n_ip;
    const int img = p.ip_image[ipb], cam = p.image_camera[img];
    if (tid < 36) Li[tid] = Linv[(long)img * 36 + tid];
    __syncthreads();
    double *Fi = F + (long)(6 * img) * ldf;
    for (int q = tid; q < mp; q += 256) {
        const int ip = ipb + q, pt = p.ip_point[ip];
        const double *u0 = Ubuf + ((long)2 * ip) * 8, *u1 = u0 + 8;
        for (int b = 0; b < 3; b++) {
            const int col = p.point_col[3 * pt + b];
            if (col < 0) continue;
            const double a0 = rowsA[(long)(2 * b) * S + ip], a1 = rowsA[(long)(2 * b + 1) * S + ip];
            double y[6];
#pragma unroll
            for (int j = 0; j < 6; j++) y[j] = u0[j] * a0 + u1[j] * a1;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                double f = 0.0;
#pragma unroll
                for (int j = k; j < 6; j++) f += Li[6 * j + k] * y[j];      // (L^-T y)[k] = sum_j Linv[j][k] y[j]
                Fi[(long)k * ldf + col] = f;
            }
        }
    }
    const int jb = p.cam_dist_begin[cam], ncr = 3 + p.cam_dist_begin[cam + 1] - jb;
    for (int c = tid; c < ncr; c += 256) {
        const int col = c < 3 ? p.io_col[3 * cam + c] : p.dist_col[jb + c - 3];
        if (col < 0) continue;
        const double *Gi = G + (long)img * 6 * SCHUR_GLD;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double f = 0.0;
#pragma unroll
            for (int j = k; j < 6; j++) f += Li[6 * j + k] * Gi[j * SCHUR_GLD + c];
            Fi[(long)k * ldf + col] = f;
        }
    }
}

hipError_t launch_schur_expand_f(hipStream_t s, const DevProblem &p, const int32_t *blk_list, int n_list, const double *rowsA,
                                 const double *Ubuf, const double *Linv, const double *G, double *F, long ldf) {
    if (n_list <= 0) return hipSuccess;
    hipLaunchKernelGGL(blk_expand_f_kernel, dim3(n_list), dim3(256), 0, s, p, blk_list, rowsA, Ubuf, Linv, G, F, ldf);
    return hipGetLastError();
}

hipError_t launch_schur_eliminate(hipStream_t s, const DevProblem &p, const int32_t *blk_list, int n_list, int max_m,
                                  const int32_t *ip_list, int n_ip_list, const double *rowsA, const double *rowsW,
                                  double *T, double sigma2, double lambda, double *Ubuf, double *Linv, double *G,
                                  double *Pp, int *info, double *diagcorr, double *xq, const PPGather *det_pp, double *Ug) {
    if (n_list <= 0) return hipSuccess;
    hipLaunchKernelGGL(blk_elim_kernel, dim3(n_list), dim3(256), 0, s, p, blk_list, rowsA, rowsW, T, sigma2, lambda, Ubuf,
                       Linv, G, info, xq, Ug);
    if (Pp)   // only when P' is wanted in memory (JAICOV_PP_MATERIALISE / the atomic point x point kernel); the gather forms it on the fly
        hipLaunchKernelGGL(blk_pprime_kernel, dim3((max_m + 63) / 64, (max_m + 63) / 64, n_list), dim3(64, 4), 0, s, p, blk_list,
                           Ubuf, sigma2, Pp);
    if (lambda > 0.0 && diagcorr && det_pp) {
        const long nt = (long)3 * p.n_points + (long)p.n_cameras * SCHUR_GLD;
        hipLaunchKernelGGL(blk_diagcorr_det_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, p, *det_pp, blk_list, n_list,
                           rowsA, Ubuf, G, diagcorr);
    } else if (lambda > 0.0 && diagcorr) {
        const long nt = (long)3 * n_ip_list + (long)n_list * SCHUR_GLD;
        hipLaunchKernelGGL(blk_diagcorr_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, p, ip_list, n_ip_list,
                           blk_list, n_list, rowsA, Ubuf, G, diagcorr);
    }
    return hipGetLastError();
}

// T' = sigma2 T - U (U' A_c): needed by the camera-side kernels only (the point x point gather forms P' from D^-1 and U); where it is
// launched -- in front of the fork or on the side stream -- is the caller's choice (assemble.hip, launch_assemble_blocks)
hipError_t launch_schur_tfix(hipStream_t s, const DevProblem &p, const int32_t *ip_list, int n_ip_list, const double *Ubuf,
                             const double *G, double sigma2, double *T) {
    const long tot = (long)2 * n_ip_list;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(blk_tfix_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p, ip_list, n_ip_list, Ubuf, G,
                       sigma2, T);
    return hipGetLastError();
}

hipError_t launch_schur_backsub(hipStream_t s, const DevProblem &p, const int32_t *blk_list, int n_list, const double *Ubuf,
                                const double *Linv, const double *vbuf, double *xE) {
    if (n_list <= 0) return hipSuccess;
    hipLaunchKernelGGL(blk_backsub_kernel, dim3(n_list), dim3(256), 0, s, p, blk_list, Ubuf, Linv, vbuf, xE);
    return hipGetLastError();
}

}  // namespace jaicov
