// Iterative refinement of the step: the residual of the UNSCALED bordered system  [[0, B], [B', N]] [kappa; dx] = [0; n]
// in twice the working precision.  gfx950 only.
//
// Why ("Accuracy at config 4"): cond(V N V) ~ 1e9 at the headline size; the Cholesky step is accurate to 2.6e-8
// there, the reference's packed Bunch-Kaufman (MathExtension.java:338-353, dspsv) to 3.6e-9.  Neither the explicit
// inverses of the diagonal tiles nor the block-recursive triangular inverse are what loses the digits
// (scripts/stability_probe.py: LAPACK's dpotrf/dpotrs lose the same) -- it is the method's constant -- so the cure is the
// classical one: r = n - N dx accumulated as unevaluated sums hi + lo (two-fold precision: products by fma, sums by
// TwoSum; Ogita/Rump/Oishi's Dot2), one forward + one backward substitution with the factor at hand (dense.hip),
// dx += correction.  One step takes the error from cond * eps to (cond * eps)^2.
//
// N is the engine's row-major lower triangle (leading dimension ld); every 128 x 128 tile is read ONCE and serves both
// the product with x (its rows) and the product of its transpose (its columns).  Partial sums go to a [block row][slot]
// table and are added in slot order by a second kernel: the residual has the same bits in every run.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace jaicov {

typedef double d4r_t __attribute__((ext_vector_type(4)));

struct RefineBorder {       // datum border of the system, d <= 7 (by value: 7 * 2 doubles)
    double kappa[7];        // unscaled multipliers (dx[0..d))
    double rk[7];           // scaled border residual  r_k = -R (B dx)
};

// (hi, lo) += a * x, error-free product, TwoSum; lo collects the rounding errors
__device__ __forceinline__ void dd_fma(double a, double x, double &hi, double &lo) {
#pragma clang fp contract(off)
    const double p = a * x;
    const double e = __builtin_fma(a, x, -p);
    const double s = hi + p;
    const double bb = s - hi;
    const double err = (hi - (s - bb)) + (p - bb);
    hi = s;
    lo += err + e;
}
// (hi, lo) += (h2, l2)
__device__ __forceinline__ void dd_add(double h2, double l2, double &hi, double &lo) {
#pragma clang fp contract(off)
    const double s = hi + h2;
    const double bb = s - hi;
    const double err = (hi - (s - bb)) + (h2 - bb);
    hi = s;
    lo += err + l2;
}
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// One workgroup per lower tile (I, J), J <= I.  Thread (ty, tx): rows ty + 16 k, columns 8 tx .. 8 tx + 7.
// P[(I * nbk + slot) * 256 + 2 * i + {0, 1}] = (hi, lo) of the contribution of block column `slot` to row 128 I + i.
// Entries of N outside rows/columns [d, U) count as zero (border and padding).
__global__ __launch_bounds__(256) void symv_dd_tile_kernel(const double *__restrict__ N, long ld, int U, int d,
                                                           const double *__restrict__ x, int nbk, double *__restrict__ P) {
    __shared__ double rowres[128][2];
    __shared__ double colres[4][128][2];
    const int t = blockIdx.x;
    int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(I + 1) * (I + 2) / 2 <= t) ++I;
    while ((long)I * (I + 1) / 2 > t) --I;
    const int J = t - I * (I + 1) / 2;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const bool diag = I == J;
    double xJ[8], xI[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int gc = J * 128 + 8 * tx + c;
        xJ[c] = (gc >= d && gc < U) ? x[gc] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int gr = I * 128 + ty + 16 * k;
        xI[k] = (gr >= d && gr < U) ? x[gr] : 0.0;
    }
    double rh[8], rl[8], ch[8], cl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) rh[k] = rl[k] = ch[k] = cl[k] = 0.0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        d4r_t a[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int k = 4 * half + kk;
            const int gr = I * 128 + ty + 16 * k;
            if (gr >= d && gr < U) {
                const double *p = N + (long)gr * ld + J * 128 + 8 * tx;
                a[kk][0] = *reinterpret_cast<const d4r_t *>(p);
                a[kk][1] = *reinterpret_cast<const d4r_t *>(p + 4);
            } else {
                a[kk][0] = a[kk][1] = (d4r_t){0.0, 0.0, 0.0, 0.0};
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int k = 4 * half + kk;
            const int r = ty + 16 * k;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int cc = 8 * tx + c;
                double v = c < 4 ? a[kk][0][c] : a[kk][1][c - 4];
                if (J * 128 + cc < d) v = 0.0;
                if (diag && cc > r) v = 0.0;                  // the strict upper part of a diagonal tile is not N
                dd_fma(v, xJ[c], rh[k], rl[k]);               // row part: (N x)_r
                if (!(diag && cc == r)) dd_fma(v, xI[k], ch[c], cl[c]);   // column part: (N' x)_c, diagonal counted once
            }
        }
    }
    // rows: sum over the 16 tx lanes (lane bits 0..3)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double h2 = shfl_xor_d(rh[k], m), l2 = shfl_xor_d(rl[k], m);
            dd_add(h2, l2, rh[k], rl[k]);
        }
    }
    // columns: sum over the 4 ty values of this wave (lane bits 4, 5), then over the 4 waves through LDS
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const double h2 = shfl_xor_d(ch[c], m), l2 = shfl_xor_d(cl[c], m);
            dd_add(h2, l2, ch[c], cl[c]);
        }
    }
    if (tx == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { rowres[ty + 16 * k][0] = rh[k]; rowres[ty + 16 * k][1] = rl[k]; }
    }
    if ((tid & 63) < 16) {
        const int w = tid >> 6;
#pragma unroll
        for (int c = 0; c < 8; c++) { colres[w][8 * tx + c][0] = ch[c]; colres[w][8 * tx + c][1] = cl[c]; }
    }
    __syncthreads();
    if (tid < 128) {
        double hi = colres[0][tid][0], lo = colres[0][tid][1];
#pragma unroll
        for (int w = 1; w < 4; w++) dd_add(colres[w][tid][0], colres[w][tid][1], hi, lo);
        if (diag) {
            dd_add(rowres[tid][0], rowres[tid][1], hi, lo);
            double *o = P + ((long)I * nbk + I) * 256 + 2 * tid;
            o[0] = hi; o[1] = lo;
        } else {
            double *o = P + ((long)I * nbk + J) * 256 + 2 * tid;
            o[0] = rowres[tid][0]; o[1] = rowres[tid][1];
            double *oc = P + ((long)J * nbk + I) * 256 + 2 * tid;
            oc[0] = hi; oc[1] = lo;
        }
    }
}

// rhs_i = V_i (n_i - (N x)_i - sum_a B_ai kappa_a) + sum_a Bh_ai rk_a  for d <= i < U, 0 elsewhere (i < order_pad).
// One workgroup per block row: 8 groups of 128 threads add every eighth slot each (fixed order), the groups are combined in
// group order -- the same bits in every run, and eight times the parallelism of one thread per row (50 -> 10 us at config 4).
__global__ __launch_bounds__(1024) void symv_dd_reduce_kernel(const double *__restrict__ P, int nbk, int U, int d,
                                                              const double *__restrict__ n, const double *__restrict__ V,
                                                              const double *__restrict__ Braw, const double *__restrict__ Bh,
                                                              long bstride, RefineBorder bd, double *__restrict__ rhs) {
    __shared__ double part[8][128][2];
    const int I = blockIdx.x, i = threadIdx.x & 127, grp = threadIdx.x >> 7, g = I * 128 + i;
    double hi = 0.0, lo = 0.0;
    const double *p = P + (long)I * nbk * 256 + 2 * i;
    for (int s = grp; s < nbk; s += 8) dd_add(p[(long)s * 256], p[(long)s * 256 + 1], hi, lo);
    part[grp][i][0] = hi; part[grp][i][1] = lo;
    __syncthreads();
    if (grp != 0) return;
    for (int q = 1; q < 8; q++) dd_add(part[q][i][0], part[q][i][1], hi, lo);
    double out = 0.0;
    if (g >= d && g < U) {
        double rh = n[g], rl = 0.0;
        dd_add(-hi, -lo, rh, rl);
        for (int a = 0; a < d; a++) dd_fma(-Braw[(long)a * bstride + g], bd.kappa[a], rh, rl);
        out = V[g] * (rh + rl);
        for (int a = 0; a < d; a++) out += Bh[(long)a * bstride + g] * bd.rk[a];
    }
    rhs[g] = out;
}

// rhs (order_pad entries) <- scaled residual of the bordered system at x (= dx, with kappa in its first d entries)
hipError_t launch_residual_dd(hipStream_t s, const double *N, long ld, int U, int d, int order_pad, const double *x,
                              const double *n, const double *V, const double *Braw, const double *Bh, long bstride,
                              const RefineBorder &bd, double *P, double *rhs) {
    const int nbk = order_pad / 128;
    hipLaunchKernelGGL(symv_dd_tile_kernel, dim3(nbk * (nbk + 1) / 2), dim3(256), 0, s, N, ld, U, d, x, nbk, P);
    hipLaunchKernelGGL(symv_dd_reduce_kernel, dim3(nbk), dim3(1024), 0, s, P, nbk, U, d, n, V, Braw, Bh, bstride, bd, rhs);
    return hipGetLastError();
}

}  // namespace jaicov
