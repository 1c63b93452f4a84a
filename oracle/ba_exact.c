/*
 * ba_exact.c -- GROUND TRUTH for the accuracy study.  TEST INFRASTRUCTURE ONLY, and NOT the reference's arithmetic.
 *
 * The reference (and ba_oracle.c, its restatement) forms the weight of a jointly dispersed group as dpptrf + dpptri of
 * D / sigma0^2 in fp64 (DirectlyObservedParameterGroup.java:82-86, MathExtension.java:304-324) and stacks A'PA in fp64
 * (PartialDerivativeFactory.java:475-505).  At BASELINE config 4 the per-image dispersions have cond(D) ~ 1e7, so ANY fp64
 * inverse of D carries a forward error of ~1e-11 .. 1e-9, and cond(V N V) ~ 1e9 turns that into ~1e-7 on Qxx: two correct
 * fp64 implementations differ from each other by that much.  To say which of two implementations is closer to the
 * mathematical answer this file assembles the SAME normal equations with
 *     P = sigma0^2 * inv(D)     Cholesky + triangular inverse in x87 extended precision (64-bit mantissa) + one Newton-Schulz
 *                               step whose residual is accumulated in twice the working precision (newton_dd),
 *     N = sum A' P A, n = sum A' P w   products and sums in extended precision, rounded ONCE to a (hi, lo) pair of doubles,
 * from the oracle's own fp64 rows A, w (oracle_rows: their relative error, ~1e-16, is five orders below that of an fp64
 * inv(D) and is part of neither implementation's difference).  oracle_inverse_residual_q certifies an inverse with a
 * binary128 residual.  tests/golden/make_exactN.py turns this into the committed fixtures tests/golden/<cfg>/<cfg>_exactN.npz.
 *
 * Only tests/ (through oracle/oracle.py) uses it.  OpenMP is used over the image blocks; nothing here is timed.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/jaicov_neq.h"

typedef long double ld;

#define EX_MAXD 64
#define EX_KLOC (12 + EX_MAXD)

/* exported by ba_oracle.c */
int oracle_rows(const jaicov_problem_desc *d, const double *vals, double sigma2, int ip, double *w2, double *A_out,
                int kloc_out, double *P4);
int oracle_num_slots(const jaicov_problem_desc *d);
void oracle_slot_columns(const jaicov_problem_desc *d, int32_t *col);

static inline size_t pidx(int r, int c) { return (size_t)r + ((size_t)c + 1) * (size_t)c / 2; }

static int local_cols(const jaicov_problem_desc *d, int ip, int32_t gcol[EX_KLOC]) {
    int img = d->ip_image[ip], pt = d->ip_point[ip], cam = d->image_camera[img];
    int jb = d->cam_dist_begin[cam], je = d->cam_dist_begin[cam + 1];
    for (int a = 0; a < 3; a++) gcol[a] = d->point_col[3 * pt + a];
    for (int a = 0; a < 3; a++) gcol[3 + a] = d->io_col[3 * cam + a];
    for (int a = 0; a < 6; a++) gcol[6 + a] = d->eo_col[6 * img + a];
    for (int j = jb; j < je; j++) gcol[12 + j - jb] = d->dist_col[j];
    return 12 + je - jb;
}

static inline ld dot_ld(const ld *a, const ld *b, int n) {
    ld s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i + 4 <= n; i += 4) { s0 += a[i] * b[i]; s1 += a[i + 1] * b[i + 1]; s2 += a[i + 2] * b[i + 2]; s3 += a[i + 3] * b[i + 3]; }
    for (; i < n; i++) s0 += a[i] * b[i];
    return (s0 + s1) + (s2 + s3);
}

/* One Newton-Schulz step X <- X + X (I - D X) on an extended-precision inverse (forward error cond(D) 2^-64 ~ 5e-13 at config 4,
 * which is no better than what the DEVICE reaches with its own compensated step): the residual R = I - D X is accumulated with
 * error-free products (FMA) and TwoSum (Ogita-Rump-Oishi Dot2: as if in twice the working precision) for X = X_hi + X_lo, so that
 * R (~5e-13) is known to ~1e-16 of itself; the correction X R then only needs fp64.  Afterwards X is good to the last bits of
 * the extended format (certified by oracle_inverse_residual_q in binary128). */
static void newton_dd(int m, const double *D, ld *X) {
    size_t mm = (size_t)m * m;
    double *Xh = (double *)malloc(sizeof(double) * mm), *Xl = (double *)malloc(sizeof(double) * mm);
    double *R = (double *)malloc(sizeof(double) * mm), *Y = (double *)calloc(mm, sizeof(double));
    double *s = (double *)malloc(sizeof(double) * (size_t)m), *c = (double *)malloc(sizeof(double) * (size_t)m);
    for (size_t i = 0; i < mm; i++) { Xh[i] = (double)X[i]; Xl[i] = (double)(X[i] - (ld)Xh[i]); }
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < m; j++) { s[j] = (i == j) ? 1.0 : 0.0; c[j] = 0.0; }
        for (int k = 0; k < m; k++) {
            const double d = -D[(size_t)i * m + k];
            const double *xh = Xh + (size_t)k * m, *xl = Xl + (size_t)k * m;
            for (int j = 0; j < m; j++) {
                double p = d * xh[j];
                double e = fma(d, xh[j], -p);                 /* p + e == d * xh exactly */
                double t = s[j] + p;
                double z = t - s[j];
                double q = (s[j] - (t - z)) + (p - z);         /* t + q == s + p exactly */
                s[j] = t;
                c[j] += (e + q) + d * xl[j];
            }
        }
        for (int j = 0; j < m; j++) R[(size_t)i * m + j] = s[j] + c[j];
    }
    for (int i = 0; i < m; i++)                                 /* Y = X_hi R (fp64: R is ~1e-12 of the identity) */
        for (int k = 0; k < m; k++) {
            const double x = Xh[(size_t)i * m + k];
            const double *r = R + (size_t)k * m;
            double *y = Y + (size_t)i * m;
            for (int j = 0; j < m; j++) y[j] += x * r[j];
        }
    for (int i = 0; i < m; i++)
        for (int j = 0; j <= i; j++) {
            ld v = 0.5L * ((X[(size_t)i * m + j] + (ld)Y[(size_t)i * m + j]) + (X[(size_t)j * m + i] + (ld)Y[(size_t)j * m + i]));
            X[(size_t)i * m + j] = v;
            X[(size_t)j * m + i] = v;
        }
    free(Xh); free(Xl); free(R); free(Y); free(s); free(c);
}

/* X = inv(D), D symmetric positive definite m x m (row-major, fp64 entries taken as exact), everything in extended precision.
 * X full row-major m x m.  Returns 0 or the 1-based index of a non-positive pivot. */
static int spd_inverse_ld(int m, const double *D, ld *X) {
    ld *C = (ld *)malloc(sizeof(ld) * (size_t)m * m);      /* lower Cholesky factor, row-major */
    ld *Wt = (ld *)calloc((size_t)m * m, sizeof(ld));      /* Wt[j][l] = inv(C)[l][j], l >= j: rows of the transpose are contiguous */
    if (!C || !Wt) { free(C); free(Wt); return -1; }
    for (int i = 0; i < m; i++) {
        for (int j = 0; j <= i; j++) {
            ld s = (ld)D[(size_t)i * m + j] - dot_ld(C + (size_t)i * m, C + (size_t)j * m, j);
            if (i == j) {
                if (s <= 0) { free(C); free(Wt); return i + 1; }
                C[(size_t)i * m + i] = sqrtl(s);
            } else
                C[(size_t)i * m + j] = s / C[(size_t)j * m + j];
        }
    }
    /* W = inv(C): W[i][j] = -(sum_{l=j}^{i-1} C[i][l] W[l][j]) / C[i][i], W[j][j] = 1 / C[j][j] */
    for (int j = 0; j < m; j++) {
        ld *wj = Wt + (size_t)j * m;
        wj[j] = 1.0L / C[(size_t)j * m + j];
        for (int i = j + 1; i < m; i++)
            wj[i] = -dot_ld(C + (size_t)i * m + j, wj + j, i - j) / C[(size_t)i * m + i];
    }
    /* X = W' W: X[a][b] = sum_{l >= max(a,b)} W[l][a] W[l][b] */
    for (int a = 0; a < m; a++)
        for (int b = 0; b <= a; b++) {
            ld s = dot_ld(Wt + (size_t)a * m + a, Wt + (size_t)b * m + a, m - a);
            X[(size_t)a * m + b] = s;
            X[(size_t)b * m + a] = s;
        }
    free(C); free(Wt);
    newton_dd(m, D, X);
    return 0;
}

/* max |I - D X / scale| with D, scale fp64 and X = X_hi + X_lo, products and sums in binary128: certifies X = scale * inv(D).
 * rows [r0, r1) only (the whole matrix costs m^3 software multiplications). */
double oracle_inverse_residual_q(int m, const double *D, const double *X_hi, const double *X_lo, double scale, int r0, int r1) {
    double worst = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(max : worst)
    for (int i = r0; i < r1; i++) {
        __float128 *acc = (__float128 *)malloc(sizeof(__float128) * (size_t)m);
        for (int j = 0; j < m; j++) acc[j] = (i == j) ? (__float128)scale : 0;
        for (int k = 0; k < m; k++) {
            __float128 dik = D[(size_t)i * m + k];
            const double *xh = X_hi + (size_t)k * m, *xl = X_lo ? X_lo + (size_t)k * m : NULL;
            for (int j = 0; j < m; j++) acc[j] -= dik * ((__float128)xh[j] + (xl ? (__float128)xl[j] : 0));
        }
        for (int j = 0; j < m; j++) {
            double a = fabs((double)(acc[j] / (__float128)scale));
            if (a > worst) worst = a;
        }
        free(acc);
    }
    return worst;
}

/* The same certificate over the WHOLE matrix at the price of a few matrix-vector products: for nvec vectors v of random signs,
 * max_i |(v - D (X v) / scale)_i| with X = X_hi + X_lo, everything in binary128 (2 m^2 software multiplications per vector instead of
 * m^3).  With R = I - D X / scale this is max_i |sum_j R_ij v_j|: a row of R whose 2-norm is rho shows up as ~rho in every probe, so a
 * handful of probes bounds every row of the residual (not its single largest entry, which the row-wise check above measures exactly).
 * The signs come from a 64-bit LCG seeded by the caller: the same numbers in every run. */
double oracle_inverse_residual_probe_q(int m, const double *D, const double *X_hi, const double *X_lo, double scale, int nvec,
                                       unsigned long long seed) {
    double worst = 0;
    __float128 *v = (__float128 *)malloc(sizeof(__float128) * (size_t)m), *y = (__float128 *)malloc(sizeof(__float128) * (size_t)m);
    unsigned long long st = seed * 6364136223846793005ULL + 1442695040888963407ULL;
    for (int t = 0; t < nvec; t++) {
        for (int j = 0; j < m; j++) {
            st = st * 6364136223846793005ULL + 1442695040888963407ULL;
            v[j] = (st >> 62) & 1 ? 1 : -1;
        }
        for (int i = 0; i < m; i++) {
            const double *xh = X_hi + (size_t)i * m, *xl = X_lo ? X_lo + (size_t)i * m : NULL;
            __float128 s = 0;
            for (int j = 0; j < m; j++) s += ((__float128)xh[j] + (xl ? (__float128)xl[j] : 0)) * v[j];
            y[i] = s;
        }
        for (int i = 0; i < m; i++) {
            const double *di = D + (size_t)i * m;
            __float128 s = (__float128)scale * v[i];
            for (int j = 0; j < m; j++) s -= (__float128)di[j] * y[j];
            double a = fabs((double)(s / (__float128)scale));
            if (a > worst) worst = a;
        }
    }
    free(v); free(y);
    return worst;
}

/* P = sigma2 * inv(D) of image block blk as a (hi, lo) pair of row-major m x m fp64 matrices */
int oracle_exact_block_weight(const jaicov_problem_desc *d, double sigma2, int blk, double *P_hi, double *P_lo) {
    int m = 2 * (d->blk_ip_begin[blk + 1] - d->blk_ip_begin[blk]);
    ld *X = (ld *)malloc(sizeof(ld) * (size_t)m * m);
    if (!X) return -1;
    int info = spd_inverse_ld(m, d->blk_disp + d->blk_disp_offset[blk], X);
    if (!info)
        for (size_t i = 0; i < (size_t)m * m; i++) {
            ld p = (ld)sigma2 * X[i];
            P_hi[i] = (double)p;
            if (P_lo) P_lo[i] = (double)(p - (ld)P_hi[i]);
        }
    free(X);
    return info;
}

/* one jointly dispersed image group, N += A' P A (upper), n += A' P w, extended precision throughout */
static int exact_image_block(const jaicov_problem_desc *d, const double *vals, double sigma2, int blk, ld *Ng, ld *ng) {
    int b = d->blk_ip_begin[blk], e = d->blk_ip_begin[blk + 1];
    int m = 2 * (e - b), U = d->n_unknowns;
    if (m == 0) return 0;
    int kmax = (e - b) * 3 + EX_KLOC, k = 0;
    int32_t *cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)kmax);
    int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)U);
    for (int i = 0; i < U; i++) pos[i] = -1;
    for (int ip = b; ip < e; ip++) {
        int32_t gcol[EX_KLOC];
        int kl = local_cols(d, ip, gcol);
        for (int l = 0; l < kl; l++)
            if (gcol[l] >= 0 && pos[gcol[l]] < 0) { pos[gcol[l]] = 0; cols[k++] = gcol[l]; }
    }
    /* ascending columns (any order gives the same sums up to the final rounding; ascending keeps the upper triangle simple) */
    for (int i = 1; i < k; i++) { int32_t c = cols[i]; int j = i - 1; while (j >= 0 && cols[j] > c) { cols[j + 1] = cols[j]; j--; } cols[j + 1] = c; }
    for (int j = 0; j < k; j++) pos[cols[j]] = j;
    /* sparse rows of A */
    int32_t *aj = (int32_t *)malloc(sizeof(int32_t) * (size_t)m * EX_KLOC);
    double *av = (double *)malloc(sizeof(double) * (size_t)m * EX_KLOC);
    int *an = (int *)calloc((size_t)m, sizeof(int));
    double *w = (double *)malloc(sizeof(double) * (size_t)m);
    for (int ip = b; ip < e; ip++) {
        double Al[2 * EX_KLOC], wl[2];
        int32_t gcol[EX_KLOC];
        oracle_rows(d, vals, sigma2, ip, wl, Al, EX_KLOC, NULL);
        int kl = local_cols(d, ip, gcol);
        for (int r = 0; r < 2; r++) {
            int row = 2 * (ip - b) + r, cnt = 0;
            for (int l = 0; l < kl; l++) {
                if (gcol[l] < 0) continue;
                aj[(size_t)row * EX_KLOC + cnt] = pos[gcol[l]];
                av[(size_t)row * EX_KLOC + cnt] = Al[r * EX_KLOC + l];
                cnt++;
            }
            an[row] = cnt;
            w[row] = wl[r];
        }
    }
    ld *X = (ld *)malloc(sizeof(ld) * (size_t)m * m);
    int info = X ? spd_inverse_ld(m, d->blk_disp + d->blk_disp_offset[blk], X) : -1;
    if (!info) {
        ld *T = (ld *)calloc((size_t)m * k, sizeof(ld)), *Pw = (ld *)calloc((size_t)m, sizeof(ld));
        ld *Nl = (ld *)calloc((size_t)k * k, sizeof(ld)), *nl = (ld *)calloc((size_t)k, sizeof(ld));
        for (int r = 0; r < m; r++) {                       /* T = P A, Pw = P w */
            ld *tr = T + (size_t)r * k, pw = 0;
            for (int q = 0; q < m; q++) {
                ld p = (ld)sigma2 * X[(size_t)r * m + q];
                const int32_t *jq = aj + (size_t)q * EX_KLOC;
                const double *vq = av + (size_t)q * EX_KLOC;
                for (int t = 0; t < an[q]; t++) tr[jq[t]] += p * (ld)vq[t];
                pw += p * (ld)w[q];
            }
            Pw[r] = pw;
        }
        for (int r = 0; r < m; r++) {                       /* Nl += A' T (upper), nl += A' Pw */
            const ld *tr = T + (size_t)r * k;
            for (int t = 0; t < an[r]; t++) {
                int i = aj[(size_t)r * EX_KLOC + t];
                ld a = (ld)av[(size_t)r * EX_KLOC + t];
                ld *ni = Nl + (size_t)i * k;
                for (int j = i; j < k; j++) ni[j] += a * tr[j];
                nl[i] += a * Pw[r];
            }
        }
#pragma omp critical(exact_scatter)
        {
            for (int i = 0; i < k; i++) {
                for (int j = i; j < k; j++) Ng[pidx(cols[i], cols[j])] += Nl[(size_t)i * k + j];
                ng[cols[i]] += nl[i];
            }
        }
        free(T); free(Pw); free(Nl); free(nl);
    }
    free(X); free(aj); free(av); free(an); free(w); free(cols); free(pos);
    return info;
}

/* All observation groups of the problem (BA:795-797) in extended precision.  Outputs: packed 'U' N and n as (hi, lo) pairs.
 * Datum rows, damping and the preconditioner are NOT applied (createNormalEquation's tail, BA:799-831, works on the result). */
int oracle_exact_accumulate(const jaicov_problem_desc *d, const double *vals, double sigma2, double *N_hi, double *N_lo,
                            double *n_hi, double *n_lo) {
    int U = d->n_unknowns;
    size_t len = (size_t)U * ((size_t)U + 1) / 2;
    ld *Ng = (ld *)calloc(len, sizeof(ld)), *ng = (ld *)calloc((size_t)U, sizeof(ld));
    if (!Ng || !ng) { free(Ng); free(ng); return -1; }
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int blk = 0; blk < d->n_image_blocks; blk++) {
        int info = exact_image_block(d, vals, sigma2, blk, Ng, ng);
        if (info) {
#pragma omp critical(exact_bad)
            bad = info;
        }
    }
    /* image points outside blocks: fp64 rows and 2 x 2 weights (PDF:296-319), extended sums */
    uint8_t *in_block = (uint8_t *)calloc((size_t)d->n_image_points + 1, 1);
    for (int b = 0; b < d->n_image_blocks; b++)
        for (int ip = d->blk_ip_begin[b]; ip < d->blk_ip_begin[b + 1]; ip++) in_block[ip] = 1;
    for (int ip = 0; ip < d->n_image_points; ip++) {
        if (in_block[ip]) continue;
        double Al[2 * EX_KLOC], wl[2], P[4];
        int32_t gcol[EX_KLOC];
        oracle_rows(d, vals, sigma2, ip, wl, Al, EX_KLOC, P);
        int kl = local_cols(d, ip, gcol);
        for (int a = 0; a < kl; a++) {
            if (gcol[a] < 0) continue;
            ld pa0 = (ld)Al[a] * P[0] + (ld)Al[EX_KLOC + a] * P[2], pa1 = (ld)Al[a] * P[1] + (ld)Al[EX_KLOC + a] * P[3];
            ng[gcol[a]] += pa0 * wl[0] + pa1 * wl[1];
            for (int c = 0; c < kl; c++) {
                if (gcol[c] < gcol[a]) continue;               /* upper triangle; fixed columns are negative */
                Ng[pidx(gcol[a], gcol[c])] += pa0 * Al[c] + pa1 * Al[EX_KLOC + c];
            }
        }
    }
    free(in_block);
    /* scale bars (PDF:210-283) */
    for (int s = 0; s < d->n_scale_bars; s++) {
        int pa = d->sb_point_a[s], pb = d->sb_point_b[s];
        const double *a = vals + 3 * pa, *b = vals + 3 * pb;
        ld dX = (ld)b[0] - a[0], dY = (ld)b[1] - a[1], dZ = (ld)b[2] - a[2];
        ld len3 = sqrtl(dX * dX + dY * dY + dZ * dZ);
        ld e6[6] = {-dX / len3, -dY / len3, -dZ / len3, dX / len3, dY / len3, dZ / len3};
        ld P = (ld)sigma2 / d->sb_var[s], w = (ld)d->sb_length[s] - len3;
        int32_t col[6];
        for (int t = 0; t < 6; t++) col[t] = t < 3 ? d->point_col[3 * pa + t] : d->point_col[3 * pb + t - 3];
        for (int i = 0; i < 6; i++) {
            if (col[i] < 0) continue;
            ng[col[i]] += e6[i] * P * w;
            for (int j = 0; j < 6; j++)
                if (col[j] >= col[i]) Ng[pidx(col[i], col[j])] += e6[i] * P * e6[j];
        }
    }
    /* directly observed groups (PDF:447-473): A = unit rows, P diagonal or sigma0^2 inv(D) */
    int32_t *slot_col = (int32_t *)malloc(sizeof(int32_t) * (size_t)oracle_num_slots(d));
    oracle_slot_columns(d, slot_col);
    for (int g = 0; g < d->n_direct_groups && !bad; g++) {
        int b = d->dg_row_begin[g], e = d->dg_row_begin[g + 1], m = e - b;
        if (m == 0) continue;
        int dense = d->dg_disp_offset && d->dg_disp_offset[g] >= 0;
        ld *X = NULL;
        if (dense) {
            X = (ld *)malloc(sizeof(ld) * (size_t)m * m);
            int info = spd_inverse_ld(m, d->dg_disp + d->dg_disp_offset[g], X);
            if (info) { bad = info; free(X); break; }
        }
        for (int r = 0; r < m; r++) {
            int cr = slot_col[d->dg_slot[b + r]];
            if (cr < 0) continue;
            for (int q = 0; q < m; q++) {
                ld p = dense ? (ld)sigma2 * X[(size_t)r * m + q] : (r == q ? (ld)sigma2 / d->dg_var[b + r] : 0);
                if (p == 0) continue;
                int sq = d->dg_slot[b + q], cq = slot_col[sq];
                ng[cr] += p * ((ld)d->dg_obs[b + q] - vals[sq]);
                if (cq >= cr) Ng[pidx(cr, cq)] += p;
            }
        }
        free(X);
    }
    free(slot_col);
    for (size_t i = 0; i < len; i++) {
        N_hi[i] = (double)Ng[i];
        N_lo[i] = (double)(Ng[i] - (ld)N_hi[i]);
    }
    for (int i = 0; i < U; i++) {
        n_hi[i] = (double)ng[i];
        n_lo[i] = (double)(ng[i] - (ld)n_hi[i]);
    }
    free(Ng); free(ng);
    return bad;
}

/* r = (b_hi + b_lo) - (A_hi + A_lo) x, A symmetric packed 'U', accumulated in extended precision, rounded once.
 * nrhs right-hand sides / solutions stored one after another (x, b, r: nrhs x n); lo parts may be NULL. */
void oracle_residual_ld2(int n, const double *ap_hi, const double *ap_lo, int nrhs, const double *x, const double *b_hi,
                         const double *b_lo, double *r) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < nrhs; q++) {
        const double *xq = x + (size_t)q * n;
        ld *acc = (ld *)malloc(sizeof(ld) * (size_t)(n > 0 ? n : 1));
        for (int i = 0; i < n; i++) acc[i] = (ld)b_hi[(size_t)q * n + i] + (b_lo ? (ld)b_lo[(size_t)q * n + i] : 0);
        size_t kk = 0;
        for (int j = 0; j < n; j++) {
            ld xj = (ld)xq[j], t = 0;
            for (int i = 0; i < j; i++) {
                ld a = (ld)ap_hi[kk + i] + (ap_lo ? (ld)ap_lo[kk + i] : 0);
                acc[i] -= a * xj;
                t += a * (ld)xq[i];
            }
            acc[j] -= t + ((ld)ap_hi[kk + j] + (ap_lo ? (ld)ap_lo[kk + j] : 0)) * xj;
            kk += (size_t)j + 1;
        }
        for (int i = 0; i < n; i++) r[(size_t)q * n + i] = (double)acc[i];
        free(acc);
    }
}

/* y = (A_hi + A_lo) v in extended precision, rounded once (the probe N.v of the fixtures) */
void oracle_matvec_ld2(int n, const double *ap_hi, const double *ap_lo, const double *v, double *y) {
    double *zero = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    oracle_residual_ld2(n, ap_hi, ap_lo, 1, v, zero, NULL, y);
    for (int i = 0; i < n; i++) y[i] = -y[i];
    free(zero);
}
