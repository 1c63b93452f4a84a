/*
 * ba_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded, fp64 restatement of the normal-equation hot path of JAICOV
 * (applied-geodesy/bundle-adjustment).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product (bundle-adjustment_amd/) never links, imports or calls it.
 *
 * PINNING STATUS: the reference ships no tests, golden vectors or fixtures for this path and
 * is Java, which this image cannot run (no JVM).  The restatement is therefore pinned by
 *   (1) independent sympy/mpmath 50-digit derivatives of the published model function (tests/golden/jacobian_rows.json),
 *   (2) the structural known answers of JAICOV/example (n = 19 945, u = 1 147, d = 6, dof = 18 804; example.htm:33-35,42)
 *       and its near-fixed-point behaviour; from the same third-party report (tests/golden/example/example.htm.gz) the
 *       a-posteriori S0 = 0.000405, the standard deviations (6 digits) and correlations (3 decimals) of the interior
 *       orientation (example.htm:77-98), and in AICON's datum the standard deviations of all object points and projection
 *       centres (4 decimals, 795 values), and the 19 944 image-coordinate residuals (6 decimals) --
 *       tests/test_host.py::test_oracle_reproduces_the_reports_*,
 *   (3) LAPACK dsytrf/dsysv/dpotrf as shipped with scipy for the packed solver restatements.
 * The third-party arithmetic the reference delegates to -- com.googlecode.matrix-toolkits-java:mtj:1.0.4 and
 * com.github.fommil.netlib:core:1.1.2 (F2jLAPACK dspsv/dsptrf/dsptrs/dsptri/dpptrf/dpptri) -- exists in
 * /root/reference only as binary jars; its published (LAPACK 3.1 reference) algorithm is restated here.
 * At that boundary parity is "unpinned" by reference-held vectors (none exist) and pinned by (3) instead.
 *
 * Each function cites the reference lines it follows (paths relative to JAICOV/src/org/applied_geodesy/):
 *   BA  adjustment/bundle/BundleAdjustment.java          PDF adjustment/bundle/derivation/PartialDerivativeFactory.java
 *   DMF .../derivation/DistortionModelFactory.java       RSF .../derivation/RadiallySymmetricDistortionModelFactory.java
 *   TDF .../derivation/TangentialDistortionModelFactory.java   ASF .../derivation/AffinityShearDistortionModelFactory.java
 *   RDF .../derivation/RadialDistanceDistortionModelFactory.java
 *   NES adjustment/NormalEquationSystem.java             MX  adjustment/MathExtension.java
 *   DOPG adjustment/bundle/parameter/DirectlyObservedParameterGroup.java
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/jaicov_neq.h"

#define ORA_MAXD 64                 /* distortion coefficients per camera the oracle accepts */
#define ORA_KLOC (12 + ORA_MAXD)    /* local parameter ids: 0-2 XYZ, 3 x0, 4 y0, 5 c, 6-11 EO, 12.. distortion */

/* Constant.EPS (adjustment/Constant.java:68-75): the loop ends with eps = 2^-53 */
double oracle_eps(void) {
    volatile double eps = 1.0, x = 2.0, y = 1.0;
    while (y < x) { eps *= 0.5; x = 1.0 + eps; }
    return eps;
}

/* ---- slot layout (include/jaicov_neq.h) -------------------------------------------------------------------- */
static inline int slot_point(const jaicov_problem_desc *d, int p) { (void)d; return 3 * p; }
static inline int slot_io(const jaicov_problem_desc *d, int c) { return 3 * d->n_points + 3 * c; }
static inline int slot_dist(const jaicov_problem_desc *d, int j) { return 3 * d->n_points + 3 * d->n_cameras + j; }
static inline int slot_eo(const jaicov_problem_desc *d, int i) { return 3 * d->n_points + 3 * d->n_cameras + d->n_dist + 6 * i; }
int oracle_num_slots(const jaicov_problem_desc *d) { return 3 * d->n_points + 3 * d->n_cameras + d->n_dist + 6 * d->n_images; }

/* column of every slot (JAICOV_COL_FIXED if none) */
void oracle_slot_columns(const jaicov_problem_desc *d, int32_t *col) {
    int s = 0;
    for (int i = 0; i < 3 * d->n_points; i++) col[s++] = d->point_col[i];
    for (int i = 0; i < 3 * d->n_cameras; i++) col[s++] = d->io_col[i];
    for (int i = 0; i < d->n_dist; i++) col[s++] = d->dist_col[i];
    for (int i = 0; i < 6 * d->n_images; i++) col[s++] = d->eo_col[i];
}

/* ---- MTJ UpperSymmPackMatrix semantics (getIndex(r,c) = r + (c+1)*c/2; add/set are no-ops for r > c) */
static inline size_t pidx(int r, int c) { return (size_t)r + ((size_t)c + 1) * (size_t)c / 2; }
static inline void pk_add(double *N, int r, int c, double v) { if (r <= c) N[pidx(r, c)] += v; }
static inline void pk_set(double *N, int r, int c, double v) { if (r <= c) N[pidx(r, c)] = v; }
static inline double pk_get(const double *N, int r, int c) { return r <= c ? N[pidx(r, c)] : N[pidx(c, r)]; }

/* ============================================================================================================
 * a1-a7: collinearity equations, analytic partials, distortion chain rule for ONE image point
 * ============================================================================================================ */
typedef struct {
    double cosOmega, sinOmega, cosPhi, sinPhi, cosKappa, sinKappa;
    double r11, r12, r13, r21, r22, r23, r31, r32, r33;
    double xs, ys, x, y, dX, dY, dZ, kx, ky, N, kxN, kyN;
    /* [0] = xs-row, [1] = ys-row partials in local order X,Y,Z,x0,y0,c,X0,Y0,Z0,omega,phi,kappa */
    double par[2][12];
} colli_t;

/* PDF:94-190 CollinearityEquationFactory.<init> */
static void collinearity(colli_t *q, double c, double x0, double y0, const double eo[6], const double pt[3]) {
    double X0 = eo[0], Y0 = eo[1], Z0 = eo[2], omega = eo[3], phi = eo[4], kappa = eo[5];
    double X = pt[0], Y = pt[1], Z = pt[2];
    q->cosOmega = cos(omega); q->sinOmega = sin(omega);
    q->cosPhi = cos(phi);     q->sinPhi = sin(phi);
    q->cosKappa = cos(kappa); q->sinKappa = sin(kappa);
    /* PDF:125-135 */
    q->r11 =  q->cosPhi * q->cosKappa;
    q->r12 = -q->cosPhi * q->sinKappa;
    q->r13 =  q->sinPhi;
    q->r21 =  q->cosOmega * q->sinKappa + q->sinOmega * q->sinPhi * q->cosKappa;
    q->r22 =  q->cosOmega * q->cosKappa - q->sinOmega * q->sinPhi * q->sinKappa;
    q->r23 = -q->sinOmega * q->cosPhi;
    q->r31 = q->sinOmega * q->sinKappa - q->cosOmega * q->sinPhi * q->cosKappa;
    q->r32 = q->sinOmega * q->cosKappa + q->cosOmega * q->sinPhi * q->sinKappa;
    q->r33 = q->cosOmega * q->cosPhi;
    /* PDF:137-152 */
    q->dX = X - X0; q->dY = Y - Y0; q->dZ = Z - Z0;
    q->kx = q->r11 * q->dX + q->r21 * q->dY + q->r31 * q->dZ;
    q->ky = q->r12 * q->dX + q->r22 * q->dY + q->r32 * q->dZ;
    q->N  = q->r13 * q->dX + q->r23 * q->dY + q->r33 * q->dZ;
    q->kxN = q->kx / q->N;
    q->kyN = q->ky / q->N;
    q->xs = -c * q->kxN;
    q->ys = -c * q->kyN;
    q->x = x0 + q->xs;
    q->y = y0 + q->ys;
    /* PDF:157-171 x-equation */
    double *px = q->par[0], *py = q->par[1];
    px[0] = -(q->r13 * q->xs + c * q->r11) / q->N;
    px[1] = -(q->r23 * q->xs + c * q->r21) / q->N;
    px[2] = -(q->r33 * q->xs + c * q->r31) / q->N;
    px[3] = 1.0; px[4] = 0.0; px[5] = -q->kxN;
    px[6] = -px[0]; px[7] = -px[1]; px[8] = -px[2];
    px[9]  = (q->xs * (q->r33 * q->dY - q->r23 * q->dZ) + c * (q->r31 * q->dY - q->r21 * q->dZ)) / q->N;
    px[10] = (q->xs * (q->ky * q->sinKappa - q->kx * q->cosKappa) + c * q->N * q->cosKappa) / q->N;
    px[11] = q->ys;
    /* PDF:175-189 y-equation */
    py[0] = -(q->r13 * q->ys + c * q->r12) / q->N;
    py[1] = -(q->r23 * q->ys + c * q->r22) / q->N;
    py[2] = -(q->r33 * q->ys + c * q->r32) / q->N;
    py[3] = 0.0; py[4] = 1.0; py[5] = -q->kyN;
    py[6] = -py[0]; py[7] = -py[1]; py[8] = -py[2];
    py[9]  = (q->ys * (q->r33 * q->dY - q->r23 * q->dZ) + c * (q->r32 * q->dY - q->r22 * q->dZ)) / q->N;
    py[10] = (q->ys * (q->ky * q->sinKappa - q->kx * q->cosKappa) - c * q->N * q->sinKappa) / q->N;
    py[11] = -q->xs;
}

/* DMF:33-101 DistortionModelFactory.apply: w -= delta; chain rule onto X,Y,Z,c,X0,Y0,Z0,omega,phi,kappa (NOT x0,y0) */
static void dmf_apply(const colli_t *q, double A[2][ORA_KLOC], double w[2], double deltaX, double deltaY,
                      double dXxs, double dXys, double dYxs, double dYys) {
    static const int targets[10] = {0, 1, 2, 5, 6, 7, 8, 9, 10, 11};
    w[0] += -deltaX;
    w[1] += -deltaY;
    for (int t = 0; t < 10; t++) {
        int l = targets[t];
        A[0][l] += dXxs * q->par[0][l] + dXys * q->par[1][l];
        A[1][l] += dYxs * q->par[0][l] + dYys * q->par[1][l];
    }
}

#define ORA_ZMAX 16   /* radial terms of one Zernike polynomial */
#define ORA_PI 3.14159265358979323846

/* MathExtension.binomial (MathExtension.java:53-64) */
static long binomial_ref(int n, int k) {
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    long result = 1;
    for (int i = 1; i <= k; i++) result = result * (n - k + i) / i;
    return result;
}

/* ZernikeCoefficient.ZernikePolynomial (ZernikeCoefficient.java:41-57): radial exponents p[], coefficients len*c[] */
static int zernike_terms(int order, int *m_out, long p[ORA_ZMAX], double cn[ORA_ZMAX]) {
    int n = (int)ceil((-3.0 + sqrt(9.0 + 8.0 * order)) / 2.0);
    int m = 2 * order - n * (n + 2);
    int halfnm = (n - abs(m)) / 2;
    double length = sqrt((1 + ((m != 0) ? 1 : 0)) * (n + 1) / ORA_PI);
    for (int k = 0; k <= halfnm && k < ORA_ZMAX; k++) {
        p[k] = n - 2 * k;
        long c = ((k % 2 == 0) ? 1 : -1) * binomial_ref(n - k, k) * binomial_ref(n - 2 * k, halfnm - k);
        cn[k] = length * (double)c;
    }
    *m_out = m;
    return halfnm + 1;
}

/*
 * PDF:285-445 getPartialDerivativeImageCoordinate without the stacking: fills the local 2 x KLOC row block, the
 * misclosure w and the 2x2 weight P (row-major; diag != 0 when rho == 0, PDF:300).  vals = slot vector.
 */
static void eval_image_point(const jaicov_problem_desc *d, const double *vals, double sigma2, int ip,
                             double A[2][ORA_KLOC], double w[2], double P[4], int *diagonal) {
    int img = d->ip_image[ip], pt = d->ip_point[ip], cam = d->image_camera[img];
    const double *io = vals + slot_io(d, cam);        /* x0, y0, c */
    const double *eo = vals + slot_eo(d, img);
    const double *xyz = vals + slot_point(d, pt);
    colli_t q;
    collinearity(&q, io[2], io[0], io[1], eo, xyz);

    /* PDF:296-319 stochastic model */
    double vx = d->ip_var_x[ip], vy = d->ip_var_y[ip], rho = d->ip_rho[ip];
    *diagonal = (rho == 0);
    if (*diagonal) {
        P[0] = sigma2 / vx; P[3] = sigma2 / vy; P[1] = P[2] = 0.0;
    } else {
        double invDet = sigma2 / ((1.0 - rho * rho) * vx * vy);
        P[0] = invDet * vy;
        P[3] = invDet * vx;
        P[1] = P[2] = -invDet * rho * sqrt(vx * vy);
    }
    /* PDF:321-322 */
    w[0] = d->ip_x[ip] - q.x;
    w[1] = d->ip_y[ip] - q.y;
    /* PDF:327-414 (A.set of the twelve base columns) */
    memset(A, 0, sizeof(double) * 2 * ORA_KLOC);
    for (int l = 0; l < 12; l++) { A[0][l] = q.par[0][l]; A[1][l] = q.par[1][l]; }

    /* PDF:417-442: models applied in DistortionModel.Type order; the flattening keeps that order in dist_kind */
    int jb = d->cam_dist_begin[cam], je = d->cam_dist_begin[cam + 1];
    const double *dv = vals + slot_dist(d, 0);
    double r0 = d->cam_r0[cam];
    double r2 = q.xs * q.xs + q.ys * q.ys;
    double r02 = r0 * r0;
    double xxs2 = 2.0 * q.xs * q.xs, yys2 = 2.0 * q.ys * q.ys, xys2 = 2.0 * q.xs * q.ys;

    int jCx = -1, jCy = -1, jBx = -1, jBy = -1;
    for (int j = jb; j < je; j++) {
        switch (d->dist_kind[j]) {
        case JAICOV_DIST_AFFINITY_CX: jCx = j; break;
        case JAICOV_DIST_AFFINITY_CY: jCy = j; break;
        case JAICOV_DIST_TANGENTIAL_BX: jBx = j; break;
        case JAICOV_DIST_TANGENTIAL_BY: jBy = j; break;
        default: break;
        }
    }
    /* ASF:37-81 */
    if (jCx >= 0 && jCy >= 0) {
        double cx = dv[jCx], cy = dv[jCy];
        double deltaX = cx * q.xs + cy * q.ys, deltaY = 0.0;
        dmf_apply(&q, A, w, deltaX, deltaY, cx, cy, 0.0, 0.0);
        A[0][12 + jCx - jb] = q.xs; A[1][12 + jCx - jb] = 0.0;
        A[0][12 + jCy - jb] = q.ys; A[1][12 + jCy - jb] = 0.0;
    }
    /* TDF:39-134 */
    if (jBx >= 0 && jBy >= 0) {
        double bx = dv[jBx], by = dv[jBy];
        double sum = 1.0;
        double deltaX = bx * (r2 + xxs2) + by * xys2;
        double deltaY = by * (r2 + yys2) + bx * xys2;
        double dXxs = 2.0 * (3.0 * bx * q.xs + by * q.ys);
        double dXys = 2.0 * (by * q.xs + bx * q.ys);
        double dYxs = 2.0 * (by * q.xs + bx * q.ys);
        double dYys = 2.0 * (bx * q.xs + 3.0 * by * q.ys);
        dmf_apply(&q, A, w, deltaX, deltaY, dXxs, dXys, dYxs, dYys);
        for (int j = jb; j < je; j++) {
            if (d->dist_kind[j] != JAICOV_DIST_TANGENTIAL_BI) continue;
            double bi = dv[j];
            int expi = d->dist_order[j];
            double ri = pow(r2, expi);
            double dTani = bi * ri;
            sum += dTani;
            double deltaXi = deltaX * dTani, deltaYi = deltaY * dTani;
            double par_xs_Bi = deltaX * ri, par_ys_Bi = deltaY * ri;
            double constTani = 2.0 * bi * expi * pow(r2, expi - 1);
            double constTanXi = deltaX * constTani, constTanYi = deltaY * constTani;
            double dXxsi = dTani * dXxs + q.xs * constTanXi;
            double dXysi = dTani * dXys + q.ys * constTanXi;
            double dYxsi = dTani * dYxs + q.xs * constTanYi;
            double dYysi = dTani * dYys + q.ys * constTanYi;
            dmf_apply(&q, A, w, deltaXi, deltaYi, dXxsi, dXysi, dYxsi, dYysi);
            A[0][12 + j - jb] = par_xs_Bi; A[1][12 + j - jb] = par_ys_Bi;
        }
        A[0][12 + jBx - jb] = sum * (r2 + xxs2); A[1][12 + jBx - jb] = sum * xys2;
        A[0][12 + jBy - jb] = sum * xys2;        A[1][12 + jBy - jb] = sum * (r2 + yys2);
    }
    /* RSF:39-90 */
    for (int j = jb; j < je; j++) {
        if (d->dist_kind[j] != JAICOV_DIST_RADIAL_AI) continue;
        double ai = dv[j];
        int expi = d->dist_order[j];
        double dRi = pow(r2, expi) - pow(r02, expi);
        double dRadi = ai * dRi;
        double deltaX = q.xs * dRadi, deltaY = q.ys * dRadi;
        double constRadi = ai * expi * pow(r2, expi - 1);
        double dXxs = xxs2 * constRadi + dRadi, dXys = xys2 * constRadi;
        double dYxs = xys2 * constRadi, dYys = yys2 * constRadi + dRadi;
        dmf_apply(&q, A, w, deltaX, deltaY, dXxs, dXys, dYxs, dYys);
        A[0][12 + j - jb] = q.xs * dRi; A[1][12 + j - jb] = q.ys * dRi;
    }
    /* RDF:39-161 */
    for (int j = jb; j < je; j++) {
        if (d->dist_kind[j] != JAICOV_DIST_DISTANCE_DI) continue;
        double di = dv[j];
        int expi = d->dist_order[j];
        double dRi = pow(r2, expi) - pow(r02, expi);
        double dDisti = (di * dRi) / q.N;
        double deltaX = q.xs * dDisti, deltaY = q.ys * dDisti;
        double parN[12] = {q.r13, q.r23, q.r33, 0, 0, 0, -q.r13, -q.r23, -q.r33,
                           -q.r33 * q.dY + q.r23 * q.dZ, q.kx * q.cosKappa - q.ky * q.sinKappa, 0.0};
        double constRadi = (di * expi * pow(r2, expi - 1)) / q.N;
        double dXxs = xxs2 * constRadi + dDisti, dXys = xys2 * constRadi;
        double dXN = -deltaX / q.N;
        double dYxs = xys2 * constRadi, dYys = yys2 * constRadi + dDisti;
        double dYN = -deltaY / q.N;
        dmf_apply(&q, A, w, deltaX, deltaY, dXxs, dXys, dYxs, dYys);
        A[0][12 + j - jb] = (q.xs * dRi) / q.N; A[1][12 + j - jb] = (q.ys * dRi) / q.N;
        static const int targets[9] = {0, 1, 2, 6, 7, 8, 9, 10, 11};
        for (int t = 0; t < 9; t++) {
            int l = targets[t];
            A[0][l] += parN[l] * dXN;
            A[1][l] += parN[l] * dYN;
        }
    }
    /* ZernikeDistortionModelFactory.java:153-227 (X, then Y: DistortionModel.Type order) */
    for (int kind = JAICOV_DIST_ZERNIKE_X; kind <= JAICOV_DIST_ZERNIKE_Y; kind++)
        for (int j = jb; j < je; j++) {
            if (d->dist_kind[j] != kind) continue;
            double xxs = q.xs * q.xs, yys = q.ys * q.ys;
            double phi = atan2(q.ys, q.xs);
            double rr2 = xxs + yys, rn2 = rr2 / r02;
            double zi = dv[j];
            long p[ORA_ZMAX]; double cn[ORA_ZMAX]; int mi;
            int nt = zernike_terms(d->dist_order[j], &mi, p, cn);
            double m = mi, sinmphi = sin(m * phi), cosmphi = cos(m * phi);
            double par_Zi = 0;
            for (int t = 0; t < nt; t++) {
                long pj = p[t];
                double cj = cn[t];
                double constC = cj * pow(rn2, (double)(pj / 2));
                double constZ = zi * cj / r02 * pow(rn2, (double)(pj / 2 - 1));
                double delta, par_delta_xs, par_delta_ys;
                if (m < 0) {
                    double constXsin = (-pj * q.xs * sinmphi + m * q.ys * cosmphi);
                    double constYsin = (-pj * q.ys * sinmphi - m * q.xs * cosmphi);
                    par_delta_xs = constZ * constXsin; par_delta_ys = constZ * constYsin;
                    delta = -zi * constC * sinmphi;
                    par_Zi += -constC * sinmphi;
                } else {
                    double constXcos = (pj * q.xs * cosmphi + m * q.ys * sinmphi);
                    double constYcos = (pj * q.ys * cosmphi - m * q.xs * sinmphi);
                    par_delta_xs = constZ * constXcos; par_delta_ys = constZ * constYcos;
                    delta = +zi * constC * cosmphi;
                    par_Zi += +constC * cosmphi;
                }
                if (kind == JAICOV_DIST_ZERNIKE_X) dmf_apply(&q, A, w, delta, 0, par_delta_xs, par_delta_ys, 0, 0);
                else dmf_apply(&q, A, w, 0, delta, 0, 0, par_delta_xs, par_delta_ys);
            }
            if (kind == JAICOV_DIST_ZERNIKE_X) A[0][12 + j - jb] = par_Zi;
            else A[1][12 + j - jb] = par_Zi;
        }
    /* ZernikeDistortionModelFactory.java:41-143 (Gradient) */
    for (int j = jb; j < je; j++) {
        if (d->dist_kind[j] != JAICOV_DIST_ZERNIKE_Z) continue;
        double xs = q.xs, ys = q.ys;
        double xxs = xs * xs, yys = ys * ys, xys = xs * ys;
        double phi = atan2(ys, xs);
        double rr2 = xxs + yys, rn2 = rr2 / r02;
        double const2rnr0 = 2.0 / rn2 / r02;
        double zi = dv[j];
        long p[ORA_ZMAX]; double cn[ORA_ZMAX]; int mi;
        int nt = zernike_terms(d->dist_order[j], &mi, p, cn);
        double m = mi, sinmphi = sin(m * phi), cosmphi = cos(m * phi);
        double par_xs_Zi = 0, par_ys_Zi = 0;
        for (int t = 0; t < nt; t++) {
            long pj = p[t];
            long constExp = (pj / 2 - 1);
            double cj = cn[t];
            double constC = cj / r02 * pow(rn2, (double)constExp);
            if (m < 0) {
                double constXsin = (-pj * xs * sinmphi + m * ys * cosmphi);
                double constYsin = (-pj * ys * sinmphi - m * xs * cosmphi);
                double deltaX = zi * constC * constXsin, deltaY = zi * constC * constYsin;
                double dXxs = zi * constC * (constExp * xs * const2rnr0 * constXsin - pj * sinmphi + m / rr2 * (pj * xys * cosmphi + m * yys * sinmphi));
                double dXys = zi * constC * (constExp * ys * const2rnr0 * constXsin + m * cosmphi - m / rr2 * (pj * xxs * cosmphi + m * xys * sinmphi));
                double dYxs = zi * constC * (constExp * xs * const2rnr0 * constYsin - m * cosmphi + m / rr2 * (pj * yys * cosmphi - m * xys * sinmphi));
                double dYys = zi * constC * (constExp * ys * const2rnr0 * constYsin - pj * sinmphi - m / rr2 * (pj * xys * cosmphi - m * xxs * sinmphi));
                dmf_apply(&q, A, w, deltaX, deltaY, dXxs, dXys, dYxs, dYys);
                par_xs_Zi += constC * constXsin; par_ys_Zi += constC * constYsin;
            } else {
                double constXcos = (pj * xs * cosmphi + m * ys * sinmphi);
                double constYcos = (pj * ys * cosmphi - m * xs * sinmphi);
                double deltaX = zi * constC * constXcos, deltaY = zi * constC * constYcos;
                double dXxs = zi * constC * (constExp * xs * const2rnr0 * constXcos + pj * cosmphi + m / rr2 * (pj * xys * sinmphi - m * yys * cosmphi));
                double dXys = zi * constC * (constExp * ys * const2rnr0 * constXcos + m * sinmphi - m / rr2 * (pj * xxs * sinmphi - m * xys * cosmphi));
                double dYxs = zi * constC * (constExp * xs * const2rnr0 * constYcos - m * sinmphi + m / rr2 * (pj * yys * sinmphi + m * xys * cosmphi));
                double dYys = zi * constC * (constExp * ys * const2rnr0 * constYcos + pj * cosmphi - m / rr2 * (pj * xys * sinmphi + m * xxs * cosmphi));
                dmf_apply(&q, A, w, deltaX, deltaY, dXxs, dXys, dYxs, dYys);
                par_xs_Zi += constC * constXcos; par_ys_Zi += constC * constYcos;
            }
        }
        A[0][12 + j - jb] = par_xs_Zi; A[1][12 + j - jb] = par_ys_Zi;
    }
}

/* local id -> global column of image point ip (JAICOV_COL_FIXED where the parameter owns none) */
static int local_columns(const jaicov_problem_desc *d, int ip, int32_t gcol[ORA_KLOC]) {
    int img = d->ip_image[ip], pt = d->ip_point[ip], cam = d->image_camera[img];
    int jb = d->cam_dist_begin[cam], je = d->cam_dist_begin[cam + 1];
    for (int a = 0; a < 3; a++) gcol[a] = d->point_col[3 * pt + a];
    for (int a = 0; a < 3; a++) gcol[3 + a] = d->io_col[3 * cam + a];
    for (int a = 0; a < 6; a++) gcol[6 + a] = d->eo_col[6 * img + a];
    for (int j = jb; j < je; j++) gcol[12 + j - jb] = d->dist_col[j];
    return 12 + je - jb;
}

/* parity hook: rows of one image point in the local layout of jaicov_neq_get_rows() */
int oracle_rows(const jaicov_problem_desc *d, const double *vals, double sigma2, int ip, double *w2, double *A_out,
                int kloc_out, double *P4) {
    double A[2][ORA_KLOC], w[2], P[4];
    int diag;
    eval_image_point(d, vals, sigma2, ip, A, w, P, &diag);
    w2[0] = w[0]; w2[1] = w[1];
    for (int r = 0; r < 2; r++)
        for (int l = 0; l < kloc_out; l++) A_out[r * kloc_out + l] = l < ORA_KLOC ? A[r][l] : 0.0;
    if (P4) memcpy(P4, P, sizeof(P));
    return diag;
}

/* ============================================================================================================
 * a10: stackNormalEquationSystem (PDF:475-505), generic: A is m x k (compact: column j of A belongs to cols[j]),
 * cols sorted ascending and unique, P m x m row-major (diag: only P[r][r] read).  N may be NULL (BA:476).
 * The loop nest and the order of the additions are the reference's.
 * ============================================================================================================ */
static void stack_neq(double *N, double *n, int m, int k, const double *A, const int32_t *cols, const double *P,
                      const double *w, int diag) {
    for (int row = 0; row < m; row++) {
        for (int ia = 0; ia < k; ia++) {
            int colAT = cols[ia];
            double aT = A[(size_t)row * k + ia];
            if (diag)
                n[colAT] += aT * P[(size_t)row * m + row] * w[row];
            else
                for (int colP = 0; colP < m; colP++) n[colAT] += aT * P[(size_t)row * m + colP] * w[colP];
            if (N) {
                for (int ib = ia; ib < k; ib++) {
                    int colA = cols[ib];
                    if (diag)
                        pk_add(N, colAT, colA, aT * P[(size_t)row * m + row] * A[(size_t)row * k + ib]);
                    else
                        for (int colP = 0; colP < m; colP++)
                            pk_add(N, colAT, colA, aT * P[(size_t)row * m + colP] * A[(size_t)colP * k + ib]);
                }
            }
        }
    }
}

/* sort (col, local) pairs ascending by col: Collections.sort(columns) PDF:476 */
static void sort_pairs(int k, int32_t *cols, int32_t *loc) {
    for (int i = 1; i < k; i++) {
        int32_t c = cols[i], l = loc[i];
        int j = i - 1;
        while (j >= 0 && cols[j] > c) { cols[j + 1] = cols[j]; loc[j + 1] = loc[j]; j--; }
        cols[j + 1] = c; loc[j + 1] = l;
    }
}

/* ---- packed SPD inverse: MX.inv(UpperSPDPackMatrix) MX:304-324 = dpptrf + dpptri ('U') ---------------------- */
/* dpptrf: A = U'U, column by column (LAPACK dpptrf.f, UPLO='U') */
int oracle_dpptrf(int n, double *ap) {
    size_t jj = 0;                                   /* 0-based index of A(j,j) */
    for (int j = 0; j < n; j++) {
        size_t jc = jj + 1 - 0;                      /* placeholder to keep names close to dpptrf.f */
        (void)jc;
        size_t colstart = (size_t)j * (j + 1) / 2;   /* A(0,j) */
        /* compute elements 0..j-1 of column j: solve U(0:j-1,0:j-1)' x = a(0:j-1,j) (dtpsv 'U','T','N') */
        for (int i = 0; i < j; i++) {
            double t = ap[colstart + i];
            size_t ci = (size_t)i * (i + 1) / 2;
            for (int l = 0; l < i; l++) t -= ap[ci + l] * ap[colstart + l];
            ap[colstart + i] = t / ap[ci + i];
        }
        double ajj = ap[colstart + j];
        for (int l = 0; l < j; l++) ajj -= ap[colstart + l] * ap[colstart + l];
        if (ajj <= 0.0) { ap[colstart + j] = ajj; return j + 1; }
        ap[colstart + j] = sqrt(ajj);
        jj = colstart + j;
    }
    return 0;
}

/* dpptri: inverse from the factor: dtptri('U','N') then inv(U)*inv(U)' (LAPACK dpptri.f, UPLO='U') */
int oracle_dpptri(int n, double *ap) {
    /* dtptri upper non-unit */
    for (int j = 0; j < n; j++) {
        size_t cj = (size_t)j * (j + 1) / 2;
        if (ap[cj + j] == 0.0) return j + 1;
    }
    for (int j = 0; j < n; j++) {
        size_t cj = (size_t)j * (j + 1) / 2;
        ap[cj + j] = 1.0 / ap[cj + j];
        double ajj = -ap[cj + j];
        /* x := T(0:j-1,0:j-1) * x  (dtpmv 'U','N','N'), T already inverted */
        for (int l = 0; l < j; l++) {
            double xl = ap[cj + l];
            if (xl != 0.0) {
                size_t cl = (size_t)l * (l + 1) / 2;
                for (int i = 0; i < l; i++) ap[cj + i] += xl * ap[cl + i];
                ap[cj + l] = xl * ap[cl + l];
            }
        }
        for (int i = 0; i < j; i++) ap[cj + i] *= ajj;
    }
    /* inv(U) * inv(U)' */
    for (int j = 0; j < n; j++) {
        size_t cj = (size_t)j * (j + 1) / 2;
        /* dspr('U', j, 1, ap(jc), 1, ap) */
        for (int c = 0; c < j; c++) {
            double xc = ap[cj + c];
            if (xc != 0.0) {
                size_t cc = (size_t)c * (c + 1) / 2;
                for (int i = 0; i <= c; i++) ap[cc + i] += ap[cj + i] * xc;
            }
        }
        double ajj = ap[cj + j];
        for (int i = 0; i <= j; i++) ap[cj + i] *= ajj;
    }
    return 0;
}

/* weight of a dense dispersion D (row-major m x m): P = inv(D / sigma2)  (DOPG:82-86).  P_out row-major full. */
static int dispersion_to_weight(int m, const double *D, double sigma2, double *P_out) {
    size_t len = (size_t)m * (m + 1) / 2;
    double *ap = (double *)malloc(len * sizeof(double));
    if (!ap) return -1;
    for (int c = 0; c < m; c++)
        for (int r = 0; r <= c; r++) ap[pidx(r, c)] = D[(size_t)r * m + c] * (1.0 / sigma2);
    int info = oracle_dpptrf(m, ap);
    if (!info) info = oracle_dpptri(m, ap);
    if (!info)
        for (int r = 0; r < m; r++)
            for (int c = 0; c < m; c++) P_out[(size_t)r * m + c] = pk_get(ap, r, c);
    free(ap);
    return info;
}
int oracle_dispersion_to_weight(int m, const double *D, double sigma2, double *P_out) {
    return dispersion_to_weight(m, D, sigma2, P_out);
}

/* ============================================================================================================
 * Observation groups.  mode: N != NULL accumulate N,n;  dx != NULL additionally accumulate omega (BA:472-491).
 * ============================================================================================================ */
typedef struct {
    const jaicov_problem_desc *d;
    const double *vals;
    double sigma2;
    double *N, *n;             /* N may be NULL */
    const double *dx;          /* may be NULL */
    double omega;
    double **blk_weight;       /* cached weights of image blocks [n_image_blocks] (row-major 2m x 2m) */
    double **dg_weight;        /* cached weights of direct groups with dispersion */
    int32_t *slot_col;
    uint8_t *ip_in_block;
} ora_ctx;

/* v = w - A dx ; omega += v' P v   (BA:480-488) */
static void omega_add(ora_ctx *c, int m, int k, const double *A, const int32_t *cols, const double *P,
                      const double *w, int diag) {
    double *v = (double *)malloc(sizeof(double) * m);
    for (int r = 0; r < m; r++) {
        double s = w[r];
        for (int j = 0; j < k; j++) s += -1.0 * A[(size_t)r * k + j] * c->dx[cols[j]];
        v[r] = s;
    }
    double om = 0.0;
    for (int r = 0; r < m; r++) {
        double pv = 0.0;
        if (diag) pv = P[(size_t)r * m + r] * v[r];
        else for (int q = 0; q < m; q++) pv += P[(size_t)r * m + q] * v[q];
        om += v[r] * pv;
    }
    c->omega += om;
    free(v);
}

static void group_emit(ora_ctx *c, int m, int k, const double *A, const int32_t *cols, const double *P,
                       const double *w, int diag) {
    if (c->n) stack_neq(c->N, c->n, m, k, A, cols, P, w, diag);
    if (c->dx) omega_add(c, m, k, A, cols, P, w, diag);
}

/* single ImageCoordinate group (PDF:285-445) */
static void group_image_point(ora_ctx *c, int ip) {
    double A[2][ORA_KLOC], w[2], P[4];
    int diag;
    int32_t gcol[ORA_KLOC], cols[ORA_KLOC], loc[ORA_KLOC];
    eval_image_point(c->d, c->vals, c->sigma2, ip, A, w, P, &diag);
    int kl = local_columns(c->d, ip, gcol), k = 0;
    for (int l = 0; l < kl; l++)
        if (gcol[l] >= 0) { cols[k] = gcol[l]; loc[k] = l; k++; }
    sort_pairs(k, cols, loc);
    double Ac[2 * ORA_KLOC];
    for (int r = 0; r < 2; r++)
        for (int j = 0; j < k; j++) Ac[r * k + j] = A[r][loc[j]];
    group_emit(c, 2, k, Ac, cols, P, w, diag);
}

/* image block: all image points [b,e) of one image under a joint dense dispersion; rows 2i, 2i+1 = x,y of point i.
 * Routed through the unchanged a10 contract with diagonalWeighting = false. */
static int group_image_block(ora_ctx *c, int blk) {
    const jaicov_problem_desc *d = c->d;
    int b = d->blk_ip_begin[blk], e = d->blk_ip_begin[blk + 1];
    int m = 2 * (e - b);
    if (m == 0) return 0;
    if (!c->blk_weight[blk]) {
        c->blk_weight[blk] = (double *)malloc(sizeof(double) * (size_t)m * m);
        int info = dispersion_to_weight(m, d->blk_disp + d->blk_disp_offset[blk], c->sigma2, c->blk_weight[blk]);
        if (info) return info;
    }
    /* union of columns */
    int kmax = (e - b) * 3 + ORA_KLOC, k = 0;
    int32_t *cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)kmax * 2);
    int32_t *tmp = cols + kmax;
    for (int ip = b; ip < e; ip++) {
        int32_t gcol[ORA_KLOC];
        int kl = local_columns(d, ip, gcol);
        for (int l = 0; l < kl; l++) {
            if (gcol[l] < 0) continue;
            int found = 0;
            for (int j = 0; j < k && !found; j++) found = cols[j] == gcol[l];
            if (!found) cols[k++] = gcol[l];
        }
    }
    for (int j = 0; j < k; j++) tmp[j] = j;
    sort_pairs(k, cols, tmp);
    double *A = (double *)calloc((size_t)m * k, sizeof(double));
    double *w = (double *)malloc(sizeof(double) * m);
    for (int ip = b; ip < e; ip++) {
        double Al[2][ORA_KLOC], wl[2], Pl[4];
        int diag;
        int32_t gcol[ORA_KLOC];
        eval_image_point(d, c->vals, c->sigma2, ip, Al, wl, Pl, &diag);
        int kl = local_columns(d, ip, gcol);
        for (int l = 0; l < kl; l++) {
            if (gcol[l] < 0) continue;
            int j = 0;
            while (cols[j] != gcol[l]) j++;
            A[(size_t)(2 * (ip - b)) * k + j] = Al[0][l];
            A[(size_t)(2 * (ip - b) + 1) * k + j] = Al[1][l];
        }
        w[2 * (ip - b)] = wl[0];
        w[2 * (ip - b) + 1] = wl[1];
    }
    group_emit(c, m, k, A, cols, c->blk_weight[blk], w, 0);
    free(A); free(w); free(cols);
    return 0;
}

/* PDF:210-283 getPartialDerivativeScaleBar */
static void group_scale_bar(ora_ctx *c, int s) {
    const jaicov_problem_desc *d = c->d;
    int pa = d->sb_point_a[s], pb = d->sb_point_b[s];
    const double *a = c->vals + slot_point(d, pa), *b = c->vals + slot_point(d, pb);
    double dX = b[0] - a[0], dY = b[1] - a[1], dZ = b[2] - a[2];
    double len = sqrt(dX * dX + dY * dY + dZ * dZ);
    double ax = dX / len, ay = dY / len, az = dZ / len;
    double P = c->sigma2 / d->sb_var[s];
    double w = d->sb_length[s] - len;
    double vals6[6] = {-ax, -ay, -az, +ax, +ay, +az};
    int32_t cols[6], loc[6];
    int k = 0;
    for (int t = 0; t < 6; t++) {
        int32_t col = t < 3 ? d->point_col[3 * pa + t] : d->point_col[3 * pb + t - 3];
        if (col >= 0) { cols[k] = col; loc[k] = t; k++; }
    }
    sort_pairs(k, cols, loc);
    double A[6];
    for (int j = 0; j < k; j++) A[j] = vals6[loc[j]];
    group_emit(c, 1, k, A, cols, &P, &w, 1);
}

/* PDF:447-473 getPartialDerivativeDirectlyObservedParameters */
static int group_direct(ora_ctx *c, int g) {
    const jaicov_problem_desc *d = c->d;
    int b = d->dg_row_begin[g], e = d->dg_row_begin[g + 1], m = e - b;
    if (m == 0) return 0;
    int dense = d->dg_disp_offset && d->dg_disp_offset[g] >= 0;
    double *P;
    if (dense) {
        if (!c->dg_weight[g]) {
            c->dg_weight[g] = (double *)malloc(sizeof(double) * (size_t)m * m);
            int info = dispersion_to_weight(m, d->dg_disp + d->dg_disp_offset[g], c->sigma2, c->dg_weight[g]);
            if (info) return info;
        }
        P = c->dg_weight[g];
    } else {
        P = (double *)calloc((size_t)m * m, sizeof(double));
        for (int r = 0; r < m; r++) P[(size_t)r * m + r] = c->sigma2 / d->dg_var[b + r];
    }
    int32_t *cols = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)m), *rowof = cols + m;
    double *w = (double *)malloc(sizeof(double) * m);
    int k = 0;
    for (int r = 0; r < m; r++) {
        int slot = d->dg_slot[b + r];
        int col = c->slot_col[slot];
        if (col >= 0) { cols[k] = col; rowof[k] = r; k++; }
        w[r] = d->dg_obs[b + r] - c->vals[slot];
    }
    sort_pairs(k, cols, rowof);
    double *A = (double *)calloc((size_t)m * (k ? k : 1), sizeof(double));
    for (int j = 0; j < k; j++) A[(size_t)rowof[j] * k + j] = 1.0;
    group_emit(c, m, k, A, cols, P, w, !dense);
    free(A); free(w); free(cols);
    if (!dense) free(P);
    return 0;
}

static int ctx_init(ora_ctx *c, const jaicov_problem_desc *d, const double *vals, double sigma2) {
    memset(c, 0, sizeof(*c));
    c->d = d; c->vals = vals; c->sigma2 = sigma2;
    c->blk_weight = (double **)calloc((size_t)d->n_image_blocks + 1, sizeof(double *));
    c->dg_weight = (double **)calloc((size_t)d->n_direct_groups + 1, sizeof(double *));
    c->slot_col = (int32_t *)malloc(sizeof(int32_t) * (size_t)oracle_num_slots(d));
    oracle_slot_columns(d, c->slot_col);
    c->ip_in_block = (uint8_t *)calloc((size_t)d->n_image_points + 1, 1);
    for (int b = 0; b < d->n_image_blocks; b++)
        for (int ip = d->blk_ip_begin[b]; ip < d->blk_ip_begin[b + 1]; ip++) c->ip_in_block[ip] = 1;
    return 0;
}
static void ctx_free(ora_ctx *c) {
    for (int b = 0; b < c->d->n_image_blocks; b++) free(c->blk_weight[b]);
    for (int g = 0; g < c->d->n_direct_groups; g++) free(c->dg_weight[g]);
    free(c->blk_weight); free(c->dg_weight); free(c->slot_col); free(c->ip_in_block);
}

/* all groups in LinkedHashSet order (BA:670-771): image points (image-major), scale bars, direct groups.
 * An image block takes the place of its first image point.  image range [ib,ie) restricts image points (sharding);
 * shared != 0 includes scale bars and direct groups. */
static int sweep_groups(ora_ctx *c, int ib, int ie, int shared) {
    const jaicov_problem_desc *d = c->d;
    int blk = 0;
    for (int ip = 0; ip < d->n_image_points; ip++) {
        int img = d->ip_image[ip];
        if (c->ip_in_block[ip]) {
            while (blk < d->n_image_blocks && d->blk_ip_begin[blk + 1] <= ip) blk++;
            if (d->blk_ip_begin[blk] == ip && img >= ib && img < ie) {
                int info = group_image_block(c, blk);
                if (info) return info;
            }
            continue;
        }
        if (img < ib || img >= ie) continue;
        group_image_point(c, ip);
    }
    if (shared) {
        for (int s = 0; s < d->n_scale_bars; s++) group_scale_bar(c, s);
        for (int g = 0; g < d->n_direct_groups; g++) {
            int info = group_direct(c, g);
            if (info) return info;
        }
    }
    return 0;
}

/* ============================================================================================================
 * a12: addDatumConditionRows (BA:493-635)
 * ============================================================================================================ */
static int datum_rows(const jaicov_problem_desc *d, const double *vals, double *N) {
    int defect = d->rank_defect;
    if (defect == 0) return 0;
    double x0 = 0, y0 = 0, z0 = 0;
    int count = 0;
    for (int p = 0; p < d->n_points; p++) {
        const int32_t *col = d->point_col + 3 * p;
        if (!d->point_datum[p] || col[0] < 0 || col[1] < 0 || col[2] < 0) continue;
        const double *v = vals + slot_point(d, p);
        x0 += v[0]; y0 += v[1]; z0 += v[2];
        count++;
    }
    if (count < 3) return JAICOV_ERR_BAD_ARGUMENT;          /* BA:515-516 IllegalArgumentException */
    x0 = x0 / (double)count; y0 = y0 / (double)count; z0 = z0 / (double)count;
    int row = 0;
    int f = d->datum_flags;
    int tx = (f & JAICOV_DATUM_TX) ? row++ : -1;
    int ty = (f & JAICOV_DATUM_TY) ? row++ : -1;
    int tz = (f & JAICOV_DATUM_TZ) ? row++ : -1;
    int rx = (f & JAICOV_DATUM_RX) ? row++ : -1;
    int ry = (f & JAICOV_DATUM_RY) ? row++ : -1;
    int rz = (f & JAICOV_DATUM_RZ) ? row++ : -1;
    int ms = (f & JAICOV_DATUM_SCALE) ? row++ : -1;
    double norm[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < d->n_points; p++) {
        const int32_t *col = d->point_col + 3 * p;
        if (!d->point_datum[p] || col[0] < 0 || col[1] < 0 || col[2] < 0) continue;
        const double *v = vals + slot_point(d, p);
        double x = v[0] - x0, y = v[1] - y0, z = v[2] - z0;
        if (tx >= 0) { pk_set(N, tx, col[0], 1.0); norm[tx] += 1.0; }
        if (ty >= 0) { pk_set(N, ty, col[1], 1.0); norm[ty] += 1.0; }
        if (tz >= 0) { pk_set(N, tz, col[2], 1.0); norm[tz] += 1.0; }
        if (rx >= 0) { pk_set(N, rx, col[1], z); pk_set(N, rx, col[2], -y); norm[rx] += z * z + y * y; }
        if (ry >= 0) { pk_set(N, ry, col[0], -z); pk_set(N, ry, col[2], x); norm[ry] += z * z + x * x; }
        if (rz >= 0) { pk_set(N, rz, col[0], y); pk_set(N, rz, col[1], -x); norm[rz] += x * x + y * y; }
        if (ms >= 0) {
            pk_set(N, ms, col[0], x); pk_set(N, ms, col[1], y); pk_set(N, ms, col[2], z);
            norm[ms] += x * x + y * y + z * z;
        }
    }
    for (int p = 0; p < d->n_points; p++) {
        const int32_t *col = d->point_col + 3 * p;
        if (!d->point_datum[p] || col[0] < 0 || col[1] < 0 || col[2] < 0) continue;
        if (tx >= 0) pk_set(N, tx, col[0], pk_get(N, tx, col[0]) / sqrt(norm[tx]));
        if (ty >= 0) pk_set(N, ty, col[1], pk_get(N, ty, col[1]) / sqrt(norm[ty]));
        if (tz >= 0) pk_set(N, tz, col[2], pk_get(N, tz, col[2]) / sqrt(norm[tz]));
        if (rx >= 0) {
            pk_set(N, rx, col[1], pk_get(N, rx, col[1]) / sqrt(norm[rx]));
            pk_set(N, rx, col[2], pk_get(N, rx, col[2]) / sqrt(norm[rx]));
        }
        if (ry >= 0) {
            pk_set(N, ry, col[0], pk_get(N, ry, col[0]) / sqrt(norm[ry]));
            pk_set(N, ry, col[2], pk_get(N, ry, col[2]) / sqrt(norm[ry]));
        }
        if (rz >= 0) {
            pk_set(N, rz, col[0], pk_get(N, rz, col[0]) / sqrt(norm[rz]));
            pk_set(N, rz, col[1], pk_get(N, rz, col[1]) / sqrt(norm[rz]));
        }
        if (ms >= 0) {
            pk_set(N, ms, col[0], pk_get(N, ms, col[0]) / sqrt(norm[ms]));
            pk_set(N, ms, col[1], pk_get(N, ms, col[1]) / sqrt(norm[ms]));
            pk_set(N, ms, col[2], pk_get(N, ms, col[2]) / sqrt(norm[ms]));
        }
    }
    return 0;
}

/* ============================================================================================================
 * a11: createNormalEquation (BA:789-834).  N packed U(U+1)/2, n[U], V[U] (diagonal preconditioner).
 * partial != 0: only the group sums of images [ib,ie) (+ shared groups if shared), no datum/damping/V
 * (what one rank contributes before the all-reduce).
 * ============================================================================================================ */
int oracle_accumulate(const jaicov_problem_desc *d, const double *vals, double sigma2, int ib, int ie, int shared,
                      double *N, double *n) {
    int U = d->n_unknowns;
    memset(N, 0, sizeof(double) * ((size_t)U * (U + 1) / 2));
    memset(n, 0, sizeof(double) * U);
    ora_ctx c;
    ctx_init(&c, d, vals, sigma2);
    c.N = N; c.n = n;
    int info = sweep_groups(&c, ib, ie, shared);
    ctx_free(&c);
    return info;
}

int oracle_finalize(const jaicov_problem_desc *d, const double *vals, double lambda, int simulation, double *N,
                    double *n, double *V) {
    int U = d->n_unknowns;
    int info = datum_rows(d, vals, N);
    if (info) return info;
    /* BA:814-822 */
    if (lambda > 0) {
        int ns = oracle_num_slots(d);
        int32_t *sc = (int32_t *)malloc(sizeof(int32_t) * ns);
        oracle_slot_columns(d, sc);
        for (int s = 0; s < ns; s++) {
            int col = sc[s];
            if (col < 0) continue;
            pk_add(N, col, col, lambda * pk_get(N, col, col));
        }
        free(sc);
    }
    /* BA:825-828 */
    double EPS = oracle_eps();
    for (int c = 0; c < U; c++) {
        double v = pk_get(N, c, c);
        V[c] = v > EPS ? 1.0 / sqrt(v) : 1.0;
    }
    if (simulation) memset(n, 0, sizeof(double) * U);       /* BA:830-831 */
    return 0;
}

int oracle_build(const jaicov_problem_desc *d, const double *vals, double sigma2, double lambda, int simulation,
                 double *N, double *n, double *V) {
    int info = oracle_accumulate(d, vals, sigma2, 0, d->n_images, 1, N, n);
    if (info) return info;
    return oracle_finalize(d, vals, lambda, simulation, N, n, V);
}

/* a13: NES.applyPrecondition (NES:82-91): m <- V m ; M[r,c] <- V_c * M[r,c] * V_r  */
void oracle_precondition(int U, const double *V, double *M, double *m) {
    for (int row = 0; row < U; row++) {
        if (m) m[row] = V[row] * m[row];
        if (M)
            for (int col = row; col < U; col++) M[pidx(row, col)] = V[col] * M[pidx(row, col)] * V[row];
    }
}

/* ============================================================================================================
 * a14: MX.solve(UpperSymmPackMatrix, DenseVector, numRows, invert) (MX:338-366) = dspsv (+ dsptri)
 * LAPACK reference algorithms, UPLO='U', packed storage, restated with 1-based index arithmetic kept in comments'
 * notation; ap is 0-based here: AP(i) == ap[i-1].
 * ============================================================================================================ */
#define AP(i) ap[(i) - 1]

static int idamax1(int n, const double *x) {   /* 1-based index of first max |x| */
    if (n < 1) return 0;
    int im = 1;
    double dm = fabs(x[0]);
    for (int i = 2; i <= n; i++)
        if (fabs(x[i - 1]) > dm) { im = i; dm = fabs(x[i - 1]); }
    return im;
}

/* dsptrf('U'): Bunch-Kaufman diagonal pivoting, A = U D U'.  ipiv 1-based semantic, stored in ipiv[k-1]. */
int oracle_dsptrf(int n, double *ap, int *ipiv) {
    const double alpha = (1.0 + sqrt(17.0)) / 8.0;
    int info = 0;
    long k = n;
    long kc = (long)(n - 1) * n / 2 + 1;
    while (k >= 1) {
        long knc = kc;
        int kstep = 1;
        long kp, kpc = 0, imax = 0, jmax;
        double absakk = fabs(AP(kc + k - 1));
        double colmax = 0.0;
        if (k > 1) {
            imax = idamax1((int)(k - 1), &AP(kc));
            colmax = fabs(AP(kc + imax - 1));
        }
        if ((absakk > colmax ? absakk : colmax) == 0.0) {
            if (info == 0) info = (int)k;
            kp = k;
        } else {
            if (absakk >= alpha * colmax) {
                kp = k;
            } else {
                double rowmax = 0.0;
                jmax = imax;
                long kx = imax * (imax + 1) / 2 + imax;
                for (long j = imax + 1; j <= k; j++) {
                    if (fabs(AP(kx)) > rowmax) { rowmax = fabs(AP(kx)); jmax = j; }
                    kx += j;
                }
                kpc = (imax - 1) * imax / 2 + 1;
                if (imax > 1) {
                    jmax = idamax1((int)(imax - 1), &AP(kpc));
                    double t = fabs(AP(kpc + jmax - 1));
                    if (t > rowmax) rowmax = t;
                }
                if (absakk >= alpha * colmax * (colmax / rowmax)) {
                    kp = k;
                } else if (fabs(AP(kpc + imax - 1)) >= alpha * rowmax) {
                    kp = imax;
                } else {
                    kp = imax;
                    kstep = 2;
                }
            }
            long kk = k - kstep + 1;
            if (kstep == 2) knc = knc - k + 1;
            if (kp != kk) {
                kpc = (kp - 1) * kp / 2 + 1;
                for (long i = 0; i < kp - 1; i++) { double t = AP(knc + i); AP(knc + i) = AP(kpc + i); AP(kpc + i) = t; }
                long kx = kpc + kp - 1;
                for (long j = kp + 1; j <= kk - 1; j++) {
                    kx = kx + j - 1;
                    double t = AP(knc + j - 1); AP(knc + j - 1) = AP(kx); AP(kx) = t;
                }
                double t = AP(knc + kk - 1); AP(knc + kk - 1) = AP(kpc + kp - 1); AP(kpc + kp - 1) = t;
                if (kstep == 2) { t = AP(kc + k - 2); AP(kc + k - 2) = AP(kc + kp - 1); AP(kc + kp - 1) = t; }
            }
            if (kstep == 1) {
                /* W(k) = U(k) D(k); A := A - W(k) 1/D(k) W(k)' ; dspr('U', k-1, -r1, AP(kc), 1, AP) */
                double r1 = 1.0 / AP(kc + k - 1);
                long kkx = 1;
                for (long j = 1; j <= k - 1; j++) {
                    double xj = AP(kc + j - 1);
                    if (xj != 0.0) {
                        double temp = -r1 * xj;
                        long kq = kkx;
                        for (long i = 1; i <= j; i++) { AP(kq) += AP(kc + i - 1) * temp; kq++; }
                    }
                    kkx += j;
                }
                for (long i = 0; i < k - 1; i++) AP(kc + i) *= r1;
            } else {
                if (k > 2) {
                    double d12 = AP(k - 1 + (k - 1) * k / 2);
                    double d22 = AP(k - 1 + (k - 2) * (k - 1) / 2) / d12;
                    double d11 = AP(k + (k - 1) * k / 2) / d12;
                    double t = 1.0 / (d11 * d22 - 1.0);
                    d12 = t / d12;
                    for (long j = k - 2; j >= 1; j--) {
                        double wkm1 = d12 * (d11 * AP(j + (k - 2) * (k - 1) / 2) - AP(j + (k - 1) * k / 2));
                        double wk = d12 * (d22 * AP(j + (k - 1) * k / 2) - AP(j + (k - 2) * (k - 1) / 2));
                        for (long i = j; i >= 1; i--)
                            AP(i + (j - 1) * j / 2) = AP(i + (j - 1) * j / 2) - AP(i + (k - 1) * k / 2) * wk -
                                                      AP(i + (k - 2) * (k - 1) / 2) * wkm1;
                        AP(j + (k - 1) * k / 2) = wk;
                        AP(j + (k - 2) * (k - 1) / 2) = wkm1;
                    }
                }
            }
        }
        if (kstep == 1) ipiv[k - 1] = (int)kp;
        else { ipiv[k - 1] = -(int)kp; ipiv[k - 2] = -(int)kp; }
        k -= kstep;
        kc = knc - k;
    }
    return info;
}

/* dsptrs('U', n, nrhs = 1): solve A x = b with the factorisation of dsptrf */
void oracle_dsptrs(int n, const double *ap, const int *ipiv, double *b) {
#define B(i) b[(i) - 1]
    long k = n;
    long kc = (long)n * (n + 1) / 2 + 1;
    while (k >= 1) {
        kc -= k;
        if (ipiv[k - 1] > 0) {
            long kp = ipiv[k - 1];
            if (kp != k) { double t = B(k); B(k) = B(kp); B(kp) = t; }
            for (long i = 1; i <= k - 1; i++) B(i) += -1.0 * AP(kc + i - 1) * B(k);     /* dger */
            B(k) *= 1.0 / AP(kc + k - 1);                                                 /* dscal */
            k -= 1;
        } else {
            long kp = -ipiv[k - 1];
            if (kp != k - 1) { double t = B(k - 1); B(k - 1) = B(kp); B(kp) = t; }
            for (long i = 1; i <= k - 2; i++) B(i) += -1.0 * AP(kc + i - 1) * B(k);
            for (long i = 1; i <= k - 2; i++) B(i) += -1.0 * AP(kc - (k - 1) + i - 1) * B(k - 1);
            double akm1k = AP(kc + k - 2);
            double akm1 = AP(kc - 1) / akm1k;
            double ak = AP(kc + k - 1) / akm1k;
            double denom = akm1 * ak - 1.0;
            double bkm1 = B(k - 1) / akm1k;
            double bk = B(k) / akm1k;
            B(k - 1) = (ak * bkm1 - bk) / denom;
            B(k) = (akm1 * bk - bkm1) / denom;
            kc = kc - k + 1;
            k -= 2;
        }
    }
    k = 1;
    kc = 1;
    while (k <= n) {
        if (ipiv[k - 1] > 0) {
            double s = 0.0;
            for (long i = 1; i <= k - 1; i++) s += AP(kc + i - 1) * B(i);                 /* dgemv 'T' */
            B(k) += -1.0 * s;
            long kp = ipiv[k - 1];
            if (kp != k) { double t = B(k); B(k) = B(kp); B(kp) = t; }
            kc += k;
            k += 1;
        } else {
            double s = 0.0, s2 = 0.0;
            for (long i = 1; i <= k - 1; i++) s += AP(kc + i - 1) * B(i);
            B(k) += -1.0 * s;
            for (long i = 1; i <= k - 1; i++) s2 += AP(kc + k + i - 1) * B(i);
            B(k + 1) += -1.0 * s2;
            long kp = -ipiv[k - 1];
            if (kp != k) { double t = B(k); B(k) = B(kp); B(kp) = t; }
            kc += 2 * k + 1;
            k += 2;
        }
    }
#undef B
}

/* y := alpha*A*x + beta*y for packed upper symmetric A of order n (dspmv 'U', beta = 0, alpha = -1 used below) */
static void dspmv_u(int n, double alpha, const double *ap, const double *x, double *y) {
    for (int i = 0; i < n; i++) y[i] = 0.0;
    long kk = 1;
    for (int j = 1; j <= n; j++) {
        double temp1 = alpha * x[j - 1], temp2 = 0.0;
        long k = kk;
        for (int i = 1; i <= j - 1; i++) {
            y[i - 1] += temp1 * AP(k);
            temp2 += AP(k) * x[i - 1];
            k++;
        }
        y[j - 1] += temp1 * AP(kk + j - 1) + alpha * temp2;
        kk += j;
    }
}

/* dsptri('U'): inverse from the factorisation of dsptrf */
int oracle_dsptri(int n, double *ap, const int *ipiv, double *work) {
    long kp = (long)n * (n + 1) / 2;
    for (int info = n; info >= 1; info--) {
        if (ipiv[info - 1] > 0 && AP(kp) == 0.0) return info;
        kp -= info;
    }
    long k = 1, kc = 1;
    while (k <= n) {
        long kcnext = kc + k;
        int kstep;
        if (ipiv[k - 1] > 0) {
            AP(kc + k - 1) = 1.0 / AP(kc + k - 1);
            if (k > 1) {
                memcpy(work, &AP(kc), sizeof(double) * (k - 1));
                dspmv_u((int)(k - 1), -1.0, ap, work, &AP(kc));
                double dot = 0.0;
                for (long i = 0; i < k - 1; i++) dot += work[i] * AP(kc + i);
                AP(kc + k - 1) -= dot;
            }
            kstep = 1;
        } else {
            double t = fabs(AP(kcnext + k - 1));
            double ak = AP(kc + k - 1) / t;
            double akp1 = AP(kcnext + k) / t;
            double akkp1 = AP(kcnext + k - 1) / t;
            double dd = t * (ak * akp1 - 1.0);
            AP(kc + k - 1) = akp1 / dd;
            AP(kcnext + k) = ak / dd;
            AP(kcnext + k - 1) = -akkp1 / dd;
            if (k > 1) {
                memcpy(work, &AP(kc), sizeof(double) * (k - 1));
                dspmv_u((int)(k - 1), -1.0, ap, work, &AP(kc));
                double dot = 0.0;
                for (long i = 0; i < k - 1; i++) dot += work[i] * AP(kc + i);
                AP(kc + k - 1) -= dot;
                dot = 0.0;
                for (long i = 0; i < k - 1; i++) dot += AP(kc + i) * AP(kcnext + i);
                AP(kcnext + k - 1) -= dot;
                memcpy(work, &AP(kcnext), sizeof(double) * (k - 1));
                dspmv_u((int)(k - 1), -1.0, ap, work, &AP(kcnext));
                dot = 0.0;
                for (long i = 0; i < k - 1; i++) dot += work[i] * AP(kcnext + i);
                AP(kcnext + k) -= dot;
            }
            kstep = 2;
            kcnext = kcnext + k + 1;
        }
        long kpv = ipiv[k - 1] < 0 ? -ipiv[k - 1] : ipiv[k - 1];
        if (kpv != k) {
            long kpc = (kpv - 1) * kpv / 2 + 1;
            for (long i = 0; i < kpv - 1; i++) { double t = AP(kc + i); AP(kc + i) = AP(kpc + i); AP(kpc + i) = t; }
            long kx = kpc + kpv - 1;
            for (long j = kpv + 1; j <= k - 1; j++) {
                kx = kx + j - 1;
                double t = AP(kc + j - 1); AP(kc + j - 1) = AP(kx); AP(kx) = t;
            }
            double t = AP(kc + k - 1); AP(kc + k - 1) = AP(kpc + kpv - 1); AP(kpc + kpv - 1) = t;
            if (kstep == 2) { t = AP(kc + k + k - 1); AP(kc + k + k - 1) = AP(kc + k + kpv - 1); AP(kc + k + kpv - 1) = t; }
        }
        k += kstep;
        kc = kcnext;
    }
    return 0;
}
#undef AP

/* r = b - A x for symmetric packed 'U' A, accumulated in long double (x87 extended: 64-bit mantissa) and rounded once:
 * the residual of extended-precision iterative refinement.  NOT part of the reference's algorithm: tests/golden/make_cfg4_truth.py
 * uses it to obtain the exact solution of the oracle's own system, against which the oracle's (dspsv) and the GPU's
 * (Cholesky) forward errors are both measured at cond(N) ~ 1e9. */
void oracle_residual_ld(int n, const double *ap, const double *x, const double *b, double *r) {
    long double *acc = (long double *)malloc(sizeof(long double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) acc[i] = (long double)b[i];
    size_t kk = 0;
    for (int j = 0; j < n; j++) {
        long double xj = (long double)x[j], t = 0.0L;
        for (int i = 0; i < j; i++) {
            long double a = (long double)ap[kk + i];
            acc[i] -= a * xj;
            t += a * (long double)x[i];
        }
        acc[j] -= t + (long double)ap[kk + j] * xj;
        kk += (size_t)j + 1;
    }
    for (int i = 0; i < n; i++) r[i] = (double)acc[i];
    free(acc);
}

/* MX:338-366: returns 0, >0 singular (MatrixSingularException), <0 illegal argument */
int oracle_solve(int U, double *N, double *n, int invert) {
    int *ipiv = (int *)malloc(sizeof(int) * (U > 0 ? U : 1));
    int info = oracle_dsptrf(U, N, ipiv);
    if (info == 0) oracle_dsptrs(U, N, ipiv, n);
    if (info == 0 && invert) {
        double *work = (double *)malloc(sizeof(double) * (U > 0 ? U : 1));
        info = oracle_dsptri(U, N, ipiv, work);
        free(work);
    }
    free(ipiv);
    return info;
}

/* a17: getOmega (BA:472-491) at the current (pre-update) parameters */
int oracle_omega(const jaicov_problem_desc *d, const double *vals, double sigma2, const double *dx, double *omega) {
    ora_ctx c;
    ctx_init(&c, d, vals, sigma2);
    c.dx = dx;
    int info = sweep_groups(&c, 0, d->n_images, 1);
    *omega = c.omega;
    ctx_free(&c);
    return info;
}

/* a17: updateUnknownParameters (BA:450-462) */
double oracle_update(const jaicov_problem_desc *d, double *vals, const double *dx) {
    int ns = oracle_num_slots(d);
    int32_t *sc = (int32_t *)malloc(sizeof(int32_t) * ns);
    oracle_slot_columns(d, sc);
    double maxAbsDx = 0;
    for (int s = 0; s < ns; s++) {
        int col = sc[s];
        if (col >= 0) {
            double dv = dx[col];
            maxAbsDx = fabs(dv) > maxAbsDx ? fabs(dv) : maxAbsDx;
            vals[s] = vals[s] + dv;
        }
    }
    free(sc);
    return maxAbsDx;
}

/* ============================================================================================================
 * a18: centroidCoordinates (BA:115-201).  "Unknown parameters" are the parameters registered by addUnknownParameter
 * (BA:645-650), i.e. those that own a column.  The mean of all unknown X (object AND camera), Y, Z is subtracted
 * (invert == 0; needs equal, non-zero counts: BA:142-151) or added back (invert != 0) to those parameters and to the
 * OBSERVED values of every directly observed coordinate (BA:179-200, whatever the state of the referenced parameter).
 * centroid[3] is output for invert == 0 and input otherwise.  Returns 0, or JAICOV_ERR_UNSUPPORTED for the
 * UnsupportedOperationException of BA:151.
 * ============================================================================================================ */
static int slot_axis(const jaicov_problem_desc *d, int slot) {
    if (slot < 3 * d->n_points) return slot % 3;                        /* OBJECT_COORDINATE_X/Y/Z */
    int e0 = slot_eo(d, 0);
    if (slot >= e0) { int k = (slot - e0) % 6; return k < 3 ? k : -1; }   /* CAMERA_COORDINATE_X/Y/Z */
    return -1;
}
int oracle_centroid(const jaicov_problem_desc *d, double *vals, double *dg_obs, int invert, double centroid[3]) {
    int ns = oracle_num_slots(d);
    int32_t *sc = (int32_t *)malloc(sizeof(int32_t) * ns);
    oracle_slot_columns(d, sc);
    if (!invert) {
        double x0 = 0, y0 = 0, z0 = 0;
        int cntX = 0, cntY = 0, cntZ = 0;
        /* unknownParameters is a LinkedHashSet in registration order (BA:667-782): points in first-seen order, ... the sum is
         * taken in slot order here; the order only moves the last bits of the mean */
        for (int s = 0; s < ns; s++) {
            if (sc[s] < 0) continue;
            switch (slot_axis(d, s)) {
                case 0: x0 += vals[s]; cntX++; break;
                case 1: y0 += vals[s]; cntY++; break;
                case 2: z0 += vals[s]; cntZ++; break;
                default: break;
            }
        }
        if (cntX == cntY && cntX == cntZ && cntY == cntZ && cntX > 0) {
            centroid[0] = x0 / cntX; centroid[1] = y0 / cntY; centroid[2] = z0 / cntZ;
        } else { free(sc); return JAICOV_ERR_UNSUPPORTED; }
    }
    double sign = invert ? 1.0 : -1.0;
    double c[3] = {sign * centroid[0], sign * centroid[1], sign * centroid[2]};
    for (int s = 0; s < ns; s++) {
        int a = slot_axis(d, s);
        if (sc[s] >= 0 && a >= 0) vals[s] = vals[s] + c[a];
    }
    for (int r = 0; r < d->n_direct_rows; r++) {
        int a = slot_axis(d, d->dg_slot[r]);
        if (a >= 0) dg_obs[r] = dg_obs[r] + c[a];
    }
    free(sc);
    return 0;
}

/* ============================================================================================================
 * f1: MatrixInversion.REDUCED / PRE_ELIMINATION.  reduceNormalEquationSystem (BA:1197-1342) image by image, literally:
 * N22 = the image's EO block, inverted with MX.inv (dpptrf + dpptri); for every parameter of the image's interior
 * orientation / distortion and of the object points it observes: n1 -= N12 inv(N22) n2, N11 -= N12 inv(N22) N21 (upper
 * part through pk_add: MTJ's add is a no-op below the diagonal).  pre_elimination != 0 additionally parks inv(N22)
 * and inv(N22) n2 in the EO rows of N and n for extractReducedParameters (BA:1258-1295).  N, n are the PRECONDITIONED
 * system (BA:238 runs first).  Returns 0 or the LAPACK info of a failed MX.inv.
 * ============================================================================================================ */
static int image_unknowns(const jaicov_problem_desc *d, int img, int32_t *out) {
    /* unknownInteriorOrientationAndDistortionParameters, then X, Y, Z of the object point of every image coordinate of the
     * image in the image's order (BA:1297-1308); a point observed twice by one image appears twice, as in the reference */
    int cam = d->image_camera[img], k = 0;
    for (int t = 0; t < 3; t++)
        if (d->io_col[3 * cam + t] >= 0) out[k++] = d->io_col[3 * cam + t];
    for (int j = d->cam_dist_begin[cam]; j < d->cam_dist_begin[cam + 1]; j++)
        if (d->dist_col[j] >= 0) out[k++] = d->dist_col[j];
    for (int ip = 0; ip < d->n_image_points; ip++) {
        if (d->ip_image[ip] != img) continue;
        for (int t = 0; t < 3; t++) {
            int c = d->point_col[3 * d->ip_point[ip] + t];
            if (c >= 0) out[k++] = c;
        }
    }
    return k;
}
int oracle_reduce(const jaicov_problem_desc *d, double *N, double *n, int pre_elimination) {
    int32_t *up = (int32_t *)malloc(sizeof(int32_t) * (size_t)(3 + d->n_dist + 3 * d->n_image_points + 8));
    for (int cam = 0; cam < d->n_cameras; cam++)                       /* BA:1198: cameras, then the camera's images */
        for (int img = 0; img < d->n_images; img++) {
            if (d->image_camera[img] != cam) continue;
            int eo[6], m = 0;
            for (int t = 0; t < 6; t++)
                if (d->eo_col[6 * img + t] >= 0) eo[m++] = d->eo_col[6 * img + t];
            double N22[21], n2[6];
            for (int r = 0; r < m; r++) {
                for (int c = r; c < m; c++) N22[pidx(r, c)] = pk_get(N, eo[r], eo[c]);
                n2[r] = n[eo[r]];
                if (pre_elimination) n[eo[r]] = 0;
            }
            int info = oracle_dpptrf(m, N22);
            if (!info) info = oracle_dpptri(m, N22);
            if (info) { free(up); return info; }
            if (pre_elimination)
                for (int r = 0; r < m; r++) {
                    double nr = n2[r], dv = N22[pidx(r, r)];
                    pk_set(N, eo[r], eo[r], dv);
                    n[eo[r]] += dv * nr;
                    for (int c = r + 1; c < m; c++) {
                        double nc = n2[c], ov = N22[pidx(r, c)];
                        pk_set(N, eo[r], eo[c], ov);
                        n[eo[r]] += ov * nc;
                        n[eo[c]] += ov * nr;
                    }
                }
            int k = image_unknowns(d, img, up);
            for (int a = 0; a < k; a++) {
                int rowN = up[a];
                double n12[6];
                for (int c = 0; c < m; c++) {
                    double dot = 0;
                    for (int r = 0; r < m; r++) dot += pk_get(N, rowN, eo[r]) * (r <= c ? N22[pidx(r, c)] : N22[pidx(c, r)]);
                    n12[c] = dot;
                }
                double nd = 0;
                for (int c = 0; c < m; c++) nd += n12[c] * n2[c];
                n[rowN] += -nd;
                for (int b = 0; b < k; b++) {
                    int colN = up[b];
                    double dot = 0;
                    for (int r = 0; r < m; r++) dot += n12[r] * pk_get(N, colN, eo[r]);
                    pk_add(N, rowN, colN, -dot);
                }
            }
        }
    free(up);
    return 0;
}

/* extractReducedParameters (BA:1344-1453): dx2 = inv(N22) n2 - inv(N22) N21 dx1 from what oracle_reduce parked */
void oracle_extract_reduced(const jaicov_problem_desc *d, const double *N, double *n) {
    int32_t *up = (int32_t *)malloc(sizeof(int32_t) * (size_t)(3 + d->n_dist + 3 * d->n_image_points + 8));
    for (int cam = 0; cam < d->n_cameras; cam++)
        for (int img = 0; img < d->n_images; img++) {
            if (d->image_camera[img] != cam) continue;
            int eo[6], m = 0;
            for (int t = 0; t < 6; t++)
                if (d->eo_col[6 * img + t] >= 0) eo[m++] = d->eo_col[6 * img + t];
            double inv[21], dx2[6];
            for (int r = 0; r < m; r++) {
                for (int c = r; c < m; c++) inv[pidx(r, c)] = pk_get(N, eo[r], eo[c]);
                dx2[r] = n[eo[r]];
            }
            int k = image_unknowns(d, img, up);
            for (int a = 0; a < k; a++) {
                int rowN = up[a];
                double n21[6] = {0, 0, 0, 0, 0, 0};
                for (int r = 0; r < m; r++) {
                    double vr = pk_get(N, rowN, eo[r]);
                    n21[r] += inv[pidx(r, r)] * vr;
                    for (int c = r + 1; c < m; c++) {
                        double vc = pk_get(N, rowN, eo[c]), irc = inv[pidx(r, c)];
                        n21[r] += irc * vc;
                        n21[c] += irc * vr;
                    }
                }
                for (int r = 0; r < m; r++) dx2[r] += -n[rowN] * n21[r];
            }
            for (int r = 0; r < m; r++) n[eo[r]] = dx2[r];
        }
    free(up);
}

/* numRows of BA:262 / BA:284 */
int oracle_reduced_rows(const jaicov_problem_desc *d) {
    int k = d->rank_defect + 3 * d->n_points;      /* objectCoordinates.size() * 3: ALL object points (quirk Q5) */
    for (int i = 0; i < 3 * d->n_cameras; i++) k += d->io_col[i] >= 0;
    for (int i = 0; i < d->n_dist; i++) k += d->dist_col[i] >= 0;
    return k;
}

/* ============================================================================================================
 * a16/a17: estimateModel loop (BA:203-387) + updateModel (BA:389-442).  invert = MatrixInversion (BA:65-70):
 * 0 NONE, 1 FULL, 2 REDUCED, 3 PRE_ELIMINATION (BA:261-271, 283-294) -- literally, including what REDUCED leaves in the
 * exterior-orientation entries of dx in the last pass.
 * Centroiding (BA:115-201) is oracle_centroid, called by the caller around this loop as BA:224-225 / 357-358 do; vals are
 * updated in place.
 * Q_out (packed, may be NULL) receives Qxx when invert != 0.
 * ============================================================================================================ */
typedef struct {
    int32_t state, iterations;
    double omega, max_abs_dx, final_lambda, seconds_total, seconds_last_pass;
} oracle_result;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int oracle_estimate(const jaicov_problem_desc *d, double *vals, double sigma2apriori, double lambda0, int max_iter,
                    int invert, int simulation, double *Q_out, oracle_result *res) {
    const double SQRT_EPS = sqrt(oracle_eps());
    int U = d->n_unknowns;
    size_t plen = (size_t)U * (U + 1) / 2;
    double *N = (double *)malloc(sizeof(double) * (plen ? plen : 1));
    double *n = (double *)malloc(sizeof(double) * (U ? U : 1));
    double *V = (double *)malloc(sizeof(double) * (U ? U : 1));
    if (!N || !n || !V) { free(N); free(n); free(V); res->state = -7; return 0; }
    double t0 = now_s();
    int deriveFirst = lambda0 > 0;                     /* BA:207 */
    double adapted = 0, damping = fabs(lambda0);
    double maxAbsDx = 0.0, lastValid = 0.0, omega = 0.0;
    int runs = max_iter - 1;
    int isEstimated = 0, complete = 0, isConverge = 1;
    if (max_iter == 0) { complete = isEstimated = 1; adapted = 0; }
    double sigma2 = sigma2apriori > 0 ? sigma2apriori : 1.0;   /* BA:221 */
    int state = 0, iter = 0;
    do {
        double tp = now_s();
        maxAbsDx = 0.0;
        iter = max_iter - runs;
        /* BA:235 createNormalEquation (BA:801-812: first damping value) */
        if (deriveFirst) { adapted = damping; deriveFirst = 0; }
        int info = oracle_build(d, vals, sigma2, adapted, simulation, N, n, V);
        if (info) { state = info < 0 ? -2 : -2; break; }     /* IllegalArgument -> SINGULAR_MATRIX (BA:304-309) */
        oracle_precondition(U, V, N, n);                       /* BA:238 */
        complete = isEstimated;
        int want_inv = complete && invert;
        if (complete) {                                        /* BA:252-280 */
            if (invert == 2 || invert == 3) {
                info = oracle_reduce(d, N, n, invert == 3);    /* BA:264 */
                if (!info) info = oracle_solve(oracle_reduced_rows(d), N, n, 1);   /* BA:266: leading numRows only */
            } else
                info = oracle_solve(U, N, n, invert == 1);     /* BA:270 */
            if (info) { state = -2; break; }
            oracle_precondition(U, V, N, n);                   /* BA:273: N too, whatever the mode */
        } else {
            if (invert == 3) {                                 /* BA:283-291 */
                info = oracle_reduce(d, N, n, 1);
                if (!info) info = oracle_solve(oracle_reduced_rows(d), N, n, 0);
                if (!info) oracle_extract_reduced(d, N, n);
            } else
                info = oracle_solve(U, N, n, 0);               /* BA:294 */
            if (info) { state = -2; break; }
            oracle_precondition(U, V, NULL, n);                /* BA:297 */
        }
        if (complete && Q_out && want_inv) memcpy(Q_out, N, sizeof(double) * plen);
        double *dx = n;
        /* ---- updateModel (BA:389-442) ---- */
        int rejected = 0;
        if (adapted > 0) {
            double alpha = 0.25 * pow(adapted, -0.05);
            alpha = alpha < 0.75 ? alpha : 0.75;
            for (int i = 0; i < U; i++) dx[i] *= alpha;
            double prevOmega = omega, curOmega;
            oracle_omega(d, vals, sigma2, dx, &curOmega);
            prevOmega = prevOmega <= 0 ? 1.7976931348623157e308 : prevOmega;
            int lmaConverge = prevOmega >= curOmega;
            omega = curOmega;
            if (lmaConverge) adapted *= 0.2;
            else {
                adapted *= 5.0;
                if (adapted > 1.0 / SQRT_EPS) { adapted = 1.0 / SQRT_EPS; omega = 0.0; }
            }
            if (!lmaConverge) { maxAbsDx = lastValid; rejected = 1; }
        }
        if (!rejected) {
            if (complete) {
                if (simulation) omega = 0.0;
                else oracle_omega(d, vals, sigma2, dx, &omega);
            }
            maxAbsDx = oracle_update(d, vals, dx);
            lastValid = maxAbsDx;
        }
        res->seconds_last_pass = now_s() - tp;
        /* ---- BA:327-353 ---- */
        if (isinf(maxAbsDx) || isnan(maxAbsDx)) { state = -2; break; }
        else if (maxAbsDx <= SQRT_EPS && runs > 0 && adapted == 0) isEstimated = 1;
        else if (runs-- <= 1) {
            if (complete) isConverge = 0;
            isEstimated = 1;
        }
        if (isEstimated || adapted <= SQRT_EPS || runs < max_iter * 0.5 + 1) adapted = 0.0;
    } while (!complete);
    if (state == 0) state = isConverge ? 1 : -4;
    res->state = state;
    res->iterations = iter;
    res->omega = omega;
    res->max_abs_dx = maxAbsDx;
    res->final_lambda = adapted;
    res->seconds_total = now_s() - t0;
    free(N); free(n); free(V);
    return 0;
}

/* ---- cpu_baseline helpers (bench.py): the "faithful" flavour allocates the dense 2 x U row of PDF:305 ------- */
/* one image-point group the way the reference does it: new DenseMatrix(2,U) zeroed, filled, stacked through
 * A.get(row, col) on the dense rows.  Returns a checksum so the work cannot be optimised away. */
double oracle_faithful_image_points(const jaicov_problem_desc *d, const double *vals, double sigma2, int ip_begin,
                                    int ip_end, double *N, double *n) {
    int U = d->n_unknowns;
    double chk = 0;
    for (int ip = ip_begin; ip < ip_end; ip++) {
        double *Ad = (double *)calloc((size_t)2 * U, sizeof(double));      /* PDF:305 column-major 2 x U */
        double A[2][ORA_KLOC], w[2], P[4];
        int diag;
        int32_t gcol[ORA_KLOC], cols[ORA_KLOC], loc[ORA_KLOC];
        eval_image_point(d, vals, sigma2, ip, A, w, P, &diag);
        int kl = local_columns(d, ip, gcol), k = 0;
        for (int l = 0; l < kl; l++)
            if (gcol[l] >= 0) { cols[k] = gcol[l]; loc[k] = l; Ad[2 * (size_t)gcol[l]] = A[0][l]; Ad[2 * (size_t)gcol[l] + 1] = A[1][l]; k++; }
        sort_pairs(k, cols, loc);
        for (int row = 0; row < 2; row++)
            for (int ia = 0; ia < k; ia++) {
                double aT = Ad[2 * (size_t)cols[ia] + row];
                if (diag) n[cols[ia]] += aT * P[row * 2 + row] * w[row];
                else for (int cp = 0; cp < 2; cp++) n[cols[ia]] += aT * P[row * 2 + cp] * w[cp];
                for (int ib = ia; ib < k; ib++) {
                    if (diag) pk_add(N, cols[ia], cols[ib], aT * P[row * 2 + row] * Ad[2 * (size_t)cols[ib] + row]);
                    else for (int cp = 0; cp < 2; cp++) pk_add(N, cols[ia], cols[ib], aT * P[row * 2 + cp] * Ad[2 * (size_t)cols[ib] + cp]);
                }
            }
        chk += Ad[2 * (size_t)cols[0]];
        free(Ad);
    }
    return chk;
}

/* "fair" flavour of an image-block group for the CPU baseline: same algebra N += A'PA, n += A'Pw on the compact
 * m x k row block (columns = the group's sorted unique columns), but as two dense products T = P A, A'T instead of
 * the reference's generic Theta(m^2 k^2) loop nest (PDF:479-502), which is unusable at m = 1000.
 * P (row-major m x m) is the cached weight (DOPG:82-86 computes it once). */
int oracle_block_weight(const jaicov_problem_desc *d, double sigma2, int blk, double *P_out) {
    int m = 2 * (d->blk_ip_begin[blk + 1] - d->blk_ip_begin[blk]);
    return dispersion_to_weight(m, d->blk_disp + d->blk_disp_offset[blk], sigma2, P_out);
}

int oracle_block_fair(const jaicov_problem_desc *d, const double *vals, double sigma2, int blk, const double *P,
                      double *N, double *n) {
    int b = d->blk_ip_begin[blk], e = d->blk_ip_begin[blk + 1];
    int m = 2 * (e - b);
    int kmax = (e - b) * 3 + ORA_KLOC, k = 0;
    int32_t *cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)kmax * 2);
    int32_t *tmp = cols + kmax;
    int U = d->n_unknowns;
    int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)U);
    for (int i = 0; i < U; i++) pos[i] = -1;
    for (int ip = b; ip < e; ip++) {
        int32_t gcol[ORA_KLOC];
        int kl = local_columns(d, ip, gcol);
        for (int l = 0; l < kl; l++)
            if (gcol[l] >= 0 && pos[gcol[l]] < 0) { pos[gcol[l]] = 0; cols[k++] = gcol[l]; }
    }
    for (int j = 0; j < k; j++) tmp[j] = j;
    sort_pairs(k, cols, tmp);
    for (int j = 0; j < k; j++) pos[cols[j]] = j;
    double *A = (double *)calloc((size_t)m * k, sizeof(double));
    double *T = (double *)calloc((size_t)m * k, sizeof(double));
    double *w = (double *)malloc(sizeof(double) * m), *Pw = (double *)calloc(m, sizeof(double));
    for (int ip = b; ip < e; ip++) {
        double Al[2][ORA_KLOC], wl[2], Pl[4];
        int diag;
        int32_t gcol[ORA_KLOC];
        eval_image_point(d, vals, sigma2, ip, Al, wl, Pl, &diag);
        int kl = local_columns(d, ip, gcol);
        for (int l = 0; l < kl; l++) {
            if (gcol[l] < 0) continue;
            A[(size_t)(2 * (ip - b)) * k + pos[gcol[l]]] = Al[0][l];
            A[(size_t)(2 * (ip - b) + 1) * k + pos[gcol[l]]] = Al[1][l];
        }
        w[2 * (ip - b)] = wl[0];
        w[2 * (ip - b) + 1] = wl[1];
    }
    for (int r = 0; r < m; r++)                      /* T = P A, Pw = P w */
        for (int q = 0; q < m; q++) {
            double p = P[(size_t)r * m + q];
            const double *aq = A + (size_t)q * k;
            double *tr = T + (size_t)r * k;
            for (int j = 0; j < k; j++) tr[j] += p * aq[j];
            Pw[r] += p * w[q];
        }
    for (int r = 0; r < m; r++) {                    /* N += A'T (upper), n += A'Pw */
        const double *ar = A + (size_t)r * k, *tr = T + (size_t)r * k;
        for (int i = 0; i < k; i++) {
            double a = ar[i];
            if (a == 0.0) continue;
            n[cols[i]] += a * Pw[r];
            for (int j = i; j < k; j++) N[pidx(cols[i], cols[j])] += a * tr[j];
        }
    }
    free(A); free(T); free(w); free(Pw); free(cols); free(pos);
    return 0;
}
