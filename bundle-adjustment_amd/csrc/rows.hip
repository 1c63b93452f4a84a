// Residual + Jacobian rows of every image point: one lane per observation, SoA in / SoA out (coalesced).
// Restates CollinearityEquationFactory (PDF:94-190), the base fill of getPartialDerivativeImageCoordinate
// (PDF:321-414) and the distortion factories (ASF:37-81, TDF:39-134, RSF:39-90, RDF:39-161 via DMF:33-101).
#include "ba_kernels.h"

namespace jaicov {

// Only 7 of the 12 base columns need accumulators: x0,y0 never change (DMF skips them) and the X0,Y0,Z0 columns are
// the exact negatives of X,Y,Z at every step (PDF:165-167,183-185; RDF:70-72).  idx: 0 X,1 Y,2 Z,3 c,4 omega,5 phi,6 kappa
struct RowAcc {
    double px[7], py[7];   // undistorted partials of xs, ys
    double ax[7], ay[7];   // accumulated row entries
    double w0, w1;
};

__device__ __forceinline__ void dmf_apply(RowAcc &a, double deltaX, double deltaY, double dXxs, double dXys,
                                          double dYxs, double dYys) {
    a.w0 += -deltaX;
    a.w1 += -deltaY;
#pragma unroll
    for (int l = 0; l < 7; l++) {
        a.ax[l] += dXxs * a.px[l] + dXys * a.py[l];
        a.ay[l] += dYxs * a.px[l] + dYys * a.py[l];
    }
}

// MathExtension.binomial (MathExtension.java:53-64)
__device__ __forceinline__ long zbinomial(int n, int k) {
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    long result = 1;
    for (int i = 1; i <= k; i++) result = result * (n - k + i) / i;
    return result;
}

__global__ __launch_bounds__(256) void rows_kernel(DevProblem p, const double *__restrict__ vals, int ip0, int count,
                                                   double *__restrict__ rowsA, double *__restrict__ rowsW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int ip = ip0 + i;
    const long S = p.n_ip;
    const int img = p.ip_image[ip], pt = p.ip_point[ip], cam = p.image_camera[img];
    const double *io = vals + slot_io(p, cam);
    const double *eo = vals + slot_eo(p, img);
    const double *xyz = vals + 3 * pt;
    const double x0 = io[0], y0 = io[1], c = io[2];
    const double X0 = eo[0], Y0 = eo[1], Z0 = eo[2];
    double sinO, cosO, sinP, cosP, sinK, cosK;
    sincos(eo[3], &sinO, &cosO);
    sincos(eo[4], &sinP, &cosP);
    sincos(eo[5], &sinK, &cosK);
    // PDF:125-135
    const double r11 = cosP * cosK, r12 = -cosP * sinK, r13 = sinP;
    const double r21 = cosO * sinK + sinO * sinP * cosK, r22 = cosO * cosK - sinO * sinP * sinK, r23 = -sinO * cosP;
    const double r31 = sinO * sinK - cosO * sinP * cosK, r32 = sinO * cosK + cosO * sinP * sinK, r33 = cosO * cosP;
    // PDF:137-152
    const double dX = xyz[0] - X0, dY = xyz[1] - Y0, dZ = xyz[2] - Z0;
    const double kx = r11 * dX + r21 * dY + r31 * dZ;
    const double ky = r12 * dX + r22 * dY + r32 * dZ;
    const double N = r13 * dX + r23 * dY + r33 * dZ;
    const double kxN = kx / N, kyN = ky / N;
    const double xs = -c * kxN, ys = -c * kyN;

    RowAcc a;
    // PDF:157-189
    a.px[0] = -(r13 * xs + c * r11) / N;
    a.px[1] = -(r23 * xs + c * r21) / N;
    a.px[2] = -(r33 * xs + c * r31) / N;
    a.px[3] = -kxN;
    a.px[4] = (xs * (r33 * dY - r23 * dZ) + c * (r31 * dY - r21 * dZ)) / N;
    a.px[5] = (xs * (ky * sinK - kx * cosK) + c * N * cosK) / N;
    a.px[6] = ys;
    a.py[0] = -(r13 * ys + c * r12) / N;
    a.py[1] = -(r23 * ys + c * r22) / N;
    a.py[2] = -(r33 * ys + c * r32) / N;
    a.py[3] = -kyN;
    a.py[4] = (ys * (r33 * dY - r23 * dZ) + c * (r32 * dY - r22 * dZ)) / N;
    a.py[5] = (ys * (ky * sinK - kx * cosK) - c * N * sinK) / N;
    a.py[6] = -xs;
#pragma unroll
    for (int l = 0; l < 7; l++) { a.ax[l] = a.px[l]; a.ay[l] = a.py[l]; }
    // PDF:321-322
    a.w0 = p.ip_x[ip] - (x0 + xs);
    a.w1 = p.ip_y[ip] - (y0 + ys);

    // ---- distortion models in DistortionModel.Type order (PDF:417-442) ------------------------------------
    const int jb = p.cam_dist_begin[cam], je = p.cam_dist_begin[cam + 1];
    const double *dv = vals + slot_dist(p, 0);
    const double r0 = p.cam_r0[cam];
    const double r2 = xs * xs + ys * ys, r02 = r0 * r0;
    const double xxs2 = 2.0 * xs * xs, yys2 = 2.0 * ys * ys, xys2 = 2.0 * xs * ys;
    int jCx = -1, jCy = -1, jBx = -1, jBy = -1;
    for (int j = jb; j < je; j++) {
        const int k = p.dist_kind[j];
        if (k == JAICOV_DIST_AFFINITY_CX) jCx = j;
        else if (k == JAICOV_DIST_AFFINITY_CY) jCy = j;
        else if (k == JAICOV_DIST_TANGENTIAL_BX) jBx = j;
        else if (k == JAICOV_DIST_TANGENTIAL_BY) jBy = j;
    }
#define OWN(j, vx, vy)                                        \
    do {                                                      \
        rowsA[(long)(2 * (12 + (j) - jb)) * S + ip] = (vx);   \
        rowsA[(long)(2 * (12 + (j) - jb) + 1) * S + ip] = (vy); \
    } while (0)
    // ASF:37-81
    if (jCx >= 0 && jCy >= 0) {
        const double cx = dv[jCx], cy = dv[jCy];
        dmf_apply(a, cx * xs + cy * ys, 0.0, cx, cy, 0.0, 0.0);
        OWN(jCx, xs, 0.0);
        OWN(jCy, ys, 0.0);
    }
    // TDF:39-134
    if (jBx >= 0 && jBy >= 0) {
        const double bx = dv[jBx], by = dv[jBy];
        double sum = 1.0;
        const double deltaX = bx * (r2 + xxs2) + by * xys2;
        const double deltaY = by * (r2 + yys2) + bx * xys2;
        const double dXxs = 2.0 * (3.0 * bx * xs + by * ys);
        const double dXys = 2.0 * (by * xs + bx * ys);
        const double dYxs = 2.0 * (by * xs + bx * ys);
        const double dYys = 2.0 * (bx * xs + 3.0 * by * ys);
        dmf_apply(a, deltaX, deltaY, dXxs, dXys, dYxs, dYys);
        for (int j = jb; j < je; j++) {
            if (p.dist_kind[j] != JAICOV_DIST_TANGENTIAL_BI) continue;
            const double bi = dv[j];
            const int e = p.dist_order[j];
            const double rim1 = ipow(r2, e - 1);
            const double ri = rim1 * r2;
            const double dTani = bi * ri;
            sum += dTani;
            const double constTani = 2.0 * bi * e * rim1;
            const double cX = deltaX * constTani, cY = deltaY * constTani;
            dmf_apply(a, deltaX * dTani, deltaY * dTani, dTani * dXxs + xs * cX, dTani * dXys + ys * cX,
                      dTani * dYxs + xs * cY, dTani * dYys + ys * cY);
            OWN(j, deltaX * ri, deltaY * ri);
        }
        OWN(jBx, sum * (r2 + xxs2), sum * xys2);
        OWN(jBy, sum * xys2, sum * (r2 + yys2));
    }
    // RSF:39-90
    for (int j = jb; j < je; j++) {
        if (p.dist_kind[j] != JAICOV_DIST_RADIAL_AI) continue;
        const double ai = dv[j];
        const int e = p.dist_order[j];
        const double rim1 = ipow(r2, e - 1);
        const double dRi = rim1 * r2 - ipow(r02, e);
        const double dRadi = ai * dRi;
        const double constRadi = ai * e * rim1;
        dmf_apply(a, xs * dRadi, ys * dRadi, xxs2 * constRadi + dRadi, xys2 * constRadi, xys2 * constRadi,
                  yys2 * constRadi + dRadi);
        OWN(j, xs * dRi, ys * dRi);
    }
    // RDF:39-161
    for (int j = jb; j < je; j++) {
        if (p.dist_kind[j] != JAICOV_DIST_DISTANCE_DI) continue;
        const double di = dv[j];
        const int e = p.dist_order[j];
        const double rim1 = ipow(r2, e - 1);
        const double dRi = rim1 * r2 - ipow(r02, e);
        const double dDisti = (di * dRi) / N;
        const double deltaX = xs * dDisti, deltaY = ys * dDisti;
        const double constRadi = (di * e * rim1) / N;
        dmf_apply(a, deltaX, deltaY, xxs2 * constRadi + dDisti, xys2 * constRadi, xys2 * constRadi,
                  yys2 * constRadi + dDisti);
        OWN(j, (xs * dRi) / N, (ys * dRi) / N);
        const double dXN = -deltaX / N, dYN = -deltaY / N;
        // dN/d(X,Y,Z) = (r13,r23,r33); dN/domega, dN/dphi (RDF:75-76); kappa: 0; c: none
        const double pN[7] = {r13, r23, r33, 0.0, -r33 * dY + r23 * dZ, kx * cosK - ky * sinK, 0.0};
#pragma unroll
        for (int l = 0; l < 7; l++) {
            if (l == 3) continue;
            a.ax[l] += pN[l] * dXN;
            a.ay[l] += pN[l] * dYN;
        }
    }
    // ZernikeDistortionModelFactory.java:41-227: X, Y (ZDF:153-227), then Gradient (ZDF:41-143), literally -- including
    // the integer division of the radial exponents (ZDF:107,176,178)
    bool any_z = false;
    for (int j = jb; j < je; j++) any_z = any_z || p.dist_kind[j] >= JAICOV_DIST_ZERNIKE_X;
    if (any_z) {
        const double xxs = xs * xs, yys = ys * ys, xys = xs * ys;
        const double phi = atan2(ys, xs);
        const double rn2 = r2 / r02;
        const double const2rnr0 = 2.0 / rn2 / r02;
        for (int kind = JAICOV_DIST_ZERNIKE_X; kind <= JAICOV_DIST_ZERNIKE_Z; kind++)
            for (int j = jb; j < je; j++) {
                if (p.dist_kind[j] != kind) continue;
                const double zi = dv[j];
                const int order = p.dist_order[j];
                const int n = (int)ceil((-3.0 + sqrt(9.0 + 8.0 * order)) / 2.0);     // ZernikeCoefficient.java:43-44
                const int mi = 2 * order - n * (n + 2);
                const int halfnm = (n - abs(mi)) / 2;
                const double length = sqrt((1 + ((mi != 0) ? 1 : 0)) * (n + 1) / 3.14159265358979323846);
                const double m = mi;
                double sinmphi, cosmphi;
                sincos(m * phi, &sinmphi, &cosmphi);
                double own_x = 0.0, own_y = 0.0;
                for (int k = 0; k <= halfnm; k++) {
                    const long pj = n - 2 * k;
                    const double cj = length * (double)(((k % 2 == 0) ? 1 : -1) * zbinomial(n - k, k) * zbinomial(n - 2 * k, halfnm - k));
                    const double cX = (mi < 0) ? (-pj * xs * sinmphi + m * ys * cosmphi) : (pj * xs * cosmphi + m * ys * sinmphi);
                    const double cY = (mi < 0) ? (-pj * ys * sinmphi - m * xs * cosmphi) : (pj * ys * cosmphi - m * xs * sinmphi);
                    if (kind != JAICOV_DIST_ZERNIKE_Z) {
                        const double constC = cj * pow(rn2, (double)(pj / 2));
                        const double constZ = zi * cj / r02 * pow(rn2, (double)(pj / 2 - 1));
                        const double az = (mi < 0) ? -constC * sinmphi : constC * cosmphi;
                        const double delta = zi * az;
                        if (kind == JAICOV_DIST_ZERNIKE_X) { dmf_apply(a, delta, 0.0, constZ * cX, constZ * cY, 0.0, 0.0); own_x += az; }
                        else { dmf_apply(a, 0.0, delta, 0.0, 0.0, constZ * cX, constZ * cY); own_y += az; }
                    } else {
                        const long ce = pj / 2 - 1;
                        const double constC = cj / r02 * pow(rn2, (double)ce);
                        double dXxs, dXys, dYxs, dYys;
                        if (mi < 0) {
                            dXxs = ce * xs * const2rnr0 * cX - pj * sinmphi + m / r2 * (pj * xys * cosmphi + m * yys * sinmphi);
                            dXys = ce * ys * const2rnr0 * cX + m * cosmphi - m / r2 * (pj * xxs * cosmphi + m * xys * sinmphi);
                            dYxs = ce * xs * const2rnr0 * cY - m * cosmphi + m / r2 * (pj * yys * cosmphi - m * xys * sinmphi);
                            dYys = ce * ys * const2rnr0 * cY - pj * sinmphi - m / r2 * (pj * xys * cosmphi - m * xxs * sinmphi);
                        } else {
                            dXxs = ce * xs * const2rnr0 * cX + pj * cosmphi + m / r2 * (pj * xys * sinmphi - m * yys * cosmphi);
                            dXys = ce * ys * const2rnr0 * cX + m * sinmphi - m / r2 * (pj * xxs * sinmphi - m * xys * cosmphi);
                            dYxs = ce * xs * const2rnr0 * cY - m * sinmphi + m / r2 * (pj * yys * sinmphi + m * xys * cosmphi);
                            dYys = ce * ys * const2rnr0 * cY + pj * cosmphi - m / r2 * (pj * xys * sinmphi + m * xxs * cosmphi);
                        }
                        const double zc = zi * constC;
                        dmf_apply(a, zc * cX, zc * cY, zc * dXxs, zc * dXys, zc * dYxs, zc * dYys);
                        own_x += constC * cX;
                        own_y += constC * cY;
                    }
                }
                OWN(j, own_x, own_y);
            }
    }
#undef OWN
    // ---- store the twelve base columns: X,Y,Z,x0,y0,c,X0,Y0,Z0,omega,phi,kappa -----------------------------
    const double ox[12] = {a.ax[0], a.ax[1], a.ax[2], 1.0, 0.0, a.ax[3], -a.ax[0], -a.ax[1], -a.ax[2], a.ax[4], a.ax[5], a.ax[6]};
    const double oy[12] = {a.ay[0], a.ay[1], a.ay[2], 0.0, 1.0, a.ay[3], -a.ay[0], -a.ay[1], -a.ay[2], a.ay[4], a.ay[5], a.ay[6]};
#pragma unroll
    for (int l = 0; l < 12; l++) {
        rowsA[(long)(2 * l) * S + ip] = ox[l];
        rowsA[(long)(2 * l + 1) * S + ip] = oy[l];
    }
    rowsW[ip] = a.w0;
    rowsW[S + ip] = a.w1;
}

hipError_t launch_rows(hipStream_t s, const DevProblem &p, const double *vals, int ip0, int count, double *rowsA,
                       double *rowsW) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(rows_kernel, dim3((count + 255) / 256), dim3(256), 0, s, p, vals, ip0, count, rowsA, rowsW);
    return hipGetLastError();
}

}  // namespace jaicov
