// fp32-input / fp32-accumulate MFMA GEMM (v_mfma_f32_16x16x4_f32) with the tiling of gemm_f64.h: 128x128 tile, 4 waves x
// (4x4 MFMA tiles), BK = 16 double-buffered through LDS.  It exists for BASELINE config 5's "fp64 vs fp32-accumulate"
// sweep of the densified J'WJ contraction (densemode.hip, assembly_mode = 2); nothing on the product path uses it.
// Lane maps (cdna_hip_programming.md 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15] as the f64 form, but C/D is the
// standard 16x16 map col = l&15, row = 4*(l>>4) + reg.
#pragma once
#include "gemm_f64.h"

namespace jaicov {

typedef float f4_t __attribute__((ext_vector_type(4)));

struct GemmArgsF {
    const float *A, *B;
    float *C;
    long lda, ldb, ldc;
    int M, N, K;         // multiples of 128 / 128 / 16
    int lower_only;
    long strideA, strideB, strideC;
};

// C = A.B (beta = 0).  Layouts as in gemm_f64.h.
template <int ALAY, int BLAY>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgsF g) {
    constexpr int TM = 128, LD_S = TM + 16, EP = 8, STAGE = GEMM_BK * 2 * LD_S;
    __shared__ float smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    int tile_row, tile_col;
    {
        const int t = blockIdx.x;
        if (g.lower_only) {
            int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while ((long)(r + 1) * (r + 2) / 2 <= t) ++r;
            while ((long)r * (r + 1) / 2 > t) --r;
            tile_row = r;
            tile_col = t - r * (r + 1) / 2;
        } else {
            const int tn = g.N / TM;
            tile_row = t / tn;
            tile_col = t - tile_row * tn;
        }
    }
    const long m0 = (long)tile_row * TM, n0 = (long)tile_col * TM;
    const float *A = g.A + (long)blockIdx.y * g.strideA, *B = g.B + (long)blockIdx.y * g.strideB;
    float *C = g.C + (long)blockIdx.y * g.strideC;
    const float *ap, *bp;
    long astep, bstep;
    int a_lds, b_lds;
    if (ALAY == LAY_KC) {
        const int row = tid & 127, kh = tid >> 7;
        ap = A + (m0 + row) * g.lda + EP * kh; astep = GEMM_BK; a_lds = (EP * kh) * LD_S + row;
    } else {
        const int kr = tid >> 4, ms = tid & 15;
        ap = A + (long)kr * g.lda + m0 + EP * ms; astep = (long)GEMM_BK * g.lda; a_lds = kr * LD_S + EP * ms;
    }
    if (BLAY == LAY_KC) {
        const int row = tid & 127, kh = tid >> 7;
        bp = B + (n0 + row) * g.ldb + EP * kh; bstep = GEMM_BK; b_lds = (EP * kh) * LD_S + row;
    } else {
        const int kr = tid >> 4, ns = tid & 15;
        bp = B + (long)kr * g.ldb + n0 + EP * ns; bstep = (long)GEMM_BK * g.ldb; b_lds = kr * LD_S + EP * ns;
    }
    f4_t ra[2], rb[2];
    auto gload = [&]() {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            ra[j] = *reinterpret_cast<const f4_t *>(ap + 4 * j);
            rb[j] = *reinterpret_cast<const f4_t *>(bp + 4 * j);
        }
        ap += astep;
        bp += bstep;
    };
    auto lstore = [&](int stage) {
        float *sa = smem + stage * STAGE + a_lds, *sb = smem + stage * STAGE + GEMM_BK * LD_S + b_lds;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (ALAY == LAY_KC) {
#pragma unroll
                for (int e = 0; e < 4; e++) sa[(4 * j + e) * LD_S] = ra[j][e];
            } else
                *reinterpret_cast<f4_t *>(sa + 4 * j) = ra[j];
            if (BLAY == LAY_KC) {
#pragma unroll
                for (int e = 0; e < 4; e++) sb[(4 * j + e) * LD_S] = rb[j][e];
            } else
                *reinterpret_cast<f4_t *>(sb + 4 * j) = rb[j];
        }
    };
    f4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};
    const int nk = g.K / GEMM_BK;
    if (nk > 0) {
        gload();
        lstore(0);
        __syncthreads();
        const int fa = (lane >> 4) * LD_S + 64 * wr + (lane & 15), fb = (lane >> 4) * LD_S + 64 * wc + (lane & 15);
        for (int kt = 0; kt < nk; kt++) {
            const int st = kt & 1;
            if (kt + 1 < nk) gload();
            const float *sa = smem + st * STAGE + fa, *sb = smem + st * STAGE + GEMM_BK * LD_S + fb;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] = sa[(4 * ks) * LD_S + 16 * i];
#pragma unroll
                for (int j = 0; j < 4; j++) b[j] = sb[(4 * ks) * LD_S + 16 * j];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nk) lstore(st ^ 1);
            __syncthreads();
        }
    }
    float *cbase = C + (m0 + 64 * wr + 4 * (lane >> 4)) * g.ldc + n0 + 64 * wc + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) cbase[(long)(16 * i + r) * g.ldc + 16 * j] = acc[i][j][r];
}

inline hipError_t gemm_f32(hipStream_t s, int alay, int blay, const GemmArgsF &g, int batch = 1) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    const int tm = g.M / 128, tn = g.N / 128;
    dim3 grid(g.lower_only ? tm * (tm + 1) / 2 : tm * tn, batch), block(256);
    if (alay == LAY_KC && blay == LAY_XC) hipLaunchKernelGGL((gemm_f32_kernel<LAY_KC, LAY_XC>), grid, block, 0, s, g);
    else if (alay == LAY_XC && blay == LAY_XC) hipLaunchKernelGGL((gemm_f32_kernel<LAY_XC, LAY_XC>), grid, block, 0, s, g);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace jaicov
