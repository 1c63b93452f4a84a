/*
 * jaicov_dense.h -- C ABI of the stand-alone dense fp64 kernels of the MI355X engine.
 *
 * Replaces the LAPACK calls JAICOV makes through MathExtension for SYMMETRIC POSITIVE DEFINITE packed matrices
 * (JAICOV/src/org/applied_geodesy/adjustment/MathExtension.java):
 *   MX.solve(UpperSPDPackMatrix N, DenseVector n, int numRows, boolean invert)   MX:239-264  (dppsv [+ dpptri])
 *   MX.inv(UpperSPDPackMatrix N, int numRows)                                    MX:304-324  (dpptrf + dpptri)
 * used by DirectlyObservedParameterGroup.getWeightMatrix (DirectlyObservedParameterGroup.java:85-86).
 * Storage is MTJ/LAPACK packed UPLO='U' column-major: index(r,c) = r + c(c+1)/2.
 */
#ifndef JAICOV_DENSE_H
#define JAICOV_DENSE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* In place: b (nrhs vectors of length n, contiguous) <- N^-1 b ; if invert != 0, ap <- N^-1 (packed).
 * Returns jaicov_status: JAICOV_ERR_SINGULAR (1) when N is not positive definite (MatrixNotSPDException, MX:250-251).
 * ms_out (optional) receives the device time of factorisation (+ substitutions, + inverse) in milliseconds. */
int jaicov_dense_spd_solve_packed(int32_t n, double *ap, double *b, int32_t nrhs, int32_t invert, double *ms_out);

/* Parity / timing hook of the fp64 MFMA GEMM family (csrc/gemm_f64.h): C = alpha op(A) op(B) + beta C, row-major C.
 * alay/blay: 0 = k contiguous, 1 = m resp. n contiguous.  M,N multiples of 128, K multiple of 16.
 * kmode: 0 full, 1 k < (tile_row+1)*128, 2 k >= tile_row*128, 3 k >= tile_col*128.  Host buffers. */
int jaicov_dense_gemm(int32_t alay, int32_t blay, int32_t M, int32_t N, int32_t K, double alpha, const double *A,
                      int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc,
                      int32_t lower_only, int32_t kmode, int32_t repeats, double *ms_out);
#ifdef __cplusplus
}
#endif
#endif
