// Batched inverse of dense SPD dispersion matrices (engine creation).  gfx950 only.
//
// Reference: DirectlyObservedParameterGroup.getWeightMatrix (DOPG:82-86) = MathExtension.inv(UpperSPDPackMatrix) (MX:304-324) =
// dpptrf + dpptri, once per group, cached.  At BASELINE config 4 that is 500 matrices of order 1000; one after another through the
// general solver they cost ~14 500 launches (4 008 single-workgroup diagonal kernels among them): ~0.5 s for 0.5 TFLOP.
// Here a chunk of `cap` matrices of one padded order is factored, inverted and multiplied TOGETHER: every launch of the blocked
// algorithm carries the matrix index as a batch dimension of the grid (~45 launches per chunk whatever its size).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jaicov {

struct BatchedSpdInverse {
    hipStream_t stream = nullptr;
    int mp = 0, nb = 0, cap = 0;    // padded order (multiple of 128), block columns, matrices per chunk
    long ld = 0, msz = 0;           // leading dimension (= mp) and doubles per matrix (mp * mp)
    double *Lb = nullptr;           // [cap] input (lower, identity padding) -> Cholesky factor
    double *Wb = nullptr;           // [cap] L^-1
    double *Qb = nullptr;           // [cap] (L L')^-1, FULL symmetric square after run()
    double *Db = nullptr;           // [cap] refinement only: copy of the input matrices, FULL squares (filled by the caller beside Lb)
    double *S1 = nullptr, *S2 = nullptr;   // [cap] refinement workspace
    bool refine = false;            // one Newton-Schulz step on every inverse, residual by error-free splitting (batchinv.hip)
    double *invd = nullptr;         // [cap][nb] inverses of the diagonal blocks
    int *d_info = nullptr;          // 0, or 1 + the first failing pivot of some matrix of the chunk
    hipError_t init(hipStream_t s, int padded_order, int matrices_per_chunk, bool with_refinement);
    void release();
    // Lb[0..count) hold the matrices (lower triangle, identity on the padding's diagonal; with `refine` also Db[0..count), full
    // squares) -> Qb[0..count) their inverses.
    // Everything is enqueued on `stream`; *info_out is valid after the stream has been synchronised (run() does not synchronise).
    hipError_t run(int count);
};

// One Newton-Schulz step Q <- Q + sym(Q (I - M Q)) with the residual formed exactly (batchinv.hip), for ONE matrix of order n <= 8192
// (multiple of 128): M, Q full symmetric squares, leading dimension ld; M is destroyed; W, T1, T2, T3 are n x ld work squares.
hipError_t newton_schulz_exact(hipStream_t stream, int n, long ld, double *M, double *Q, double *W, double *T1, double *T2, double *T3);

}  // namespace jaicov
