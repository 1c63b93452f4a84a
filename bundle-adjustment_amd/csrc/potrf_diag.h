// Diagonal-block kernel body of the blocked Cholesky (dense.hip, cholflow.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_f64.h"

namespace jaicov {

// ---------------------------------------------------------------------------------------------------------------
// Diagonal block: Cholesky of a 128x128 SPD block held in LDS by one workgroup (4 waves), plus the inverse of its
// factor.  A (global, lower part) <- L ; inv_out (128x128 row-major, zeros above the diagonal) <- L^-1.
//
// Blocked with 16-wide panels: the 16x16 diagonal block is factored AND inverted in the registers of wave 0
// (lane i = row i, operands broadcast with v_readlane), the panel solve and the trailing update run as
// v_mfma_f64_16x16x4 tile products straight out of LDS (row stride 129 doubles: conflict-free fragment reads).
// The inverse is built diagonal by diagonal: W[I][J] = -Wdd[I] * sum_{K=J}^{I-1} L[I][K] W[K][J]; the accumulator
// of the first product IS the B fragment of the second (C/D rows (l>>4)+4r == B rows 4ks+(l>>4)), so it never
// leaves the registers.  Off-diagonal W tiles are parked transposed in the (unused) strict upper part of S.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DP = 129;   // padded LDS row
constexpr int WDP = 17;   // padded row of the 16x16 diagonal-block inverses

// broadcast from a wave-uniform lane: two v_readlane_b32 (scalar path) instead of ds_bpermute round trips
__device__ __forceinline__ double bcast(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// 16x16 Cholesky + inverse of the diagonal block at (c0,c0) in the registers of ONE wave (lane i < 16 = row i).
// The reciprocal square root replaces sqrt + divide on the 128-step critical path (v_rsq_f64 + Newton, ~1 ulp).
// Round 4: the forward substitution X = L^-1 (lane c = column c of X) no longer FOLLOWS the factorisation as a second chain of 16
// dependent steps; step j of it runs inside step j of the factorisation: it needs column j of L only -- the very values
// L[k][j], k > j, that the right-looking update of the factor fetches (v_readlane for k = j + 1, LDS broadcast loads for the rest),
// so the two share their loads, and its products are independent of the factor's next pivot: they fill the latency of the
// rsqrt / Newton / readlane chain instead of waiting behind it.  (The chain workgroup of the dataflow factorisation spends 41 of its
// 70 us per block column here: at orders where the chain binds -- config 3, the first and last block columns of config 4 -- this
// kernel's time is the factorisation's.)
__device__ __forceinline__ void chol16_inv(double *S, double *Wd_p, int c0, int lane, int *info, int blk) {
    const int l15 = lane & 15;
    double a[16], x[16], sres[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = S[(c0 + l15) * DP + c0 + k];
#pragma unroll
    for (int i = 0; i < 16; i++) sres[i] = (l15 == i) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        double d = bcast(a[j], j);
        if (!(d > 0.0)) {   // not positive definite (also NaN): MatrixNotSPDException / info > 0
            if (lane == 0) atomicCAS(info, 0, blk * 128 + c0 + j + 1);
            d = 1.0;
        }
        double rl = rsqrt(d);
        rl = rl * (1.5 - 0.5 * d * rl * rl);         // one more Newton step: full fp64 accuracy
        const double ljj = d * rl;
        a[j] = (l15 == j) ? ljj : a[j] * rl;
        x[j] = sres[j] * rl;                         // row j of X = L^-1 (lane c: column c)
        // column j of L goes to LDS; the next column, which the following step waits for, gets its multiplier by
        // v_readlane (short latency), the other 14-j columns read theirs back from LDS as broadcast loads (pipelined)
        if (lane < 16 && lane >= j) S[(c0 + lane) * DP + c0 + j] = a[j];
        if (j + 1 < 16) {
            const double l1 = bcast(a[j], j + 1);    // L[j+1][j]
            a[j + 1] -= a[j] * l1;
            sres[j + 1] -= l1 * x[j];
        }
#pragma unroll
        for (int k = j + 2; k < 16; k++) {
            const double lk = S[(c0 + k) * DP + c0 + j];      // L[k][j], the same for every lane
            a[k] -= a[j] * lk;
            sres[k] -= lk * x[j];
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) Wd_p[k * WDP + lane] = x[k];          // Wdd[row k][col lane]
    }
}

// trailing-update tile (R,Q) of panel p:  S[R][Q] -= S[R][p] S[Q][p]'
__device__ __forceinline__ void diag_update_tile(double *S, int R, int Q, int c0, int l15, int l4) {
    d4_t acc;
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = S[(16 * R + l4 + 4 * r) * DP + 16 * Q + l15];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const double av = -S[(16 * R + l15) * DP + c0 + 4 * ks + l4];
        const double bv = S[(16 * Q + l15) * DP + c0 + 4 * ks + l4];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) S[(16 * R + l4 + 4 * r) * DP + 16 * Q + l15] = acc[r];
}

// The four parts of potrf_diag_body, callable one by one (cholflow.hip's chain workgroups split them between two CUs).  tid =
// threadIdx.x; a caller that runs them in a loop passes a copy the compiler cannot see through, so that the hundreds of
// addresses derived from it are not hoisted out of that loop and kept alive in registers across it.
// diag_load: block -> LDS (lower part, zeros above the diagonal).
__device__ __forceinline__ void diag_load(const double *A, long ld, double *S, const int tid) {
    // block -> LDS: 16 independent 16-byte loads in flight per thread (a rolled load->store loop serialises on the
    // memory latency: 64 round trips, ~45 us)
#pragma unroll
    for (int half = 0; half < 2; half++) {
        d2_t buf[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx2 = tid + 256 * (16 * half + i);
            buf[i] = *reinterpret_cast<const d2_t *>(A + (long)(idx2 >> 6) * ld + 2 * (idx2 & 63));
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx2 = tid + 256 * (16 * half + i);
            const int r = idx2 >> 6, c = 2 * (idx2 & 63);
            S[r * DP + c] = (c <= r) ? buf[i].x : 0.0;
            S[r * DP + c + 1] = (c + 1 <= r) ? buf[i].y : 0.0;
        }
    }
    __syncthreads();
}

// panel-solve tile: S[R][p] <- S[R][p] * Wdd[p]'  (L21 = A21 inv(L11)' with the inverted 16x16 diagonal block)
__device__ __forceinline__ void diag_solve_tile(double *S, const double *Wd, int R, int p, int l15, int l4) {
    const int c0 = 16 * p;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const double av = S[(16 * R + l15) * DP + c0 + 4 * ks + l4];
        const double bv = Wd[p * 16 * WDP + l15 * WDP + 4 * ks + l4];   // B[k][j] = Wdd[j][k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) S[(16 * R + l4 + 4 * r) * DP + c0 + l15] = acc[r];
}

// diag_factor: S (lower part) <- L in place, Wd <- inverses of the eight 16x16 diagonal blocks of L.
// Per 16-column panel p the critical path is: row tile (p+1, p) of the panel solve -> the next diagonal block (p+1, p+1) minus its
// square -> Cholesky + inverse of that block.  Round 4: wave 0 walks that path ALONE, tile by tile through LDS (a wave's LDS operations
// execute in order: no workgroup barrier between its steps), while waves 1-3 solve the other row tiles of the panel; one barrier; then
// wave 0 factors the diagonal block while waves 1-3 update everything else behind the panel (block column p+1 below the diagonal and
// the trailing tiles); one barrier.  Two barriers per panel where round 1's form had three, and the diagonal block's update no longer
// waits for the whole block column.  (potrf in the chain workgroup: 41 -> 28 us with the fused 16x16 factor / inverse, -> see DESIGN.md.)
__device__ __forceinline__ void diag_factor(double *S, double *Wd, int *info, int blk, int dbg, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    if (wave == 0 && !(dbg & 1)) chol16_inv(S, Wd, 0, lane, info, blk);
    __syncthreads();
    for (int p = 0; p < 7; p++) {
        if (dbg & 2) break;
        const int c0 = 16 * p;
        if (wave == 0) {
            diag_solve_tile(S, Wd, p + 1, p, l15, l4);                  // L[p+1][p]
            diag_update_tile(S, p + 1, p + 1, c0, l15, l4);            // (p+1, p+1) -= L[p+1][p] L[p+1][p]'
        } else {
            for (int R = p + 1 + wave; R < 8; R += 3) diag_solve_tile(S, Wd, R, p, l15, l4);     // L[R][p], R >= p + 2
        }
        __syncthreads();
        if (wave == 0) {
            if (!(dbg & 1)) chol16_inv(S, Wd + (p + 1) * 16 * WDP, 16 * (p + 1), lane, info, blk);
        } else {
            // block column p+1 below the diagonal, then the trailing tiles (R, Q), p + 2 <= Q <= R <= 7: nt tiles in all
            const int rem = 6 - p, nc = rem, nt = nc + rem * (rem + 1) / 2;
            for (int t = wave - 1; t < nt; t += 3) {
                if (t < nc) {
                    diag_update_tile(S, p + 2 + t, p + 1, c0, l15, l4);
                } else {
                    const int u = t - nc;
                    int rr = 0;
                    while ((rr + 1) * (rr + 2) / 2 <= u) ++rr;
                    diag_update_tile(S, p + 2 + rr, p + 2 + (u - rr * (rr + 1) / 2), c0, l15, l4);
                }
            }
        }
        __syncthreads();
    }
}

// diag_store_factor: factor back to global (lower part; the strict upper part of a diagonal block is never read by anyone)
__device__ __forceinline__ void diag_store_factor(const double *S, double *A, long ld, const int tid) {
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        const int idx2 = tid + 256 * i;
        const int r = idx2 >> 6, c = 2 * (idx2 & 63);
        if (c <= r) {
            d2_t v;
            v.x = S[r * DP + c];
            v.y = S[r * DP + c + 1];
            *reinterpret_cast<d2_t *>(A + (long)r * ld + c) = v;
        }
    }
}

// diag_inverse: inv_out <- L^-1 from the factor in S and the diagonal-block inverses in Wd (the strict upper part of S is
// used as parking space).  inv_out must be zero above the diagonal on entry.
// WT: the result leaves with write-through stores (every wave then drains, barrier, flag: no cache-wide release needed).
template <bool WT = false>
__device__ __forceinline__ void diag_inverse(double *S, const double *Wd, double *inv_out, int dbg, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // ---- inverse, one block diagonal after the other ---------------------------------------------------------
    for (int t = 1; t < 8; t++) {
        if (dbg & 4) break;
        for (int J = wave; J < 8 - t; J += 4) {
            const int I = J + t;
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {   // K = J: L[I][J] * Wdd[J]
                const double av = S[(16 * I + l15) * DP + 16 * J + 4 * ks + l4];
                const double bv = Wd[J * 16 * WDP + (4 * ks + l4) * WDP + l15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
            for (int K = J + 1; K < I; K++) {
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const double av = S[(16 * I + l15) * DP + 16 * K + 4 * ks + l4];
                    const double bv = S[(16 * J + l15) * DP + 16 * K + 4 * ks + l4];   // W[K][J] parked transposed
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
            }
            d4_t acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const double av = Wd[I * 16 * WDP + l15 * WDP + 4 * ks + l4];
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, acc[ks], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) S[(16 * J + l15) * DP + 16 * I + l4 + 4 * r] = -acc2[r];
        }
        __syncthreads();
    }
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        const int idx2 = tid + 256 * i;
        const int r = idx2 >> 6, c = 2 * (idx2 & 63);
        if (c <= r) {
            d2_t v;
            v.x = ((r >> 4) == (c >> 4)) ? Wd[(r >> 4) * 16 * WDP + (r & 15) * WDP + (c & 15)] : S[c * DP + r];
            v.y = (c + 1 > r) ? 0.0
                              : (((r >> 4) == ((c + 1) >> 4)) ? Wd[(r >> 4) * 16 * WDP + (r & 15) * WDP + ((c + 1) & 15)]
                                                              : S[(c + 1) * DP + r]);
            // (s_nop 1: the compiler does not know this is a store and would let the next instruction reuse the data registers
            // while the store still reads their upper half -- cholflow.hip, store_wt2)
            if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(inv_out + r * 128 + c), "v"(v) : "memory");
            else *reinterpret_cast<d2_t *>(inv_out + r * 128 + c) = v;
        }
    }
}

// inv_out must be zero above the diagonal on entry (the buffer is zero-filled once at allocation).
// S: 128 x DP doubles, Wd: 8 x 16 x WDP doubles of LDS owned by the calling workgroup (256 threads).
__device__ __forceinline__ void potrf_diag_body(double *A, long ld, double *inv_out, int *info, int blk, int dbg, double *S, double *Wd) {
    const int tid = threadIdx.x;
    diag_load(A, ld, S, tid);
    diag_factor(S, Wd, info, blk, dbg, tid);
    diag_store_factor(S, A, ld, tid);
    diag_inverse(S, Wd, inv_out, dbg, tid);
}

// Inverses of the eight 16x16 diagonal blocks of a factor that is already in S (a workgroup that did not factor it itself):
// wave w takes blocks w and w + 4; lane c < 16 holds column c of the inverse (forward substitution as in chol16_inv).
__device__ __forceinline__ void diag_block_inverses(const double *S, double *Wd, const int tid) {
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15;
    for (int b = wave; b < 8; b += 4) {
        const int c0 = 16 * b;
        double x[16], sres[16];
#pragma unroll
        for (int i = 0; i < 16; i++) sres[i] = (l15 == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            x[k] = sres[k] / S[(c0 + k) * DP + c0 + k];
#pragma unroll
            for (int i = k + 1; i < 16; i++) sres[i] -= S[(c0 + i) * DP + c0 + k] * x[k];
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; k++) Wd[b * 16 * WDP + k * WDP + lane] = x[k];
        }
    }
}

}  // namespace jaicov
