// Dense fp64 SPD solver kernels + drivers (see dense.h).  gfx950 only.
#include "dense.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "gemm_f64.h"
#include "potrf_diag.h"

namespace jaicov {

#define HIPCHK(x)                                  \
    do {                                           \
        hipError_t _e = (x);                       \
        if (_e != hipSuccess) return _e;           \
    } while (0)

__global__ __launch_bounds__(256) void potrf_diag_kernel(double *A, long ld, double *inv_out, int *info, int blk, int dbg) {
    __shared__ double S[128 * DP];
    __shared__ double Wd[8 * 16 * WDP];
    potrf_diag_body(A, ld, inv_out, info, blk, dbg, S, Wd);
}

// ---------------------------------------------------------------------------------------------------------------
// substitution (nrhs <= 8 right-hand sides).
// Forward: none.  The right-hand sides ride through the factorisation as 128 extra rows below the matrix
// (DenseSolver::aug): chol([[M, Y'], [Y, *]]) = [[L, 0], [Z, *]] with Z = Y L^-T, i.e. row q of Z is (L^-1 y_q)'.
// Backward: L' X = Z in ONE launch.  Workgroup b owns block column k = nb-1-b: it streams the blocks L[j][k], j > k,
// (double-buffered in registers) and subtracts L[j][k]' x_j as soon as the owner of column j has published x_j, then
// applies the inverse of its diagonal block and publishes x_k.  The data are their own flags: X is preset to an
// all-ones bit pattern, results are written and polled with relaxed agent-scope atomics (coherent across the XCDs,
// no cache-wide fences), a value counts as published once it differs from the pattern.  A workgroup only ever waits
// for workgroups with a smaller index, which the dispatcher starts first: the chain cannot deadlock whatever the
// residency; a bounded spin turns a lost predecessor into NaNs instead of a hang.  Per chain link the latency is one
// store-to-load round trip + one 128x128 block instead of a kernel launch.
// ---------------------------------------------------------------------------------------------------------------
constexpr unsigned long long BS_UNSET = ~0ull;
constexpr int BS_SPIN_MAX = 1 << 22;

// NR = right-hand sides carried (1, 2, 4 or 8: the smallest that holds nrhs; an adjustment without datum defect has ONE,
// and the 8-wide form spent 2 of its 8 us per link of the chain on seven columns of zeros)
template <int NR>
__device__ __forceinline__ void bs_row(const double *p, double (&v)[NR]) {
    if constexpr (NR % 4 == 0) {
#pragma unroll
        for (int h = 0; h < NR / 4; h++) {
            const d4_t t = *reinterpret_cast<const d4_t *>(p + 4 * h);
            v[4 * h] = t[0]; v[4 * h + 1] = t[1]; v[4 * h + 2] = t[2]; v[4 * h + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int q = 0; q < NR; q++) v[q] = p[q];
    }
}
template <int NR>
__global__ __launch_bounds__(256) void backsolve_chain_kernel(const double *__restrict__ L, long ld,
                                                              const double *__restrict__ invd, const double *__restrict__ Z,
                                                              long zs, double *X, long xs, int nb, int nrhs, const int *abort_word) {
    __shared__ __attribute__((aligned(32))) double xj[2][128][NR];     // x_j (then v_k), [row][rhs]
    // behind an abandoned dataflow factorisation (cholflow.hip: abort word set) the factor is garbage and the host will repeat
    // everything: leave at once (X keeps its "not published" pattern, which reads as NaN)
    if (abort_word && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    __shared__ __attribute__((aligned(32))) double red[4][128][NR];    // partial sums
    const int tid = threadIdx.x;
    const int k = nb - 1 - (int)blockIdx.x;
    const int c2 = tid & 63, qd = tid >> 6;        // block phase: columns 2*c2, 2*c2+1; rows 32*qd .. 32*qd+31
    const int di = tid & 127, dh = tid >> 7;       // diagonal phase: output row di, half dh of the sum
    const int sr = tid & 127, sq = tid >> 7;       // staging: row sr, right-hand sides sq, sq+2, sq+4, sq+6
    // inverse of the diagonal block and the right-hand side of this block, fetched ahead
    double dinv[64], zk[4];
    {
        const double *w = invd + (long)k * 16384 + (long)(64 * dh) * 128 + di;
#pragma unroll
        for (int r = 0; r < 64; r++) dinv[r] = w[(long)r * 128];
#pragma unroll
        for (int i = 0; i < 4; i++) zk[i] = sq + 2 * i < nrhs ? Z[(long)(sq + 2 * i) * zs + k * 128 + sr] : 0.0;   // nrhs <= NR
    }
    double acc[2][NR];
#pragma unroll
    for (int q = 0; q < NR; q++) acc[0][q] = acc[1][q] = 0.0;
    d2_t blk[32], nblk[32];
    int j = nb - 1;
    if (j > k) {
        const double *lp = L + (long)(j * 128 + 32 * qd) * ld + k * 128 + 2 * c2;
#pragma unroll
        for (int r = 0; r < 32; r++) blk[r] = *reinterpret_cast<const d2_t *>(lp + (long)r * ld);
    }
    for (; j > k; --j) {
        if (j - 1 > k) {
            const double *lp = L + (long)((j - 1) * 128 + 32 * qd) * ld + k * 128 + 2 * c2;
#pragma unroll
            for (int r = 0; r < 32; r++) nblk[r] = *reinterpret_cast<const d2_t *>(lp + (long)r * ld);
        }
        {   // wait for x_j: every thread polls the (up to four) values it stages
            unsigned long long b[4];
            const unsigned long long *xp = reinterpret_cast<const unsigned long long *>(X) + j * 128 + sr;
            int spin = 0;
            bool ready;
            do {
                ready = true;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    b[i] = 0;
                    if (sq + 2 * i < nrhs) {
                        // every 1024th poll by a read-modify-write at agent scope: never served from a stale line of this
                        // XCD's L2 (cholflow.hip, "Visibility": plain polls were seen to miss a set flag about once in 300
                        // factorisations)
                        const unsigned long long *q = xp + (long)(sq + 2 * i) * xs;
                        b[i] = (spin & 1023) == 1023
                                   ? __hip_atomic_fetch_or(const_cast<unsigned long long *>(q), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && b[i] != BS_UNSET;
                    }
                }
                if (!ready) __builtin_amdgcn_s_sleep(2);
            } while (!ready && ++spin < BS_SPIN_MAX);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (sq + 2 * i < NR) xj[j & 1][sr][sq + 2 * i] = __longlong_as_double((long long)b[i]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; r++) {
            double xv[NR];
            bs_row<NR>(&xj[j & 1][32 * qd + r][0], xv);
#pragma unroll
            for (int q = 0; q < NR; q++) {
                acc[0][q] += blk[r].x * xv[q];
                acc[1][q] += blk[r].y * xv[q];
            }
        }
#pragma unroll
        for (int r = 0; r < 32; r++) blk[r] = nblk[r];
    }
#pragma unroll
    for (int q = 0; q < NR; q++) {
        red[qd][2 * c2][q] = acc[0][q];
        red[qd][2 * c2 + 1][q] = acc[1][q];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int q = sq + 2 * i;
        if (q < NR) xj[0][sr][q] = zk[i] - (red[0][sr][q] + red[1][sr][q] + red[2][sr][q] + red[3][sr][q]);
    }
    __syncthreads();
    // x_k[di] = sum_r W[r][di] v[r]
    double out[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) out[q] = 0.0;
#pragma unroll
    for (int r = 0; r < 64; r++) {
        double vv[NR];
        bs_row<NR>(&xj[0][64 * dh + r][0], vv);
#pragma unroll
        for (int q = 0; q < NR; q++) out[q] += dinv[r] * vv[q];
    }
    if (dh == 1) {
#pragma unroll
        for (int q = 0; q < NR; q++) red[0][di][q] = out[q];
    }
    __syncthreads();
    if (dh == 0) {
        unsigned long long *xp = reinterpret_cast<unsigned long long *>(X) + k * 128 + di;
#pragma unroll
        for (int q = 0; q < NR; q++) {
            if (q >= nrhs) break;
            unsigned long long bits = (unsigned long long)__double_as_longlong(out[q] + red[0][di][q]);
            if (bits == BS_UNSET) bits = 0x7FF8000000000000ull;
            __hip_atomic_store(xp + (long)q * xs, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Forward substitution L z = b for ONE right-hand side in one launch: the mirror image of the chain above, for right-hand
// sides that did not ride through the factorisation (iterative refinement, engine.hip: the residual exists only after the
// first solution).  Workgroup k owns block ROW k: it streams the blocks L[k][j], j < k (rows of 1 KB: one cache line per
// 16 lanes), subtracts L[k][j] z_j as soon as the owner of row j has published z_j, then applies the inverse of its
// diagonal block.  Same flag protocol (Z preset to the all-ones pattern, relaxed agent-scope atomics), same progress
// argument (a workgroup waits only for workgroups with a smaller index) and the same bounded spin.
__global__ __launch_bounds__(256) void forwardsolve_chain_kernel(const double *__restrict__ L, long ld,
                                                                 const double *__restrict__ invd, const double *__restrict__ B,
                                                                 double *Z, int nb) {
    __shared__ __attribute__((aligned(16))) double zj[2][128];      // z_j
    __shared__ double red[128][65];                                 // per-lane partial sums of the 128 rows
    __shared__ double part[2][128];
    __shared__ double bk[128];
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    const int c2 = tid & 63, qd = tid >> 6;        // block phase: columns 2*c2, 2*c2+1; rows 32*qd .. 32*qd+31
    const int di = tid & 127, dh = tid >> 7;       // diagonal phase: output row di, columns 64*dh .. 64*dh+63
    double dinv[64];
    {
        const double *w = invd + (long)k * 16384 + (long)di * 128 + 64 * dh;     // row di of inv(L_kk)
#pragma unroll
        for (int c = 0; c < 64; c += 2) {
            const d2_t t = *reinterpret_cast<const d2_t *>(w + c);
            dinv[c] = t.x; dinv[c + 1] = t.y;
        }
        if (tid < 128) bk[tid] = B[k * 128 + tid];
    }
    double acc[32];
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = 0.0;
    d2_t blk[32], nblk[32];
    if (k > 0) {
        const double *lp = L + (long)(k * 128 + 32 * qd) * ld + 2 * c2;
#pragma unroll
        for (int r = 0; r < 32; r++) blk[r] = *reinterpret_cast<const d2_t *>(lp + (long)r * ld);
    }
    for (int j = 0; j < k; ++j) {
        if (j + 1 < k) {
            const double *lp = L + (long)(k * 128 + 32 * qd) * ld + (j + 1) * 128 + 2 * c2;
#pragma unroll
            for (int r = 0; r < 32; r++) nblk[r] = *reinterpret_cast<const d2_t *>(lp + (long)r * ld);
        }
        if (tid < 128) {   // wait for z_j
            const unsigned long long *zp = reinterpret_cast<const unsigned long long *>(Z) + j * 128 + tid;
            unsigned long long b;
            int spin = 0;
            do {
                b = (spin & 1023) == 1023
                        ? __hip_atomic_fetch_or(const_cast<unsigned long long *>(zp), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                        : __hip_atomic_load(zp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b == BS_UNSET) __builtin_amdgcn_s_sleep(2);
            } while (b == BS_UNSET && ++spin < BS_SPIN_MAX);
            zj[j & 1][tid] = __longlong_as_double((long long)b);
        }
        __syncthreads();
        const d2_t z = *reinterpret_cast<const d2_t *>(&zj[j & 1][2 * c2]);
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] += blk[r].x * z.x + blk[r].y * z.y;
#pragma unroll
        for (int r = 0; r < 32; r++) blk[r] = nblk[r];
    }
#pragma unroll
    for (int r = 0; r < 32; r++) red[32 * qd + r][c2] = acc[r];
    __syncthreads();
    {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 32; c++) s += red[di][32 * dh + c];
        part[dh][di] = s;
    }
    __syncthreads();
    double out = 0.0;
#pragma unroll
    for (int c = 0; c < 64; c++) {
        const int cc = 64 * dh + c;
        out += dinv[c] * (bk[cc] - (part[0][cc] + part[1][cc]));
    }
    __syncthreads();                 // every read of part[] is done before it is reused for the two halves of `out`
    if (dh == 1) part[0][di] = out;
    __syncthreads();
    if (dh == 0) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(out + part[0][di]);
        if (bits == BS_UNSET) bits = 0x7FF8000000000000ull;
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(Z) + k * 128 + di, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The backward chain for ONE right-hand side: a polling wave and seven streaming waves per workgroup (round 3).
//
// A link of the chain above takes 3.4 us although the hop itself -- agent-scope store in one workgroup, agent-scope load in
// another -- takes 0.42-0.53 us inside an XCD and 0.59-0.65 us across XCDs (scripts/micro/hop_latency.hip; the same test shows that a
// workgroup-scope load (sc0) is served by the CU's L1 and NEVER sees another workgroup's store, so there is no cheaper poll for
// neighbours on one XCD).  The rest is how a workgroup waits and what it still has to do once the last result it depends on is
// there (JAICOV_CHAIN_TRACE prints the link times and the phases of a workgroup):
//  * every workgroup consumes the results in order, one 128 KB block of L per result: it issues the next block's loads, polls
//    (the poll's s_waitcnt vmcnt(0) also waits for the loads just issued: loads return in order), meets at a barrier, does 64
//    FMAs per thread, copies the next block's registers into the current ones (which waits for them once more).  Polls issued
//    ahead of the loads do not help: every REPEATED poll still queues behind them;
//  * behind its last block it reduces the partial sums, multiplies with inv(L_kk), and only then publishes: four barriers,
//    each of which (__syncthreads) also waits for every global load in flight.
// Here the two kinds of waiting are separated and most of the work behind the last block is moved in front of it:
//  * wave 0 does nothing but poll: it has no other memory operation in flight, runs up to CH_RD results ahead of the stream and
//    hands them over through a ring in LDS (values + a sequence word per slot; a count of the waves that are done with a slot
//    lets it be reused);
//  * waves 1..7 stream the blocks (18 or 19 rows each), two buffers used in turn, every refill issued right behind the last use
//    of its buffer.  They wait for results by spinning on the sequence word in LDS -- lgkmcnt, not vmcnt: their block loads stay
//    in flight -- and for a block with the exact s_waitcnt vmcnt the compiler derives from straight-line code (all loads are
//    unconditional; past the end of the stream they re-read one 16-byte piece that sits in the L1).  No workgroup barrier in
//    the loop;
//  * the LAST blocks of every stream (CH_PM = 5, or 4 where that makes the rest an even number) are multiplied into the diagonal
//    inverse beforehand,
//        x_k = W_k'(z_k - S) - sum_m P_m[k]' x_{k+m},   P_m[k] = L[k+m][k] W_k,   S = the sum over the other blocks
//    (DenseSolver::premultiply(): one batched launch of 128^3 MFMA GEMMs after the factorisation), so that u = W_k'(z_k - S) --
//    the reduction over the waves, three barriers, the product with inv(L_kk) from LDS: ~2.5 us -- is formed four or five links
//    before the workgroup's turn.  The pre-multiplied blocks are streamed like the others (same buffers, same ring); what is
//    left of a link is: the polling wave sees x_{k+1}, hands it over, 38 FMAs per streaming thread, ONE exchange through LDS +
//    barrier, store.  The barriers behind the stream order LDS traffic only (chain_lds_barrier()).
// Same flag protocol towards the other workgroups (results preset to the all-ones pattern, relaxed agent-scope atomics), same
// progress argument (a workgroup waits only for workgroups with a smaller index) and the same bounded spin as above.
// What it gives at config 4 (118 links): 0.41 -> 0.31 ms, links of 2.6 us on average (1.5 - 2.1 and 2.7 - 3.9 in turn; the first
// links of a chain, whose successors are waiting, take 1.2 - 1.5).  That average is what ONE CU can stream: every workgroup reads
// one 128 KB block per link, and 128 KB / 2.6 us = 50 GB/s is the rate a single CU reaches on this chip whatever is in flight
// (the longest column alone, 15 MB, takes 0.3 ms).  With CH_PM = 4, 5, 6 the two-link sum stays at 5.2 - 5.4 us, only its split
// between the two links moves.  Going below takes two CUs per block column: backsolve_chain8_kernel<2>.  Steps on the way, all measured: polling wave + ring + two
// pre-multiplied blocks resident behind the stream 0.345 ms (the u-phase still ended after the predecessor had published);
// every wave polling the last two results itself instead of the polling wave: no gain; touching the pre-multiplied blocks into the
// L2 ahead of their loads: 0.41 ms (the polls queue behind the touches).  The forward chain has this form too where two
// workgroups can share a block row (forwardsolve_chain8_kernel: 10 row sums per lane instead of 19, which spilled).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned long long chain_poll(const unsigned long long *p, int spin) {
    // every 1024th poll by a read-modify-write at agent scope (see backsolve_chain_kernel)
    return (spin & 1023) == 1023 ? __hip_atomic_fetch_or(const_cast<unsigned long long *>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_publish(double *dst, double v) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == BS_UNSET) bits = 0x7FF8000000000000ull;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int CHAIN8_THREADS = 512;
constexpr int CH_RD = 8;              // slots of the ring between the polling wave and the streaming waves
constexpr int CH_SW = 7;              // streaming waves
constexpr int CH_ROWS = 19;           // rows of a block per streaming wave: waves 1..5 take 18, waves 6 and 7 take 19
#ifndef JAICOV_CH_PM
#define JAICOV_CH_PM 5
#endif
constexpr int CH_PM = JAICOV_CH_PM;              // blocks below a diagonal block that are multiplied into its inverse (DenseSolver::premultiply)
struct ChainRing {
    double xs[CH_RD][128];            // published results, in the order of the stream
    int ready[CH_RD];                 // 1 + index of the stream entry the slot holds
    int consumed[CH_RD];              // streaming waves that are done with the slot, counted over all its uses
};
__device__ __forceinline__ void chain_ring_init(ChainRing &rg, int tid) {
    if (tid < CH_RD) { rg.ready[tid] = 0; rg.consumed[tid] = 0; }
}
// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load in flight (s_waitcnt
// vmcnt(0)): behind the stream those are the two pre-multiplied blocks and the first polls for the predecessors' results, which
// the u-phase does not need -- its three barriers cost 7 us that way (JAICOV_CHAIN_TRACE), on the chain's critical path.
__device__ __forceinline__ void chain_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the polling wave: entry i of the stream = the 128 values at src
__device__ __forceinline__ void chain_ring_put(ChainRing &rg, int i, const double *src, int lane) {
    const int slot = i & (CH_RD - 1);
    if (i >= CH_RD) {       // every streaming wave is done with what the slot held
        const int need = CH_SW * (i / CH_RD);
        while (__hip_atomic_load(&rg.consumed[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
    }
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(src) + lane;
    unsigned long long b0, b1;
    int spin = 0;
    do {
        b0 = chain_poll(p, spin);
        b1 = chain_poll(p + 64, spin);
        if (b0 == BS_UNSET || b1 == BS_UNSET) __builtin_amdgcn_s_sleep(1);
    } while ((b0 == BS_UNSET || b1 == BS_UNSET) && ++spin < BS_SPIN_MAX);
    rg.xs[slot][lane] = __longlong_as_double((long long)b0);
    rg.xs[slot][lane + 64] = __longlong_as_double((long long)b1);
    __hip_atomic_store(&rg.ready[slot], i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // one wave: every lane writes the same word
}
// a streaming wave: waits for entry i, returns its slot.  No time limit of its own: the polling wave's waits are the bounded
// ones (chain_ring_put gives up on a lost predecessor after BS_SPIN_MAX polls and hands over the "not published" pattern, a NaN),
// and it sets `ready` for every entry whatever happened, so this loop ends whenever that one does -- and a predecessor that is
// merely LATE is waited for as long as the polling wave waits, instead of being replaced by stale ring contents after a shorter
// spin of this wave's own (a finite but wrong x_k; the tests rely on a lost link showing up as NaNs).
// (No poll limit of its own: the only producer is this workgroup's polling wave, whose own wait for the predecessor IS limited
// (BS_SPIN_MAX) and which then hands over the "not published" pattern -- every slot is filled after a bounded time, with a NaN at worst.)
__device__ __forceinline__ int chain_ring_get(ChainRing &rg, int i) {
    const int slot = i & (CH_RD - 1);
    while (__hip_atomic_load(&rg.ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != i + 1) __builtin_amdgcn_s_sleep(0);
    return slot;
}
__device__ __forceinline__ void chain_ring_done(ChainRing &rg, int slot, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(&rg.consumed[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// SPLIT = 2: workgroups 2 p and 2 p + 1 share chain position p; each streams the 64 columns of every block that belong to its 64
// outputs (half the load instructions per link: the chain with one workgroup per column runs at what ONE CU can stream, a 128 KB
// block per 2.6 us; loading 10 of a wave's 19 rows -- wrong results, right timing -- gave 1.05 us per link), forms its half of
// v = z_k - S, passes it to the other through `xch` (preset to the "not published" pattern like X; once per column, several links
// before the workgroup's turn), and publishes its half of x_k.  Used when the whole grid is resident at once (2 nb <= CUs): a
// workgroup then also waits for its partner, which has the higher index in one of the two cases (progress: the partner is the
// next workgroup the dispatcher starts).  0.31 -> 0.265 ms at config 4, links of 2.2 us: what is left is the way of a result from
// the store through the successor's polling wave, ring, products and barrier to its store.
template <int SPLIT>
__global__ __launch_bounds__(CHAIN8_THREADS) void backsolve_chain8_kernel(const double *__restrict__ L, long ld, const double *__restrict__ invd,
                                                                          const double *__restrict__ PM, const double *__restrict__ Z,
                                                                          double *X, double *xch, int nb, const int *abort_word, long long *trace) {
    __shared__ double red[8][128];
    __shared__ double comb[4][128];
    __shared__ double vv[128];
    __shared__ double Wl[128 * 129 / 2];           // inv(L_kk), lower triangle packed by rows
    __shared__ ChainRing ring;
    if (abort_word && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = SPLIT == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;          // position in the chain
    const int half = SPLIT == 2 ? (int)(blockIdx.x & 1) : 0;                        // which 64 outputs are this workgroup's
    const int k = nb - 1 - pos;
    if (SPLIT == 2 && half == 1) trace = nullptr;
    // The c = pos predecessors x_{k+1} .. x_{nb-1} are ONE stream, in the order they are published (x_{nb-1} first):
    //   entries 0 .. n_plain-1   blocks L[nb-1-i][k] of the factor, summed into S;  u = W_k'(z_k - S) is formed behind them,
    //   entries n_plain .. c-1   the pre-multiplied blocks P_m, m = c - i = npre .. 1, whose products go straight into x_k.
    // npre = CH_PM or CH_PM - 1, whichever makes n_plain even (the first loop then has no half step), or all of them.
    const int c = pos;
    const int npre = c <= CH_PM ? c : (((c - CH_PM) & 1) ? CH_PM - 1 : CH_PM);
    const int n_plain = c - npre;
    const int di = tid & 127, h4 = tid >> 7;       // diagonal phase: output di, operand quarter h4
    // streaming waves 1..7: rows r0 .. r0 + nrows - 1, columns 2*lane, 2*lane+1
    const int nrows = wave <= 5 ? 18 : 19;
    const int last_row = nrows - 1;                // buffer entry 18 of a wave with 18 rows repeats row 17 (its product is dropped)
    const int r0 = wave == 0 ? 0 : (wave <= 5 ? 18 * (wave - 1) : 90 + 19 * (wave - 6));
#define CHAIN8_STAMP(q) if (trace && tid == 64) trace[nb + 8 * pos + (q)] = wall_clock64();
#define CHAIN8_STAMP_T(q, t) if (trace && tid == (t)) trace[nb + 8 * pos + (q)] = wall_clock64();
    CHAIN8_STAMP(0);
    chain_ring_init(ring, tid);
    {   // inv(L_kk) -> LDS
        const double *w = invd + (long)k * 16384 + (long)(32 * h4) * 128 + di;
        double tw[32];
#pragma unroll
        for (int r = 0; r < 32; r++) tw[r] = w[(long)r * 128];
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int rr = 32 * h4 + r;
            if (di <= rr) Wl[rr * (rr + 1) / 2 + di] = tw[r];
        }
    }
    double zk = 0.0;
    if (tid < 128) zk = Z[k * 128 + tid];
    __syncthreads();
    const char *lcol = reinterpret_cast<const char *>(L + (long)r0 * ld + k * 128);
    const long bstep = 128 * ld * 8;               // bytes from one block row to the next
    // SPLIT = 1: a lane holds columns 2 lane, 2 lane + 1 of the wave's 18 / 19 rows, one row per load instruction.
    // SPLIT = 2: a lane holds columns 64 half + 2 (lane & 31), + 1 of every second row (lanes 0..31 the even, 32..63 the odd rows of
    //            the wave's share): TWO rows per load instruction, ten instructions per block -- what a CU can stream is set by
    //            the number of its load instructions (lane requests), not by their bytes: 8-byte loads of 64 columns x 19 rows gave
    //            0.27 ms, no better than the bytes of the unsplit chain would suggest.
    constexpr int NBUF = SPLIT == 2 ? (CH_ROWS + 1) / 2 : CH_ROWS;
    const int hr = SPLIT == 2 ? lane >> 5 : 0;                  // row parity of this lane
    const unsigned colbytes = SPLIT == 2 ? 8u * (unsigned)(64 * half + 2 * (lane & 31)) : 16u * (unsigned)lane;
    const bool own = SPLIT == 1 || (tid >> 6) == half;          // (of a thread tid < 128: its output is this workgroup's)
    const char *dummy = reinterpret_cast<const char *>(invd);
    const double *pmk = PM + (long)k * 16384;      // P_m of this column: pmk + (m - 1) * nb * 16384
    const long pmstep = (long)nb * 16384;
    d2_t A[NBUF], B[NBUF];                         // two stream buffers
    double acc0 = 0.0, acc1 = 0.0;                 // S (streamed blocks of the factor), then the pre-multiplied part
    // Stream entry i into a buffer: a block of the factor, a pre-multiplied block, or (past the end) nothing.  Addresses =
    // wave-uniform row base (scalar registers) + a 32-bit lane offset.  Loads past the end are issued all the same (straight-line
    // code keeps the s_waitcnt counts exact) but read one 16-byte piece that is in the L1.  A __builtin_amdgcn_sched_barrier(0)
    // stands on either side: without it the scheduler moves the refill loads about, and the register copies the compiler places on
    // the loop's back edge (it renames one buffer: 38 v_mov_b64) came behind an s_waitcnt vmcnt(0) -- the refill just issued had to
    // land before the loop went on, and the second buffer bought nothing; with it the copies wait for the older refill only.
#define CHAIN8_ENTRY(buf, i)                                                                                               \
    {                                                                                                                      \
        const bool fac = (i) < n_plain, pre = !fac && (i) < c;                                                             \
        const char *bp = fac ? lcol + (long)(nb - 1 - (i)) * bstep                                                         \
                             : (pre ? reinterpret_cast<const char *>(pmk + (long)(c - (i) - 1) * pmstep + (long)r0 * 128) : dummy); \
        const long st = fac ? ld * 8 : (pre ? 1024 : 0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if (SPLIT == 2) {                                                                                                  \
            const unsigned vo = (fac || pre) ? colbytes : 0u;                                                              \
            _Pragma("unroll") for (int r = 0; r < NBUF; r++) {                                                             \
                /* row 2 r + hr of the wave's share; past its end the last row again (that product is dropped: x = 0) */   \
                const int row = 2 * r + hr < last_row ? 2 * r + hr : last_row;                                             \
                buf[r] = *reinterpret_cast<const d2_t *>(bp + (long)row * st + vo);                                        \
            }                                                                                                              \
        } else {                                                                                                           \
            const unsigned vo = (fac || pre) ? colbytes : 0u;                                                              \
            _Pragma("unroll") for (int r = 0; r < NBUF; r++)                                                               \
                buf[r] = *reinterpret_cast<const d2_t *>(bp + (long)(r < 18 ? r : last_row) * st + vo);                    \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
#define CHAIN8_USE(buf, slot)                                                        \
    {                                                                                \
        const double xl = ring.xs[slot][xlane];     /* one LDS read per lane, then scalar broadcasts */ \
        const double xv = xdrop ? 0.0 : xl;         /* entries past the wave's share */ \
        if (SPLIT == 2) {                                                            \
            _Pragma("unroll") for (int r = 0; r < NBUF; r++) {                       \
                const double x0 = readlane_f64(xv, 2 * r), x1 = readlane_f64(xv, 2 * r + 1); \
                const double xr = hr ? x1 : x0;                                      \
                acc0 += buf[r].x * xr;                                               \
                acc1 += buf[r].y * xr;                                               \
            }                                                                        \
        } else {                                                                     \
            _Pragma("unroll") for (int r = 0; r < NBUF; r++) {                       \
                const double xr = readlane_f64(xv, r);                               \
                acc0 += buf[r].x * xr;                                               \
                acc1 += buf[r].y * xr;                                               \
            }                                                                        \
        }                                                                            \
    }
    const int xlane = r0 + (lane < nrows ? lane : last_row);
    const bool xdrop = lane >= nrows;               // (v_readlane indices run to 2 NBUF - 1 = 19)
    // ---- the blocks of the factor -----------------------------------------------------------------------------------
    if (wave == 0) {
        for (int i = 0; i < n_plain; i++) chain_ring_put(ring, i, X + (long)(nb - 1 - i) * 128, lane);
        CHAIN8_STAMP_T(6, 0);
    } else {
        CHAIN8_ENTRY(A, 0);
        CHAIN8_ENTRY(B, 1);
        for (int i = 0; i < n_plain; i += 2) {      // n_plain is even
            const int sa = chain_ring_get(ring, i);
            CHAIN8_USE(A, sa);
            chain_ring_done(ring, sa, lane);
            CHAIN8_ENTRY(A, i + 2);
            const int sb = chain_ring_get(ring, i + 1);
            CHAIN8_USE(B, sb);
            chain_ring_done(ring, sb, lane);
            CHAIN8_ENTRY(B, i + 3);
        }
        CHAIN8_STAMP(1);
        CHAIN8_STAMP_T(7, 448);
        if (SPLIT == 2) {       // the two row parities of a column pair sit 32 lanes apart
            acc0 += __shfl_xor(acc0, 32, 64);
            acc1 += __shfl_xor(acc1, 32, 64);
            if (lane < 32) { red[wave][64 * half + 2 * lane] = acc0; red[wave][64 * half + 2 * lane + 1] = acc1; }
        } else { red[wave][2 * lane] = acc0; red[wave][2 * lane + 1] = acc1; }
        acc0 = acc1 = 0.0;                          // from here on: the pre-multiplied part
    }
    // ---- u = W_k'(z_k - S): t + 2 links before this workgroup's turn ----------------------------------------------------
    chain_lds_barrier();
    CHAIN8_STAMP(2);
    if (tid < 128) {
        if (own) {
            double s = 0.0;
#pragma unroll
            for (int q = 1; q <= CH_SW; q++) s += red[q][tid];
            vv[tid] = zk - s;
            if (SPLIT == 2) chain_publish(xch + (long)k * 128 + tid, zk - s);
        } else {        // the other workgroup's half of v
            const unsigned long long *xp = reinterpret_cast<const unsigned long long *>(xch) + (long)k * 128 + tid;
            unsigned long long b;
            int spin = 0;
            do {
                b = chain_poll(xp, spin);
                if (b == BS_UNSET) __builtin_amdgcn_s_sleep(2);
            } while (b == BS_UNSET && ++spin < BS_SPIN_MAX);
            vv[tid] = __longlong_as_double((long long)b);
        }
    }
    chain_lds_barrier();
    CHAIN8_STAMP(3);
    {   // branch-free: entries above the diagonal read the row's first entry and are dropped
        double o0 = 0.0, o1 = 0.0;
        int base = 16 * h4 * (32 * h4 + 1);                       // rr (rr + 1) / 2 at rr = 32 h4
#pragma unroll 4
        for (int r = 0; r < 32; r += 2) {
            const int rr = 32 * h4 + r;
            const double w0 = Wl[base + (di <= rr ? di : 0)];
            const double w1 = Wl[base + rr + 1 + (di <= rr + 1 ? di : 0)];
            o0 += (di <= rr ? w0 : 0.0) * vv[rr];
            o1 += (di <= rr + 1 ? w1 : 0.0) * vv[rr + 1];
            base += 2 * rr + 3;
        }
        comb[h4][di] = o0 + o1;
    }
    chain_lds_barrier();
    double u = 0.0;
    if (tid < 128) u = (comb[0][tid] + comb[1][tid]) + (comb[2][tid] + comb[3][tid]);
    CHAIN8_STAMP(4);
    // ---- the pre-multiplied blocks (at most CH_PM): from here on a link of the chain is poll -> LDS -> 38 FMAs -> LDS -> store ------
    if (wave == 0) {
        // (touching the pre-multiplied blocks into the L2 ahead of their loads was tried here: 0.31 -> 0.41 ms, the polls queue behind it)
        for (int i = n_plain; i < c; i++) chain_ring_put(ring, i, X + (long)(nb - 1 - i) * 128, lane);
    } else {
#define CHAIN8_PRE(buf, e)                                      \
    if ((e) < npre) {                                           \
        const int sl = chain_ring_get(ring, n_plain + (e));     \
        CHAIN8_USE(buf, sl);                                    \
        chain_ring_done(ring, sl, lane);                        \
        CHAIN8_ENTRY(buf, n_plain + (e) + 2);                   \
    }
        CHAIN8_PRE(A, 0);
        CHAIN8_PRE(B, 1);
        CHAIN8_PRE(A, 2);
        CHAIN8_PRE(B, 3);
        CHAIN8_PRE(A, 4);
        CHAIN8_PRE(B, 5);
        static_assert(CH_PM <= 6, "one CHAIN8_PRE per pre-multiplied block");
#undef CHAIN8_PRE
        CHAIN8_STAMP(5);
        if (SPLIT == 2) {       // the two row parities of a column pair sit 32 lanes apart
            acc0 += __shfl_xor(acc0, 32, 64);
            acc1 += __shfl_xor(acc1, 32, 64);
            if (lane < 32) { red[wave][64 * half + 2 * lane] = acc0; red[wave][64 * half + 2 * lane + 1] = acc1; }
        } else { red[wave][2 * lane] = acc0; red[wave][2 * lane + 1] = acc1; }
    }
    if (c > 0) {
        chain_lds_barrier();
        if (tid < 128 && own) {
            double s = 0.0;
#pragma unroll
            for (int q = 1; q <= CH_SW; q++) s += red[q][tid];
            u -= s;
        }
    }
    if (tid < 128 && own) chain_publish(X + (long)k * 128 + tid, u);
#undef CHAIN8_STAMP
#undef CHAIN8_STAMP_T
#undef CHAIN8_USE
#undef CHAIN8_ENTRY
    if (trace && tid == 0) trace[pos] = wall_clock64();          // JAICOV_CHAIN_TRACE: when each link was published (100 MHz)
}

// ---------------------------------------------------------------------------------------------------------------
// The forward chain L z = b for ONE right-hand side in the form of backsolve_chain8_kernel<2>: two workgroups per block row k, each
// owning 64 of its 128 outputs; polling wave + ring; the blocks next to the diagonal (CH_PM or CH_PM - 1 of them) pre-multiplied,
//     z_k = W_k (b_k - S) - sum_m Ft_m[k]' z_{k-m},   Ft_m[k] = (W_k L[k][k-m])'   (rows: index into z_{k-m}, columns: outputs),
// and u = W_k (b_k - S) formed ahead of them.  The blocks of the factor are used as rows here (L[k][i] z_i: a workgroup streams the
// 64 rows of its outputs, whole 1 KB rows per load instruction, 9 or 10 rows per streaming wave, one partial sum per row and lane,
// reduced over the lanes through LDS in the u-phase); the pre-multiplied blocks as in the backward chain (64 columns, two rows per
// load instruction).  Both kinds go through the same two buffers of ten entries.  Only launched when the whole grid is resident
// at once (2 nb <= CUs); forwardsolve_chain_kernel otherwise.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CHF_ROWS = 10;          // rows of a block per streaming wave in the forward chain: waves 1..6 take 9, wave 7 takes 10
__global__ __launch_bounds__(CHAIN8_THREADS) void forwardsolve_chain8_kernel(const double *__restrict__ L, long ld, const double *__restrict__ invd,
                                                                             const double *__restrict__ FT, const double *__restrict__ Bv,
                                                                             double *Z, double *xch, int nb) {
    __shared__ double red[64][65];                 // per-lane partial sums of the 64 rows; the partial sums of the pre-multiplied part ([8][128]) afterwards
    __shared__ double part[8][64];
    __shared__ double comb[8][64];
    __shared__ double vv[128];
    __shared__ double Wl[128 * 129 / 2];           // inv(L_kk), lower triangle packed by columns: (row, c) at c * 128 - c (c - 1) / 2 + (row - c)
    __shared__ ChainRing ring;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = blockIdx.x >> 1, half = blockIdx.x & 1;      // block row = position in the chain; which 64 outputs
    const int c = k;                               // predecessors z_0 .. z_{k-1}, published in this order: ONE stream
    const int npre = c <= CH_PM ? c : (((c - CH_PM) & 1) ? CH_PM - 1 : CH_PM);
    const int n_plain = c - npre;                  // entries i < n_plain: L[k][i] with z_i; then Ft_m[k], m = c - i, with z_i
    const int di = tid & 127, h4 = tid >> 7;
    // streaming waves 1..7, blocks of the factor: rows 64 half + r0f .. + nrf - 1 (all 128 columns: 2 lane, 2 lane + 1)
    const int nrf = wave <= 6 ? 9 : 10, r0f = wave == 0 ? 0 : 9 * (wave - 1);
    // pre-multiplied blocks: rows r0 .. r0 + nrows - 1 (every second one per lane half), columns 64 half + 2 (lane & 31), + 1
    const int nrows = wave <= 5 ? 18 : 19, last_row = nrows - 1;
    const int r0 = wave == 0 ? 0 : (wave <= 5 ? 18 * (wave - 1) : 90 + 19 * (wave - 6));
    const int hr = lane >> 5;
    chain_ring_init(ring, tid);
    {
        const double *w = invd + (long)k * 16384 + (long)di * 128 + 32 * h4;     // row di of inv(L_kk)
        d4_t t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = *reinterpret_cast<const d4_t *>(w + 4 * q);
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const int cc = 32 * h4 + q;
            if (cc <= di) Wl[cc * 128 - cc * (cc - 1) / 2 + (di - cc)] = t[q >> 2][q & 3];
        }
    }
    double bk = 0.0;
    if (tid < 128) bk = Bv[k * 128 + tid];
    __syncthreads();
    const bool own = (tid >> 6) == half;           // (of a thread tid < 128: its output is this workgroup's)
    const char *lrow = reinterpret_cast<const char *>(L + (long)(k * 128 + 64 * half + r0f) * ld);
    const char *dummy = reinterpret_cast<const char *>(invd);
    const double *ftk = FT + (long)k * 16384;      // Ft_m of this block row: ftk + (m - 1) * nb * 16384
    const long pmstep = (long)nb * 16384;
    const unsigned cb_fac = 16u * (unsigned)lane, cb_pre = 8u * (unsigned)(64 * half + 2 * (lane & 31));
    d2_t A[CHF_ROWS], B[CHF_ROWS];
    static_assert(CHF_ROWS == (CH_ROWS + 1) / 2, "one pair of buffers serves both kinds of blocks");
    double acc[CHF_ROWS];
#pragma unroll
    for (int r = 0; r < CHF_ROWS; r++) acc[r] = 0.0;
    double p0 = 0.0, p1 = 0.0;
    // stream entry i into a buffer (see backsolve_chain8_kernel): a block of the factor (rows of this workgroup), a pre-multiplied
    // block (columns of this workgroup), or nothing
#define CHAINF_ENTRY(buf, i)                                                                                               \
    {                                                                                                                      \
        const bool fac = (i) < n_plain, pre = !fac && (i) < c;                                                             \
        const char *bp = fac ? lrow + (long)(i) * 1024                                                                     \
                             : (pre ? reinterpret_cast<const char *>(ftk + (long)(c - (i) - 1) * pmstep + (long)r0 * 128) : dummy); \
        const long st = fac ? ld * 8 : (pre ? 1024 : 0);                                                                   \
        const unsigned vo = fac ? cb_fac : (pre ? cb_pre : 0u);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int r = 0; r < CHF_ROWS; r++) {                                                             \
            const int rf = r < nrf ? r : nrf - 1, rp = 2 * r + hr < last_row ? 2 * r + hr : last_row;                      \
            buf[r] = *reinterpret_cast<const d2_t *>(bp + (long)(fac ? rf : rp) * st + vo);                                \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
    if (wave == 0) {
        for (int i = 0; i < n_plain; i++) chain_ring_put(ring, i, Z + (long)i * 128, lane);
    } else {
#define CHAINF_USE(buf, slot)                                                                                              \
    {                                                                                                                      \
        const d2_t z = *reinterpret_cast<const d2_t *>(&ring.xs[slot][2 * lane]);                                          \
        _Pragma("unroll") for (int r = 0; r < CHF_ROWS; r++) acc[r] = __builtin_fma(buf[r].y, z.y, __builtin_fma(buf[r].x, z.x, acc[r])); \
    }
        CHAINF_ENTRY(A, 0);
        CHAINF_ENTRY(B, 1);
        for (int i = 0; i < n_plain; i += 2) {      // n_plain is even
            const int sa = chain_ring_get(ring, i);
            CHAINF_USE(A, sa);
            chain_ring_done(ring, sa, lane);
            CHAINF_ENTRY(A, i + 2);
            const int sb = chain_ring_get(ring, i + 1);
            CHAINF_USE(B, sb);
            chain_ring_done(ring, sb, lane);
            CHAINF_ENTRY(B, i + 3);
        }
#undef CHAINF_USE
#pragma unroll
        for (int r = 0; r < CHF_ROWS; r++)
            if (r < nrf) red[r0f + r][lane] = acc[r];
    }
    // ---- u = W_k (b_k - S) for this workgroup's outputs: npre links before its turn ------------------------------------
    chain_lds_barrier();
    {   // row sums: thread (row, group of eight lanes)
        const int rl = tid & 63, g = tid >> 6;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) s += red[rl][8 * g + q];
        part[g][rl] = s;
    }
    chain_lds_barrier();
    if (tid < 128) {
        if (own) {
            const int rl = tid & 63;
            const double s = ((part[0][rl] + part[1][rl]) + (part[2][rl] + part[3][rl])) + ((part[4][rl] + part[5][rl]) + (part[6][rl] + part[7][rl]));
            vv[tid] = bk - s;
            chain_publish(xch + (long)k * 128 + tid, bk - s);
        } else {        // the other workgroup's half of v
            const unsigned long long *xp = reinterpret_cast<const unsigned long long *>(xch) + (long)k * 128 + tid;
            unsigned long long b;
            int spin = 0;
            do {
                b = chain_poll(xp, spin);
                if (b == BS_UNSET) __builtin_amdgcn_s_sleep(2);
            } while (b == BS_UNSET && ++spin < BS_SPIN_MAX);
            vv[tid] = __longlong_as_double((long long)b);
        }
    }
    chain_lds_barrier();
    {   // thread (output 64 half + (tid & 63), columns 16 g .. 16 g + 15)
        const int o = 64 * half + (tid & 63), g = tid >> 6;
        double out = 0.0;
#pragma unroll 4
        for (int q = 0; q < 16; q++) {
            const int cc = 16 * g + q;
            const double w = Wl[cc * 128 - cc * (cc - 1) / 2 + (cc <= o ? o - cc : 0)];
            out += (cc <= o ? w : 0.0) * vv[cc];
        }
        comb[g][tid & 63] = out;
    }
    chain_lds_barrier();
    double u = 0.0;
    if (tid < 128 && own) {
        const int ol = tid & 63;
        u = ((comb[0][ol] + comb[1][ol]) + (comb[2][ol] + comb[3][ol])) + ((comb[4][ol] + comb[5][ol]) + (comb[6][ol] + comb[7][ol]));
    }
    // ---- the pre-multiplied blocks --------------------------------------------------------------------------------------------
    double *red2 = &red[0][0];                     // [8][128]; every read of red[][] is three barriers back
    if (wave == 0) {
        for (int i = n_plain; i < c; i++) chain_ring_put(ring, i, Z + (long)i * 128, lane);
    } else {
        const int xlane = r0 + (lane < nrows ? lane : last_row);
        const bool xdrop = lane >= nrows;
#define CHAINF_PRE(buf, e)                                                                   \
    if ((e) < npre) {                                                                        \
        const int sl = chain_ring_get(ring, n_plain + (e));                                  \
        const double xl = ring.xs[sl][xlane];                                                \
        const double xv = xdrop ? 0.0 : xl;                                                  \
        _Pragma("unroll") for (int r = 0; r < CHF_ROWS; r++) {                               \
            const double x0 = readlane_f64(xv, 2 * r), x1 = readlane_f64(xv, 2 * r + 1);     \
            const double xr = hr ? x1 : x0;                                                  \
            p0 += buf[r].x * xr;                                                             \
            p1 += buf[r].y * xr;                                                             \
        }                                                                                    \
        chain_ring_done(ring, sl, lane);                                                     \
        CHAINF_ENTRY(buf, n_plain + (e) + 2);                                                \
    }
        CHAINF_PRE(A, 0);
        CHAINF_PRE(B, 1);
        CHAINF_PRE(A, 2);
        CHAINF_PRE(B, 3);
        CHAINF_PRE(A, 4);
        CHAINF_PRE(B, 5);
#undef CHAINF_PRE
        p0 += __shfl_xor(p0, 32, 64);
        p1 += __shfl_xor(p1, 32, 64);
        if (lane < 32) { red2[wave * 128 + 64 * half + 2 * lane] = p0; red2[wave * 128 + 64 * half + 2 * lane + 1] = p1; }
    }
    if (c > 0) {
        chain_lds_barrier();
        if (tid < 128 && own) {
            double s = 0.0;
#pragma unroll
            for (int q = 1; q <= CH_SW; q++) s += red2[q * 128 + tid];
            u -= s;
        }
    }
    if (tid < 128 && own) chain_publish(Z + (long)k * 128 + tid, u);
#undef CHAINF_ENTRY
}

__global__ void copy_diag_blocks_kernel(const double *invd, double *W, long ld) {
    const int k = blockIdx.x;
    for (int idx = threadIdx.x; idx < 128 * 128; idx += blockDim.x) {
        const int r = idx >> 7, c = idx & 127;
        W[(long)(k * 128 + r) * ld + k * 128 + c] = invd[(long)k * 16384 + idx];
    }
}

// M[c][r] = M[r][c] for c < r, tile by tile through LDS
__global__ __launch_bounds__(256) void symmetrize_kernel(double *M, long ld, int nt) {
    __shared__ double tile[32][33];
    // lower-triangle tile index
    const int t = blockIdx.x;
    int tr = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(tr + 1) * (tr + 2) / 2 <= t) ++tr;
    while ((long)tr * (tr + 1) / 2 > t) --tr;
    const int tc = t - tr * (tr + 1) / 2;
    (void)nt;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8) tile[y][tx] = M[(long)(tr * 32 + y) * ld + tc * 32 + tx];
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int r = tc * 32 + y, c = tr * 32 + tx;   // transposed position
        if (c > r) M[(long)r * ld + c] = tile[tx][y];
    }
}

// JAICOV_FACTOR_FORM = streams | two_step | one_kernel | chain2 | chain3: the forms of the factorisation other than the default (dataflow, chain
// form), each named by a case of tests/test_gpu_parity.py (test_factor_tile_by_tile, test_config3_step_against_oracle).  Read at
// every call: the tests switch it between engines of one process.
namespace {
std::mutex g_stream_mutex;
std::map<std::pair<int, int>, std::vector<hipStream_t>> g_stream_pool;      // (device, kind) -> idle streams
// The idle streams are destroyed when this library's static objects are (at exit, BEFORE those of the HIP runtime it depends on and before
// a profiler's finalisation: under rocprofv3 a process that left streams to the runtime's own teardown ended in a segmentation fault
// inside __cxa_finalize, after the tool had written its output).
std::vector<hipStream_t> g_all_streams;       // every stream stream_acquire has created and nobody has destroyed: idle ones and those of live (or leaked) engines
std::map<hipStream_t, int> g_stream_kind;      // ... its kind (census below)
std::map<hipStream_t, int> g_stream_device;    // ... and the device each belongs to (a stream is filed under ITS device whatever the caller's current one is)
struct StreamPoolCleanup {
    ~StreamPoolCleanup() {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (hipStream_t s : g_all_streams) (void)hipStreamDestroy(s);
        g_all_streams.clear();
        g_stream_device.clear();
        g_stream_kind.clear();
        g_stream_pool.clear();
    }
} g_stream_pool_cleanup;
}

hipStream_t stream_acquire(int kind) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        std::vector<hipStream_t> &idle = g_stream_pool[{dev, kind}];
        if (!idle.empty()) {
            hipStream_t s = idle.back();
            idle.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    hipError_t err = hipSuccess;
    if (kind == STREAM_PLAIN) {
        err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    } else if (kind == STREAM_HIGH_PRIORITY) {
        int least = 0, greatest = 0;
        err = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (err == hipSuccess) err = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
    } else {
        // CU reservation (measured on MI355X/ROCm 7.2: bit i of the 256-bit mask is CU i/8 of XCD i%8): the trailing
        // updates get CUs 0..30 of every XCD, the diagonal-block kernel CU 31 of every XCD.  (Reserving fewer CUs was
        // measured: one or two reserved CUs cost 5 ms per factorisation at config 4, eight cost the update 3 % of the chip.)
        uint32_t mask[8];
        for (int w = 0; w < 8; w++) mask[w] = kind == STREAM_UPDATE_CUS ? 0xFFFFFFFFu : 0u;
        mask[7] = kind == STREAM_UPDATE_CUS ? 0x00FFFFFFu : 0xFF000000u;
        err = hipExtStreamCreateWithCUMask(&s, 8, mask);
    }
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    g_all_streams.push_back(s);
    g_stream_device[s] = dev;
    g_stream_kind[s] = kind;
    return s;
}

void stream_release(int kind, hipStream_t s) {
    if (!s) return;
    if (hipStreamSynchronize(s) != hipSuccess) {      // a stream in an error state is not worth keeping
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(g_stream_mutex);
            g_all_streams.erase(std::remove(g_all_streams.begin(), g_all_streams.end(), s), g_all_streams.end());
            g_stream_device.erase(s);
            g_stream_kind.erase(s);
        }
        hipStreamDestroy(s);
        return;
    }
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        const auto it = g_stream_device.find(s);
        std::vector<hipStream_t> &idle = g_stream_pool[{it == g_stream_device.end() ? 0 : it->second, kind}];
        const bool masked = kind == STREAM_UPDATE_CUS || kind == STREAM_DIAGONAL_CUS;
        if (!masked || (int)idle.size() < STREAM_MASKED_IDLE_MAX) {
            idle.push_back(s);
            return;
        }
        // a CU-masked stream beyond the few that are kept: its hardware queue goes back to the device (dense.h)
        g_all_streams.erase(std::remove(g_all_streams.begin(), g_all_streams.end(), s), g_all_streams.end());
        g_stream_device.erase(s);
        g_stream_kind.erase(s);
    }
    (void)hipStreamDestroy(s);
}

// tests / DESIGN.md: streams of this library in the process, [kind] alive (held by solvers or idle) and [4 + kind] idle in the pool;
// kinds: 0 plain, 1 high priority, 2 CU-masked (trailing updates), 3 CU-masked (diagonal blocks) -- the masked ones are hardware queues
extern "C" void jaicov_debug_stream_census(int *out8) {
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    for (int i = 0; i < 8; i++) out8[i] = 0;
    for (const auto &kv : g_stream_kind)
        if (kv.second >= 0 && kv.second < 4) out8[kv.second]++;
    for (const auto &kv : g_stream_pool)
        if (kv.first.second >= 0 && kv.first.second < 4) out8[4 + kv.first.second] += (int)kv.second.size();
}

int factor_form() {
    const char *e = getenv("JAICOV_FACTOR_FORM");
    if (!e) return FACTOR_DEFAULT;
    if (!strcmp(e, "streams")) return FACTOR_STREAMS;
    if (!strcmp(e, "two_step")) return FACTOR_TWO_STEP;
    if (!strcmp(e, "one_kernel")) return FACTOR_ONE_KERNEL;
    if (!strcmp(e, "chain2")) return FACTOR_CHAIN2;          // the chain form with two chain workgroups (no third one for the second subdiagonal)
    if (!strcmp(e, "chain3")) return FACTOR_CHAIN3;          // ... with three, whatever the order (default: below 80 block columns)
    return FACTOR_DEFAULT;
}

// ---------------------------------------------------------------------------------------------------------------
hipError_t DenseSolver::init(hipStream_t s, int n_padded, bool with_inverse, bool with_rhs_rows, const DenseSolver *share) {
    stream = s;
    aug = with_rhs_rows;
    nfact = n_padded;
    n = n_padded + (aug ? 128 : 0);
    ld = n;
    owns = true;
    const size_t sq = (size_t)n * ld * sizeof(double);
    HIPCHK(hipMalloc(&L, sq));
    HIPCHK(hipMalloc(&invd, (size_t)(nfact / 128) * 16384 * sizeof(double)));
    HIPCHK(hipMemset(invd, 0, (size_t)(nfact / 128) * 16384 * sizeof(double)));
    HIPCHK(hipMalloc(&xch, (size_t)nfact * sizeof(double)));
    HIPCHK(hipEventCreateWithFlags(&pm_e0, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&pm_done, hipEventDisableTiming));
    HIPCHK(hipMalloc(&pm, (size_t)(nfact / 128) * 2 * CH_PM * 16384 * sizeof(double)));
    HIPCHK(hipMemset(pm, 0, (size_t)(nfact / 128) * 2 * CH_PM * 16384 * sizeof(double)));
    pm_ready = false;
    HIPCHK(hipMalloc(&d_info, sizeof(int)));

    // Which factorisation: both are bound by a chain of one link per block column while the order is small (dataflow: the
    // chain workgroup's potrf -> solve -> update, ~80-95 us; streams: diagonal kernel + two dependent launches, ~80 us), the
    // dataflow form then pays its start-up (two launches, a host handshake, ~0.15 ms).  Measured on MI355X
    // (scripts/flow_trace.py, ms per factorisation dataflow / streams): order 1024 0.76 / 0.62, 2048 1.32 / 1.28,
    // 3072 1.94 / 2.00, 3712 2.37 / 2.50, 5120 3.33 / 3.75, 6144 4.07 / 4.78, 8192 5.95 / 7.92, 15104 22.3 / 26.3
    // -> from 24 block columns on (rounds 2 and 3).  Round 4 (diagonal block 41 -> 27 us, third chain workgroup below 48 block columns):
    // order 640 0.43 / 0.39, 1024 0.65 / 0.59, 1152 0.70 / 0.69, 1280 0.75 / 0.77, 1408 0.82 / 0.83, 1536 0.82 / 0.92, 2048 1.10 / 1.25,
    // 2560 1.32 / 1.59, 3072 1.55 / 1.94 -> from 12 block columns on.
    const int flow_from = getenv("JAICOV_FLOW_MIN_BLOCKS") ? atoi(getenv("JAICOV_FLOW_MIN_BLOCKS")) : 12;      // (the tests lower it)
    const bool flow_wanted = factor_form() != FACTOR_STREAMS && nfact / 128 >= flow_from;
    borrowed_streams = share != nullptr && share->owns && !share->borrowed_streams && share->pstream != nullptr;
    own_ustream = own_dstream = false;
    if (borrowed_streams) {
        pstream = share->pstream;
    } else {
        pstream = stream_acquire(STREAM_HIGH_PRIORITY);
        if (!pstream) return hipErrorUnknown;
    }
    {
        // reserved CUs: CU 31 of every XCD for the diagonal blocks, the other 31 for the trailing updates (stream_acquire).  The masked
        // stream of the trailing updates is only held by a solver that factorises by the stream-scheduled form: the dataflow form
        // (flow_wanted) launches its tile kernel on the ordinary stream, and every CU-masked stream is a hardware queue (dense.h).
        // A masked stream the other solver of the engine holds is used (only where this solver needs one); what it does not hold is
        // acquired here and is then this solver's own to release.
        if (nfact >= 2048 || flow_wanted) {
            reserved_cus = 8;
            dstream = borrowed_streams ? share->dstream : nullptr;
            if (!dstream) { dstream = stream_acquire(STREAM_DIAGONAL_CUS); own_dstream = dstream != nullptr; }
            if (!flow_wanted) {
                ustream = borrowed_streams ? share->ustream : nullptr;
                if (!ustream) { ustream = stream_acquire(STREAM_UPDATE_CUS); own_ustream = ustream != nullptr; }
            }
            if (!dstream || (!flow_wanted && !ustream)) {
                if (own_ustream) stream_release(STREAM_UPDATE_CUS, ustream);
                if (own_dstream) stream_release(STREAM_DIAGONAL_CUS, dstream);
                ustream = dstream = nullptr;
                own_ustream = own_dstream = false;
            }
        }
    }
    // XCD-aware tile orders of every trailing-update size this solver can launch (one allocation, built here so that the
    // first factorisation does not stop for host work)
    if (xcd_maps && n / 128 >= 24) {
        std::vector<int2> all;
        std::vector<std::pair<int, std::pair<size_t, int>>> where;
        for (int T = 24; T <= n / 128; T++) {
            const std::vector<int2> m = xcd_tile_map(T);
            where.push_back({T, {all.size(), (int)m.size()}});
            all.insert(all.end(), m.begin(), m.end());
        }
        HIPCHK(hipMalloc(&tile_map_store, all.size() * sizeof(int2)));
        HIPCHK(hipMemcpy(tile_map_store, all.data(), all.size() * sizeof(int2), hipMemcpyHostToDevice));
        for (auto &w : where) tile_maps.emplace(w.first, std::make_pair(tile_map_store + w.second.first, w.second.second));
    }
    if (with_inverse) {
        HIPCHK(hipMalloc(&W, sq));
        HIPCHK(hipMalloc(&Q, sq));
    }
    if (flow_wanted && dstream) {
        hipError_t fe = flow_init();
        if (fe != hipSuccess) {      // the stream-scheduled factorisation remains: it wants the masked stream of the trailing updates after all
            flow_release(); (void)hipGetLastError();
            if (!ustream) { ustream = stream_acquire(STREAM_UPDATE_CUS); own_ustream = ustream != nullptr; }
        }
    }
    return hipSuccess;
}

void DenseSolver::release() {
    if (!owns) return;
    flow_release();
    hipFree(L); hipFree(invd); hipFree(d_info); hipFree(W); hipFree(Q); hipFree(pm); hipFree(xch);
    hipFree(tile_map_store);
    tile_map_store = nullptr;
    for (auto &kv : trtri_maps) hipFree(kv.second);
    trtri_maps.clear();
    tile_maps.clear();
    for (auto ev : prof_ev) hipEventDestroy(ev);
    prof_ev.clear();
    for (auto ev : sync_ev) hipEventDestroy(ev);
    sync_ev.clear();
    if (!borrowed_streams) stream_release(STREAM_HIGH_PRIORITY, pstream);
    if (own_ustream) stream_release(STREAM_UPDATE_CUS, ustream);
    if (own_dstream) stream_release(STREAM_DIAGONAL_CUS, dstream);
    pstream = ustream = dstream = nullptr;
    borrowed_streams = own_ustream = own_dstream = false;
    L = invd = W = Q = nullptr;
    if (pm_e0) hipEventDestroy(pm_e0);
    if (pm_done) hipEventDestroy(pm_done);
    pm_e0 = pm_done = nullptr; pm_wait = false;
    pm = nullptr; pm_ready = false; xch = nullptr;
    d_info = nullptr;
    owns = false;
}

// Two-level right-looking Cholesky with one-panel lookahead.
// Outer panels of `nbo` columns (trailing update with K = nbo: enough flops per byte of C to be MFMA-bound); inside a
// panel a left-looking sweep over 128-column blocks:
//   block column kk -= L[kk:n, k0:kk] L[kk:kk+128, k0:kk]'     (fp64 MFMA GEMM, K grows to nbo-128)
//   diagonal block factor + inverse (one workgroup, LDS)        L21 = A21 inv(L11)'  (GEMM with the inverse)
// Lookahead: the trailing update of panel s is split into (a) the columns of panel s+1 and (b) the rest; panel s+1 is
// factored on a second, high-priority stream as soon as (a) is done, while (b) keeps the chip busy.
hipEvent_t DenseSolver::next_event() {
    if (ev_used >= sync_ev.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        sync_ev.push_back(e);
    }
    return sync_ev[ev_used++];
}

// Panel factorisation of block columns [K0,K1).  GEMMs go to `st`; when a reserved-CU stream exists the diagonal-block
// kernel (which needs a whole CU's LDS and would otherwise starve behind the trailing update's workgroups) runs there,
// chained with events.
hipError_t DenseSolver::panel(hipStream_t st, int K0, int K1) {
    // In the tail (short trailing updates, the panel chain is the critical path) the diagonal kernel stays on the panel
    // stream: the two event hops per block cost more than the contention they avoid.
    constexpr int tail_rows = 6144;
    for (int k = K0; k < K1; k++) {
        double *Akk = L + (long)(k * 128) * ld + k * 128;
        const int rows_k = n - k * 128;
        // ... except for the first block of a panel: it becomes ready at the moment the trailing update is launched, and a
        // workgroup that wants a whole CU's LDS then waits for the update's first tiles to retire (130 us instead of 49)
        constexpr bool first_split = true;
        // (only while the update still has more workgroups than the chip has slots: 31 tile rows = 496 tiles)
        const bool split = dstream != nullptr && st == pstream && (rows_k > tail_rows || (first_split && k == K0 && rows_k > 4096));
        // While the trailing update still hides the panel (many rows left) the panel GEMMs take the 128-tile: it costs
        // the update fewer CU slots per flop than the 64-tile latency variant, which is for the critical-path regime.
        constexpr int bulk_rows = 9216;
        // in between (the update still shares the chip) the 64-tile; the 32-tile only once the panel runs alone
        const int small = (st == pstream && rows_k > bulk_rows) ? 0 : ((st == pstream && rows_k > tail_rows) ? 1 : -1);
        if (k > K0) {
            GemmArgs c{};
            c.A = L + (long)(k * 128) * ld + K0 * 128; c.lda = ld;      // L[k*128:n, K0*128 : k*128]
            c.B = c.A; c.ldb = ld;                                       // first 128 rows of the same strip
            c.C = Akk; c.ldc = ld; c.M = rows_k; c.N = 128; c.K = (k - K0) * 128;
            c.alpha = -1.0; c.beta = 1.0; c.lower_only = 0; c.kmode = KMODE_FULL;
            HIPCHK(gemm_f64(st, LAY_KC, LAY_KC, c, 1, small));
        }
        hipStream_t ds = st;
        if (split) {
            hipEvent_t e = next_event();
            HIPCHK(hipEventRecord(e, st));
            HIPCHK(hipStreamWaitEvent(dstream, e, 0));
            ds = dstream;
        }
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, ds, Akk, ld, invd + (long)k * 16384, d_info, k, 0);
        if (split) {
            hipEvent_t e = next_event();
            HIPCHK(hipEventRecord(e, dstream));
            HIPCHK(hipStreamWaitEvent(st, e, 0));
        }
        const int rows = rows_k - 128;
        if (rows <= 0) break;
        double *A21 = L + (long)((k + 1) * 128) * ld + k * 128;
        GemmArgs g{};
        // L21 = A21 * inv(L11)'   (in place: one column tile, every workgroup reads and writes only its own rows)
        g.A = A21; g.lda = ld; g.B = invd + (long)k * 16384; g.ldb = 128; g.C = A21; g.ldc = ld;
        g.M = rows; g.N = 128; g.K = 128; g.alpha = 1.0; g.beta = 0.0; g.lower_only = 0; g.kmode = KMODE_FULL;
        HIPCHK(gemm_f64(st, LAY_KC, LAY_KC, g, 1, small));
    }
    return hipGetLastError();
}

hipError_t DenseSolver::timed_gemm(hipStream_t st, const GemmArgs &u, double flops, int small) {
    if (profile) {
        if (prof_used + 2 > prof_ev.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
            prof_ev.push_back(a); prof_ev.push_back(b);
        }
        HIPCHK(hipEventRecord(prof_ev[prof_used], st));
    }
    HIPCHK(gemm_f64(st, LAY_KC, LAY_KC, u, 1, small, 1));   // own kernel symbol: gemm_f64_kernel<0, 0, 128, 128, 1> (64, 64 in the tail)
    if (profile) {
        HIPCHK(hipEventRecord(prof_ev[prof_used + 1], st));
        prof_flops.push_back(flops);
        prof_used += 2;
    }
    return hipSuccess;
}

int DenseSolver::first_panel_cols() const {
    const int nb = nfact / 128;
    const int k = nbo / 128 > 0 ? nbo / 128 : 1;
    return 128 * (k < nb ? k : nb);
}

// Call before the first WRITE into L for a new factorisation (the engine's scaled copy, a dispersion load): the Ft half of the last
// premultiply() may still be reading the old factor on pstream.  potrf() does the same for callers that fill L through potrf's own
// first tile load (flow_set_source), where nothing writes L earlier.
hipError_t DenseSolver::begin_refactor() {
    pm_ready = false;
    if (pm_wait) { HIPCHK(hipStreamWaitEvent(stream, pm_done, 0)); pm_wait = false; }
    return hipSuccess;
}

hipError_t DenseSolver::potrf(hipEvent_t first_ready, hipEvent_t all_ready) {
    HIPCHK(begin_refactor());
    if (flow_ready) return potrf_flow(all_ready);
    return potrf_streams(first_ready, all_ready);
}

hipError_t DenseSolver::potrf_streams(hipEvent_t first_ready, hipEvent_t all_ready) {
    const int nb = nfact / 128;   // diagonal blocks; rows run to n (the right-hand-side rows below the matrix included)
    // Panel width: one width for the whole factorisation (wide panels while many rows remain, narrow ones in the tail: measured
    // flat +- 0.3 ms at config 4, DESIGN_HISTORY.md)
    const int bo = nbo / 128 > 0 ? nbo / 128 : 1;
    auto width = [&](int) { return bo; };
    const bool la = lookahead && pstream != nullptr && nb > bo;
    ev_used = 0;
    hipStream_t sp = la ? pstream : stream;                 // panel GEMMs
    hipStream_t su = la && ustream ? ustream : stream;      // trailing updates (all CUs but the reserved ones)
    hipEvent_t e_start = next_event();
    if (la) {
        if (first_ready && all_ready) {                     // the caller is still filling the columns behind the first panel
            HIPCHK(hipStreamWaitEvent(sp, first_ready, 0));
            if (su != stream) HIPCHK(hipStreamWaitEvent(su, all_ready, 0));
        } else {
            HIPCHK(hipEventRecord(e_start, stream));
            HIPCHK(hipStreamWaitEvent(sp, e_start, 0));
            if (su != stream) HIPCHK(hipStreamWaitEvent(su, e_start, 0));
        }
    }
    HIPCHK(hipMemsetAsync(d_info, 0, sizeof(int), sp));    // before the first diagonal kernel, which runs on sp or behind it
    int K0 = 0, K1 = width(0) < nb ? width(0) : nb;
    HIPCHK(panel(sp, 0, K1));
    hipEvent_t e_panel = next_event();
    if (la) HIPCHK(hipEventRecord(e_panel, sp));
    while (K1 < nb) {
        const int K2 = (K1 + width(K1) < nb) ? K1 + width(K1) : nb;     // end of the next panel
        if (la) HIPCHK(hipStreamWaitEvent(su, e_panel, 0));
        const int Kw = (K1 - K0) * 128;
        // (a) columns of the next panel: rows >= K1, cols [K1,K2)
        GemmArgs a{};
        a.A = L + (long)(K1 * 128) * ld + K0 * 128; a.lda = ld; a.B = a.A; a.ldb = ld;
        a.C = L + (long)(K1 * 128) * ld + K1 * 128; a.ldc = ld;
        a.M = n - K1 * 128; a.N = (K2 - K1) * 128; a.K = Kw; a.alpha = -1.0; a.beta = 1.0; a.lower_only = 0; a.kmode = KMODE_FULL;
        HIPCHK(gemm_f64(su, LAY_KC, LAY_KC, a));
        if (la) {
            hipEvent_t e = next_event();
            HIPCHK(hipEventRecord(e, su));
            HIPCHK(hipStreamWaitEvent(sp, e, 0));
        }
        HIPCHK(panel(sp, K1, K2));
        if (la) {
            e_panel = next_event();
            HIPCHK(hipEventRecord(e_panel, sp));
        }
        // (b) the rest of the trailing matrix: rows, cols >= K2 (lower tiles)
        const int rows = n - K2 * 128;
        if (rows > 0 && K2 < nb) {
            GemmArgs u{};
            u.A = L + (long)(K2 * 128) * ld + K0 * 128; u.lda = ld; u.B = u.A; u.ldb = ld;
            u.C = L + (long)(K2 * 128) * ld + K2 * 128; u.ldc = ld;
            u.M = rows; u.N = rows; u.K = Kw; u.alpha = -1.0; u.beta = 1.0; u.lower_only = 1; u.kmode = KMODE_FULL;
            if (xcd_maps && rows / 128 >= 24) {
                const int T = rows / 128;
                auto it = tile_maps.find(T);
                if (it != tile_maps.end()) {
                    u.tile_map = it->second.first;
                    u.n_map = it->second.second;
                }
            }
            HIPCHK(timed_gemm(su, u, (double)rows * ((double)rows + 1.0) * (double)Kw));
        }
        K0 = K1;
        K1 = K2;
    }
    if (la) {   // the main stream continues only after the last panel and the last update
        hipEvent_t e1 = next_event(), e2 = next_event();
        HIPCHK(hipEventRecord(e1, sp));
        HIPCHK(hipStreamWaitEvent(stream, e1, 0));
        if (su != stream) {
            HIPCHK(hipEventRecord(e2, su));
            HIPCHK(hipStreamWaitEvent(stream, e2, 0));
        }
    }
    return hipGetLastError();
}

// The polling-wave chains pay two batched GEMM launches per factorisation (~0.1 ms) for ~1.2 us per link and chain: from this many
// block columns on (config 2, 6 block columns: 0.085 -> 0.14 ms per pass with them; config 3, 29: 0.28 -> 0.27)
static int chain8_min_nb() {         // (read at every call: the tests lower it)
    const char *e = getenv("JAICOV_CHAIN8_MIN_NB");
    return e ? atoi(e) : 24;
}

// the backward chain for one right-hand side: two workgroups per block column when the whole grid is resident at once
bool DenseSolver::chain8_split() const {
    const int nb = nfact / 128;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        if (cus <= 0) cus = 1;
    }
    return xch && nb >= 8 && 2 * nb <= cus;
}

hipError_t DenseSolver::launch_chain8(const double *Zrow, double *X, const int *abort_word, long long *trace) {
    const int nb = nfact / 128;
    if (chain8_split()) {
        HIPCHK(hipMemsetAsync(xch, 0xFF, (size_t)nfact * sizeof(double), stream));
        hipLaunchKernelGGL(backsolve_chain8_kernel<2>, dim3(2 * nb), dim3(CHAIN8_THREADS), 0, stream, L, ld, invd, pm, Zrow, X, xch, nb, abort_word, trace);
    } else {
        hipLaunchKernelGGL(backsolve_chain8_kernel<1>, dim3(nb), dim3(CHAIN8_THREADS), 0, stream, L, ld, invd, pm, Zrow, X, xch, nb, abort_word, trace);
    }
    return hipGetLastError();
}

// The CH_PM blocks below every diagonal block multiplied into its inverse, once per factorisation, for backsolve_chain8_kernel:
// P_m[k] = L[k+m][k] W_k, m = 1 .. CH_PM.  pm = [P_1[0] .. P_1[nb-1]][P_2[0] ..] .., 128 x 128 row-major each (blocks that do not
// exist stay zero).  ONE launch: batches (k, m - 1), those with k + m >= nb left out.
hipError_t DenseSolver::premultiply() {
    if (pm_ready || !pm) return hipSuccess;
    const int nb = nfact / 128;
    if (nb > 1) {
        GemmArgs p{};
        p.A = L + (long)128 * ld; p.lda = ld; p.strideA = 128 * (ld + 1); p.strideA2 = (long)128 * ld;     // L[k+m][k]   (KC)
        p.B = invd; p.ldb = 128; p.strideB = 16384; p.strideB2 = 0;                                       // W_k (k, j) row-major (XC)
        p.C = pm; p.ldc = 128; p.strideC = 16384; p.strideC2 = (long)nb * 16384;
        p.M = p.N = p.K = 128; p.alpha = 1.0; p.beta = 0.0; p.kmode = KMODE_FULL;
        p.batch_sum_limit = nb - 1;                 // k + (m - 1) <= nb - 2
        HIPCHK(gemm_f64(stream, LAY_KC, LAY_XC, p, nb - 1, 0, 0, std::min(CH_PM, nb - 1)));
        // Ft_m[k] = (W_k L[k][k-m])' = L[k][k-m]' W_k', k >= m: behind the P blocks, batches (k - m, m - 1)
        GemmArgs f{};
        f.A = L + (long)128 * ld; f.lda = ld; f.strideA = 128 * (ld + 1); f.strideA2 = (long)128 * ld;     // L[k][k-m] read transposed (XC)
        f.B = invd + 16384; f.ldb = 128; f.strideB = 16384; f.strideB2 = 16384;                           // W_k' (kk, j) = W_k[j][kk] (KC)
        f.C = pm + (size_t)CH_PM * nb * 16384 + 16384; f.ldc = 128; f.strideC = 16384; f.strideC2 = (long)nb * 16384 + 16384;
        f.M = f.N = f.K = 128; f.alpha = 1.0; f.beta = 0.0; f.kmode = KMODE_FULL;
        f.batch_sum_limit = nb - 1;
        // the Ft blocks are needed by the forward chain of the refinement only: on the side stream, beside the first backward chain
        // and the residual (solve_rhs waits for pm_done)
        constexpr bool side = true;
        if (side && pstream && pm_e0 && pm_done) {
            HIPCHK(hipEventRecord(pm_e0, stream));
            HIPCHK(hipStreamWaitEvent(pstream, pm_e0, 0));
            HIPCHK(gemm_f64(pstream, LAY_XC, LAY_KC, f, nb - 1, 0, 0, std::min(CH_PM, nb - 1)));
            HIPCHK(hipEventRecord(pm_done, pstream));
            pm_wait = true;
        } else {
            HIPCHK(gemm_f64(stream, LAY_XC, LAY_KC, f, nb - 1, 0, 0, std::min(CH_PM, nb - 1)));
        }
    }
    pm_ready = true;
    return hipGetLastError();
}

hipError_t DenseSolver::backsolve_aug(double *X, long xs, int nrhs) {
    if (!aug || nrhs < 1 || nrhs > DENSE_MAX_RHS) return hipErrorInvalidValue;
    const int nb = nfact / 128;
    HIPCHK(hipMemsetAsync(X, 0xFF, (size_t)nrhs * xs * sizeof(double), stream));   // "not yet published"
    const int *ab = flow_ready ? flow_flags + 1 : nullptr;      // cholflow.hip FLOW_ABORT
    if (nrhs <= 1 && pm && nb >= chain8_min_nb()) {
        HIPCHK(premultiply());
        HIPCHK(launch_chain8(rhs_row(0), X, ab, nullptr));
    } else if (nrhs <= 1) hipLaunchKernelGGL(backsolve_chain_kernel<1>, dim3(nb), dim3(256), 0, stream, L, ld, invd, rhs_row(0), ld, X, xs, nb, nrhs, ab);
    else if (nrhs <= 2) hipLaunchKernelGGL(backsolve_chain_kernel<2>, dim3(nb), dim3(256), 0, stream, L, ld, invd, rhs_row(0), ld, X, xs, nb, nrhs, ab);
    else if (nrhs <= 4) hipLaunchKernelGGL(backsolve_chain_kernel<4>, dim3(nb), dim3(256), 0, stream, L, ld, invd, rhs_row(0), ld, X, xs, nb, nrhs, ab);
    else hipLaunchKernelGGL(backsolve_chain_kernel<8>, dim3(nb), dim3(256), 0, stream, L, ld, invd, rhs_row(0), ld, X, xs, nb, nrhs, ab);
    return hipGetLastError();
}

// x = (L L')^-1 b for one right-hand side against the factor at hand: forward chain into `tmp`, backward chain into X
// (both of length nfact; b, tmp, X distinct device vectors).
hipError_t DenseSolver::solve_rhs(const double *b, double *tmp, double *X) {
    const int nb = nfact / 128;
    HIPCHK(hipMemsetAsync(tmp, 0xFF, (size_t)nfact * sizeof(double), stream));
    HIPCHK(hipMemsetAsync(X, 0xFF, (size_t)nfact * sizeof(double), stream));
    if (pm && nb >= chain8_min_nb()) {
        HIPCHK(premultiply());
        static const bool tracing = getenv("JAICOV_CHAIN_TRACE") != nullptr;     // development: link times of the backward chain on stderr
        long long *tr = nullptr;
        if (tracing) { HIPCHK(hipMalloc(&tr, (size_t)9 * nb * sizeof(long long))); HIPCHK(hipMemsetAsync(tr, 0, (size_t)9 * nb * sizeof(long long), stream)); }
        if (pm_wait) { HIPCHK(hipStreamWaitEvent(stream, pm_done, 0)); pm_wait = false; }
        if (chain8_split()) {
            HIPCHK(hipMemsetAsync(xch, 0xFF, (size_t)nfact * sizeof(double), stream));
            hipLaunchKernelGGL(forwardsolve_chain8_kernel, dim3(2 * nb), dim3(CHAIN8_THREADS), 0, stream, L, ld, invd, pm + (size_t)CH_PM * nb * 16384, b, tmp, xch, nb);
        } else {
            hipLaunchKernelGGL(forwardsolve_chain_kernel, dim3(nb), dim3(256), 0, stream, L, ld, invd, b, tmp, nb);
        }
        HIPCHK(launch_chain8(tmp, X, flow_ready ? flow_flags + 1 : nullptr, tr));   // behind an abandoned factorisation: leave at once (as backsolve_aug)
        if (tracing) {
            std::vector<long long> h((size_t)9 * nb);
            HIPCHK(hipMemcpyAsync(h.data(), tr, h.size() * sizeof(long long), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            hipFree(tr);
            fprintf(stderr, "[jaicov chain trace] backward, nb %d, link times in 10 ns:", nb);
            for (int q = 1; q < nb; q++) fprintf(stderr, " %lld", h[q] - h[q - 1]);
            fprintf(stderr, "\n[jaicov chain trace] per position: start, stream done (wave 1), barrier 1, barrier 2, u ready, tails done, polling wave done, stream done (wave 7), relative to the predecessor's publication (10 ns)\n");
            for (int q = 1; q < nb; q += (q < 12 ? 1 : 9)) {
                fprintf(stderr, "   pos %3d:", q);
                for (int c = 0; c < 8; c++) fprintf(stderr, " %6lld", h[(size_t)nb + 8 * q + c] - h[q - 1]);
                fprintf(stderr, "   published %6lld\n", h[q] - h[q - 1]);
            }
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(forwardsolve_chain_kernel, dim3(nb), dim3(256), 0, stream, L, ld, invd, b, tmp, nb);
    hipLaunchKernelGGL(backsolve_chain_kernel<1>, dim3(nb), dim3(256), 0, stream, L, ld, invd, tmp, (long)nfact, X, (long)nfact, nb, 1, flow_ready ? flow_flags + 1 : (const int *)nullptr);
    return hipGetLastError();
}

// W = L^-1 (lower), level by level from the inverted diagonal blocks upwards.  At level h (blocks of h x 128 columns)
// every pair [lo, lo+h) | [lo+h, lo+2h) is merged independently:  W21 = -W22 (L21 W11).  All full pairs of a level
// have the same shape and a constant address stride, so a level is two batched launches (product T = L21 W11 into the
// W21-position of Q, which is free until lauum() fills it; then W21 = -W22 T); a ragged last pair gets its own two.
// Tile orders of the two products of a merge, longest k-range first (the tiles of these launches do triangular work: in
// T = L21 W11 the k-range of a tile shrinks with its COLUMN, in W21 = -W22 T it grows with its ROW).  In the kernel's
// row-major default the last tiles dispatched include the longest ones, and the launch ends with one long tile's time on a
// nearly empty chip (1-2 ms of the top level's 10 ms at config 4); longest-first leaves only short tiles for the tail.
const int2 *DenseSolver::trtri_tile_order(int tm, int tn, int kind) {
    const long key = ((long)kind << 40) | ((long)tm << 20) | tn;
    auto it = trtri_maps.find(key);
    if (it != trtri_maps.end()) return it->second;
    std::vector<int2> m;
    m.reserve((size_t)tm * tn);
    if (kind == 0) {            // KMODE_GE_COL: column 0 has the longest k-range
        for (int j = 0; j < tn; j++)
            for (int i = 0; i < tm; i++) m.push_back(make_int2(i, j));
    } else {                    // KMODE_LE_ROW: the last row has the longest k-range
        for (int i = tm - 1; i >= 0; i--)
            for (int j = 0; j < tn; j++) m.push_back(make_int2(i, j));
    }
    int2 *d = nullptr;
    if (hipMalloc(&d, m.size() * sizeof(int2)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, m.data(), m.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
    trtri_maps.emplace(key, d);
    return d;
}

hipError_t DenseSolver::trtri() {
    const int nb = nfact / 128;
    if (!Q) return hipErrorInvalidValue;
    constexpr bool lpt = true;      // longest k-range first (row-major order: 24.7 instead of 18.5 ms at config 4)
    HIPCHK(hipMemsetAsync(W, 0, (size_t)n * ld * sizeof(double), stream));
    hipLaunchKernelGGL(copy_diag_blocks_kernel, dim3(nb), dim3(256), 0, stream, invd, W, ld);
    for (int h = 1; h < nb; h *= 2) {
        const int full = nb / (2 * h);                         // pairs with both halves complete
        const long pair_stride = (long)(2 * h) * 128 * (ld + 1);
        auto merge = [&](int lo, int mid, int hi, int batch) -> hipError_t {
            const int M = (hi - mid) * 128, N = (mid - lo) * 128;
            GemmArgs t{};
            t.A = L + (long)(mid * 128) * ld + lo * 128; t.lda = ld;          // L21 (KC)
            t.B = W + (long)(lo * 128) * ld + lo * 128; t.ldb = ld;           // W11 (k,j) row-major = XC, lower-triangular
            t.C = Q + (long)(mid * 128) * ld + lo * 128; t.ldc = ld;
            t.M = M; t.N = N; t.K = N; t.alpha = 1.0; t.beta = 0.0; t.kmode = KMODE_GE_COL;
            t.strideA = t.strideB = t.strideC = pair_stride;
            if (lpt && (M / 128) * (N / 128) >= 64) { t.tile_map = trtri_tile_order(M / 128, N / 128, 0); t.n_map = t.tile_map ? (M / 128) * (N / 128) : 0; }
            HIPCHK(gemm_f64(stream, LAY_KC, LAY_XC, t, batch));
            GemmArgs w{};
            w.A = W + (long)(mid * 128) * ld + mid * 128; w.lda = ld;         // W22 (KC), lower-triangular
            w.B = t.C; w.ldb = ld;                                            // T (k,j) row-major = XC
            w.C = W + (long)(mid * 128) * ld + lo * 128; w.ldc = ld;
            w.M = M; w.N = N; w.K = M; w.alpha = -1.0; w.beta = 0.0; w.kmode = KMODE_LE_ROW;
            w.strideA = w.strideB = w.strideC = pair_stride;
            if (lpt && (M / 128) * (N / 128) >= 64) { w.tile_map = trtri_tile_order(M / 128, N / 128, 1); w.n_map = w.tile_map ? (M / 128) * (N / 128) : 0; }
            return gemm_f64(stream, LAY_KC, LAY_XC, w, batch);
        };
        if (full > 0) HIPCHK(merge(0, h, 2 * h, full));
        const int lo = full * 2 * h, mid = lo + h;
        if (mid < nb) HIPCHK(merge(lo, mid, nb, 1));          // ragged pair: second half shorter than h
    }
    return hipGetLastError();
}

hipError_t DenseSolver::lauum() {
    GemmArgs g{};
    g.A = W; g.lda = ld; g.B = W; g.ldb = ld; g.C = Q; g.ldc = ld;
    g.M = nfact; g.N = nfact; g.K = nfact; g.alpha = 1.0; g.beta = 0.0; g.lower_only = 1; g.kmode = KMODE_GE_ROW;
    return gemm_f64(stream, LAY_XC, LAY_XC, g);
}

hipError_t DenseSolver::symmetrize(double *M) {
    const int nt = nfact / 32;
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nt * (nt + 1) / 2), dim3(256), 0, stream, M, ld, nt);
    return hipGetLastError();
}

void DenseSolver::prof_collect() {
    for (size_t i = 0; i + 1 < prof_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, prof_ev[i], prof_ev[i + 1]) == hipSuccess) {
            stat_launches += 1.0;
            stat_ms += ms;
            stat_flops += prof_flops[i / 2];
        }
    }
    prof_used = 0;
    prof_flops.clear();
    if (flow_timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, flow_t0, flow_t1) == hipSuccess) {
            const double o = flops_order > 0 ? flops_order : (double)nfact;
            stat_launches += 1.0;
            stat_ms += ms;
            stat_flops += o * o * o / 3.0;
        }
        flow_timed = false;
    }
}

int DenseSolver::fetch_info() {
    int h = -1;
    if (hipMemcpyAsync(&h, d_info, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
    int cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // cholflow.hip's control words: [1] abort, [4] / [5] flags that the plain poll missed
    if (flow_ready && hipMemcpyAsync(cw, flow_flags, sizeof(cw), hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    const int ab = cw[1];
    flow_stale_events += cw[4];
    flow_stale_confirmed += cw[5];
    flow_rescued += cw[6];
    if (cw[4] && getenv("JAICOV_VERBOSE"))
        fprintf(stderr, "jaicov: dataflow factorisation: %d flag(s) found by the read-modify-write poll, %d of them still invisible to the plain poll, %d after more than 1 ms of waiting\n", cw[4], cw[5], cw[6]);
    if (ab != 0) {            // a wait of the dataflow factorisation ran into its time limit
        flow_report_stall();
        return -9;
    }
    return h;
}

// fp64 MFMA issue-rate ceiling: every wave runs `iters` x 16 independent v_mfma_f64_16x16x4 on register operands
__global__ __launch_bounds__(256) void mfma_peak_kernel(double *out, int iters) {
    d4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    // iters < 0: random full-mantissa operands that change every iteration (DVFS: zero/constant data clocks higher)
    const bool rnd = iters < 0;
    if (rnd) iters = -iters;
    unsigned long long st = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 256ull * blockIdx.x);
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (rnd) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            a = __longlong_as_double((long long)((st >> 12) | 0x3FF0000000000000ull)) - 1.5;
            b = __longlong_as_double((long long)(((st * 0x2545F4914F6CDD1Dull) >> 12) | 0x3FF0000000000000ull)) - 1.5;
        }
#pragma unroll
        for (int i = 0; i < 16; i++)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[0] = s;
}

hipError_t mfma_peak_bench(int blocks, int iters, float *ms_out, double *tflops) {
    double *out = nullptr;
    HIPCHK(hipMalloc(&out, 8));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, out, iters);
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, out, iters);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms_out, e0, e1);
    *tflops = (double)blocks * 4.0 * (iters < 0 ? -iters : iters) * 16.0 * 2048.0 / (*ms_out * 1e-3) / 1e12;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(out);
    return hipGetLastError();
}

// does hipExtStreamCreateWithCUMask work here?  runs the MFMA peak kernel on a stream restricted by `mask` (8 words)
hipError_t cumask_bench(const uint32_t *mask, int blocks, int iters, float *ms_out, double *tflops) {
    hipStream_t st;
    HIPCHK(hipExtStreamCreateWithCUMask(&st, 8, mask));
    double *out = nullptr;
    HIPCHK(hipMalloc(&out, 8));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms_out, e0, e1);
    *tflops = (double)blocks * 4.0 * iters * 16.0 * 2048.0 / (*ms_out * 1e-3) / 1e12;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(out);
    hipStreamDestroy(st);
    return hipGetLastError();
}

// timing hook for the diagonal-block kernel (diagnostics): `iters` back-to-back launches on an SPD 128x128 block
hipError_t diag_kernel_bench(int dbg, int iters, float *ms_out) {
    double *A = nullptr, *inv = nullptr;
    int *info = nullptr;
    HIPCHK(hipMalloc(&A, 16384 * sizeof(double)));
    HIPCHK(hipMalloc(&inv, 16384 * sizeof(double)));
    HIPCHK(hipMalloc(&info, sizeof(int)));
    double *h = new double[16384];
    for (int r = 0; r < 128; r++)
        for (int c = 0; c < 128; c++) h[r * 128 + c] = (r == c) ? 4.0 : 0.5 / (1.0 + (r > c ? r - c : c - r));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(inv, 0, 16384 * sizeof(double));
    hipMemset(info, 0, sizeof(int));
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; i++) {
        hipMemcpyAsync(A, h, 16384 * sizeof(double), hipMemcpyHostToDevice, nullptr);
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, nullptr, A, 128L, inv, info, 0, dbg);
    }
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms_out, e0, e1);
    *ms_out /= iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(A); hipFree(inv); hipFree(info);
    delete[] h;
    return hipGetLastError();
}

}  // namespace jaicov
