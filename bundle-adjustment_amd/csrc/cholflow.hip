// Dataflow Cholesky: the whole factorisation L L' = M (plus the right-hand-side rows below M) as TWO concurrent launches
// (from 24 block columns on; dense.hip's stream-scheduled potrf_streams() below that).
// Replaces the arithmetic of dpptrf / the factorisation half of dspsv (MathExtension.java:248,348) like dense.hip's
// stream-scheduled potrf(), with the dependencies carried by flags in memory instead of streams, events and launches.
//
//   chol_tile_kernel    persistent workgroups (two per CU, the register / LDS footprint of gemm_f64_kernel's 128-tile).
//                       A workgroup draws tasks from one ticket counter.  A task is one 128 x 128 tile (i, j) of the lower
//                       triangle and a range [k0, k1) of block columns: the tile is loaded into the accumulators ONCE,
//                       C -= L[i][k] L[j][k]' is applied for every k of the range as soon as both operand tiles are
//                       final (left-looking: K grows to the whole width of the matrix, so the C tile's load / store and
//                       the pipeline fill are paid once per tile, not once per 512 columns), then the tile is finished:
//                       a diagonal tile goes to the diagonal kernel, an off-diagonal one is multiplied by inv(L_jj)'
//                       and published.
//   potrf_chain_kernel  the default companion (chain form): two workgroups on reserved CUs (150 KB of LDS each).  Workgroup 0
//                       keeps the critical chain potrf(c) -> L[c+1][c] -> diagonal tile (c+1, c+1) -> potrf(c+1) inside one
//                       CU (LDS + accumulators, no trip through memory and flags between the links); workgroup 1 inverts
//                       the factors.  The tile kernel hands the two tiles over as partial visits (`applied` flags).
//   potrf_diag_chain_kernel   the first companion (JAICOV_FLOW_CHAIN=0): ONE workgroup: for c = 0, 1, ...: waits for the
//                       updated diagonal tile c, factors and inverts it (potrf_diag.h), publishes L_cc and inv(L_cc).
//   chol_tile_kernel<.., true>   the one-kernel form for hosts on which two kernels cannot run side by side.
//
// Order and progress.  Tickets are handed out in the order of the host-built task list, which is a topological order of
// the dependencies (column-major: a task only ever waits for tiles of earlier tickets, and for the companion kernel, which
// only waits for tasks).  A ticket is drawn by a RUNNING workgroup, so every ticket below a waiting workgroup's own is
// held by a workgroup that is running or has finished: no deadlock whatever the residency, and no assumption about the
// dispatch order (tests/test_flow_schedule.py replays every form's list against a model of the companion).  The companion
// runs on its own CU-masked stream and the host launches the tile kernel only once it is resident.
// Every wait is bounded (wall clock): a stall sets the abort word, every waiter leaves, the host repeats the factorisation
// or reports an error (engine.hip, solve).
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility; the per-XCD L2s are not coherent, a CU's L1 is never
// refreshed by other CUs' stores).  Producer: every byte that another workgroup will read is stored write-through
// (`sc1`: 8-byte agent-scope atomic stores, or 16 bytes by inline asm -- store_wt2, with the wait states the compiler cannot
// know about), every storing wave drains (`s_waitcnt vmcnt(0)`), the workgroup's barrier, then one lane stores the flag
// (system scope).  Consumer: one lane / one wave polls with relaxed `sc1` loads -- in the slow branch also at system scope and
// by a read-modify-write, see flow_spin --, then ONE agent-scope acquire (invalidates this CU's L1) + `s_waitcnt vmcnt(0)` +
// barrier, then plain loads.  No stale line can sit in a reader's L2: a tile is read before it is final only by the
// workgroup that writes it next, write-through stores drop the line from the writer's L2, and a final tile is never written
// again.  DESIGN.md section 4 ("Visibility") records the rare stalls that were observed and what they turned out to be (not
// visibility: tile workgroups dispatched last on the XCDs of the chain workgroups standing still inside the product loop;
// they take no tickets any more, see the top of chol_tile_kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "dense.h"
#include "gemm_f64.h"
#include "potrf_diag.h"

namespace jaicov {

#define HIPCHK(x)                                  \
    do {                                           \
        hipError_t _e = (x);                       \
        if (_e != hipSuccess) return _e;           \
    } while (0)

// control words
enum { FLOW_TICKET = 0, FLOW_ABORT = 1, FLOW_DIAG_NEXT = 2, FLOW_CHAIN_AT = 3, FLOW_STALE = 4, FLOW_STALE_CONFIRMED = 5, FLOW_RESCUED = 6, FLOW_WG_OFF = 7, FLOW_CTRL_WORDS = 16 };   // RESCUED: hits of the read-modify-write poll after > 1 ms of waiting   // CHAIN_AT: column << 4 | stage of the chain workgroup
constexpr int FLOW_FIN = 1 << 20;        // task.w = k1 | FLOW_FIN: finish the tile after the updates
// Split update ranges (the late block columns' tiles: each has ~100 block-column steps to do one after the other, is drawn late -- tickets go
// out in column order -- and the chip drains while the last of them crawl through their ranges).  task.w | FLOW_PART: a PARTIAL-SUM task: zeros
// instead of the tile, the block columns [k0, k1) subtracted, the sum stored to partial buffer (task.z >> 12) and its flag set.  The tile's
// own task takes the LAST piece of the range, then adds the (task.w >> 22 & 7) partial sums of buffers (task.z >> 12) ... in buffer order
// (a fixed order: same bits in every run) before it stores / finishes the tile.  k0 = task.z & 4095.
constexpr int FLOW_PART = 1 << 21;
constexpr int FLOW_NPART_SHIFT = 22;
constexpr int FLOW_BUF_SHIFT = 12;
constexpr int FLOW_LDS = 144;            // LDS row stride of both operands (gemm_f64.h: == 16 mod 32 doubles)
constexpr int FLOW_STAGE = GEMM_BK * 2 * FLOW_LDS;

struct FlowArgs {
    double *L;
    long ld;
    double *invd;            // [nb][128][128] inverses of the diagonal blocks
    const int4 *tasks;       // {i, j, k0, k1 | FLOW_FIN}
    int n_tasks;
    int nb;                  // diagonal blocks
    int fs;                  // row stride of done / applied
    int *ctrl;               // FLOW_CTRL_WORDS control words
    int *done;               // [row blocks][fs]  L[i][k] is final (done[k][k]: set by the diagonal kernel, inv(L_kk) too)
    int *applied;            // [row blocks][fs]  applied - 1 block columns have been subtracted from the stored tile (partial visits; 0 = not stored yet)
    int *diag_ready;         // [nb] the updated diagonal tile is in memory
    int *factored;           // [nb] chain form: L_cc is in memory (the inverse follows, done[c][c])
    int *info;               // first failing pivot (dense.h)
    int *alive;              // host-visible words: workgroup b of the diagonal / chain kernel stores `seq` into alive[b] when it starts (potrf_flow's handshake)
    int seq;
    // optional: the matrix is not in L yet but is M = V N V + Bh' Bh (identity on the d border rows and on the padding) of a
    // source square N (NES.applyPrecondition, NES:82-91, fused into the first load of every tile; engine.hip scale_copy_kernel)
    const double *src;       // N, row-major lower, leading dimension src_ld; null = the tiles are in L
    long src_ld;
    const double *V;         // [>= 128 nb]
    const double *Bh;        // [d][bstride]
    int d, U, bstride;
    double *diag_scratch;    // INLINE_DIAG: [grid][128 x DP + 8 x 16 x WDP] work arrays of potrf_diag_body in global memory
    const double *zeros;     // 64 zeros
    double *scratch;         // [grid][128 x 128] per workgroup: operand of the multiplication by inv(L_jj)'
    double *partial;         // [partial buffers][128 x 128] partial sums of split update ranges (FLOW_PART)
    int *pflag;              // [partial buffers] the partial sum is in memory
    long long timeout;       // wall-clock ticks (100 MHz) a wait may take before the factorisation is abandoned
    int fake_a;              // timing experiment (wrong results): every tile reads row block j's strip as its A operand too
    int crit_prio;           // s_setprio level of the tasks on the critical chain; 0 = none
    int keep;                // chain form: blocks >= keep on an XCD that hosts a chain workgroup leave at once (0: none do)
    int second_wg;           // chain form: there is a third chain workgroup
    int second_update;       // chain form: workgroup 2 also subtracts its tile from tile (c+2, c+1)
    int inv_wt;              // chain form: the inverses leave write-through (1) or plainly behind a release (0)
    int crit_span;           // ... which are the tiles (i, j) with i <= j + crit_span
    int *wgstate;            // [grid] where each workgroup is: ticket << 12 | k << 4 | stage (flow_report_stall reads it after a stall)
    long long *ctrace;       // optional [nb][8], chain kernel: potrf start, factor done, operands there, solve done, update done (wall clock)
    long long *trace;        // optional [n_tasks][8]: start, C loaded, updates done, end (wall clock), ticks spent waiting, cycles, block, HW_ID | XCC_ID << 32
};

__device__ __forceinline__ int flow_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the same word read by a read-modify-write at agent scope: performed where all XCDs agree (never served from an L2 line)
__device__ __forceinline__ int flow_ld_rmw(const int *p) { return __hip_atomic_fetch_or(const_cast<int *>(p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// flags are stored at SYSTEM scope (sc0 sc1: past every cache of the device) and the fallback poll reads them the same way:
// see "Visibility" -- polls were seen not to find flags that memory held as set, with agent-scope stores and polls
__device__ __forceinline__ void flow_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int flow_ld_sys(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_wt(double *p, double v) {   // write-through (sc1) store of one double
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// The same for 16 bytes (no builtin for a 16-byte atomic store).  The compiler does not know that this asm is a vector-memory
// store, so it inserts none of the wait states such a store needs:
//   after it: a store of more than 8 bytes reads the upper half of its data registers a few cycles after issue; a VALU
//     instruction that overwrites them right behind the store (the register allocator reuses them at once) changes what is
//     stored.  Seen exactly so: the second double of some 16-byte pieces of L[c+2][c] wrong, lanes 12-15 of every 16, last block
//     column only (where the stores follow each other most densely).  `s_nop 1` = the two wait states LLVM inserts for real stores.
//   before it: a register written by an MFMA needs 18+ wait states before a memory instruction may read it (no interlock);
//     the data usually come through v_accvgpr_read (a VALU instruction the compiler handles), but nothing guarantees that.
__device__ __forceinline__ void store_wt2(double *p, d2_t v) {
    asm volatile("s_nop 15\n\ts_nop 7\n\tglobal_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// pause between two polls of a wait that has lasted more than 32 polls, in units of 64 clocks (s_sleep)
#ifndef FLOW_LONG_NAP
#define FLOW_LONG_NAP 40
#endif

#define FLOW_OPAQUE_TID(name) int name = tid; asm volatile("" : "+v"(name))

// The lane whose wait runs out: set the abort word and, if it is the first, copy every workgroup's state word next to it (what the
// others were doing at THAT moment; flow_report_stall prints it beside what they had reached when the kernel ended).
__device__ __forceinline__ void flow_give_up(int *ctrl) {
    if (atomicCAS(ctrl + FLOW_ABORT, 0, 2) == 0) {
        int *wg = ctrl + ctrl[FLOW_WG_OFF];
        for (int b = 0; b < 1024; b++) wg[1024 + b] = flow_ld(wg + b);
    }
}

// One lane: spin until *flag >= want.  Returns false when the factorisation is abandoned (abort word set / timeout).
__device__ __forceinline__ bool flow_spin(const int *flag, int want, int *ctrl, long long timeout, long long *waited) {
    if (flow_ld(flag) >= want) return true;
    const long long t0 = wall_clock64();
    int spins = 0;
    bool ok = true;
    for (;;) {
        if (flow_ld(flag) >= want) break;
        if (++spins < 32) __builtin_amdgcn_s_sleep(4);
        else {
            __builtin_amdgcn_s_sleep(FLOW_LONG_NAP);
            if ((spins & 31) == 0) {
                if (flow_ld_sys(flag) >= want || flow_ld_rmw(flag) >= want) {      // set; did the plain poll only just miss it, or does it still not see it?
                    atomicAdd(ctrl + FLOW_STALE, 1);
                    if (flow_ld(flag) < want) atomicAdd(ctrl + FLOW_STALE_CONFIRMED, 1);
                    if (wall_clock64() - t0 > 100000) atomicAdd(ctrl + FLOW_RESCUED, 1);
                    break;
                }
                if (flow_ld(ctrl + FLOW_ABORT) != 0) { ok = false; break; }
                if (wall_clock64() - t0 > timeout) { flow_give_up(ctrl); ok = false; break; }
            }
        }
    }
    if (waited) *waited += wall_clock64() - t0;
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
// a real call: the diagonal block's register appetite (potrf_diag.h wants all 512) must not reach the tile loop's allocation
__device__ __attribute__((noinline)) void flow_inline_diag(double *A, long ld, double *inv_out, int *info, int blk, double *ws) {
    potrf_diag_body(A, ld, inv_out, info, blk, 0, ws, ws + 128 * DP);
}

// PF = k-steps the operand loads run ahead of the MFMAs (1 or 2).  INLINE_DIAG: no diagonal kernel beside this one -- the
// workgroup that finishes the updates of a diagonal tile factors and inverts it itself, with the work arrays of potrf_diag.h in
// global memory instead of 150 KB of LDS (several times slower per block: the chain then bounds the whole factorisation).  For
// hosts on which two kernels cannot run at the same time (rocprofv3 --pmc serialises every dispatch; AMD_SERIALIZE_KERNEL): the
// factorisation stays correct there, and the tile kernel's HBM counters can be collected at all.
template <int PF, bool INLINE_DIAG>
__global__ __launch_bounds__(256, 2) void chol_tile_kernel(FlowArgs g) {
    __shared__ double smem[2 * FLOW_STAGE];
    __shared__ int s_msg[4];
    const int tid = threadIdx.x;
    const int bid = (int)blockIdx.x;
    // Everything derived from the thread index is recomputed where it is used, from a copy of tid the compiler cannot see
    // through: hoisted out of the persistent loop those values (dozens of them) stay live across the MFMA loop, the kernel
    // spills, and a kernel with a scratch segment is admitted with fewer waves per shader engine (measured: 458 of the 496
    // workgroups resident).

    d4_t acc[4][4];
    // acc += asign * A B' over nk k-steps of 16: A(x, k) = Ap[x * lda + k], B(y, k) = Bp[y * ldb + k], x, y < 128.  The sign
    // rides on the A operand's way into LDS, so that the accumulators always hold the tile itself (never its negative:
    // a sign applied at load / store time costs a second set of 128 registers around the epilogue).
    auto accumulate = [&](const double *Ap, long lda, const double *Bp, long ldb, int nk, double asign) __attribute__((always_inline)) {
        FLOW_OPAQUE_TID(tq);
        // operand staging (gemm_f64.h, KC layout: 8 lanes fetch the 16 k of one row, transposed on the way into LDS)
        const int seg = tq & 7, rbase = tq >> 3, lane = tq & 63, wave = tq >> 6;
        const int odd = (seg & 1) * FLOW_LDS;
        const int st_lds = (2 * seg) * FLOW_LDS + rbase + 8 * ((seg >> 1) & 1) + 4 * (seg >> 2);
        const int fa = (lane >> 4) * FLOW_LDS + 64 * (wave >> 1) + (lane & 15);
        const int fb = (lane >> 4) * FLOW_LDS + 64 * (wave & 1) + (lane & 15);
        const double *ap = Ap + (long)rbase * lda + 2 * seg;
        const double *bp = Bp + (long)rbase * ldb + 2 * seg;
        // Operands travel global -> registers -> LDS, TWO k-steps ahead: the A strips of the tiles of one column are all
        // different (no reuse in L2 as in gemm_f64.h's super-tiles), they come from HBM, and one k-step (~4 us) of lead did
        // not always cover that latency with every CU streaming (the left-looking loop ran at 86 % of the matrix pipe).
        d2_t ra[2][4], rb[2][4];
#define FLOW_GLOAD(S)                                                                                        \
    do {                                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 4; q++) ra[S][q] = *reinterpret_cast<const d2_t *>(ap + (long)(32 * q) * lda); \
        _Pragma("unroll") for (int q = 0; q < 4; q++) rb[S][q] = *reinterpret_cast<const d2_t *>(bp + (long)(32 * q) * ldb); \
        ++ls;                                                                                                \
        const long adv = ls < nk ? GEMM_BK : 0;   /* past the last step: the same addresses again, never beyond */ \
        ap += adv;                                                                                           \
        bp += adv;                                                                                           \
    } while (0)
#define FLOW_LSTORE(S, STAGE)                                                                                \
    do {                                                                                                     \
        double *sa = smem + (STAGE) * FLOW_STAGE + st_lds;                                                   \
        double *sb = sa + GEMM_BK * FLOW_LDS;                                                                \
        _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                      \
            sa[32 * q + odd] = asign * ra[S][q].x;                                                           \
            sa[32 * q + FLOW_LDS - odd] = asign * ra[S][q].y;                                                \
            sb[32 * q + odd] = rb[S][q].x;                                                                   \
            sb[32 * q + FLOW_LDS - odd] = rb[S][q].y;                                                        \
        }                                                                                                    \
    } while (0)
#define FLOW_FRAG(STAGE, KS, A_, B_)                                                                         \
    do {                                                                                                     \
        const double *sa = smem + (STAGE) * FLOW_STAGE + fa;                                                 \
        const double *sb = smem + (STAGE) * FLOW_STAGE + GEMM_BK * FLOW_LDS + fb;                            \
        const int sh = 8 * ((KS) & 1) + 4 * ((KS) >> 1);                                                     \
        _Pragma("unroll") for (int x = 0; x < 4; x++) A_[x] = sa[(4 * (KS)) * FLOW_LDS + 16 * x + sh];       \
        _Pragma("unroll") for (int y = 0; y < 4; y++) B_[y] = sb[(4 * (KS)) * FLOW_LDS + 16 * y + sh];       \
    } while (0)
#define FLOW_MM(A_, B_)                                                                                      \
    do {                                                                                                     \
        _Pragma("unroll") for (int x = 0; x < 4; x++)                                                        \
            _Pragma("unroll") for (int y = 0; y < 4; y++)                                                    \
                acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_[x], B_[y], acc[x][y], 0, 0, 0);          \
    } while (0)
#define FLOW_MFMA(STAGE)                                                                                     \
    do {                                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 4; ks++) {                                                   \
            double a[4], b[4];                                                                               \
            FLOW_FRAG(STAGE, ks, a, b);                                                                      \
            FLOW_MM(a, b);                                                                                   \
        }                                                                                                    \
    } while (0)
        // nk is a multiple of 8.  The loads are issued unconditionally (the last two of a run re-read the last step): behind
        // a branch the compiler's s_waitcnt placement has to assume the loads were NOT issued and waits for vmcnt(0) before
        // the LDS stores, i.e. for the loads of two steps ahead as well.
        int ls = 0;                    // steps whose loads have been issued
        FLOW_GLOAD(0);                 // step 0
        if (PF == 2) {
            FLOW_GLOAD(1);             // step 1
            FLOW_LSTORE(0, 0);
            __syncthreads();
            for (int kt = 0; kt < nk; kt += 2) {
                FLOW_GLOAD(0);         // step kt + 2
                FLOW_MFMA(0);          // step kt
                FLOW_LSTORE(1, 1);     // step kt + 1
                __syncthreads();
                FLOW_GLOAD(1);         // step kt + 3
                FLOW_MFMA(1);          // step kt + 1
                FLOW_LSTORE(0, 0);     // step kt + 2 (after the last step: unused)
                __syncthreads();
            }
        } else {
            FLOW_LSTORE(0, 0);
            __syncthreads();
            for (int kt = 0; kt < nk; kt += 2) {
                FLOW_GLOAD(0);         // step kt + 1
                FLOW_MFMA(0);          // step kt
                FLOW_LSTORE(0, 1);
                __syncthreads();
                FLOW_GLOAD(0);         // step kt + 2
                FLOW_MFMA(1);          // step kt + 1
                FLOW_LSTORE(0, 0);
                __syncthreads();
            }
        }
#undef FLOW_GLOAD
#undef FLOW_LSTORE
#undef FLOW_MFMA
#undef FLOW_FRAG
#undef FLOW_MM
    };

    // The workgroups dispatched last on an XCD that hosts a chain workgroup take no tickets.  Those XCDs hold fewer tile
    // workgroups than the others (a CU is taken, and the dispatcher stops at the first workgroup that does not fit), so the last
    // ones of the grid stay queued there for the whole run -- and about once in 2 000-3 600 factorisations one of the last
    // RESIDENT ones (always block 448, 449, 456 or 464 of 512) was seen to stand still inside the product loop, with its flags
    // ready, until the time limit emptied the chip ("Visibility").  Leaving early, they let the queued
    // ones through (which leave too): nothing is left queued behind a full shader engine while the factorisation runs
    // (16 000 factorisations without a stall since; 7-8 expected at the earlier rate).
    if (!INLINE_DIAG && g.keep > 0 && (int)blockIdx.x >= g.keep) {
        const int xcc = 1 + (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf);
        if (xcc == flow_ld(g.ctrl + 9) || xcc == flow_ld(g.ctrl + 11) || (g.second_wg && xcc == flow_ld(g.ctrl + 13))) return;
    }
    for (;;) {
        if (tid == 0) s_msg[0] = atomicAdd(g.ctrl + FLOW_TICKET, 1);
        __syncthreads();
        const int t = s_msg[0];
        __syncthreads();
        if (t >= g.n_tasks) break;
        const int4 tk = g.tasks[t];
        const int ti = tk.x, tj = tk.y, k0 = tk.z & ((1 << FLOW_BUF_SHIFT) - 1), k1 = tk.w & (FLOW_FIN - 1);
        const int pbuf = tk.z >> FLOW_BUF_SHIFT, npart = (tk.w >> FLOW_NPART_SHIFT) & 7;
        const bool part = (tk.w & FLOW_PART) != 0;
        const bool first_visit = k0 == 0 || npart > 0;     // the tile has not been stored by an earlier visit
#define FLOW_STAGE_MARK(kk, st) do { if (tid == 0) flow_st(g.wgstate + bid, (t << 12) | ((kk) << 4) | (st)); } while (0)
        FLOW_STAGE_MARK(k0, 1);
        const bool fin = (tk.w & FLOW_FIN) != 0;
        long long waited = 0, t_start = 0, t_c = 0, t_upd = 0, c_c = 0, c_upd = 0, w_upd = 0;
        int n_runs = 0;
        if (g.trace) t_start = wall_clock64();
        bool ok = true;
        // the chain diag(j) -> L[j+1][j] -> tile (j+1,j+1) -> diag(j+1) bounds the whole factorisation whenever the trailing work is
        // short: its two tile tasks take the matrix pipe ahead of the workgroup they share their SIMDs with
        const bool critical = ti <= tj + g.crit_span && g.crit_prio > 0;
        if (critical) __builtin_amdgcn_s_setprio(3);
        // ---- one or two phases, each: accumulators <- memory, runs of MFMA work, accumulators -> memory.  Phase 0: the tile
        //      and its updates.  Phase 1 (off-diagonal tile that is to be finished): zeros, then C inv(L_jj)' with C read back
        //      from this workgroup's scratch tile.  ONE load site, ONE instance of the pipelined loop and ONE store site, and the
        //      accumulators' live range never crosses a phase: anything else made the compiler shuffle and spill them.
        int k = k0;
        for (int phase = 0;; phase++) {
            // -- what has to be there before the accumulators are loaded
            if ((phase == 0 && !first_visit && !part) || phase == 1) {
                FLOW_STAGE_MARK(k, 2);
                if (tid == 0) {
                    // phase 0: an earlier (partial) visit wrote the tile; phase 1: inv(L_jj) from the diagonal kernel
                    const int *f = phase == 0 ? g.applied + (long)ti * g.fs + tj : g.done + (long)tj * g.fs + tj;
                    const bool r = flow_spin(f, phase == 0 ? k0 + 1 : 1, g.ctrl, g.timeout, &waited);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    drain_stores();
                    s_msg[1] = r ? 1 : 0;
                }
                __syncthreads();
                ok = s_msg[1] != 0;
                __syncthreads();
                if (!ok) break;
            }
            {
                FLOW_OPAQUE_TID(tq);
                const long trow = 64 * (tq >> 7) + ((tq & 63) >> 4), tcol = 64 * ((tq >> 6) & 1) + (tq & 15);
                // phase 1 starts from zeros: read like a tile (row stride 0) so that there is one load site
                const bool scaled = phase == 0 && first_visit && !part && g.src != nullptr && ti < g.nb;   // first load of a matrix tile from N
                const double *src = phase != 0 || part ? g.zeros + (tq & 15)
                                    : scaled   ? g.src + ((long)ti * 128 + trow) * g.src_ld + (long)tj * 128 + tcol
                                               : g.L + ((long)ti * 128 + trow) * g.ld + (long)tj * 128 + tcol;
                const long sld = phase != 0 || part ? 0 : (scaled ? g.src_ld : g.ld);
#pragma unroll
                for (int x = 0; x < 4; x++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double *rowp = src + (long)(16 * x + 4 * r) * sld;
#pragma unroll
                        for (int y = 0; y < 4; y++) acc[x][y][r] = rowp[16 * y];
                    }
                if (scaled) {
                    const int R0 = ti * 128 + (int)trow, C0 = tj * 128 + (int)tcol;
                    double vc[4];
#pragma unroll
                    for (int y = 0; y < 4; y++) vc[y] = g.V[C0 + 16 * y];
#pragma unroll
                    for (int x = 0; x < 4; x++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int R = R0 + 16 * x + 4 * r;
                            const double vr = g.V[R];
#pragma unroll
                            for (int y = 0; y < 4; y++) {
                                const int Cc = C0 + 16 * y;
                                double v = vr * acc[x][y][r] * vc[y];
                                for (int a = 0; a < g.d; a++) v += g.Bh[(long)a * g.bstride + R] * g.Bh[(long)a * g.bstride + Cc];
                                const bool inside = R >= g.d && R < g.U && Cc >= g.d;
                                acc[x][y][r] = inside ? v : (R == Cc ? 1.0 : 0.0);
                            }
                        }
                }
            }
            if (g.trace && phase == 0) { t_c = wall_clock64(); c_c = clock64(); }
            // -- runs: phase 0: the block columns [k, k1) as their operand tiles become final; phase 1: one run of 8 k-steps
            for (bool more = true; more;) {
                const double *Ap, *Bp;
                long lda, ldb;
                int nk;
                double asign;
                if (phase == 0) {
                    if (k >= k1) break;
                    FLOW_STAGE_MARK(k, 3);
                    if (tid < 64) {
                        FLOW_OPAQUE_TID(lane);
                        const int *fi = g.done + (long)ti * g.fs, *fj = g.done + (long)tj * g.fs;
                        const long long t0 = wall_clock64();
                        int spins = 0, cnt = 0;
                        for (;;) {
                            const int idx = k + lane;
                            const bool ready = idx < k1 && flow_ld(fi + idx) != 0 && flow_ld(fj + idx) != 0;
                            const unsigned long long m = __ballot(ready);
                            cnt = m == ~0ull ? 64 : __builtin_ctzll(~m);
                            if (cnt > 0) break;
                            if (++spins < 32) __builtin_amdgcn_s_sleep(4);
                            else {
                                __builtin_amdgcn_s_sleep(FLOW_LONG_NAP);
                                if ((spins & 31) == 0) {
                                    if (wall_clock64() - t0 > 100000) {       // > 1 ms: whatever this CU / XCD still holds of the flag lines goes
                                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                                        drain_stores();
                                    }
                                    const bool ready2 = idx < k1 && (flow_ld_sys(fi + idx) != 0 || flow_ld_rmw(fi + idx) != 0) &&
                                                        (flow_ld_sys(fj + idx) != 0 || flow_ld_rmw(fj + idx) != 0);
                                    const unsigned long long m2 = __ballot(ready2);
                                    cnt = m2 == ~0ull ? 64 : __builtin_ctzll(~m2);
                                    if (cnt > 0) {
                                        const bool again = idx < k1 && flow_ld(fi + idx) != 0 && flow_ld(fj + idx) != 0;
                                        const unsigned long long m3 = __ballot(again);
                                        if (lane == 0) {
                                            atomicAdd(g.ctrl + FLOW_STALE, 1);
                                            if ((m3 & 1ull) == 0) atomicAdd(g.ctrl + FLOW_STALE_CONFIRMED, 1);   // the plain poll still misses it
                                            if (wall_clock64() - t0 > 100000) atomicAdd(g.ctrl + FLOW_RESCUED, 1);   // ... had missed it for > 1 ms
                                        }
                                        break;
                                    }
                                    if (flow_ld(g.ctrl + FLOW_ABORT) != 0) break;
                                    if (wall_clock64() - t0 > g.timeout) { if (lane == 0) flow_give_up(g.ctrl); break; }
                                }
                            }
                        }
                        if (spins > 0) waited += wall_clock64() - t0;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        drain_stores();
                        if (lane == 0) {
                            s_msg[1] = cnt;
                            s_msg[2] = spins;
                        }
                    }
                    __syncthreads();
                    const int cnt = s_msg[1];
                    // a task that had to wait for an operand is at the dependency front: what it does next (this block column,
                    // then the product with the inverse) is what the tiles to its right are waiting for -- it takes the matrix
                    // pipe ahead of the workgroup it shares the CU with, which is (usually) still deep in its own updates
                    if (g.crit_prio >= 2 && s_msg[2] > 0) __builtin_amdgcn_s_setprio(3);
                    __syncthreads();
                    if (cnt == 0) { ok = false; break; }
                    ++n_runs;
                    FLOW_STAGE_MARK(k, 4);
                    Ap = g.L + (long)(g.fake_a ? tj : ti) * 128 * g.ld + (long)k * 128;
                    Bp = g.L + (long)tj * 128 * g.ld + (long)k * 128;
                    lda = ldb = g.ld;
                    nk = 8 * cnt;
                    asign = -1.0;             // C -= L[i][k] L[j][k]'
                    k += cnt;
                } else {
                    Ap = g.scratch + (long)bid * 16384;
                    Bp = g.invd + (long)tj * 16384;
                    lda = ldb = 128;
                    nk = 8;
                    asign = 1.0;              // L[i][j] = C inv(L_jj)'
                    more = false;
                }
                accumulate(Ap, lda, Bp, ldb, nk, asign);
            }
            if (!ok) break;
            // -- the partial sums of the other pieces of a split range (written by earlier tickets), in buffer order
            if (phase == 0 && npart > 0) {
                for (int q = 0; q < npart; q++) {
                    FLOW_STAGE_MARK(k, 9);
                    if (tid == 0) {
                        const bool r = flow_spin(g.pflag + pbuf + q, 1, g.ctrl, g.timeout, &waited);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        drain_stores();
                        s_msg[1] = r ? 1 : 0;
                    }
                    __syncthreads();
                    ok = s_msg[1] != 0;
                    __syncthreads();
                    if (!ok) break;
                    FLOW_OPAQUE_TID(tq);
                    const double *pp = g.partial + (long)(pbuf + q) * 16384 + (64 * (tq >> 7) + ((tq & 63) >> 4)) * 128 + 64 * ((tq >> 6) & 1) + (tq & 15);
#pragma unroll
                    for (int x = 0; x < 4; x++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
#pragma unroll
                            for (int y = 0; y < 4; y++) acc[x][y][r] += pp[(16 * x + 4 * r) * 128 + 16 * y];
                }
                if (!ok) break;
            }
            if (g.trace && phase == 0) { t_upd = wall_clock64(); c_upd = clock64(); w_upd = waited; }
            // -- where the accumulators go, and which flag tells whom
            FLOW_STAGE_MARK(k, 5 + phase);
            const bool to_scratch = phase == 0 && fin && ti != tj;
            {
                FLOW_OPAQUE_TID(tq);
                const long trow = 64 * (tq >> 7) + ((tq & 63) >> 4), tcol = 64 * ((tq >> 6) & 1) + (tq & 15);
                double *dst = part         ? g.partial + (long)pbuf * 16384 + trow * 128 + tcol
                              : to_scratch ? g.scratch + (long)bid * 16384 + trow * 128 + tcol
                                           : g.L + ((long)ti * 128 + trow) * g.ld + (long)tj * 128 + tcol;
                const long dld = part || to_scratch ? 128 : g.ld;
#pragma unroll
                for (int x = 0; x < 4; x++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        double *rowp = dst + (long)(16 * x + 4 * r) * dld;
#pragma unroll
                        for (int y = 0; y < 4; y++) store_wt(rowp + 16 * y, acc[x][y][r]);
                    }
            }
            drain_stores();
            if (to_scratch) continue;     // phase 1 follows (its wait comes with a barrier)
            __syncthreads();
            if (INLINE_DIAG && phase == 0 && fin) {
                // the updated diagonal tile is in memory (written through): factor + invert it here, work arrays in this
                // workgroup's global scratch; plain stores, published behind an agent-scope release like the diagonal kernel's
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    drain_stores();
                }
                __syncthreads();
                double *ws = g.diag_scratch + (long)bid * (128 * DP + 8 * 16 * WDP);
                flow_inline_diag(g.L + (long)tj * 128 * g.ld + (long)tj * 128, g.ld, g.invd + (long)tj * 16384, g.info, tj, ws);
                drain_stores();
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    drain_stores();
                    flow_st(g.done + (long)tj * g.fs + tj, 1);
                }
                break;
            }
            if (tid == 0) {
                if (part) flow_st(g.pflag + pbuf, 1);                                      // a partial sum, for the tile's own task
                else if (phase == 1) flow_st(g.done + (long)ti * g.fs + tj, 1);            // L[i][j] is final
                else if (fin) flow_st(g.diag_ready + tj, 1);                               // updated diagonal tile, for the diagonal kernel
                else flow_st(g.applied + (long)ti * g.fs + tj, k1 + 1);                    // partial visit (chain form: the tile is the chain workgroup's from here)
            }
            break;
        }
        if (g.crit_prio > 0) __builtin_amdgcn_s_setprio(0);
        FLOW_STAGE_MARK(k, ok ? 7 : 8);
        if (!ok) break;
        if (g.trace && tid == 0) {
            long long *tr = g.trace + 8 * (long)t;
            tr[0] = t_start; tr[1] = t_c; tr[2] = t_upd; tr[3] = wall_clock64(); tr[4] = waited;
            tr[5] = c_upd - c_c;   // shader-clock cycles of the update phase (against tr[2] - tr[1] at 100 MHz: the clock it ran at)
            tr[6] = (long long)bid | ((long long)n_runs << 16) | (w_upd << 32);   // block, runs, ticks waited before the updates were done
            tr[7] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |                              // HW_ID
                    ((long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf) << 32);   // XCC_ID
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void potrf_diag_chain_kernel(FlowArgs g) {
    __shared__ double S[128 * DP];
    __shared__ double Wd[8 * 16 * WDP];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    if (tid == 0 && g.alive) __hip_atomic_store(g.alive, g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // resident: the tile kernel may come
    for (int c = 0; c < g.nb; c++) {
        if (tid == 0) {
            const bool r = flow_spin(g.diag_ready + c, 1, g.ctrl, g.timeout, nullptr);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_stores();
            s_ok = r ? 1 : 0;
        }
        __syncthreads();
        const bool ok = s_ok != 0;
        __syncthreads();
        if (!ok) return;
        potrf_diag_body(g.L + (long)c * 128 * g.ld + (long)c * 128, g.ld, g.invd + (long)c * 16384, g.info, c, 0, S, Wd);
        drain_stores();
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            drain_stores();
            flow_st(g.done + (long)c * g.fs + c, 1);
            flow_st(g.ctrl + FLOW_DIAG_NEXT, c + 1);
        }
    }
}

// L[c+1][c] = T L_cc^-T for the tile T at Tg, transposed: Y = L_cc^-1 T' by block forward substitution with the factor in S and
// the inverses of its 16x16 diagonal blocks in Wd; block row q is finished (times inv(L_qq)), then subtracted from the block
// rows below.  Y stays in the accumulators throughout: the C/D registers of a finished block row are the B operand of the
// products that follow.  Index trick: the MFMA puts row l4 + 4r of a 16-row result into register r of lane group l4; feeding
// the A operand (blocks of L_cc) with rows AND columns permuted by pi(4a + b) = 4b + a makes that register hold row 4 l4 + r
// instead, so a lane owns four CONSECUTIVE columns of its row of T: 32-byte pieces in memory, whole lines per instruction,
// for the load and for the store (8-byte pieces cost 8 us more per link of the chain).
// wave w: rows 32w .. 32w+31 of T.  y[p][cc][r] = element (32w + 16cc + l15, 16p + 4 l4 + r).
__device__ __forceinline__ void chain_tile_load(const double *Tg, long ld, d4_t (&y)[8][2], const int tid) {
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const double *tp = Tg + (long)(32 * wave + l15) * ld + 4 * l4;
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const d2_t lo = *reinterpret_cast<const d2_t *>(tp + (long)(16 * cc) * ld + 16 * p);
            const d2_t hi = *reinterpret_cast<const d2_t *>(tp + (long)(16 * cc) * ld + 16 * p + 2);
            y[p][cc][0] = lo.x; y[p][cc][1] = lo.y; y[p][cc][2] = hi.x; y[p][cc][3] = hi.y;
        }
}
__device__ __forceinline__ void chain_tile_solve(d4_t (&y)[8][2], const double *S, const double *Wd, const int tid) {
    const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int pi15 = 4 * (l15 & 3) + (l15 >> 2);
#pragma unroll
    for (int q = 0; q < 8; q++) {
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            d4_t z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const double av = Wd[q * 16 * WDP + pi15 * WDP + 4 * l4 + ks];
                z = __builtin_amdgcn_mfma_f64_16x16x4f64(av, y[q][cc][ks], z, 0, 0, 0);
            }
            y[q][cc] = z;
        }
#pragma unroll
        for (int p = q + 1; p < 8; p++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const double av = -S[(16 * p + pi15) * DP + 16 * q + 4 * l4 + ks];
#pragma unroll
                for (int cc = 0; cc < 2; cc++) y[p][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, y[q][cc][ks], y[p][cc], 0, 0, 0);
            }
    }
}
__device__ __forceinline__ void chain_tile_store(double *Tg, long ld, const d4_t (&y)[8][2], const int tid) {   // write-through
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *tp = Tg + (long)(32 * wave + l15) * ld + 4 * l4;
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            d2_t lo, hi;
            lo.x = y[p][cc][0]; lo.y = y[p][cc][1]; hi.x = y[p][cc][2]; hi.y = y[p][cc][3];
            store_wt2(tp + (long)(16 * cc) * ld + 16 * p, lo);
            store_wt2(tp + (long)(16 * cc) * ld + 16 * p + 2, hi);
        }
}

// t -= (tile in S) y, all in the transposed layout of chain_tile_load: for a tile T = t', A = y' and the 128 x 128 tile B in S
// (row-major, stride DP) this is T -= A B'.  512 MFMAs per wave.
__device__ __forceinline__ void chain_tile_update(d4_t (&t)[8][2], const d4_t (&y)[8][2], const double *S, const int tid) {
    const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int pi15 = 4 * (l15 & 3) + (l15 >> 2);
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
        for (int p = 0; p < 8; p++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const double av = -S[(16 * p + pi15) * DP + 16 * q + 4 * l4 + ks];
#pragma unroll
                for (int cc = 0; cc < 2; cc++) t[p][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, y[q][cc][ks], t[p][cc], 0, 0, 0);
            }
}
// a whole 128 x 128 tile -> S (row-major, stride DP)
__device__ __forceinline__ void chain_tile_to_lds(const double *A, long ld, double *S, const int tid) {
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
            S[r * DP + c] = buf[i].x;
            S[r * DP + c + 1] = buf[i].y;
        }
    }
}

// One link of the chain after potrf(c): S holds L_cc, Wd the inverses of its 16x16 diagonal blocks; tile (c+1, c) and the
// diagonal tile (c+1, c+1) are in memory.  Leaves L[c+1][c] in memory (write-through, drained) and the updated diagonal tile in
// S.
__device__ __forceinline__ void chain_solve_update(double *Tg, long ld, long long *tr, double *S, double *Wd, const int tid) {
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    // the 36 lower 16x16 tiles of the diagonal tile, nine per wave: wave w takes the tile rows w (w + 1 tiles) and 7 - w (8 - w)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#define CHAIN_TILE_R(i) ((i) <= wv ? wv : 7 - wv)
#define CHAIN_TILE_Q(i) ((i) <= wv ? (i) : (i) - wv - 1)
    const double *Cg = Tg + 128;
    d4_t y[8][2], cacc[9];
    chain_tile_load(Tg, ld, y, tid);
    if (tr) {
        drain_stores();
        tr[5] = wall_clock64();
    }
    chain_tile_solve(y, S, Wd, tid);
    if (tr) tr[6] = wall_clock64();
    // the diagonal tile (c+1, c+1) is fetched only now: loaded before the solve its 72 registers joined the solve's in one
    // allocation and the chain loop spilled (28 VGPRs).  Its latency hides behind the write-through stores of L[c+1][c] and the
    // two barriers that follow.
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) cacc[i][r] = Cg[(long)(16 * CHAIN_TILE_R(i) + l4 + 4 * r) * ld + 16 * CHAIN_TILE_Q(i) + l15];
    chain_tile_store(Tg, ld, y, tid);
    if (tr) tr[3] = wall_clock64();
    __syncthreads();          // every wave has read L_cc and Wd for the last time
    // ... and into S, row-major, as both operands of the update of the diagonal tile
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++)
#pragma unroll
            for (int r = 0; r < 4; r++) S[(32 * wave + 16 * cc + l15) * DP + 16 * p + 4 * l4 + r] = y[p][cc][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double *sa = S + (16 * CHAIN_TILE_R(i) + l15) * DP + l4, *sb = S + (16 * CHAIN_TILE_Q(i) + l15) * DP + l4;
#pragma unroll
        for (int k = 0; k < 32; k++) cacc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-sa[4 * k], sb[4 * k], cacc[i], 0, 0, 0);
    }
    __syncthreads();          // every wave has read L[c+1][c] for the last time: the next diagonal block takes S over
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = l4 + 4 * r;
            S[(16 * CHAIN_TILE_R(i) + row) * DP + 16 * CHAIN_TILE_Q(i) + l15] = (CHAIN_TILE_R(i) == CHAIN_TILE_Q(i) && l15 > row) ? 0.0 : cacc[i][r];
        }
    drain_stores();           // L[c+1][c] has left
    __syncthreads();
}
#undef CHAIN_TILE_R
#undef CHAIN_TILE_Q

__device__ __forceinline__ void chain_factor(double *Acc, long ld, int *info, int c, double *S, double *Wd, const int tid, long long *tr) {
    diag_factor(S, Wd, info, c, 0, tid);
    if (tr) tr[7] = wall_clock64();
    // L_cc to memory, write-through (no cache-wide release on the chain); the flag follows once every wave has drained
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        const int idx2 = tid + 256 * i;
        const int r = idx2 >> 6, cc = 2 * (idx2 & 63);
        if (cc <= r) {
            d2_t v;
            v.x = S[r * DP + cc];
            v.y = S[r * DP + cc + 1];
            store_wt2(Acc + (long)r * ld + cc, v);
        }
    }
}

// The companion roles of the chain kernel (entered once per launch, by workgroups 1 and 2).  Round 2's build spilled 28 VGPRs inside
// the chain loop; what removed them was moving ONE load in chain_solve_update (the diagonal tile after the solve instead of
// before it).  Making the roles real calls was tried on the way and is not needed: it bought nothing and cost a 744-byte call frame.
__device__ __forceinline__ void chain_role_inverses(const FlowArgs *gp, double *S, double *Wd, int *s_okp) {
    const FlowArgs &g = *gp;
    const int tid = threadIdx.x;
    // ---- the inverses ----------------------------------------------------------------------------------------------
    for (int c = 0; c < g.nb; c++) {
        if (tid == 0) {
            const bool r = flow_spin(g.factored + c, 1, g.ctrl, g.timeout, nullptr);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_stores();
            (*s_okp) = r ? 1 : 0;
        }
        __syncthreads();
        const bool ok = (*s_okp) != 0;
        __syncthreads();
        if (!ok) return;
        FLOW_OPAQUE_TID(tq);      // see potrf_diag.h: nothing derived from the thread index is to live across this loop
        diag_load(g.L + (long)c * 128 * g.ld + (long)c * 128, g.ld, S, tq);
        diag_block_inverses(S, Wd, tq);
        __syncthreads();
        if (g.inv_wt) {
            diag_inverse<true>(S, Wd, g.invd + (long)c * 16384, 0, tq);      // write-through
            drain_stores();
            __syncthreads();      // every wave's part has landed; and S may be loaded again
            if (tid == 64) {
                flow_st(g.done + (long)c * g.fs + c, 1);
                flow_st(g.ctrl + FLOW_DIAG_NEXT, c + 1);
            }
        } else {
            diag_inverse<false>(S, Wd, g.invd + (long)c * 16384, 0, tq);
            drain_stores();
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                drain_stores();
                flow_st(g.done + (long)c * g.fs + c, 1);
                flow_st(g.ctrl + FLOW_DIAG_NEXT, c + 1);
            }
            __syncthreads();
        }
    }
    return;
}
__device__ __forceinline__ void chain_role_second(const FlowArgs *gp, double *S, double *Wd, int *s_okp) {
    const FlowArgs &g = *gp;
    const int tid = threadIdx.x;
    // ---- the second subdiagonal: workgroup 2 finishes tile (c+2, c) by the same block forward substitution as the chain
    //      (from L_cc itself: it does not wait for the inverse) and subtracts it from tile (c+2, c+1), the tile the chain
    //      workgroup needs next -- the path that used to take three tile workgroups in a row (inverse -> product with
    //      the inverse -> update) and set the period of the chain.
    for (int c = 0; c + 2 < g.nb; c++) {
        if (tid == 0) {
            const bool r = flow_spin(g.factored + c, 1, g.ctrl, g.timeout, nullptr);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_stores();
            (*s_okp) = r ? 1 : 0;
        }
        __syncthreads();
        if ((*s_okp) == 0) return;
        __syncthreads();
        d4_t y[8][2];
        {
            FLOW_OPAQUE_TID(tq);
            diag_load(g.L + (long)c * 128 * g.ld + (long)c * 128, g.ld, S, tq);
            diag_block_inverses(S, Wd, tq);
        }
        __syncthreads();
        if (tid == 0) {
            const bool r = flow_spin(g.applied + (long)(c + 2) * g.fs + c, c + 1, g.ctrl, g.timeout, nullptr);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_stores();
            (*s_okp) = r ? 1 : 0;
        }
        __syncthreads();       // also: Wd is complete
        if ((*s_okp) == 0) return;
        __syncthreads();
        double *T2 = g.L + (long)(c + 2) * 128 * g.ld + (long)c * 128;      // tile (c+2, c)
        {
            FLOW_OPAQUE_TID(tq);
            chain_tile_load(T2, g.ld, y, tq);
            chain_tile_solve(y, S, Wd, tq);
            chain_tile_store(T2, g.ld, y, tq);
        }
        drain_stores();
        __syncthreads();       // L[c+2][c] has landed; every wave has read L_cc for the last time
        if (tid == 64) flow_st(g.done + (long)(c + 2) * g.fs + c, 1);
        if (!g.second_update) continue;
        if (tid == 0) {
            bool r = flow_spin(g.applied + (long)(c + 2) * g.fs + c + 1, c + 1, g.ctrl, g.timeout, nullptr);
            r = r && flow_spin(g.done + (long)(c + 1) * g.fs + c, 1, g.ctrl, g.timeout, nullptr);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_stores();
            (*s_okp) = r ? 1 : 0;
        }
        __syncthreads();
        if ((*s_okp) == 0) return;
        __syncthreads();
        {
            FLOW_OPAQUE_TID(tq);
            d4_t t[8][2];
            chain_tile_to_lds(g.L + (long)(c + 1) * 128 * g.ld + (long)c * 128, g.ld, S, tq);      // L[c+1][c]
            chain_tile_load(T2 + 128, g.ld, t, tq);                                                 // tile (c+2, c+1)
            __syncthreads();
            chain_tile_update(t, y, S, tq);
            chain_tile_store(T2 + 128, g.ld, t, tq);
        }
        drain_stores();
        __syncthreads();
        if (tid == 64) flow_st(g.applied + (long)(c + 2) * g.fs + c + 1, c + 2);
    }
    return;
}

// ---------------------------------------------------------------------------------------------------------------
// Chain form of the diagonal kernel: TWO workgroups, each on a reserved CU of its own.
//
//   workgroup 0   keeps the whole critical chain of the factorisation inside one CU: potrf(c) in LDS -> L[c+1][c] = T L_cc^-T
//                 -> the diagonal tile (c+1, c+1) minus L[c+1][c] L[c+1][c]' -> potrf(c+1) ... without a trip through memory,
//                 flags and other workgroups in between (that trip: store, flag, poll, acquire, load, three times per block
//                 column, 95-122 us per column where this loop needs about 50).  The tile kernel delivers T = tile (c+1, c) with
//                 all its updates and the diagonal tile (c+1, c+1) with the updates of the block columns < c (partial visits).
//                 The solve runs transposed, Y = L_cc^-1 T', by block forward substitution on the 16x16 diagonal-block
//                 inverses that the factorisation leaves in Wd -- the full inverse of L_cc is not needed here -- with Y in the
//                 accumulators from the load to the store: the C/D layout of a finished block row of Y is the B operand of the
//                 products that subtract it from the block rows below (C/D rows (l>>4)+4r == B rows 4ks+(l>>4)).
//   workgroup 1   inverts the factors (the tile kernel multiplies the other tiles of a block column by inv(L_cc)', a plain MFMA
//                 product) beside the chain instead of inside it.
__global__ __launch_bounds__(256) void potrf_chain_kernel(FlowArgs g) {
    __shared__ double S[128 * DP];
    __shared__ double Wd[8 * 16 * WDP];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    if (tid == 0) {      // where this workgroup runs: HW_ID, XCC_ID + 1 (the tile kernel and flow_report_stall read them) ...
        flow_st(g.ctrl + 8 + 2 * blockIdx.x, (int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
        flow_st(g.ctrl + 9 + 2 * blockIdx.x, 1 + (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf));
        drain_stores();
        // ... then: resident, the tile kernel may come
        if (g.alive) __hip_atomic_store(g.alive + blockIdx.x, g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (blockIdx.x == 1) { chain_role_inverses(&g, S, Wd, &s_ok); return; }
    if (blockIdx.x == 2) { chain_role_second(&g, S, Wd, &s_ok); return; }
    // ---- the chain -------------------------------------------------------------------------------------------------
    if (tid == 0) {
        const bool r = flow_spin(g.applied, 1, g.ctrl, g.timeout, nullptr);      // tile (0, 0): scaled, in L
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        drain_stores();
        s_ok = r ? 1 : 0;
    }
    __syncthreads();
    if (s_ok == 0) return;
    __syncthreads();
    diag_load(g.L, g.ld, S, tid);
    for (int c = 0; c < g.nb; c++) {
        long long *tr = g.ctrace && tid == 0 ? g.ctrace + 8 * (long)c : nullptr;
        if (tr) tr[0] = wall_clock64();
        if (tid == 0) flow_st(g.ctrl + FLOW_CHAIN_AT, (c << 4) | 1);
        {
            FLOW_OPAQUE_TID(tq);
            chain_factor(g.L + (long)c * 128 * g.ld + (long)c * 128, g.ld, g.info, c, S, Wd, tq, tr);
        }
        if (tr) tr[1] = wall_clock64();
        // L_cc is on its way (write-through): the inverse workgroup may have it as soon as every wave's stores have landed --
        // before this workgroup starts to wait for anything else
        drain_stores();
        __syncthreads();
        if (tid == 64) flow_st(g.factored + c, 1);
        if (c + 1 >= g.nb) break;
        if (tid == 0) {
            flow_st(g.ctrl + FLOW_CHAIN_AT, (c << 4) | 2);
            bool r = flow_spin(g.applied + (long)(c + 1) * g.fs + c, c + 1, g.ctrl, g.timeout, nullptr);
            r = r && flow_spin(g.applied + (long)(c + 1) * g.fs + c + 1, c + 1, g.ctrl, g.timeout, nullptr);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_stores();
            s_ok = r ? 1 : 0;
        }
        __syncthreads();
        const bool ok = s_ok != 0;
        if (!ok) return;
        if (tr) tr[2] = wall_clock64();
        if (tid == 0) flow_st(g.ctrl + FLOW_CHAIN_AT, (c << 4) | 3);
        {
            FLOW_OPAQUE_TID(tq);
            chain_solve_update(g.L + (long)(c + 1) * 128 * g.ld + (long)c * 128, g.ld, tr, S, Wd, tq);
        }
        if (tid == 64) flow_st(g.done + (long)(c + 1) * g.fs + c, 1);
        if (tr) tr[4] = wall_clock64();
    }
}

// Which tasks get their update range split (flow_schedule).  Round 5.  Tickets go out in column order, so the tiles of the late block columns --
// each with ~100 block-column steps to do one after the other, 3+ ms -- are drawn last; the chip drained while they crawled through their
// ranges (mean resident tasks in the last two tenths of the span: 417 and 138 of 496), and the chain waited for exactly those tiles.  Cutting
// the range in two, the first half as an independent partial-sum task at the head of the column (its operands are final long before: it never
// waits, 30.2 us per step), halves the serial length of every late task.  Measured (ms per factorisation, without / with, same box):
// 80 block columns 8.33 / 8.18, 100: 14.4 / 13.8, 118: 22.5 / 21.4, 142: 36.7 / 35.5; 64: 5.20 / 5.40, 40: 2.63 / 2.80 (the partial sums' extra
// store + load and the later start of the column's own tasks cost more than the shorter tail saves).  At 118 block columns the first split
// column anywhere in 40 .. 72 gives the same time (21.35-21.45), 88: 21.8; three pieces 22.3, four 23.3 (worse than none: the tile's own task
// then spends most of its life waiting for operand columns, holding a slot).  JAICOV_FLOW_SPLIT="m:from" is a test hook.
void flow_split_rule(int nb, int *m, int *from) {
    *m = 1; *from = 1 << 30;
    if (nb >= 80) { *m = 2; *from = nb / 2; }
    if (const char *e = getenv("JAICOV_FLOW_SPLIT")) {
        int a = 1, b = 0;
        if (sscanf(e, "%d:%d", &a, &b) >= 1) { *m = a < 1 ? 1 : a; *from = b; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Task list: left-looking.  w = 1: column-major; column j: the diagonal tile first, then the tiles below it (the tile right
// below feeds the next diagonal tile: it is the one the chain waits for), the right-hand-side rows last.  w > 1: groups of w
// columns; the tiles of the group's diagonal blocks first, then row by row the w tiles of a row, which stream the same A
// strip L[i][0..j) at the same time (the second reader finds it in L2 / the Infinity Cache).  Either way a tile only depends
// on tiles that come earlier in the list.
// split_m > 1: the update range of every task of the block columns >= split_from is cut into split_m pieces (when each piece still
// has >= 8 block columns); the first split_m - 1 become partial-sum tasks (FLOW_PART) at the HEAD of their column's tasks -- their
// operands are tiles of much earlier columns, so they run at once and in parallel with each other --, the tile's own task keeps the last
// piece and adds the sums.  *n_bufs = partial buffers needed.
static std::vector<int4> flow_schedule(int nb, int row_blocks, int w, bool chain, int second, int split_m = 1, int split_from = 1 << 30, int *n_bufs = nullptr) {
    std::vector<int4> tasks;
    tasks.reserve((size_t)nb * (row_blocks + 1) / 2 + row_blocks);
    if (w < 1) w = 1;
    if (split_m > 8) split_m = 8;
    if (w > 1) split_m = 1;
    int bufs = 0;
    // chain form (potrf_chain_kernel): the diagonal tile (j, j) is visited for the block columns < j - 1 only and the tile below
    // it, (j + 1, j), is not finished: the chain workgroup takes both from there.  With the optional third workgroup
    // (`second` = 1 / 2) the tile (j + 2, j) is not finished either, and (2) the tile (j + 1, j) lacks its last update.
    auto task = [&](int i, int j) {
        if (chain && i == j) return make_int4(i, j, 0, std::max(j - 1, 0));
        if (chain && i == j + 1 && i < nb) return make_int4(i, j, 0, second >= 2 ? std::max(j - 1, 0) : j);   // second = 2: the last update comes from workgroup 2
        if (chain && second >= 1 && i == j + 2 && i < nb) return make_int4(i, j, 0, j);     // all updates; workgroup 2 finishes it
        return make_int4(i, j, 0, j | FLOW_FIN);
    };
    std::vector<std::vector<int4>> head(nb), own(nb);      // per block column: partial-sum tasks emitted at its head, its tiles' own tasks
    for (int j0 = 0; j0 < nb; j0 += w) {
        const int j1 = std::min(nb, j0 + w);
        auto add = [&](int i, int j) {
            int4 t = task(i, j);
            const int k1 = t.w & (FLOW_FIN - 1);
            if (split_m > 1 && j >= split_from && k1 >= 8 * split_m) {
                int a = 0;
                for (int q = 1; q < split_m; q++) {
                    const int c = (int)((long)k1 * q / split_m);
                    head[j].push_back(make_int4(i, j, a | ((bufs + q - 1) << FLOW_BUF_SHIFT), c | FLOW_PART));
                    a = c;
                }
                t.z = a | (bufs << FLOW_BUF_SHIFT);
                t.w |= (split_m - 1) << FLOW_NPART_SHIFT;
                bufs += split_m - 1;
            }
            own[j0].push_back(t);
        };
        for (int j = j0; j < j1; j++)
            for (int i = j; i < j1; i++) add(i, j);
        for (int i = j1; i < row_blocks; i++)
            for (int j = j0; j < j1; j++) add(i, j);
    }
    for (int j = 0; j < nb; j++) {
        tasks.insert(tasks.end(), head[j].begin(), head[j].end());
        tasks.insert(tasks.end(), own[j].begin(), own[j].end());
    }
    if (n_bufs) *n_bufs = bufs;
    return tasks;
}

}  // namespace jaicov
// debug / tests (no device needed): the task list of the dataflow factorisation, 4 ints per task {i, j, k0, k1 | FIN << 20};
// returns the number of tasks (the first `cap` are written)
extern "C" int jaicov_debug_flow_tasks2(int nb, int row_blocks, int w, int chain, int second, int split_m, int split_from, int *out, int cap) {
    if (nb < 1 || row_blocks < nb) return -1;
    const std::vector<int4> t = jaicov::flow_schedule(nb, row_blocks, w, chain != 0, second, split_m, split_from);
    for (size_t q = 0; q < t.size() && (int)q < cap; q++) {
        out[4 * q] = t[q].x; out[4 * q + 1] = t[q].y; out[4 * q + 2] = t[q].z; out[4 * q + 3] = t[q].w;
    }
    return (int)t.size();
}
extern "C" int jaicov_debug_flow_tasks(int nb, int row_blocks, int w, int chain, int second, int *out, int cap) {
    return jaicov_debug_flow_tasks2(nb, row_blocks, w, chain, second, 1, 1 << 30, out, cap);
}
// the split the solver uses for nb block columns: *m pieces for the tasks of the block columns >= *from (m = 1: none)
extern "C" void jaicov_debug_flow_split(int nb, int *m, int *from) { jaicov::flow_split_rule(nb, m, from); }
namespace jaicov {

// Can two kernels of this process run at the same time?  Probed once: a kernel that waits (at most 50 ms) for a word that a
// second kernel on another stream sets.  Under rocprofv3 --pmc (every dispatch serialised) the second one only starts after
// the first has given up.
__global__ void flow_probe_wait_kernel(int *word, int *result) {
    const long long t0 = wall_clock64();
    int seen = 0;
    while (wall_clock64() - t0 < 5000000LL) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(20);
    }
    *result = seen;
}
__global__ void flow_probe_set_kernel(int *word) { __hip_atomic_store(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

bool DenseSolver::flow_kernels_overlap() {
    static int cached = -1;
    if (factor_form() == FACTOR_ONE_KERNEL) return false;                 // JAICOV_FACTOR_FORM=one_kernel: force the one-kernel form
    if (cached >= 0) return cached != 0;
    int *d = nullptr, h[2] = {0, 0};
    hipStream_t s2 = nullptr;
    bool ok = hipMalloc(&d, 2 * sizeof(int)) == hipSuccess && hipMemset(d, 0, 2 * sizeof(int)) == hipSuccess &&
              hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(flow_probe_wait_kernel, dim3(1), dim3(1), 0, dstream, d, d + 1);
        hipLaunchKernelGGL(flow_probe_set_kernel, dim3(1), dim3(1), 0, s2, d);
        ok = hipStreamSynchronize(dstream) == hipSuccess && hipStreamSynchronize(s2) == hipSuccess &&
             hipMemcpy(h, d, 2 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (s2) hipStreamDestroy(s2);
    if (d) hipFree(d);
    cached = ok && h[1] == 1 ? 1 : 0;
    if (!cached && getenv("JAICOV_VERBOSE")) fprintf(stderr, "jaicov: kernels do not overlap on this host: dataflow factorisation runs as one kernel with inline diagonal blocks\n");
    return cached != 0;
}

// ---- residency of the tile kernel beside the chain workgroups, MEASURED (round 5) -----------------------------------------------------
// The `keep` rule of potrf_flow (the last workgroups dealt to an XCD that hosts a chain workgroup leave without a ticket, so that the ones
// queued behind its full shader engine get in and leave too) used to assume the MI355X's shape: 8 XCDs x 4 shader engines, 64 blocks dealt
// per XCD, "the last eight".  It is now derived from what a launch of the tile kernel's footprint actually does next to stand-ins of the
// chain workgroups: XCDs seen, shader engines per XCD, blocks dealt to an XCD, and how many of them stay QUEUED on an XCD whose reserved CU
// is taken.  Once per process and (grid, chain workgroups).
struct FlowResidency { int n_xcd = 0, n_se = 0, dealt = 0, resident_min = 0, queued_max = 0, keep = 0, se_cap = 0; bool valid = false; };

__global__ __launch_bounds__(256) void flow_residency_standin_kernel(int *alive, const int *release) {
    __shared__ double S[128 * DP];                 // the chain kernel's LDS: a whole CU
    __shared__ double Wd[8 * 16 * WDP];
    S[threadIdx.x] = 0.0; Wd[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(alive + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < 20000000LL) __builtin_amdgcn_s_sleep(100);
        if (S[1] + Wd[1] != 0.0) alive[0] = 2;     // (keeps the arrays)
    }
}
// counts: [0..15] workgroups that started before the release (resident at once), [16..31] workgroups dealt, [32..47] highest shader-engine id, per XCC;
// [64 + 8 xcc + se] resident at once per shader engine
__global__ __launch_bounds__(256, 2) void flow_residency_probe_kernel(int *counts, const int *release) {
    __shared__ double smem[2 * FLOW_STAGE];        // the tile kernel's LDS: two workgroups per CU
    smem[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int xcc = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf);
        const int se = (int)((__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) >> 13) & 7);
        const bool early = __hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0;
        if (early) { atomicAdd(counts + xcc, 1); atomicAdd(counts + 64 + 8 * xcc + se, 1); }
        atomicAdd(counts + 16 + xcc, 1);
        atomicMax(counts + 32 + xcc, se);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < 20000000LL) __builtin_amdgcn_s_sleep(100);
        if (smem[1] != 0.0) counts[48] = 1;
    }
}

static FlowResidency flow_measure_residency(hipStream_t stream, hipStream_t dstream, int grid, int chain_wgs) {
    static std::vector<std::pair<std::pair<int, int>, FlowResidency>> cache;
    static std::mutex cache_mutex;                      // engines may be created from several host threads
    std::lock_guard<std::mutex> lock(cache_mutex);      // (held across the measurement: two probes at once would measure each other)
    for (auto &c : cache)
        if (c.first == std::make_pair(grid, chain_wgs)) return c.second;
    FlowResidency r;
    int *host = nullptr, *counts = nullptr;
    int h[192] = {0};
    bool ok = hipHostMalloc((void **)&host, 8 * sizeof(int), hipHostMallocMapped) == hipSuccess && hipMalloc(&counts, 192 * sizeof(int)) == hipSuccess &&
              hipMemset(counts, 0, 192 * sizeof(int)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (ok) {
        for (int i = 0; i < 8; i++) host[i] = 0;
        hipLaunchKernelGGL(flow_residency_standin_kernel, dim3(chain_wgs), dim3(256), 0, dstream, host, host + 4);
        const auto t0 = std::chrono::steady_clock::now();
        auto alive = [&]() { for (int i = 0; i < chain_wgs; i++) if (__atomic_load_n(host + i, __ATOMIC_ACQUIRE) == 0) return false; return true; };
        while (!alive() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0) std::this_thread::yield();
        ok = alive();
        if (ok) {
            hipLaunchKernelGGL(flow_residency_probe_kernel, dim3(grid), dim3(256), 0, stream, counts, host + 4);
            std::this_thread::sleep_for(std::chrono::milliseconds(12));     // every workgroup that fits has started and counted itself (3 ms were not always enough right after a big engine's allocations)
        }
        __atomic_store_n(host + 4, 1, __ATOMIC_RELEASE);
        ok = hipStreamSynchronize(stream) == hipSuccess && hipStreamSynchronize(dstream) == hipSuccess && ok &&
             hipMemcpy(h, counts, 192 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (host) hipHostFree(host);
    if (counts) hipFree(counts);
    if (ok) {
        int full = 0;
        for (int x = 0; x < 16; x++)
            if (h[16 + x] > 0) { r.n_xcd++; r.dealt = std::max(r.dealt, h[16 + x]); r.n_se = std::max(r.n_se, h[32 + x] + 1); full = std::max(full, h[x]); }
        r.resident_min = full;
        for (int x = 0; x < 16; x++)
            if (h[16 + x] > 0) { r.resident_min = std::min(r.resident_min, h[x]); r.queued_max = std::max(r.queued_max, h[16 + x] - h[x]); }
        // the rule: of the blocks dealt to an XCD that hosts a chain workgroup, the last  dealt - shader engines x (capacity of the engine
        // whose CU is taken)  leave at once -- on the MI355X 64 - 4 x 14 = 8, i.e. keep = 448 of 512, the value the soaks of rounds 3-5 ran
        // with.  Nothing queued anywhere (another device shape, another footprint): no rule.
        r.se_cap = 1 << 30;      // fewest workgroups resident at once on one shader engine (the one whose CU a chain workgroup holds)
        for (int x = 0; x < 16; x++)
            for (int q = 0; q < r.n_se; q++)
                if (h[16 + x] > 0) r.se_cap = std::min(r.se_cap, h[64 + 8 * x + q]);
        // How many stay queued on such an XCD depends on where the dispatcher's round-robin over the shader engines stands when the kernel
        // starts: blocks are dealt to the engines in turn and the queue stops at the first block that does not fit, i.e. when the engine with
        // the taken CU (capacity se_cap) is offered its (se_cap + 1)-th block -- after n_se * se_cap + (0 .. n_se - 1) blocks: 56 .. 59 of 64 on
        // the MI355X (59 in a fresh process, 56-58 when the first solver is measured at the end of a config-4 engine's creation).  The rule takes
        // the worst case of that geometry, which a later launch may meet whatever this one saw.
        const int leave = r.queued_max > 0 ? std::max(r.queued_max, r.dealt - r.n_se * std::min(r.se_cap, r.dealt)) : 0;
        r.keep = leave > 0 && r.n_xcd > 0 && grid % r.n_xcd == 0 && r.dealt > leave ? r.n_xcd * (r.dealt - leave) : 0;
        r.valid = true;
    }
    (void)hipGetLastError();
    if (getenv("JAICOV_VERBOSE"))
        fprintf(stderr, "jaicov: residency of the tile kernel beside %d chain workgroups (grid %d): %s; %d XCDs, %d shader engines each, %d blocks dealt per XCD, "
                "at least %d resident at once (%d on the fullest-booked shader engine), at most %d queued -> blocks >= %d of a chain workgroup's XCD leave without a ticket\n",
                chain_wgs, grid, r.valid ? "measured" : "NOT measured", r.n_xcd, r.n_se, r.dealt, r.resident_min, r.se_cap, r.queued_max, r.keep);
    // (a measurement taken while another engine of the process was computing can come out short of residents: it is used -- a smaller
    // `keep` costs a few workers, never correctness -- but not remembered, so that the next solver measures again)
    if (r.valid && r.queued_max <= 3 * std::max(1, r.n_se)) cache.push_back({{grid, chain_wgs}, r});
    return r;
}
extern "C" int jaicov_debug_flow_residency(int *out8) {       // tests / DESIGN.md: {valid, XCDs, shader engines, dealt, resident min, queued max, keep, 0}
    hipStream_t s = nullptr, d = stream_acquire(STREAM_DIAGONAL_CUS);
    if (!d || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    const FlowResidency r = flow_measure_residency(s, d, std::min(1024, 2 * cus), 2);
    hipStreamDestroy(s);
    stream_release(STREAM_DIAGONAL_CUS, d);
    out8[0] = r.valid; out8[1] = r.n_xcd; out8[2] = r.n_se; out8[3] = r.dealt; out8[4] = r.resident_min; out8[5] = r.queued_max; out8[6] = r.keep; out8[7] = r.se_cap;
    return 0;
}

hipError_t DenseSolver::flow_init() {
    const int nb = nfact / 128, row_blocks = n / 128;
    if (!dstream) return hipErrorNotSupported;
    // chain form unless kernels cannot run side by side (one-kernel form, diagonal blocks inline) or it is switched off
    flow_one_kernel = !flow_kernels_overlap();
    flow_chain = !flow_one_kernel && factor_form() != FACTOR_TWO_STEP;
    // Third chain workgroup for the second subdiagonal: it finishes tile (c+2, c) by the chain's own forward substitution (from L_cc
    // itself: no wait for the inverse workgroup) and subtracts it from tile (c+2, c+1), the tile the chain workgroup needs next.
    // Rounds 2 and 3 measured it flat (22.4 +- 0.3 ms at order 15 104) and round 4's prune removed it -- for a day: once potrf of the
    // 128-block had fallen from 41 to 27 us (potrf_diag.h) the chain workgroup's own cycle (59 us) was no longer the one that
    // binds at small orders; the path potrf(c-1) -> inverse -> product with the inverse -> last update of tile (c+1, c) was (the chain
    // waited 6-9 us per block column for that tile at order 3 072), and that is the path this workgroup shortens: order 3 072
    // 1.70 -> 1.58 ms.  At order 8 192 it is neutral (5.3 vs 5.4 ms) and at 15 104 it changes nothing (22.7 vs 22.8 ms: the tiles it
    // needs are late by the same path one diagonal further out, and the row chain of the tile kernel binds).  Crossover measured at the end of
    // round 4 (ms, two / three chain workgroups): 40 block columns 2.88 / 2.65, 48: 3.60 / 3.39, 56: 4.42 / 4.20, 64: 5.36 / 5.21, 72: 6.65 / 6.53,
    // 80: 8.25 / 8.21, 118: 22.3 / 22.7 -> used below 80 block columns.  JAICOV_FACTOR_FORM=chain2 / chain3 (test hooks): the chain form with two / three workgroups at any order.
    flow_second = flow_chain && factor_form() != FACTOR_CHAIN2 && (nb < 80 || factor_form() == FACTOR_CHAIN3) ? 2 : 0;
    int split_m = 1, split_from = 1 << 30;
    flow_split_rule(nb, &split_m, &split_from);
    flow_partials = 0;
    const std::vector<int4> tasks = flow_schedule(nb, row_blocks, 1, flow_chain, flow_second, split_m, split_from, &flow_partials);
    flow_tasks = (int)tasks.size();
    flow_task_host = tasks;
    HIPCHK(hipMalloc(&flow_task_list, tasks.size() * sizeof(int4)));
    HIPCHK(hipMemcpy(flow_task_list, tasks.data(), tasks.size() * sizeof(int4), hipMemcpyHostToDevice));
    flow_fs = nb;
    flow_words = (size_t)FLOW_CTRL_WORDS + 2 * (size_t)row_blocks * nb + 2 * nb + 2048 + (size_t)flow_partials;   // ... + one state word per workgroup, and room for their copy at the moment a wait runs out
    // The flags live in FINE-GRAINED device memory (coherent across the XCDs while a kernel runs): in ordinary (coarse-grained)
    // memory polls of every flavour -- sc1, system scope, read-modify-write, with an acquire in between -- were seen to miss
    // flags that memory held as set, about once in 1 000-2 400 factorisations ("Visibility").
    flow_flags = nullptr;
    if (hipExtMallocWithFlags((void **)&flow_flags, flow_words * sizeof(int), hipDeviceMallocFinegrained) != hipSuccess) {
        flow_flags = nullptr;
        (void)hipGetLastError();
    }
    if (!flow_flags) {
        // (seen to matter: in ordinary memory a flag store can stay invisible to the other XCDs' polls for as long as the kernel runs)
        fprintf(stderr, "jaicov: no fine-grained device memory for the flags of the dataflow factorisation (%zu words): ordinary memory, rare stalls possible\n", flow_words);
        HIPCHK(hipMalloc(&flow_flags, flow_words * sizeof(int)));
    }
    HIPCHK(hipMemset(flow_flags, 0, flow_words * sizeof(int)));
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    flow_grid = 2 * cus;             // two workgroups per CU; the CU of the diagonal kernel takes none, the surplus stays queued (harmless)
    if (flow_grid < 1) flow_grid = 1;
    if (flow_grid > 1024) flow_grid = 1024;
    HIPCHK(hipMalloc(&flow_scratch, ((size_t)flow_grid * 16384 + 64) * sizeof(double)));
    HIPCHK(hipMemset(flow_scratch + (size_t)flow_grid * 16384, 0, 64 * sizeof(double)));
    if (flow_partials > 0) HIPCHK(hipMalloc(&flow_partial, (size_t)flow_partials * 16384 * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&flow_alive, 4 * sizeof(int), hipHostMallocMapped));
    flow_alive[0] = flow_alive[1] = flow_alive[2] = flow_alive[3] = 0;
    HIPCHK(hipEventCreateWithFlags(&flow_e0, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&flow_e1, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&flow_t0));
    HIPCHK(hipEventCreate(&flow_t1));
    flow_keep = flow_chain ? flow_measure_residency(stream, dstream, flow_grid, flow_second ? 3 : 2).keep : 0;
    flow_ready = true;
    if (getenv("JAICOV_FLOW_TRACE_ON")) HIPCHK(flow_enable_trace(true));     // per-task timestamps, read by flow_report_stall
    return hipSuccess;
}

void DenseSolver::flow_release() {
    if (flow_task_list) hipFree(flow_task_list);
    if (flow_flags) hipFree(flow_flags);
    if (flow_scratch) hipFree(flow_scratch);
    if (flow_partial) hipFree(flow_partial);
    flow_partial = nullptr;
    if (flow_trace) hipFree(flow_trace);
    if (flow_diag_scratch) hipFree(flow_diag_scratch);
    flow_diag_scratch = nullptr;
    if (flow_alive) hipHostFree(flow_alive);
    flow_alive = nullptr;
    for (hipEvent_t e : {flow_e0, flow_e1, flow_t0, flow_t1})
        if (e) hipEventDestroy(e);
    flow_task_list = nullptr; flow_flags = nullptr; flow_scratch = nullptr; flow_trace = nullptr;
    flow_e0 = flow_e1 = flow_t0 = flow_t1 = nullptr;
    flow_ready = false;
}

hipError_t DenseSolver::potrf_flow(hipEvent_t all_ready) {
    const int nb = nfact / 128, row_blocks = n / 128;
    FlowArgs g{};
    g.L = L; g.ld = ld; g.invd = invd;
    g.tasks = flow_task_list; g.n_tasks = flow_tasks; g.nb = nb; g.fs = flow_fs;
    g.ctrl = flow_flags;
    g.done = flow_flags + FLOW_CTRL_WORDS;
    g.applied = g.done + (size_t)row_blocks * nb;
    g.diag_ready = g.applied + (size_t)row_blocks * nb;
    g.factored = g.diag_ready + nb;
    g.wgstate = g.factored + nb;
    g.pflag = g.wgstate + 2048;
    g.partial = flow_partial;
    g.info = d_info;
    g.scratch = flow_scratch;
    g.zeros = flow_scratch + (size_t)flow_grid * 16384;
    // time limit of a single wait, 100 MHz ticks: 0.25 s + 10 x the time the whole factorisation should take at 20 TFLOP/s
    // (order 15 104: 0.25 + 0.57 s... the longest ordinary wait there is ~2 ms)
    const double expect_ms = (double)nfact * nfact * nfact / 3.0 / 20e9;
    g.timeout = 100000LL * (getenv("JAICOV_FLOW_TIMEOUT_MS") ? atoi(getenv("JAICOV_FLOW_TIMEOUT_MS")) : (int)(250 + 10 * expect_ms));   // (the tests set 0)
    g.src = flow_src; g.src_ld = flow_src_ld; g.V = flow_V; g.Bh = flow_Bh; g.d = flow_d; g.U = flow_U; g.bstride = flow_bstride;
    flow_src = nullptr;      // one factorisation only
    g.trace = flow_trace;
    g.ctrace = flow_trace ? flow_trace + 8 * (size_t)flow_tasks : nullptr;
    g.fake_a = 0;
    g.crit_prio = 1;
    g.second_update = flow_second >= 2 ? 1 : 0;
    g.second_wg = flow_second ? 1 : 0;
    // which of the last workgroups of a chain workgroup's XCD take no ticket: from the residency measured beside stand-ins of the chain
    // workgroups (flow_measure_residency; 448 of 512 on the MI355X), 0 = none (nothing stays queued on this device, or not measurable)
    g.keep = flow_chain ? flow_keep : 0;
    g.inv_wt = 1;
    g.crit_span = flow_chain ? 2 : 1;
    g.alive = flow_alive;
    g.seq = ++flow_seq;
    HIPCHK(hipMemsetAsync(flow_flags, 0, flow_words * sizeof(int), stream));
    flow_wg_off = (int)(g.wgstate - g.ctrl);
    HIPCHK(hipMemcpyAsync(flow_flags + FLOW_WG_OFF, &flow_wg_off, sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(d_info, 0, sizeof(int), stream));
    if (flow_one_kernel) {
        // Kernels cannot run side by side here (every dispatch serialised: counter collection, a debugging environment): the
        // diagonal kernel and the tile kernel would wait for each other until the time limit.  ONE kernel, diagonal blocks inline.
        if (!flow_diag_scratch) HIPCHK(hipMalloc(&flow_diag_scratch, (size_t)flow_grid * (128 * DP + 8 * 16 * WDP) * sizeof(double)));
        g.diag_scratch = flow_diag_scratch;
        g.alive = nullptr;
        if (all_ready) HIPCHK(hipStreamWaitEvent(stream, all_ready, 0));
        if (profile) HIPCHK(hipEventRecord(flow_t0, stream));
        hipLaunchKernelGGL((chol_tile_kernel<1, true>), dim3(flow_grid), dim3(256), 0, stream, g);
        if (profile) {
            HIPCHK(hipEventRecord(flow_t1, stream));
            flow_timed = true;
        }
        return hipGetLastError();
    }
    HIPCHK(hipEventRecord(flow_e0, stream));
    HIPCHK(hipStreamWaitEvent(dstream, flow_e0, 0));
    if (all_ready) HIPCHK(hipStreamWaitEvent(dstream, all_ready, 0));
    if (flow_chain) hipLaunchKernelGGL(potrf_chain_kernel, dim3(flow_second ? 3 : 2), dim3(256), 0, dstream, g);
    else hipLaunchKernelGGL(potrf_diag_chain_kernel, dim3(1), dim3(256), 0, dstream, g);
    HIPCHK(hipGetLastError());
    // Residency.  The diagonal kernel needs a whole CU's LDS and runs on the stream whose CU mask holds one CU of every XCD.
    // The tile kernel runs on the ORDINARY stream with all CUs: on a stream masked to the other 248 CUs only 458 of 496
    // workgroups became resident (measured: workgroups are dealt round-robin to the four shader engines of an XCD, the engine
    // that holds the reserved CU takes 14, and the first workgroup that does not fit holds back the rest of that XCD's queue);
    // unmasked, every CU takes two.  So that the tile kernel cannot fill the reserved CUs before the diagonal kernel has one,
    // the host launches it only after the diagonal kernel has said that it is running (one word of host-visible memory).
    {
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        while (__atomic_load_n(flow_alive, __ATOMIC_ACQUIRE) != g.seq ||
               (flow_chain && (__atomic_load_n(flow_alive + 1, __ATOMIC_ACQUIRE) != g.seq ||
                               (flow_second && __atomic_load_n(flow_alive + 2, __ATOMIC_ACQUIRE) != g.seq)))) {
            if ((++spins & 1023) == 0) {
                const double waited_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (waited_s > 30.0) {
                    // the companion never became resident: tell it to leave (its waits watch the abort word), wait for it, and join
                    // the streams again, so that a later factorisation cannot clear the flags under a kernel that still runs
                    const int two = 2;
                    (void)hipMemcpyAsync(flow_flags + FLOW_ABORT, &two, sizeof(int), hipMemcpyHostToDevice, pstream ? pstream : stream);
                    (void)hipStreamSynchronize(pstream ? pstream : stream);
                    (void)hipStreamSynchronize(dstream);
                    (void)hipEventRecord(flow_e1, dstream);
                    (void)hipStreamWaitEvent(stream, flow_e1, 0);
                    return hipErrorLaunchTimeOut;
                }
                if (waited_s > 2e-4) std::this_thread::yield();      // the companion waits for earlier work on `stream`: do not burn the core meanwhile
            }
        }
    }
    if (profile) HIPCHK(hipEventRecord(flow_t0, stream));
    // operand loads TWO k-steps ahead (chol_tile_kernel<2, false>, 256 VGPRs, no scratch).  Measured equal to one step of lead in round 2, when
    // the backlog of the late columns hid everything; with the split update ranges the latency of the short runs at the dependency front shows:
    // 21.6 -> 21.35 ms at 118 block columns, 5.30 -> 5.19 at 64, equal at 24 (round 5)
    hipLaunchKernelGGL((chol_tile_kernel<2, false>), dim3(flow_grid), dim3(256), 0, stream, g);
    if (profile) {
        HIPCHK(hipEventRecord(flow_t1, stream));
        flow_timed = true;
    }
    HIPCHK(hipEventRecord(flow_e1, dstream));
    HIPCHK(hipStreamWaitEvent(stream, flow_e1, 0));
    return hipGetLastError();
}

// Error path (fetch_info found the abort word set): where did the factorisation stop?
void DenseSolver::flow_report_stall() {
    if (!flow_flags) return;
    const int nb = nfact / 128, row_blocks = n / 128;
    std::vector<int> f(flow_words);
    if (hipMemcpy(f.data(), flow_flags, flow_words * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return;
    const int *done = f.data() + FLOW_CTRL_WORDS, *applied = done + (size_t)row_blocks * nb;
    const int *diag_ready = applied + (size_t)row_blocks * nb, *factored = diag_ready + nb;
    int c_inv = 0, c_fac = 0, c_sub = 0;
    while (c_inv < nb && done[(size_t)c_inv * nb + c_inv]) ++c_inv;
    while (c_fac < nb && factored[c_fac]) ++c_fac;
    while (c_sub + 1 < nb && done[(size_t)(c_sub + 1) * nb + c_sub]) ++c_sub;
    int first_open = -1, open_row = -1;
    for (int j = 0; j < nb && first_open < 0; j++)
        for (int i = j + 1; i < row_blocks; i++)
            if (!done[(size_t)i * nb + j]) { first_open = j; open_row = i; break; }
    fprintf(stderr,
            "jaicov: dataflow factorisation abandoned (abort %d): ticket %d of %d, chain form %d, blocks %d; inverses published %d, "
            "factors published %d, subdiagonal tiles published %d, first unfinished tile (%d, %d), chain workgroup at column %d stage %d",
            f[FLOW_ABORT], f[FLOW_TICKET], flow_tasks, (int)flow_chain, nb, c_inv, c_fac, c_sub, open_row, first_open, f[FLOW_CHAIN_AT] >> 4, f[FLOW_CHAIN_AT] & 15);
    if (flow_chain && c_fac < nb) {
        const int c = c_fac > 0 ? c_fac - 1 : 0;      // the column the chain workgroup was (probably) working on
        if (c + 1 < nb)
            fprintf(stderr, "; column %d: tile below stored with %d updates, next diagonal tile with %d (wanted %d)", c,
                    applied[(size_t)(c + 1) * nb + c] - 1, applied[(size_t)(c + 1) * nb + c + 1] - 1, c);
    }
    fprintf(stderr, "\n");
    for (int b = 0; b < 3; b++) {
        const unsigned hw = (unsigned)f[8 + 2 * b];
        if (flow_chain && (b < 2 || flow_second))
            fprintf(stderr, "jaicov:   chain workgroup %d runs on [xcc %d se %u sh %u cu %u]\n", b, f[9 + 2 * b] - 1, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf);
    }
    // the oldest tickets still in flight, and what their workgroups were doing (1 drawn, 2 waiting for an earlier visit / the
    // inverse, 3 polling operand flags at block column k, 4 products, 5 / 6 storing after phase 0 / 1, 7 finished, 8 gave up)
    const int *wg = factored + nb;
    std::vector<std::pair<int, int>> open;
    for (int b = 0; b < std::min(flow_grid, 1024); b++)
        if (wg[b] != 0 && (wg[b] & 15) != 7) open.push_back({wg[b] >> 12, b});
    std::sort(open.begin(), open.end());
    std::vector<long long> tr;
    if (flow_trace) {
        tr.resize((size_t)flow_tasks * 8);
        if (hipMemcpy(tr.data(), flow_trace, tr.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) tr.clear();
    }
    long long t_last = 0;
    for (size_t q = 0; q + 7 < tr.size(); q += 8) t_last = std::max(t_last, tr[q + 3]);
    auto ticket_of = [&](int i, int j) {
        for (int t = 0; t < flow_tasks; t++)
            if (flow_task_host[t].x == i && flow_task_host[t].y == j) return t;
        return -1;
    };
    // What every workgroup was doing at the MOMENT the first wait ran out (flow_give_up's copy of the state words): the ones that were not
    // waiting then -- inside the products, storing, adding partial sums -- are the ones the others were waiting for.
    {
        int by_stage[16] = {0};
        std::vector<std::pair<int, int>> busy;
        for (int b = 0; b < std::min(flow_grid, 1024); b++) {
            const int w = wg[1024 + b];
            if (w == 0) continue;
            by_stage[w & 15]++;
            if ((w & 15) == 4 || (w & 15) == 5 || (w & 15) == 6 || (w & 15) == 1) busy.push_back({w >> 12, b});
        }
        std::sort(busy.begin(), busy.end());
        fprintf(stderr, "jaicov:   when the first wait ran out: workgroups by stage [1 drawn %d, 2 waits for a visit / the inverse %d, 3 polls operands %d, 4 products %d, 5 / 6 stores %d / %d, 7 between tasks %d, 9 waits for partial sums %d]\n",
                by_stage[1], by_stage[2], by_stage[3], by_stage[4], by_stage[5], by_stage[6], by_stage[7], by_stage[9]);
        for (size_t q = 0; q < busy.size() && q < 8; q++) {
            const int w = wg[1024 + busy[q].second], t = w >> 12;
            const int4 tk = (t >= 0 && t < (int)flow_task_host.size()) ? flow_task_host[t] : make_int4(-1, -1, 0, 0);
            fprintf(stderr, "jaicov:     not waiting then: workgroup %d, ticket %d = tile (%d, %d)%s, stage %d at block column %d; now: stage %d at block column %d\n", busy[q].second, t, tk.x, tk.y,
                    (tk.w & FLOW_PART) ? " partial sum" : "", w & 15, (w >> 4) & 255, wg[busy[q].second] & 15, (wg[busy[q].second] >> 4) & 255);
        }
    }
    for (size_t q = 0; q < open.size() && q < 6; q++) {
        const int w = wg[open[q].second], t = w >> 12;
        const int4 tk = flow_task_host.empty() ? make_int4(-1, -1, 0, 0) : flow_task_host[t];
        fprintf(stderr, "jaicov:   workgroup %d: ticket %d = tile (%d, %d) to block column %d%s, stage %d at block column %d\n", open[q].second, t, tk.x,
                tk.y, tk.w & (FLOW_FIN - 1), (tk.w & FLOW_FIN) ? " + finish" : "", w & 15, (w >> 4) & 255);
        // with JAICOV_FLOW_TRACE_ON: when were the two operand tiles it was waiting for finished (ms before the last recorded event)?
        const int kk = (w >> 4) & 255;
        if (!tr.empty() && (w & 15) == 8 && kk < tk.y)
            for (int side = 0; side < 2; side++) {
                const int oi = side == 0 ? tk.x : tk.y, ot = ticket_of(oi, kk);
                if (ot < 0) continue;
                const long long *r = &tr[(size_t)ot * 8];
                const unsigned hw = (unsigned)(r[7] & 0xffffffff);
                const int ws = wg[1024 + (int)(r[6] & 0xffff)];
                fprintf(stderr, "jaicov:     (when the first wait ran out that workgroup was at: ticket %d, block column %d, stage %d)\n", ws >> 12, (ws >> 4) & 255, ws & 15);
                fprintf(stderr, "jaicov:     operand tile (%d, %d) = ticket %d on workgroup %d [xcc %d se %u sh %u cu %u simd %u wave %u]: started %.3f ms, tile loaded %.3f ms, updates done %.3f ms, finished %.3f ms before the end; it waited %.3f ms\n",
                        oi, kk, ot, (int)(r[6] & 0xffff), (int)((r[7] >> 32) & 0xf), (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3, hw & 0xf,
                        (t_last - r[0]) / 1e5, (t_last - r[1]) / 1e5, (t_last - r[2]) / 1e5, r[3] ? (t_last - r[3]) / 1e5 : -1.0, r[4] / 1e5);
            }
    }
}

// debug: per-task timeline of the next factorisations (scripts/flow_trace.py)
hipError_t DenseSolver::flow_enable_trace(bool on) {
    if (on && !flow_trace) {
        const size_t words = ((size_t)flow_tasks + nfact / 128) * 8;      // per task, then per block column (chain kernel)
        HIPCHK(hipMalloc(&flow_trace, words * sizeof(long long)));
        HIPCHK(hipMemset(flow_trace, 0, words * sizeof(long long)));
    } else if (!on && flow_trace) {
        hipFree(flow_trace);
        flow_trace = nullptr;
    }
    return hipSuccess;
}

}  // namespace jaicov
