// Dataflow Cholesky: the whole factorisation L L' = M (plus the right-hand-side rows below M) as TWO concurrent launches.
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
//   potrf_diag_chain_kernel   ONE workgroup on a reserved CU (it needs 150 KB of LDS): for c = 0, 1, ...: waits for the
//                       updated diagonal tile c, factors and inverts it (potrf_diag.h), publishes L_cc and inv(L_cc).
//
// Order and progress.  Tickets are handed out in the order of the host-built task list, which is a topological order of
// the dependencies (column-major: a task only ever waits for tiles of earlier tickets, and for the diagonal kernel, which
// only waits for a task).  A ticket is drawn by a RUNNING workgroup, so every ticket below a waiting workgroup's own is
// held by a workgroup that is running or has finished: no deadlock whatever the residency, and no assumption about the
// dispatch order.  The diagonal kernel runs on its own CU-masked stream, so it is resident whatever the tile kernel fills.
// Every spin is bounded (wall clock): a stall sets the abort word, every waiter leaves, the host reports an error.
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility; the per-XCD L2s are not coherent, a CU's L1 is never
// refreshed by other CUs' stores).  Producer: every byte that another workgroup will read is stored write-through
// (`sc1`, agent-scope relaxed atomic stores: the line leaves the storing XCD's L2), every storing wave drains
// (`s_waitcnt vmcnt(0)`), the workgroup's barrier, then one lane stores the flag (`sc1`).  The diagonal kernel stores
// plainly and publishes behind an agent-scope release.  Consumer: one lane polls with relaxed `sc1` loads, then ONE
// agent-scope acquire (invalidates this CU's L1) + `s_waitcnt vmcnt(0)` + barrier, then plain loads.  No stale line can
// sit in a reader's L2: a tile is read either by the one workgroup that also writes it next (its own XCD's L2, and the
// write drops the line), or only after it has become final, after which it is never written again.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

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
enum { FLOW_TICKET = 0, FLOW_ABORT = 1, FLOW_DIAG_NEXT = 2, FLOW_CTRL_WORDS = 16 };
constexpr int FLOW_FIN = 1 << 20;        // task.w = k1 | FLOW_FIN: finish the tile after the updates
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
    int *applied;            // [row blocks][fs]  block columns [0, applied) have been subtracted from the stored tile (partial visits)
    int *diag_ready;         // [nb] the updated diagonal tile is in memory
    int *info;               // first failing pivot (dense.h)
    double *scratch;         // [grid][128 x 128] per workgroup: operand of the multiplication by inv(L_jj)'
    long long timeout;       // wall-clock ticks (100 MHz) a wait may take before the factorisation is abandoned
    long long *trace;        // optional [n_tasks][8]: start, C loaded, updates done, end (wall clock), ticks spent waiting, XCC id
};

__device__ __forceinline__ int flow_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flow_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_wt(double *p, double v) {   // write-through (sc1) store of one double
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
            __builtin_amdgcn_s_sleep(40);
            if ((spins & 31) == 0) {
                if (flow_ld(ctrl + FLOW_ABORT) != 0) { ok = false; break; }
                if (wall_clock64() - t0 > timeout) { flow_st(ctrl + FLOW_ABORT, 2); ok = false; break; }
            }
        }
    }
    if (waited) *waited += wall_clock64() - t0;
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void chol_tile_kernel(FlowArgs g) {
    __shared__ double smem[2 * FLOW_STAGE];
    __shared__ int s_msg[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // operand staging (gemm_f64.h, KC layout: 8 lanes fetch the 16 k of one row, transposed on the way into LDS)
    const int seg = tid & 7, rbase = tid >> 3;
    const int odd = (seg & 1) * FLOW_LDS;
    const int st_lds = (2 * seg) * FLOW_LDS + rbase + 8 * ((seg >> 1) & 1) + 4 * (seg >> 2);
    const int fa = (lane >> 4) * FLOW_LDS + 64 * wr + (lane & 15);
    const int fb = (lane >> 4) * FLOW_LDS + 64 * wc + (lane & 15);
    double *const my_scratch = g.scratch + (long)blockIdx.x * 16384;

    d4_t acc[4][4];
    // acc += A B' over nk k-steps of 16: A(x, k) = Ap[x * lda + k], B(y, k) = Bp[y * ldb + k], x, y < 128
    auto accumulate = [&](const double *Ap, long lda, const double *Bp, long ldb, int nk) __attribute__((always_inline)) {
        const double *ap = Ap + (long)rbase * lda + 2 * seg;
        const double *bp = Bp + (long)rbase * ldb + 2 * seg;
        d2_t ra[4], rb[4];
        auto gload = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; q++) ra[q] = *reinterpret_cast<const d2_t *>(ap + (long)(32 * q) * lda);
#pragma unroll
            for (int q = 0; q < 4; q++) rb[q] = *reinterpret_cast<const d2_t *>(bp + (long)(32 * q) * ldb);
            ap += GEMM_BK;
            bp += GEMM_BK;
        };
        auto lstore = [&](int stage) __attribute__((always_inline)) {
            double *sa = smem + stage * FLOW_STAGE + st_lds;
            double *sb = sa + GEMM_BK * FLOW_LDS;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                sa[32 * q + odd] = ra[q].x;
                sa[32 * q + FLOW_LDS - odd] = ra[q].y;
                sb[32 * q + odd] = rb[q].x;
                sb[32 * q + FLOW_LDS - odd] = rb[q].y;
            }
        };
        gload();
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt++) {
            const int st = kt & 1;
            if (kt + 1 < nk) gload();
            const double *sa = smem + st * FLOW_STAGE + fa;
            const double *sb = smem + st * FLOW_STAGE + GEMM_BK * FLOW_LDS + fb;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                double a[4], b[4];
                const int sh = 8 * (ks & 1) + 4 * (ks >> 1);
#pragma unroll
                for (int x = 0; x < 4; x++) a[x] = sa[(4 * ks) * FLOW_LDS + 16 * x + sh];
#pragma unroll
                for (int y = 0; y < 4; y++) b[y] = sb[(4 * ks) * FLOW_LDS + 16 * y + sh];
#pragma unroll
                for (int x = 0; x < 4; x++)
#pragma unroll
                    for (int y = 0; y < 4; y++) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
            }
            if (kt + 1 < nk) lstore(st ^ 1);
            __syncthreads();
        }
    };

    for (;;) {
        if (tid == 0) s_msg[0] = atomicAdd(g.ctrl + FLOW_TICKET, 1);
        __syncthreads();
        const int t = s_msg[0];
        __syncthreads();
        if (t >= g.n_tasks) break;
        const int4 tk = g.tasks[t];
        const int ti = tk.x, tj = tk.y, k0 = tk.z, k1 = tk.w & (FLOW_FIN - 1);
        const bool fin = (tk.w & FLOW_FIN) != 0;
        long long waited = 0, t_start = 0, t_c = 0, t_upd = 0;
        if (g.trace) t_start = wall_clock64();
        bool ok = true;
        // ---- the tile as it stands ------------------------------------------------------------------------------
        if (k0 > 0) {   // an earlier (partial) visit wrote it: wait for that visit
            if (tid == 0) {
                const bool r = flow_spin(g.applied + (long)ti * g.fs + tj, k0, g.ctrl, g.timeout, &waited);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                drain_stores();
                s_msg[1] = r ? 1 : 0;
            }
            __syncthreads();
            ok = s_msg[1] != 0;
            __syncthreads();
            if (!ok) break;
        }
        double *const ctile = g.L + ((long)ti * 128 + 64 * wr + (lane >> 4)) * g.ld + (long)tj * 128 + 64 * wc + (lane & 15);
#pragma unroll
        for (int x = 0; x < 4; x++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double *rowp = ctile + (long)(16 * x + 4 * r) * g.ld;
#pragma unroll
                for (int y = 0; y < 4; y++) acc[x][y][r] = -rowp[16 * y];
            }
        if (g.trace) t_c = wall_clock64();
        // ---- runs of MFMA work: updates with the block columns whose operand tiles are final, then (off-diagonal tile that is
        //      to be finished) the multiplication by inv(L_jj)'.  ONE instance of the pipelined loop serves both. ------------
        int k = k0;
        bool solved = false;          // the accumulators hold L[i][j] (after the multiplication by inv(L_jj)')
        for (;;) {
            const double *Ap, *Bp;
            long lda, ldb;
            int nk;
            if (k < k1) {
                if (wave == 0) {
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
                            __builtin_amdgcn_s_sleep(40);
                            if ((spins & 31) == 0) {
                                if (flow_ld(g.ctrl + FLOW_ABORT) != 0) break;
                                if (wall_clock64() - t0 > g.timeout) { flow_st(g.ctrl + FLOW_ABORT, 2); break; }
                            }
                        }
                    }
                    if (spins > 0) waited += wall_clock64() - t0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    drain_stores();
                    if (lane == 0) s_msg[1] = cnt;
                }
                __syncthreads();
                const int cnt = s_msg[1];
                __syncthreads();
                if (cnt == 0) { ok = false; break; }
                Ap = g.L + (long)ti * 128 * g.ld + (long)k * 128;
                Bp = g.L + (long)tj * 128 * g.ld + (long)k * 128;
                lda = ldb = g.ld;
                nk = 8 * cnt;
                k += cnt;
            } else {
                if (g.trace && !solved) t_upd = wall_clock64();
                // what leaves the accumulators, where to, and which flag tells whom
                double *dst;
                long dld;
                double sign;
                int *flag;
                int flag_value = 1;
                const bool to_scratch = fin && ti != tj && !solved;
                if (to_scratch) {   // C for the multiplication by inv(L_jj)': through this workgroup's scratch tile
                    dst = my_scratch + (64 * wr + (lane >> 4)) * 128 + 64 * wc + (lane & 15); dld = 128; sign = -1.0; flag = nullptr;
                } else if (solved) {   // L[i][j]
                    dst = ctile; dld = g.ld; sign = 1.0; flag = g.done + (long)ti * g.fs + tj;
                } else if (fin) {      // updated diagonal tile, for the diagonal kernel
                    dst = ctile; dld = g.ld; sign = -1.0; flag = g.diag_ready + tj;
                } else {               // partial visit
                    dst = ctile; dld = g.ld; sign = -1.0; flag = g.applied + (long)ti * g.fs + tj; flag_value = k1;
                }
#pragma unroll
                for (int x = 0; x < 4; x++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        double *rowp = dst + (long)(16 * x + 4 * r) * dld;
#pragma unroll
                        for (int y = 0; y < 4; y++) store_wt(rowp + 16 * y, sign * acc[x][y][r]);
                    }
                drain_stores();
                if (!to_scratch) {
                    __syncthreads();
                    if (tid == 0) flow_st(flag, flag_value);
                    break;
                }
                if (tid == 0) {
                    const bool r = flow_spin(g.done + (long)tj * g.fs + tj, 1, g.ctrl, g.timeout, &waited);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    drain_stores();
                    s_msg[1] = r ? 1 : 0;
                }
                __syncthreads();
                ok = s_msg[1] != 0;
                __syncthreads();
                if (!ok) break;
#pragma unroll
                for (int x = 0; x < 4; x++)
#pragma unroll
                    for (int y = 0; y < 4; y++) acc[x][y] = (d4_t){0.0, 0.0, 0.0, 0.0};
                Ap = my_scratch;
                Bp = g.invd + (long)tj * 16384;
                lda = ldb = 128;
                nk = 8;
                solved = true;
            }
            accumulate(Ap, lda, Bp, ldb, nk);
        }
        if (!ok) break;
        if (g.trace && tid == 0) {
            long long *tr = g.trace + 8 * (long)t;
            tr[0] = t_start; tr[1] = t_c; tr[2] = t_upd; tr[3] = wall_clock64(); tr[4] = waited;
            tr[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf;   // XCC_ID
            tr[6] = blockIdx.x;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void potrf_diag_chain_kernel(FlowArgs g) {
    __shared__ double S[128 * DP];
    __shared__ double Wd[8 * 16 * WDP];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
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

// ---------------------------------------------------------------------------------------------------------------
// Task list: left-looking, column-major.  Column j: the diagonal tile first, then the tiles below it (the tile right below
// feeds the next diagonal tile: it is the one the chain waits for), the right-hand-side rows last.
static std::vector<int4> flow_schedule(int nb, int row_blocks) {
    std::vector<int4> tasks;
    tasks.reserve((size_t)nb * (row_blocks + 1) / 2 + row_blocks);
    for (int j = 0; j < nb; j++)
        for (int i = j; i < row_blocks; i++) tasks.push_back(make_int4(i, j, 0, j | FLOW_FIN));
    return tasks;
}

hipError_t DenseSolver::flow_init() {
    const int nb = nfact / 128, row_blocks = n / 128;
    if (!ustream || !dstream) return hipErrorNotSupported;
    const std::vector<int4> tasks = flow_schedule(nb, row_blocks);
    flow_tasks = (int)tasks.size();
    HIPCHK(hipMalloc(&flow_task_list, tasks.size() * sizeof(int4)));
    HIPCHK(hipMemcpy(flow_task_list, tasks.data(), tasks.size() * sizeof(int4), hipMemcpyHostToDevice));
    flow_fs = nb;
    flow_words = (size_t)FLOW_CTRL_WORDS + 2 * (size_t)row_blocks * nb + nb;
    HIPCHK(hipMalloc(&flow_flags, flow_words * sizeof(int)));
    HIPCHK(hipMemset(flow_flags, 0, flow_words * sizeof(int)));
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    flow_grid = 2 * (cus - 8);       // two workgroups per CU of the update stream's mask (dense.hip: CU 31 of every XCD is reserved)
    if (const char *e = getenv("JAICOV_FLOW_GRID")) flow_grid = atoi(e);
    if (flow_grid < 1) flow_grid = 1;
    HIPCHK(hipMalloc(&flow_scratch, (size_t)flow_grid * 16384 * sizeof(double)));
    HIPCHK(hipEventCreateWithFlags(&flow_e0, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&flow_e1, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&flow_e2, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&flow_t0));
    HIPCHK(hipEventCreate(&flow_t1));
    flow_ready = true;
    return hipSuccess;
}

void DenseSolver::flow_release() {
    if (flow_task_list) hipFree(flow_task_list);
    if (flow_flags) hipFree(flow_flags);
    if (flow_scratch) hipFree(flow_scratch);
    if (flow_trace) hipFree(flow_trace);
    for (hipEvent_t e : {flow_e0, flow_e1, flow_e2, flow_t0, flow_t1})
        if (e) hipEventDestroy(e);
    flow_task_list = nullptr; flow_flags = nullptr; flow_scratch = nullptr; flow_trace = nullptr;
    flow_e0 = flow_e1 = flow_e2 = flow_t0 = flow_t1 = nullptr;
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
    g.info = d_info;
    g.scratch = flow_scratch;
    g.timeout = 100000000LL * (getenv("JAICOV_FLOW_TIMEOUT_S") ? atoi(getenv("JAICOV_FLOW_TIMEOUT_S")) : 10);
    g.trace = flow_trace;
    HIPCHK(hipMemsetAsync(flow_flags, 0, flow_words * sizeof(int), stream));
    HIPCHK(hipMemsetAsync(d_info, 0, sizeof(int), stream));
    HIPCHK(hipEventRecord(flow_e0, stream));
    HIPCHK(hipStreamWaitEvent(dstream, flow_e0, 0));
    HIPCHK(hipStreamWaitEvent(ustream, flow_e0, 0));
    if (all_ready) {
        HIPCHK(hipStreamWaitEvent(dstream, all_ready, 0));
        HIPCHK(hipStreamWaitEvent(ustream, all_ready, 0));
    }
    hipLaunchKernelGGL(potrf_diag_chain_kernel, dim3(1), dim3(256), 0, dstream, g);
    if (profile) HIPCHK(hipEventRecord(flow_t0, ustream));
    hipLaunchKernelGGL(chol_tile_kernel, dim3(flow_grid), dim3(256), 0, ustream, g);
    if (profile) {
        HIPCHK(hipEventRecord(flow_t1, ustream));
        flow_timed = true;
    }
    HIPCHK(hipEventRecord(flow_e1, dstream));
    HIPCHK(hipEventRecord(flow_e2, ustream));
    HIPCHK(hipStreamWaitEvent(stream, flow_e1, 0));
    HIPCHK(hipStreamWaitEvent(stream, flow_e2, 0));
    return hipGetLastError();
}

// debug: per-task timeline of the next factorisations (scripts/flow_trace.py)
hipError_t DenseSolver::flow_enable_trace(bool on) {
    if (on && !flow_trace) {
        HIPCHK(hipMalloc(&flow_trace, (size_t)flow_tasks * 8 * sizeof(long long)));
        HIPCHK(hipMemset(flow_trace, 0, (size_t)flow_tasks * 8 * sizeof(long long)));
    } else if (!on && flow_trace) {
        hipFree(flow_trace);
        flow_trace = nullptr;
    }
    return hipSuccess;
}

}  // namespace jaicov
