// Dense fp64 SPD machinery on the device: blocked Cholesky, multi-rhs substitution, triangular inverse, L^-T L^-1.
// Storage: row-major, LOWER triangle, order n (multiple of 128), leading dimension ld.  The same memory read
// column-major is LAPACK's UPLO='U' -- which is how MTJ's UpperSymmPackMatrix (packed) maps onto it.
// Replaces the arithmetic of MathExtension.solve / MathExtension.inv (MathExtension.java:239-264,304-324,338-366),
// i.e. netlib dppsv/dpptrf/dpptri and (through the bordered formulation in engine.hip) dspsv/dsptri.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <vector>

namespace jaicov {

struct GemmArgs;

constexpr int DENSE_NB = 128;      // diagonal block handled by one workgroup in LDS
constexpr int DENSE_MAX_RHS = 8;   // rhs vectors the substitution kernels carry at once
enum { FACTOR_DEFAULT = 0, FACTOR_STREAMS = 1, FACTOR_TWO_STEP = 2, FACTOR_ONE_KERNEL = 3, FACTOR_CHAIN2 = 4, FACTOR_CHAIN3 = 5 };
void flow_split_rule(int nb, int *m, int *from);   // cholflow.hip
int factor_form();                 // JAICOV_FACTOR_FORM (dense.hip): test hook for the non-default forms of the factorisation

// Streams are kept for the life of the process (dense.hip): creating one costs 12-35 ms on this stack (a hardware queue is set up), destroying
// one 5 ms, and an engine needs eight -- 100 ms of the 225 ms an engine creation took at BASELINE config 4.  stream_acquire hands out a
// stream nobody else holds (a cached one if there is one of that kind for the current device, else a new one; nullptr if it cannot
// be created); stream_release synchronises it and puts it back.  CU-masked streams are the exception to "kept": each is a hardware queue
// of its own, idle or not, and a process that holds more than ~16 has ALL its queues time-sliced (round 5: 94 block columns 19.5 ms with 8
// idle ones beside the solver's, 28.8 with 16, 58 with 32; scripts/queue_count_probe.py) -- at most STREAM_MASKED_IDLE_MAX of a kind stay
// in the pool, the rest are destroyed on release.
enum { STREAM_PLAIN = 0, STREAM_HIGH_PRIORITY = 1, STREAM_UPDATE_CUS = 2, STREAM_DIAGONAL_CUS = 3, STREAM_KINDS = 4 };
constexpr int STREAM_MASKED_IDLE_MAX = 2;
hipStream_t stream_acquire(int kind);
void stream_release(int kind, hipStream_t s);

struct DenseSolver {
    hipStream_t stream = nullptr;
    int n = 0;                 // rows of the storage, multiple of 128: nfact, plus 128 right-hand-side rows if `aug`
    int nfact = 0;             // padded order of the matrix that is factorised
    bool aug = false;          // right-hand sides are carried through the factorisation as rows nfact.. of L
    long ld = 0;
    int nbo = 512;             // outer panel width of the factorisation (multiple of 128)
    bool lookahead = true;     // factor panel s+1 on `pstream` while the rest of trailing update s runs
    hipStream_t pstream = nullptr;   // panel GEMMs (high priority)
    hipStream_t ustream = nullptr;   // trailing updates: all CUs except the reserved ones (CU mask)
    hipStream_t dstream = nullptr;   // diagonal-block kernel: the reserved CUs
    std::vector<hipEvent_t> sync_ev;
    size_t ev_used = 0;
    hipEvent_t next_event();
    std::map<int, std::pair<int2 *, int>> tile_maps;   // XCD-aware tile order of the trailing update, by tile rows
    bool xcd_maps = true;
    int2 *tile_map_store = nullptr;
    std::map<long, int2 *> trtri_maps;   // tile orders of the triangular inverse's products, longest k-range first (dense.hip)
    const int2 *trtri_tile_order(int tm, int tn, int kind);
    double *L = nullptr;       // n x ld : input SPD matrix (lower) -> Cholesky factor (lower)
    double *invd = nullptr;    // (n/128) x 128 x 128 : inverses of the diagonal blocks of L
    double *pm = nullptr;      // 2 x 5 x (n/128) x 128 x 128 : the chains' pre-multiplied blocks P_1 .. P_5, Ft_1 .. Ft_5 (premultiply(), dense.hip CH_PM)
    double *xch = nullptr;     // n : the two workgroups of a block column pass their halves of v through this (backsolve_chain8_kernel<2>)
    hipError_t launch_chain8(const double *Zrow, double *X, const int *abort_word, long long *trace);
    bool chain8_split() const;  // two workgroups per block column / row in the one-right-hand-side chains (the grid must be resident at once)
    bool pm_ready = false;     // pm belongs to the factor at hand (potrf() clears it)
    hipEvent_t pm_e0 = nullptr, pm_done = nullptr;   // the Ft half of premultiply() runs on pstream
    bool pm_wait = false;      // ... and `stream` has not waited for it yet
    hipError_t premultiply();
    double *W = nullptr;       // n x ld : L^-1 (lower), only for the inverse
    double *Q = nullptr;       // n x ld : (L L')^-1 (lower tiles valid; symmetrize() fills the rest); workspace of trtri()
    int *d_info = nullptr;     // first failing pivot (1-based), 0 = ok
    bool owns = false;
    bool borrowed_streams = false;   // pstream (and what it holds of ustream / dstream) belong to another solver (init's `share`)
    bool own_ustream = false, own_dstream = false;   // ... the masked streams this solver acquired itself
    // optional per-launch profiling of the trailing update (HIP events on `stream`)
    bool profile = false;
    std::vector<hipEvent_t> prof_ev;     // pairs
    std::vector<double> prof_flops;
    size_t prof_used = 0;
    double stat_launches = 0, stat_ms = 0, stat_flops = 0;
    void prof_collect();                 // call after the stream has been synchronised
    double flops_order = 0;              // real (unpadded) order of the matrix, for the algorithmic flop count n^3/3; 0 = nfact

    // dataflow factorisation (cholflow.hip): the whole potrf as two concurrent launches, dependencies as flags in memory
    bool flow_ready = false, flow_timed = false, flow_chain = false, flow_one_kernel = false;
    int flow_second = 0;                 // third chain workgroup (cholflow.hip): 0 = none, 2 = it finishes tile (c+2, c) and subtracts it from (c+2, c+1)
    int flow_wg_off = 0;                 // offset of the per-workgroup state words in flow_flags (read by the device)
    long long flow_stale_events = 0, flow_stale_confirmed = 0, flow_rescued = 0;   // flags that only the read-modify-write poll saw (fetch_info)
    std::vector<int4> flow_task_host;    // the task list (flow_report_stall)
    int reserved_cus = 8;                // CUs kept free of the update stream for the diagonal-block kernel
    int4 *flow_task_list = nullptr;
    int flow_tasks = 0, flow_fs = 0, flow_grid = 0;
    int flow_keep = 0;                   // blocks >= this on an XCD that hosts a chain workgroup take no ticket (measured at flow_init, cholflow.hip)
    int *flow_flags = nullptr;           // control words, done / applied flags, diag_ready
    size_t flow_words = 0;
    double *flow_scratch = nullptr;
    double *flow_partial = nullptr;      // partial sums of split update ranges (cholflow.hip, FLOW_PART)
    int flow_partials = 0;               // ... how many 128 x 128 buffers
    long long *flow_trace = nullptr;
    double *flow_diag_scratch = nullptr; // one-kernel form (kernels cannot overlap on this host): work arrays of the inline diagonal blocks
    bool flow_kernels_overlap();         // probed once per process
    int *flow_alive = nullptr;           // host-visible: sequence number of the last diagonal kernel that has started
    int flow_seq = 0;
    hipEvent_t flow_e0 = nullptr, flow_e1 = nullptr, flow_t0 = nullptr, flow_t1 = nullptr;
    // source of the NEXT potrf(): M = V N V + Bh' Bh read straight from N by the tile kernel (no scaled copy into L first)
    const double *flow_src = nullptr, *flow_V = nullptr, *flow_Bh = nullptr;
    long flow_src_ld = 0;
    int flow_d = 0, flow_U = 0, flow_bstride = 0;
    void flow_set_source(const double *N, long ldN, const double *V, const double *Bh, int bstride, int d, int U) {
        flow_src = N; flow_src_ld = ldN; flow_V = V; flow_Bh = Bh; flow_bstride = bstride; flow_d = d; flow_U = U;
    }
    hipError_t flow_init();
    void flow_release();
    hipError_t potrf_flow(hipEvent_t all_ready);
    hipError_t potrf_streams(hipEvent_t first_ready, hipEvent_t all_ready);   // the stream / event scheduled factorisation
    hipError_t flow_enable_trace(bool on);
    void flow_report_stall();            // one line on stderr: how far the abandoned factorisation got

    // `share`: a solver of the same engine that is never at work at the same time (the full-order and the EO-reduced solver of an engine):
    // its side streams are used instead of streams of this solver's own -- a CU-masked stream is a HARDWARE QUEUE, and beyond ~16 of them
    // in a process the scheduler time-slices the queues, persistent kernels included ("Hardware queues")
    hipError_t init(hipStream_t s, int n_padded, bool with_inverse, bool with_rhs_rows = false, const DenseSolver *share = nullptr);
    double *rhs_row(int q) const { return L + (long)(nfact + q) * ld; }   // row q of the right-hand sides / of Z = Y L^-T
    void release();
    hipError_t panel(hipStream_t st, int K0, int K1);
    hipError_t timed_gemm(hipStream_t st, const GemmArgs &u, double flops, int small = 0);
    // L <- chol(L); info via fetch_info().  first_ready / all_ready (optional, events on `stream`): the first panel's
    // columns [0, first_panel_cols()) resp. the whole matrix are in place -- the first panel then factors while the
    // caller is still filling the rest.
    hipError_t potrf(hipEvent_t first_ready = nullptr, hipEvent_t all_ready = nullptr);
    hipError_t begin_refactor();                            // before the first write into L for a new factorisation (waits for the side stream's readers of the old one)
    int first_panel_cols() const;
    hipError_t backsolve_aug(double *X, long xs, int nrhs);  // L' X = Z, Z = the rhs rows after potrf(); X rows have stride xs
    hipError_t solve_rhs(const double *b, double *tmp, double *X);   // X = (L L')^-1 b, one right-hand side (forward + backward chain)
    hipError_t trtri();                                     // W <- L^-1
    hipError_t lauum();                                     // Q <- W' W (lower tiles)
    hipError_t symmetrize(double *M);                       // copy lower -> upper
    int fetch_info();                                       // synchronises the stream; > 0 failing pivot, -9 dataflow stalled
};

}  // namespace jaicov
