// Bundle-adjustment device kernels: residual/Jacobian rows, normal-equation assembly, omega.  gfx950 only.
// Reference arithmetic (paths relative to JAICOV/src/org/applied_geodesy/adjustment/bundle/derivation/):
//   PartialDerivativeFactory.java (PDF), DistortionModelFactory.java (DMF), RadiallySymmetric.. (RSF),
//   Tangential.. (TDF), AffinityShear.. (ASF), RadialDistance..DistortionModelFactory.java (RDF).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/jaicov_neq.h"

namespace jaicov {

constexpr int MAXD = JAICOV_MAX_DIST_PER_CAMERA;   // 20
constexpr int KROW = 12 + MAXD;                    // local row layout: X,Y,Z,x0,y0,c,X0,Y0,Z0,omega,phi,kappa,dist..
constexpr int KC_MAX = 9 + MAXD;                   // shared (camera + EO) local columns: io(3), eo(6), dist
constexpr int KC_LD = KC_MAX + 1;                  // + the misclosure column in T = P [A_c | w]
constexpr int SEG = 256;                           // image points per assembly segment
constexpr int SCHUR_GLD = 24;                      // row length of G = U' [A_io, A_dist | w] (schur.hip)

// Device-side view of jaicov_problem_desc (device pointers) + derived tables
struct DevProblem {
    int U, Upad, d, n_points, n_cameras, n_images, n_dist, n_ip, n_blocks, n_sb, n_dg, n_dg_rows, n_slots;
    long ld;
    const int32_t *point_col, *io_col, *cam_dist_begin, *dist_kind, *dist_order, *dist_col, *image_camera, *eo_col;
    const double *cam_r0;
    const int32_t *ip_image, *ip_point;
    const double *ip_x, *ip_y, *ip_var_x, *ip_var_y, *ip_rho;
    const int32_t *blk_ip_begin;       // [n_blocks+1]
    const int64_t *blk_w_offset;       // [n_blocks] offset of the block's D^-1 (m x m row-major) in blk_w; < 0: block-diagonal weights, see ip_w3
    const double *blk_w;
    const double *ip_w3;               // [3 n_ip] or null: (w00, w01, w11) of inv([[vx, rho s], [rho s, vy]]) per image point -- the weights of an
                                       // ORDINARY image served as an image block (engine.hip): 2 x 2 blocks on the diagonal, nothing else stored
    const int32_t *sb_a, *sb_b;
    const double *sb_len, *sb_var;
    const int32_t *dg_row_begin, *dg_slot;
    const double *dg_obs, *dg_var;
    const int64_t *dg_w_offset;        // -1 = diagonal
    const double *dg_w;                // D^-1 of dense groups
    const int32_t *slot_col;           // [n_slots]
};

// tables of the atomics-free point x point gather (assemble.hip, blk_pp_gather_kernel); null = use the atomic kernel
struct PPRecord {          // one (object point, image block) incidence, 32 bytes
    int32_t ipb, mp, lp, pad;   // first image point of the block, points in the block, local index of the point
    int64_t poff;               // offset of the block's Dinv in blk_w (< 0: block-diagonal weights, DevProblem::ip_w3)
    int64_t pad2;
};
#ifndef JAICOV_PP_CW
#define JAICOV_PP_CW 1664
#endif
#ifndef JAICOV_PP_NT
#define JAICOV_PP_NT 256
#endif
constexpr int PP_CW = JAICOV_PP_CW;   // DEFAULT columns of one LDS strip of the point x point gather (3 rows x cw doubles of LDS; PPGather::cw)
constexpr int PP_NT = JAICOV_PP_NT;   // threads per workgroup of the gather
struct PPGather {
    const int32_t *pt_ip_begin = nullptr;   // [n_points+1] CSR over recs
    const PPRecord *recs = nullptr;         // image order within a point
    const int32_t *ipcol = nullptr;         // [3][n_ip] column of X / Y / Z of the point seen by image point ip
    // the engine stores the points of a dense block in column order, so the partners that fall into column chunk c of
    // record o are the block positions range[2*(o*n_chunks+c)] .. range[..+1] (already cut at the row's own column:
    // only the lower triangle is assembled)
    const int32_t *range = nullptr;
    int cmin = 0, n_chunks = 0;
    int cw = PP_CW;            // columns of one LDS strip (runtime: JAICOV_PP_CW; multiple of 4, the LDS strip is 3 * cw doubles)
    const double *ug = nullptr; // SchurBufs::Ug (coalesced U_q loads): REQUIRED when the gather forms P' = sigma2 Dinv - U U' itself (EO pre-elimination
                                // without SchurBufs::materialise); launch_assemble_blocks refuses the launch otherwise
    int xcd_map = 0;           // != 0: 1-D grid, every XCD works on its own chunks only (blk_pp_gather_kernel)
    // plain != 0: the rows cmin..cmax of N are point rows only and nobody else has written their columns >= cmin yet;
    // the strips are then STORED (zeros included) instead of added, which saves zeroing that part of N and reading it back
    int cmax = -1, plain = 0;
    // deterministic variant (engine option): the waves of a workgroup add their images' products to the strip in turn, image order
    int det = 0;
};

// buffers of the per-image EO pre-elimination (schur.hip); Pp == nullptr -> mode off
struct SchurBufs {
    double *U = nullptr, *Linv = nullptr, *G = nullptr, *Pp = nullptr, *diagcorr = nullptr;
    double *Ug = nullptr;     // the same U once more, laid out for the point x point gather: [6][n_ip] pairs (see blk_elim_kernel); blk_elim_kernel
                              // skips it when null, the fused gather needs it (PPGather::ug)
    double *xq = nullptr;     // [6 * images] n_E / diag(N_EE): what the reference's REDUCED last pass leaves in dx (engine option)
    int *info = nullptr;
    double lambda = 0.0;
    bool active = false;      // this pass pre-eliminates the exterior orientations
    int materialise = 0;      // P' = sigma2 Dinv - U U' is written out (Pp) instead of being formed inside the point x point gather
};

// assembly_mode = 1 (densemode.hip): workspace and driver of the densified MFMA contraction of the image groups
struct DenseMode {
    void *Ppad = nullptr, *Apad = nullptr, *Bbuf = nullptr, *Sbuf = nullptr;   // double, or float when fp32
    int32_t *cmap = nullptr;
    int mpad = 0, kpad = 0, batch = 0;
    bool fp32 = false;                       // assembly_mode = 2: fp32 operands and fp32 MFMA accumulation
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipError_t init(int max_m, int max_k1, int n_blocks, bool single);
    void release();
    hipError_t assemble(hipStream_t s, const DevProblem &p, const int32_t *blk_list, int n_list, const double *rowsA,
                        const double *rowsW, double sigma2, double *N, double *n, float *gemm_ms);
};

enum { ASSEMBLY_DEFAULT = 0, ASSEMBLY_T_VECTOR = 1, ASSEMBLY_NO_FORK = 2, ASSEMBLY_MATERIALISE = 3 };
int assembly_form();      // JAICOV_ASSEMBLY_FORM (assemble.hip): test hook for the alternative forms of the dense-block assembly

// ---- slot layout ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_io(const DevProblem &p, int c) { return 3 * p.n_points + 3 * c; }
__device__ __forceinline__ int slot_dist(const DevProblem &p, int j) { return 3 * p.n_points + 3 * p.n_cameras + j; }
__device__ __forceinline__ int slot_eo(const DevProblem &p, int i) {
    return 3 * p.n_points + 3 * p.n_cameras + p.n_dist + 6 * i;
}

__device__ __forceinline__ double ipow(double b, int e) {   // Math.pow(double,int) for small e >= 0
    double r = 1.0;
    for (int i = 0; i < e; i++) r *= b;
    return r;
}

// lower-storage atomic add: N[max(r,c)][min(r,c)]
__device__ __forceinline__ void nadd(double *N, long ld, int r, int c, double v) {
    const int hi = r > c ? r : c, lo = r > c ? c : r;
    unsafeAtomicAdd(N + (long)hi * ld + lo, v);
}

}  // namespace jaicov
