/*
 * jaicov_neq.h -- C ABI of the MI355X normal-equation engine for JAICOV-style bundle adjustment.
 *
 * This is the drop-in boundary for ONE path of applied-geodesy/bundle-adjustment (JAICOV): the per-iteration
 * Gauss-Newton / Levenberg-Marquardt inner loop of BundleAdjustment.estimateModel().  The reference has no FFI for
 * this path (pure Java), so the boundary is defined here; every entry point names the reference code it replaces
 * (paths relative to JAICOV/src/org/applied_geodesy/):
 *
 *   BA  = adjustment/bundle/BundleAdjustment.java
 *   PDF = adjustment/bundle/derivation/PartialDerivativeFactory.java
 *   NES = adjustment/NormalEquationSystem.java
 *   MX  = adjustment/MathExtension.java
 *   DOPG= adjustment/bundle/parameter/DirectlyObservedParameterGroup.java
 *
 * Conventions
 *   - plain C, no torch / C++ types in any signature; all pointers are HOST pointers unless a name says "device".
 *   - every function returns a jaicov_status (0 = ok).  Nothing throws across the boundary (BA:304-315 maps the
 *     reference's exceptions to EstimationStateType codes the same way).
 *   - one engine per BundleAdjustment, externally synchronised (the reference is single-threaded, BA:203 is
 *     single-shot).
 *   - matrices leave the engine in MTJ / LAPACK packed order, UPLO='U', column-major:
 *         index(r,c) = r + c*(c+1)/2 ,  r <= c        (UpperSymmPackMatrix.getData(), MX:341-342)
 *     order U = u + d: rows/columns 0..d-1 are the datum border (BA:493-635), unknowns follow (BA:776-781).
 *   - a parameter that is FIXED in the reference (column == Integer.MAX_VALUE, UnknownParameter.java:27) carries
 *     JAICOV_COL_FIXED here; free parameters carry their reference column (already shifted by d).
 */
#ifndef JAICOV_NEQ_H
#define JAICOV_NEQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JAICOV_NEQ_ABI_VERSION 1
#define JAICOV_COL_FIXED (-1)
#define JAICOV_MAX_DIST_PER_CAMERA 20   /* distortion coefficients per camera the device kernels are built for */

/* Status codes.  <0 : bad argument (reference: IllegalArgumentException, MX:352,363);
 *                >0 : numerical failure (MatrixSingularException / MatrixNotSPDException, MX:350,361);
 *                OOM and device errors are distinct (BA:370-375 OUT_OF_MEMORY).                         */
typedef enum jaicov_status {
    JAICOV_OK                 = 0,
    JAICOV_ERR_BAD_ARGUMENT   = -1,
    JAICOV_ERR_BAD_STATE      = -2,   /* call order violated (e.g. solve before build)                  */
    JAICOV_ERR_UNSUPPORTED    = -3,
    JAICOV_ERR_OUT_OF_MEMORY  = -4,
    JAICOV_ERR_DEVICE         = -5,   /* HIP runtime error; text via jaicov_neq_last_error()            */
    JAICOV_ERR_NO_DEVICE      = -6,   /* no gfx950 device / HIP runtime unusable: the engine never falls back to CPU */
    JAICOV_ERR_SINGULAR       = 1,    /* factorisation hit a non-positive / zero pivot (info > 0)       */
    JAICOV_ERR_NOT_FINITE     = 2     /* NaN/Inf in the step (BA:327-331)                               */
} jaicov_status;

/* Distortion coefficient kinds, in the reference's APPLICATION order (DistortionModel.Type ordinal,
 * camera/distortion/DistortionModel.java:29-37; Camera.java:50 sorts by it).                            */
typedef enum jaicov_dist_kind {
    JAICOV_DIST_AFFINITY_CX   = 0,    /* AffinityShearDistortionModel Cx  (ASF:37-81)                   */
    JAICOV_DIST_AFFINITY_CY   = 1,
    JAICOV_DIST_TANGENTIAL_BX = 2,    /* TangentialDistortionModel Bx,By,Bi (TDF:39-134)                */
    JAICOV_DIST_TANGENTIAL_BY = 3,
    JAICOV_DIST_TANGENTIAL_BI = 4,
    JAICOV_DIST_RADIAL_AI     = 5,    /* RadiallySymmetricDistortionModel Ai (RSF:39-90)                */
    JAICOV_DIST_DISTANCE_DI   = 6,    /* RadialDistanceDistortionModel Di (RDF:39-161)                  */
    /* ZernikeDistortionModel.X / .Y / .Gradient (ZernikeDistortionModelFactory.java:41-227); dist_order = the single
     * index `order` >= 1 of ZernikeCoefficient (n, m from Schwiegerling Eq. 2:100/101, ZernikeCoefficient.java:41-57).
     * Restated literally, including the integer division `pj/2` of the radial exponents (ZDF:107,176,178): for odd
     * radial orders the reference's value and its chain-rule factors are not derivatives of one another.          */
    JAICOV_DIST_ZERNIKE_X     = 7,
    JAICOV_DIST_ZERNIKE_Y     = 8,
    JAICOV_DIST_ZERNIKE_Z     = 9
} jaicov_dist_kind;

/* Datum flags = which inner-constraint rows exist (RankDefect.java:119-130, BA:523-530 order).          */
enum {
    JAICOV_DATUM_TX = 1, JAICOV_DATUM_TY = 2, JAICOV_DATUM_TZ = 4,
    JAICOV_DATUM_RX = 8, JAICOV_DATUM_RY = 16, JAICOV_DATUM_RZ = 32, JAICOV_DATUM_SCALE = 64
};

/*
 * Immutable structure of one adjustment, flattened from the Java object graph after
 * BA.prepareUnknownParameters() (BA:667-782) has numbered rows and columns.
 *
 * Parameter VALUES travel separately in one "slot" vector of length jaicov_neq_num_slots():
 *     [ 3*n_points (X,Y,Z per point) | 3*n_cameras (x0,y0,c) | n_dist | 6*n_images (X0,Y0,Z0,omega,phi,kappa) ]
 * Every parameter, free or fixed, owns a slot; free ones additionally own a column.
 */
typedef struct jaicov_problem_desc {
    uint32_t struct_size;          /* sizeof(jaicov_problem_desc), checked                               */
    int32_t  n_unknowns;           /* U = u + d, order of N (BA:791)                                     */
    int32_t  rank_defect;          /* d (BA:776)                                                         */
    int32_t  datum_flags;          /* JAICOV_DATUM_* of the FREE defects; popcount == rank_defect        */

    int32_t  n_points, n_cameras, n_images, n_dist;
    int32_t  n_image_points, n_image_blocks, n_scale_bars, n_direct_groups, n_direct_rows;

    /* object points (ObjectCoordinate.java) */
    const int32_t *point_col;      /* [3*n_points] column of X,Y,Z or JAICOV_COL_FIXED                   */
    const uint8_t *point_datum;    /* [n_points]  ObjectCoordinate.isDatum()                             */

    /* cameras (Camera.java, InteriorOrientation.java:60-82 order x0,y0,c) */
    const int32_t *io_col;         /* [3*n_cameras]                                                      */
    const double  *cam_r0;         /* [n_cameras] r0 of the radial / distance models (Camera.java:45)    */
    const int32_t *cam_dist_begin; /* [n_cameras+1] range of this camera's coefficients in dist_*        */
    const int32_t *dist_kind;      /* [n_dist] jaicov_dist_kind, per camera in application order         */
    const int32_t *dist_order;     /* [n_dist] polynomial order i of Ai/Bi/Di (PolynomialCoefficient.java) */
    const int32_t *dist_col;       /* [n_dist]                                                           */

    /* images (Image.java, ExteriorOrientation.java:37-46 order X0,Y0,Z0,omega,phi,kappa) */
    const int32_t *image_camera;   /* [n_images]                                                         */
    const int32_t *eo_col;         /* [6*n_images]                                                       */

    /* image points = ImageCoordinate observation groups, image-major (BA:670-693): rows 2k, 2k+1         */
    const int32_t *ip_image;       /* [n_image_points]                                                   */
    const int32_t *ip_point;       /* [n_image_points]                                                   */
    const double  *ip_x, *ip_y;    /* observed xp, yp                                                    */
    const double  *ip_var_x, *ip_var_y;   /* variances (ImageCoordinate.java:48-49)                      */
    const double  *ip_rho;         /* correlation coefficient in (-1,1) (ImageCoordinate.java:40)        */

    /* image blocks: contiguous ranges of image points of ONE image observed with a joint, fully
     * populated dispersion (the largest W the reference's a10 contract can express).
     * Image points outside every block are ordinary 2x2 groups.                                         */
    const int32_t *blk_ip_begin;   /* [n_image_blocks+1] ranges must be disjoint and ascending           */
    const int64_t *blk_disp_offset;/* [n_image_blocks] offset (in doubles) into blk_disp                 */
    const double  *blk_disp;       /* row-major symmetric (2m x 2m) dispersion, rows x0,y0,x1,y1,...     */

    /* scale bars (ScaleBar.java, PDF:210-283) */
    const int32_t *sb_point_a, *sb_point_b;   /* [n_scale_bars]                                          */
    const double  *sb_length, *sb_var;

    /* directly observed parameter groups (DOPG, PDF:447-473) */
    const int32_t *dg_row_begin;   /* [n_direct_groups+1]                                                */
    const int32_t *dg_slot;        /* [n_direct_rows] slot of the referenced unknown parameter           */
    const double  *dg_obs;         /* [n_direct_rows] observed value                                     */
    const double  *dg_var;         /* [n_direct_rows] variance (diag of D when a dispersion is given)    */
    const int64_t *dg_disp_offset; /* [n_direct_groups] offset into dg_disp, or -1 = diagonal (DOPG:71-78) */
    const double  *dg_disp;        /* row-major symmetric (m x m) dispersion matrices (DOPG:49-61)       */
} jaicov_problem_desc;

typedef struct jaicov_engine_options {
    uint32_t struct_size;
    int32_t  device;               /* HIP device ordinal                                                 */
    /* observation sharding: this engine accumulates only images [image_begin,image_end);
     * scale bars, directly observed groups, datum, damping and V are applied by the rank with
     * apply_shared != 0 ... after the reduction they are identical everywhere.  -1/-1 = all images.     */
    int32_t  image_begin, image_end;
    int32_t  apply_shared;
    int32_t  assembly_mode;        /* 0 = structure-aware (default); 1 = J'WJ of the jointly dispersed image groups as a
                                      dense contraction A'(PA) on the fp64 matrix cores (PDF:486-498 literally; ~50x the
                                      arithmetic, for the MFMA-utilisation figure and as a second path for parity tests);
                                      2 = mode 1 with operands rounded to fp32 and fp32 MFMA accumulation (BASELINE config 5's
                                      precision sweep; NOT a product mode, results differ at the 1e-7 level by design)       */
    int32_t  block_size;           /* factorisation block NB; 0 = default                                */
    int32_t  reduced_reference_quirk; /* != 0: a solve with JAICOV_INVERT_REDUCED returns in the exterior-orientation entries of dx
                                      what the reference's last pass leaves there in MatrixInversion.REDUCED -- the UNSOLVED
                                      right-hand side scaled twice by the preconditioner, V_c^2 n_c (BA:261-267 solve only the
                                      leading numRows, BA:273 scales all of dx; BA:430, 450-461 then use these entries for Omega
                                      and the update) -- instead of the back-substituted step.  Default 0.      */
    int32_t  deterministic;        /* 0 = default = ON (since round 4), < 0 = off.  ON: the assembly of the jointly dispersed image groups sums
                                      in a fixed order (image order), so two runs give the same bits in N, n, dx and Qxx -- the reference is
                                      bit-reproducible, and cond(N) ~ 1e9 turns run-to-run differences of 1e-16 in N into 1e-9 in Qxx on the
                                      smallest test scenes.  The waves of a workgroup of the point x point gather form their images' products
                                      concurrently and pass a turn word in LDS for the adds (assemble.hip); the small reductions use fixed-order
                                      second kernels.  Costs 0.3 ms per pass at config 4 (assembly 2.74 -> 3.0 ms; round 3's form with workgroup
                                      barriers: 0.7, round 2's: 1.7).  OFF: LDS / memory fp64 atomics in arrival order.                        */
    int32_t  refinement;           /* iterative refinement of the step in jaicov_neq_solve: 0 = default (ONE step), < 0 = none, k > 0 = k steps
                                      (at most 4).  A step computes the residual n - N dx (and the datum border's) of the unscaled system
                                      in two-fold precision on the device and solves for the correction with the factor at hand (one forward
                                      and one backward substitution, csrc/refine.hip): the error of dx against the exact solution of the
                                      assembled system falls from cond * eps (2.6e-8 at config 4, where the reference's dspsv, MX:338-353, is
                                      at 3.6e-9) to ~1e-12.  Costs ~1 ms per pass at config 4.                  */
    int32_t  ordinary_group_elimination; /* 0 = default: when the whole problem qualifies (every image has >= 3 observations, at most 2048 per
                                      image without a dispersion of its own, the exterior orientations own the trailing columns and are
                                      not directly observed) the exterior orientations of images whose points are ORDINARY ImageCoordinate
                                      groups (diagonal / 2 x 2 weights) are pre-eliminated on the device like those of jointly dispersed
                                      images -- reduceNormalEquationSystem, BA:1197-1342, serves every image --: such an image is held as an
                                      image block whose block-diagonal inv(D) is kept as 2 x 2 blocks (24 bytes per observation) -- provided the
                                      6 I exterior-orientation columns are at least two 128-column blocks of the factorisation (below that the
                                      block kernels' launches cost more than the smaller system saves: BASELINE config 2).  > 0: eliminate at any
                                      size.  < 0: off, ordinary groups are assembled one by one into the full-order system (rounds 1-3).    */
    int32_t  dispersion_refinement; /* 0 = default: every inverse dispersion (DOPG:82-86) gets one Newton-Schulz step X <- X + X (I - D X) at
                                      create, with the residual formed by error-free splitting on the fp64 matrix cores (batchinv.hip): the
                                      forward error of inv(D) falls from cond(D) * eps (3e-11 at config 4, the reference's dpptrf + dpptri the
                                      same) to ~1e-14, and with it the error of N = A' inv(D) A against the exactly assembled system.  Four
                                      more GEMMs per matrix (+80 ms of engine creation at config 4).  < 0: off.                              */
    int32_t  expansion_exchange;   /* != 0 on a SHARDED engine (image_begin/image_end): the caller promises to sum
                                      jaicov_neq_expansion_buffer() over the ranks between accumulate and the inverting solve, so that
                                      MatrixInversion.FULL is expanded from the reduced inverse on a shard too (JAICOV_INVERT_FULL_EXPANDED;
                                      without the promise a shard falls back to the literal order-U route).  Ignored on an unsharded engine. */
    int32_t  inverse_refinement;   /* 0 = default: an inverting solve of order <= 8192 (padded) adds one Newton-Schulz step Q <- Q + Q (I - M Q) to the
                                      inverse of the scaled system, with the residual formed exactly by error-free splitting on the fp64 matrix
                                      cores (batchinv.hip): the inverse from a Cholesky factor at cond ~ 4e8 (BASELINE config 3) is 2-3e-9 from the
                                      exact inverse where the reference's dspsv + dsptri reaches 5e-10; with the step it is at the rounding of its
                                      entries.  Four GEMMs of the order (+4 ms at config 3, once per adjustment).  Larger orders are left alone: at
                                      config 4 the reference itself is 2e-7 from the truth, the engine 2e-8.  < 0: off.    */
    int32_t  reserved[1];
} jaicov_engine_options;

typedef struct jaicov_engine jaicov_engine;

/* --- lifetime ---------------------------------------------------------------------------------------- */

/* One-time upload of the immutable structure; inverts the dense dispersions to weights on the device
 * (DOPG:82-86 does dpptrf+dpptri once and caches).  opts may be NULL.                                    */
int  jaicov_neq_create(const jaicov_problem_desc *desc, const jaicov_engine_options *opts, jaicov_engine **out);
void jaicov_neq_destroy(jaicov_engine *e);                      /* idempotent on NULL                    */
const char *jaicov_neq_last_error(const jaicov_engine *e);      /* text of the last failure, never NULL  */
int  jaicov_neq_abi_version(void);
size_t jaicov_neq_num_slots(const jaicov_engine *e);
size_t jaicov_neq_packed_length(const jaicov_engine *e);        /* U*(U+1)/2                             */

/* --- parameter values (Parameter.value of every UnknownParameter) ------------------------------------ */
int jaicov_neq_set_parameters(jaicov_engine *e, const double *slots, size_t n_slots);
int jaicov_neq_get_parameters(jaicov_engine *e, double *slots, size_t n_slots);

/* --- one pass of the loop body (BA:228-355) ---------------------------------------------------------- */

/* replaces BA.createNormalEquation() (BA:789-834): residual + Jacobian rows (PDF:94-190,285-445 and the
 * distortion factories), N += A'PA, n += A'Pw (PDF:475-505), datum rows (BA:493-635), LM damping
 * N[c,c] *= (1+lambda) (BA:814-822), preconditioner V (BA:825-828), SIMULATION zeroes n (BA:830-831).
 * = accumulate + (host all-reduce for multi-GPU) + finalize.                                             */
int jaicov_neq_build(jaicov_engine *e, double sigma2apriori, double lambda, int simulation);
int jaicov_neq_accumulate(jaicov_engine *e, double sigma2apriori);
/* accumulate with the LM damping value known up front (needed when the engine pre-eliminates the exterior-orientation
 * blocks per image while assembling, the device-side form of MatrixInversion.PRE_ELIMINATION, BA:283-291,1197-1453). */
int jaicov_neq_accumulate2(jaicov_engine *e, double sigma2apriori, double lambda);
/* MatrixInversion (BA:65-70) as the `invert` argument of prepare_inverse / solve / estimate:
 * NONE: no cofactor matrix.  FULL: Qxx = K^-1 of all u + d unknowns (BA:268-271).  REDUCED: the block of Qxx that belongs
 * to the datum border, the object points, the interior orientation and the distortion parameters, i.e. the inverse of
 * the system from which the exterior orientations were eliminated -- what MatrixInversion.REDUCED and PRE_ELIMINATION
 * keep in the final pass (BA:261-267: reduceNormalEquationSystem + solve(N, n, numRows, true)); when the engine cannot
 * pre-eliminate (jaicov_neq_reduced_order() == U) REDUCED is served by the full inverse.                           */
#define JAICOV_INVERT_NONE    0
#define JAICOV_INVERT_FULL    1
#define JAICOV_INVERT_REDUCED 2
/* FULL_EXPANDED: the same matrix as FULL (order U, all unknowns), computed from the EO-reduced system instead of from a
 * factorisation of the unreduced one: Q_RR = S^-1 (the REDUCED inverse), Q_ER = -F Q_RR, Q_EE = N_EE^-1 - Q_ER F' with
 * F = N_EE^-1 N_ER, N_EE block diagonal 6 x 6 per image (the block formulas behind BA:1197-1453).  The build keeps the EO
 * pre-elimination (jaicov_neq_get_normal returns the reduced system).  At config 4 it is both faster (no order-18 014 factorisation)
 * and ~50x more accurate than FULL (profiles/r03_cfg4_accuracy.json: the unreduced Cholesky loses 2e-7 where the reduced one
 * loses 4e-9).  jaicov_neq_estimate and the host mirror's estimateModel() use it for MatrixInversion.FULL; where the engine cannot
 * pre-eliminate, or holds only a shard of the images, it is served as FULL.  JAICOV_FULL_LITERAL=1 forces FULL everywhere.        */
#define JAICOV_INVERT_FULL_EXPANDED 3
/* Announces the `invert` value of the solve after the NEXT build (the final pass, BA:252-280): FULL makes that build
 * assemble the full system instead of the EO-reduced one.  estimateModel knows this before it builds (BA:250
 * estimateCompleteModel = isEstimated).                                                                            */
int jaicov_neq_prepare_inverse(jaicov_engine *e, int inverse_follows);
/* Order of the cofactor matrix the last inverting solve left on the device: U (FULL) or jaicov_neq_reduced_order()
 * (REDUCED on a pre-eliminated system); -1 without one.                                                            */
int jaicov_neq_cofactor_order(const jaicov_engine *e);
/* Order of the system the last accumulate assembled: U, or the first EO column when the EO blocks were pre-eliminated
 * (then every rank's solve returns dx with only ITS images' EO entries filled; a multi-GPU host sums dx[order..U)).   */
int jaicov_neq_reduced_order(const jaicov_engine *e);
int jaicov_neq_finalize(jaicov_engine *e, double sigma2apriori, double lambda, int simulation);

/* Device buffer holding this rank's partial normal equations between accumulate and finalize: one contiguous
 * array of *count doubles: N packed 'U' (k(k+1)/2, k = jaicov_neq_reduced_order()) followed by n (k) and, when the
 * exterior orientations were pre-eliminated, by the k diagonal corrections the LM damping needs.  A multi-GPU host sums it over ranks in place
 * (ncclAllReduce, sum, double -- SURVEY 8(e)); finalize() picks the summed values up again.  The call synchronises
 * the engine stream so that the collective may run on any other stream.                                          */
int jaicov_neq_reduce_buffer(jaicov_engine *e, void **device_ptr, size_t *count);
/* The same without the host-side wait: *stream receives the engine's hipStream_t; the buffer is complete in stream order.
 * A host that enqueues its collective in that order (ncclAllReduce on this stream, or on a stream that waits for an event
 * recorded here and is waited for before finalize) keeps the whole pass asynchronous, so the launch work of the
 * factorisation overlaps the assembly and the collective instead of following them.                                */
int jaicov_neq_reduce_buffer_async(jaicov_engine *e, void **device_ptr, size_t *count, void **stream);

/* replaces NES.applyPrecondition (NES:82-91) + MX.solve(N,n,numRows,invert) (MX:338-366) + the reverse
 * preconditioning (BA:273,297).  dx_out[U] (border entries = Lagrange multipliers, as dspsv leaves them).
 * invert = JAICOV_INVERT_*: != NONE keeps Qxx on the device (BA:274).                                     */
int jaicov_neq_solve(jaicov_engine *e, int invert, double *dx_out);

/* After a solve on a system whose exterior orientations were pre-eliminated (jaicov_neq_reduced_order() < U): the DEVICE array of
 * the EO steps this engine back-substituted, 6 doubles per image in image order (*count = 6 * images), zero for the images of
 * other ranks.  dx_out[reduced_order + i] of the solve holds the same numbers.  A multi-GPU host sums the array over the ranks in
 * place (ncclAllReduce) and copies it into the tail of dx, instead of sending the host copy back to the device for the collective. */
int jaicov_neq_eo_step_buffer(jaicov_engine *e, void **device_ptr, size_t *count);
/* Sharded final pass with JAICOV_INVERT_FULL_EXPANDED (BA:268-271 on several GPUs): after jaicov_neq_accumulate of a pass announced
 * with jaicov_neq_prepare_inverse(e, JAICOV_INVERT_FULL_EXPANDED), the DEVICE array [F | L_E^-1] -- F = N_EE^-1 N_ER, one 6-row band
 * per image, dense [6 images padded to 128][reduced order padded to 128], then 36 doubles per image -- with this engine's images'
 * entries set and zeros elsewhere.  The caller sums it over the ranks in place (one all-reduce, 371 MB at config 4) before
 * jaicov_neq_solve(e, JAICOV_INVERT_FULL_EXPANDED, ...).  Needs engine option expansion_exchange on a shard.                     */
int jaicov_neq_expansion_buffer(jaicov_engine *e, void **device_ptr, size_t *count);

/* replaces BA.getOmega(dx) (BA:472-491): sum over groups of (w - A dx)' P (w - A dx) at the CURRENT
 * (pre-update) parameters.                                                                               */
int jaicov_neq_omega(jaicov_engine *e, double sigma2apriori, const double *dx, double *omega);

/* replaces BA.updateUnknownParameters(dx) (BA:450-462): x[col] += dx[col] on the device copy of the slots;
 * returns max |dx| over unknown columns.                                                                 */
int jaicov_neq_update(jaicov_engine *e, const double *dx, double *max_abs_dx);

/* --- results ----------------------------------------------------------------------------------------- */
/* N (after finalize, before preconditioning) and n in packed 'U' order, for UpperSymmPackMatrix.getData().
 * len = U(U+1)/2.  When the last accumulate pre-eliminated the exterior orientations (jaicov_neq_reduced_order() < U)
 * the arrays hold the REDUCED system in their leading reduced_order() rows/columns (packed 'U': the leading
 * k(k+1)/2 entries) and zeros elsewhere -- what reduceNormalEquationSystem (BA:1197-1340) leaves in N11, n1; call
 * jaicov_neq_prepare_inverse(e, JAICOV_INVERT_FULL) before the build to obtain the unreduced N.                    */
int jaicov_neq_get_normal(jaicov_engine *e, double *N_packed, size_t len, double *n, size_t U);
/* Qxx (BA:1177 getCofactorMatrix), packed 'U', order jaicov_neq_cofactor_order().  Requires an inverting solve. */
int jaicov_neq_get_cofactor(jaicov_engine *e, double *Q_packed, size_t len);
/* sub-matrix gather Q[idx[i], idx[j]] into a dense row-major k x k buffer (what MatlabResultWriter.java:210-221
 * and DefaultResultWriter.java:126-155 read element-wise).                                               */
int jaicov_neq_get_cofactor_sub(jaicov_engine *e, const int32_t *idx, int32_t k, double *out);
/* the same scaled on the device by the a-posteriori variance factor: sigma2 * Qxx[idx, idx], the dispersion block the
 * writers print (DefaultResultWriter.java:126-155 `sigma2apost * cofactor.get(row, column)`) -- SURVEY 8(f) f2.      */
int jaicov_neq_get_dispersion_sub(jaicov_engine *e, double sigma2_aposteriori, const int32_t *idx, int32_t k, double *out);
/* compact residual/Jacobian rows of image point ip (debug / parity): w[2], A[2*(12+JAICOV_MAX_DIST_PER_CAMERA)]
 * in local order X,Y,Z,x0,y0,c,X0,Y0,Z0,omega,phi,kappa,dist...                                          */
int jaicov_neq_get_rows(jaicov_engine *e, int32_t ip_begin, int32_t ip_count, double *w, double *A);

/* --- whole loop (BA.estimateModel, BA:203-387) ------------------------------------------------------- */
typedef struct jaicov_estimate_options {
    uint32_t struct_size;
    int32_t  max_iterations;       /* DefaultValue.java:25 = 5000                                        */
    int32_t  invert;               /* JAICOV_INVERT_* = MatrixInversion (BA:65-70)                       */
    int32_t  simulation;           /* EstimationType.SIMULATION (BA:830)                                 */
    double   lambda0;              /* setLevenbergMarquardtDampingValue (BA:1189)                        */
    double   sigma2apriori;        /* BA:98,641                                                          */
} jaicov_estimate_options;

typedef struct jaicov_estimate_result {
    int32_t  state;                /* EstimationStateType id (adjustment/EstimationStateType.java:25-42):
                                      1 ERROR_FREE_ESTIMATION, -1 INTERRUPT, -2 SINGULAR_MATRIX, -4 NO_CONVERGENCE,
                                      -7 OUT_OF_MEMORY                                                    */
    int32_t  iterations;
    double   omega;
    double   max_abs_dx;
    double   final_lambda;
    double   seconds_total, seconds_last_pass;
} jaicov_estimate_result;

int jaicov_neq_estimate(jaicov_engine *e, const jaicov_estimate_options *opts, jaicov_estimate_result *res);
/* BundleAdjustment.interrupt() (BA:1455-1457): sets the cooperative cancel flag; jaicov_neq_estimate polls it where the
 * reference polls `interrupt` (after the build, BA:240, and after the update, BA:320), ends with state -1 (INTERRUPT) and
 * clears it.  The only entry point that may be called from another thread while a call on the same engine is running.     */
int jaicov_neq_cancel(jaicov_engine *e);

/* timing of the stages of the last pass in milliseconds (HIP events on the engine stream):
 * [0] rows  [1] assembly  [2] finalize  [3] factorisation  [4] solve  [5] inverse  [6] omega  [7] total  */
int jaicov_neq_last_timings(jaicov_engine *e, double *ms, int32_t n);

/* Per-kernel profiling of the dominant kernel: the dataflow Cholesky's tile kernel (chol_tile_kernel, ONE persistent launch per
 * factorisation from 12 block columns on; for smaller orders the trailing-update GEMM launches of the stream-scheduled form).  When
 * enabled, every such launch is bracketed by a pair of HIP events on the stream it runs on; jaicov_neq_kernel_stats reads the sums.   */
int jaicov_neq_set_profiling(jaicov_engine *e, int enable);
/* What jaicov_neq_create spent (ms, wall clock of the host): [0] the whole call, [1] host time inside the uploads of the dense
 * dispersions (pageable host memory -> device), [2] dispersions -> weights altogether (upload + batched inversion, DOPG:82-86),
 * [3] validation, tables and structure upload, [4] work buffers + the full-order solver, [5] EO pre-elimination buffers + the
 * reduced solver.                                                                                                              */
int jaicov_neq_create_timings(jaicov_engine *e, double *ms, int32_t n);
/* Parity hook for DOPG:82-86 / MX:304-324: inv(D) of image block `block` as the engine caches it (the reference caches
 * sigma0^2 times it), row-major m x m with m = 2 * (points of the block), rows and columns in the caller's observation order;
 * len must be m * m.                                                                                                            */
int jaicov_neq_get_block_weight(jaicov_engine *e, int32_t block, double *out, size_t len);
/* stats: [0] launches of the dominant kernel, [1] their summed device ms, [2] their summed algorithmic flops (order^3 / 3 per dataflow
 * factorisation; rows (rows + 1) K per lower-triangular update of the stream-scheduled form), since the last reset (needs
 * jaicov_neq_set_profiling).  With n >= 6 and assembly_mode = 1: [3] passes, [4] summed device ms of the two J'WJ GEMM launches per batch of
 * images, [5] summed algorithmic flops of SURVEY 8(d)'s dense-group row, sum_g 2 m^2 (k+1) + m (k+1)(k+2).
 * With n >= 10, health counters of the dataflow factorisation since jaicov_neq_create (not reset): [6] factorisations that were
 * abandoned on the device (a bounded wait ran out) and repeated, [7] flags that only the slow-path poll found, [8] of those the
 * ones the plain poll still missed, [9] the ones found after more than 1 ms of waiting.  A healthy run has [6] == 0 ([7]..[9] are informational:
 * the slow-path poll also finds flags that were simply set late); bench.py prints them and flags a line whose run repeated a factorisation.
 * With n >= 11: [10] the relative size of the last refinement correction, max |correction| / max |dx| (= the error the unrefined step had).
 * With n >= 12: [11] the refinement steps per solve the engine runs (option `refinement` after clamping).  With n >= 13: [12] the strip width
 * (columns) of the point x point gather, chosen at create from how an image's points spread over the columns.                          */
int jaicov_neq_kernel_stats(jaicov_engine *e, double *stats, int32_t n, int reset);

/* THE ENVIRONMENT IS NOT AN INTERFACE.  The library reads a handful of variables; none of them is a supported way to configure a run:
 *   test hooks that select a SECOND numerical path so that the test-suite can hold it to the same parity as the default
 *   (tests/test_gpu_parity.py names each): JAICOV_FACTOR_FORM (form of the factorisation: streams, two_step, one_kernel, chain2, chain3),
 *   JAICOV_ASSEMBLY_FORM (t_vector, materialise), JAICOV_FLOW_MIN_BLOCKS (smallest order that takes the dataflow factorisation),
 *   JAICOV_CHAIN8_MIN_NB (smallest order that takes the polling-wave substitution chains), JAICOV_FLOW_SPLIT ("m:from": split update
 *   ranges of the dataflow factorisation), JAICOV_FLOW_TIMEOUT_MS (time limit of a wait: the tests of the repeat path set 0);
 *   diagnostics: JAICOV_VERBOSE, JAICOV_FLOW_TRACE_ON, JAICOV_CHAIN_TRACE, JAICOV_TRACE_TAG, JAICOV_TRACE_WARM.
 * Results are the same (to rounding) under every one of them; what a host may configure is jaicov_engine_options.                        */

#ifdef __cplusplus
}
#endif
#endif /* JAICOV_NEQ_H */
