/*
 * JNI shim between NativeNormalEquationEngine (Java) and the C ABI of include/jaicov_neq.h.
 * The build image has no JDK, so this file is not built by __graft_entry__.build(); on a box with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include jaicov_jni.c \
 *       -L../../bundle-adjustment_amd/csrc -ljaicov_neq -o libjaicov_jni.so
 * tests/test_jni_binding.py keeps it honest without one: every `native` method of the Java class has exactly one
 * Java_... function here (same arity), every jaicov_neq_* called here is declared in the header, and the file compiles
 * against a minimal jni.h (tests/jni_stub, test infrastructure only).
 *
 * Array policy.  create() COPIES the structure arrays (Get<Type>ArrayElements / Release with JNI_ABORT, one array at a
 * time, nothing held across jaicov_neq_create's device work except those copies).  The per-call double[] arguments are
 * pinned with GetPrimitiveArrayCritical for the duration of ONE jaicov_neq_* call; those calls make no JNI calls and the
 * engine is externally synchronised (one engine per BundleAdjustment), which is what the critical-region rules ask for.
 * Every pinned pointer is released with the pointer it was pinned with.
 */
#include <jni.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "jaicov_neq.h"

#define ENG(h) ((jaicov_engine *)(intptr_t)(h))
#define NAT(name) Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_##name

static void throw_status(JNIEnv *e, int status, const char *msg) {
    const char *cls = status > 0 ? "no/uib/cipr/matrix/MatrixSingularException"          /* MX:350,361 */
                     : status == JAICOV_ERR_OUT_OF_MEMORY ? "java/lang/OutOfMemoryError"  /* BA:370-375 */
                     : status == JAICOV_ERR_BAD_ARGUMENT ? "java/lang/IllegalArgumentException" /* MX:352,363 */
                                                         : "java/lang/IllegalStateException";
    jclass c = (*e)->FindClass(e, cls);
    if (c) (*e)->ThrowNew(e, c, msg ? msg : "jaicov engine error");
}

/* one structure array of the problem description: a copy (or pin) obtained with the typed Get...Elements call */
typedef struct { jarray arr; void *ptr; char kind; jsize len; } held_t;

static int hold(JNIEnv *e, jobject obj, jclass c, const char *name, char kind, held_t *h) {
    char sig[3] = {'[', kind, 0};
    jfieldID f = (*e)->GetFieldID(e, c, name, sig);
    h->arr = NULL; h->ptr = NULL; h->kind = kind; h->len = 0;
    if (!f) return -1;                                   /* NoSuchFieldError is pending */
    h->arr = (jarray)(*e)->GetObjectField(e, obj, f);
    if (!h->arr) return 0;                               /* optional array absent */
    h->len = (*e)->GetArrayLength(e, h->arr);
    switch (kind) {
        case 'I': h->ptr = (*e)->GetIntArrayElements(e, (jintArray)h->arr, NULL); break;
        case 'J': h->ptr = (*e)->GetLongArrayElements(e, (jlongArray)h->arr, NULL); break;
        case 'B': h->ptr = (*e)->GetByteArrayElements(e, (jbyteArray)h->arr, NULL); break;
        default:  h->ptr = (*e)->GetDoubleArrayElements(e, (jdoubleArray)h->arr, NULL); break;
    }
    return h->ptr ? 0 : -1;                              /* OutOfMemoryError is pending */
}

static void unhold(JNIEnv *e, held_t *h) {
    if (!h->arr || !h->ptr) return;
    switch (h->kind) {
        case 'I': (*e)->ReleaseIntArrayElements(e, (jintArray)h->arr, (jint *)h->ptr, JNI_ABORT); break;
        case 'J': (*e)->ReleaseLongArrayElements(e, (jlongArray)h->arr, (jlong *)h->ptr, JNI_ABORT); break;
        case 'B': (*e)->ReleaseByteArrayElements(e, (jbyteArray)h->arr, (jbyte *)h->ptr, JNI_ABORT); break;
        default:  (*e)->ReleaseDoubleArrayElements(e, (jdoubleArray)h->arr, (jdouble *)h->ptr, JNI_ABORT); break;
    }
    h->ptr = NULL;
}

/* EngineOptions (Java) -> jaicov_engine_options, field for field; returns -1 with a JNI exception pending when a field is missing */
static int read_options(JNIEnv *e, jobject opt, jaicov_engine_options *o) {
    memset(o, 0, sizeof(*o));
    o->struct_size = sizeof(*o);
    o->image_begin = o->image_end = -1;
    o->apply_shared = 1;
    if (!opt) return 0;                                  /* null = all defaults, all images, device 0 */
    jclass oc = (*e)->GetObjectClass(e, opt);
    static const struct { const char *name; size_t off; } I[] = {
        {"device", offsetof(jaicov_engine_options, device)},
        {"imageBegin", offsetof(jaicov_engine_options, image_begin)},
        {"imageEnd", offsetof(jaicov_engine_options, image_end)},
        {"assemblyMode", offsetof(jaicov_engine_options, assembly_mode)},
        {"blockSize", offsetof(jaicov_engine_options, block_size)},
        {"reducedReferenceQuirk", offsetof(jaicov_engine_options, reduced_reference_quirk)},
        {"deterministic", offsetof(jaicov_engine_options, deterministic)},
        {"refinement", offsetof(jaicov_engine_options, refinement)},
        {"ordinaryGroupElimination", offsetof(jaicov_engine_options, ordinary_group_elimination)},
        {"dispersionRefinement", offsetof(jaicov_engine_options, dispersion_refinement)},
        {"expansionExchange", offsetof(jaicov_engine_options, expansion_exchange)},
        {"inverseRefinement", offsetof(jaicov_engine_options, inverse_refinement)}};
    for (size_t q = 0; q < sizeof(I) / sizeof(I[0]); q++) {
        jfieldID f = (*e)->GetFieldID(e, oc, I[q].name, "I");
        if (!f) return -1;
        *(int32_t *)((char *)o + I[q].off) = (int32_t)(*e)->GetIntField(e, opt, f);
    }
    jfieldID fs = (*e)->GetFieldID(e, oc, "applyShared", "Z");
    if (!fs) return -1;
    o->apply_shared = (*e)->GetBooleanField(e, opt, fs) ? 1 : 0;
    return 0;
}

JNIEXPORT jlong JNICALL NAT(create)(JNIEnv *e, jclass k, jobject d, jobject options) {
    (void)k;
    jaicov_engine_options o;
    if (read_options(e, options, &o)) return 0;
    jclass c = (*e)->GetObjectClass(e, d);
    jaicov_problem_desc p;
    memset(&p, 0, sizeof(p));
    p.struct_size = sizeof(p);
    jfieldID fu = (*e)->GetFieldID(e, c, "numberOfUnknowns", "I"), fr = (*e)->GetFieldID(e, c, "rankDefect", "I"),
             fd = (*e)->GetFieldID(e, c, "datumFlags", "I");
    if (!fu || !fr || !fd) return 0;
    p.n_unknowns = (*e)->GetIntField(e, d, fu);
    p.rank_defect = (*e)->GetIntField(e, d, fr);
    p.datum_flags = (*e)->GetIntField(e, d, fd);
    enum { POINT_COL, POINT_DATUM, IO_COL, CAM_R0, CAM_DIST_BEGIN, DIST_KIND, DIST_ORDER, DIST_COL, IMAGE_CAMERA, EO_COL, IP_IMAGE,
           IP_POINT, IP_X, IP_Y, IP_VX, IP_VY, IP_RHO, BLK_BEGIN, BLK_OFF, BLK_DISP, SB_A, SB_B, SB_LEN, SB_VAR, DG_BEGIN, DG_SLOT,
           DG_OBS, DG_VAR, DG_OFF, DG_DISP, N_HELD };
    static const struct { const char *name; char kind; } F[N_HELD] = {
        {"pointColumn", 'I'}, {"pointDatum", 'B'}, {"interiorColumn", 'I'}, {"cameraR0", 'D'}, {"cameraDistortionBegin", 'I'},
        {"distortionKind", 'I'}, {"distortionOrder", 'I'}, {"distortionColumn", 'I'}, {"imageCamera", 'I'}, {"exteriorColumn", 'I'},
        {"imagePointImage", 'I'}, {"imagePointPoint", 'I'}, {"x", 'D'}, {"y", 'D'}, {"varianceX", 'D'}, {"varianceY", 'D'}, {"rho", 'D'},
        {"blockBegin", 'I'}, {"blockDispersionOffset", 'J'}, {"blockDispersion", 'D'}, {"scaleBarA", 'I'}, {"scaleBarB", 'I'},
        {"scaleBarLength", 'D'}, {"scaleBarVariance", 'D'}, {"directRowBegin", 'I'}, {"directSlot", 'I'}, {"directObservation", 'D'},
        {"directVariance", 'D'}, {"directDispersionOffset", 'J'}, {"directDispersion", 'D'}};
    held_t a[N_HELD];
    int n = 0, bad = 0;
    for (; n < N_HELD && !bad; n++) bad = hold(e, d, c, F[n].name, F[n].kind, &a[n]);
    jaicov_engine *eng = NULL;
    int rc = JAICOV_ERR_BAD_ARGUMENT;
    char msg[512];
    msg[0] = 0;
    if (!bad) {
        p.point_col = (const int32_t *)a[POINT_COL].ptr;          p.point_datum = (const uint8_t *)a[POINT_DATUM].ptr;
        p.io_col = (const int32_t *)a[IO_COL].ptr;                p.cam_r0 = (const double *)a[CAM_R0].ptr;
        p.cam_dist_begin = (const int32_t *)a[CAM_DIST_BEGIN].ptr; p.dist_kind = (const int32_t *)a[DIST_KIND].ptr;
        p.dist_order = (const int32_t *)a[DIST_ORDER].ptr;        p.dist_col = (const int32_t *)a[DIST_COL].ptr;
        p.image_camera = (const int32_t *)a[IMAGE_CAMERA].ptr;    p.eo_col = (const int32_t *)a[EO_COL].ptr;
        p.ip_image = (const int32_t *)a[IP_IMAGE].ptr;            p.ip_point = (const int32_t *)a[IP_POINT].ptr;
        p.ip_x = (const double *)a[IP_X].ptr;                     p.ip_y = (const double *)a[IP_Y].ptr;
        p.ip_var_x = (const double *)a[IP_VX].ptr;                p.ip_var_y = (const double *)a[IP_VY].ptr;
        p.ip_rho = (const double *)a[IP_RHO].ptr;
        p.blk_ip_begin = (const int32_t *)a[BLK_BEGIN].ptr;       p.blk_disp_offset = (const int64_t *)a[BLK_OFF].ptr;
        p.blk_disp = (const double *)a[BLK_DISP].ptr;
        p.sb_point_a = (const int32_t *)a[SB_A].ptr;              p.sb_point_b = (const int32_t *)a[SB_B].ptr;
        p.sb_length = (const double *)a[SB_LEN].ptr;              p.sb_var = (const double *)a[SB_VAR].ptr;
        p.dg_row_begin = (const int32_t *)a[DG_BEGIN].ptr;        p.dg_slot = (const int32_t *)a[DG_SLOT].ptr;
        p.dg_obs = (const double *)a[DG_OBS].ptr;                 p.dg_var = (const double *)a[DG_VAR].ptr;
        p.dg_disp_offset = (const int64_t *)a[DG_OFF].ptr;        p.dg_disp = (const double *)a[DG_DISP].ptr;
        /* the n_* counts follow from the array lengths */
        p.n_points = a[POINT_COL].len / 3;
        p.n_cameras = a[IO_COL].len / 3;
        p.n_dist = a[DIST_KIND].len;
        p.n_images = a[IMAGE_CAMERA].len;
        p.n_image_points = a[IP_IMAGE].len;
        p.n_image_blocks = a[BLK_BEGIN].len > 0 ? a[BLK_BEGIN].len - 1 : 0;
        p.n_scale_bars = a[SB_A].len;
        p.n_direct_groups = a[DG_BEGIN].len > 0 ? a[DG_BEGIN].len - 1 : 0;
        p.n_direct_rows = a[DG_SLOT].len;
        rc = jaicov_neq_create(&p, &o, &eng);            /* copies everything it keeps */
        if (rc != JAICOV_OK) {
            strncpy(msg, eng ? jaicov_neq_last_error(eng) : "jaicov_neq_create failed", sizeof(msg) - 1);
            msg[sizeof(msg) - 1] = 0;
            jaicov_neq_destroy(eng);
            eng = NULL;
        }
    }
    while (n > 0) unhold(e, &a[--n]);
    if (bad) return 0;                                   /* the JNI exception of the failed lookup / copy is pending */
    if (rc != JAICOV_OK) { throw_status(e, rc, msg); return 0; }
    return (jlong)(intptr_t)eng;
}

JNIEXPORT void JNICALL NAT(destroy)(JNIEnv *e, jclass k, jlong h) { (void)e; (void)k; jaicov_neq_destroy(ENG(h)); }
JNIEXPORT jstring JNICALL NAT(lastError)(JNIEnv *e, jclass k, jlong h) { (void)k; return (*e)->NewStringUTF(e, jaicov_neq_last_error(ENG(h))); }
JNIEXPORT jlong JNICALL NAT(numSlots)(JNIEnv *e, jclass k, jlong h) { (void)e; (void)k; return (jlong)jaicov_neq_num_slots(ENG(h)); }
JNIEXPORT jlong JNICALL NAT(packedLength)(JNIEnv *e, jclass k, jlong h) { (void)e; (void)k; return (jlong)jaicov_neq_packed_length(ENG(h)); }

JNIEXPORT jint JNICALL NAT(setParameters)(JNIEnv *e, jclass k, jlong h, jdoubleArray s) {
    (void)k;
    jsize n = (*e)->GetArrayLength(e, s);
    double *p = (double *)(*e)->GetPrimitiveArrayCritical(e, s, NULL);
    if (!p) return JAICOV_ERR_OUT_OF_MEMORY;
    int rc = jaicov_neq_set_parameters(ENG(h), p, (size_t)n);
    (*e)->ReleasePrimitiveArrayCritical(e, s, p, JNI_ABORT);
    return rc;
}
JNIEXPORT jint JNICALL NAT(getParameters)(JNIEnv *e, jclass k, jlong h, jdoubleArray s) {
    (void)k;
    jsize n = (*e)->GetArrayLength(e, s);
    double *p = (double *)(*e)->GetPrimitiveArrayCritical(e, s, NULL);
    if (!p) return JAICOV_ERR_OUT_OF_MEMORY;
    int rc = jaicov_neq_get_parameters(ENG(h), p, (size_t)n);
    (*e)->ReleasePrimitiveArrayCritical(e, s, p, 0);
    return rc;
}
JNIEXPORT jint JNICALL NAT(build)(JNIEnv *e, jclass k, jlong h, jdouble s2, jdouble lambda, jboolean sim) {
    (void)e; (void)k;
    return jaicov_neq_build(ENG(h), s2, lambda, sim ? 1 : 0);
}
JNIEXPORT jint JNICALL NAT(accumulate)(JNIEnv *e, jclass k, jlong h, jdouble s2, jdouble lambda) {
    (void)e; (void)k;
    return jaicov_neq_accumulate2(ENG(h), s2, lambda);
}
JNIEXPORT jint JNICALL NAT(finish)(JNIEnv *e, jclass k, jlong h, jdouble s2, jdouble lambda, jboolean sim) {
    (void)e; (void)k;
    return jaicov_neq_finalize(ENG(h), s2, lambda, sim ? 1 : 0);
}
/* out[0] = DEVICE address of the packed partial normal equations, out[1] = number of doubles: what the host hands to
 * ncclAllReduce (sum, double) between accumulate() and finish() on a multi-GPU node */
JNIEXPORT jint JNICALL NAT(reduceBuffer)(JNIEnv *e, jclass k, jlong h, jlongArray out) {
    (void)k;
    void *ptr = NULL;
    size_t cnt = 0;
    int rc = jaicov_neq_reduce_buffer(ENG(h), &ptr, &cnt);
    jlong v[2] = {(jlong)(intptr_t)ptr, (jlong)cnt};
    if ((*e)->GetArrayLength(e, out) < 2) return JAICOV_ERR_BAD_ARGUMENT;
    (*e)->SetLongArrayRegion(e, out, 0, 2, v);
    return rc;
}
/* device pointer + count (+ stream) triples of the three exchange buffers of a sharded run */
static jint put_longs(JNIEnv *e, jlongArray out, int rc, const jlong *v, jsize n) {
    if ((*e)->GetArrayLength(e, out) < n) return JAICOV_ERR_BAD_ARGUMENT;
    (*e)->SetLongArrayRegion(e, out, 0, n, v);
    return rc;
}
/* out = {device address, count, hipStream_t}: the collective is enqueued on that stream, no host wait (jaicov_neq_reduce_buffer_async) */
JNIEXPORT jint JNICALL NAT(reduceBufferAsync)(JNIEnv *e, jclass k, jlong h, jlongArray out) {
    (void)k;
    void *ptr = NULL, *st = NULL;
    size_t cnt = 0;
    int rc = jaicov_neq_reduce_buffer_async(ENG(h), &ptr, &cnt, &st);
    jlong v[3] = {(jlong)(intptr_t)ptr, (jlong)cnt, (jlong)(intptr_t)st};
    return put_longs(e, out, rc, v, 3);
}
/* the sharded FINAL pass of MatrixInversion.FULL (BA:268-271 per rank): [F | L_E^-1] of this rank's images, summed over the ranks */
JNIEXPORT jint JNICALL NAT(expansionBuffer)(JNIEnv *e, jclass k, jlong h, jlongArray out) {
    (void)k;
    void *ptr = NULL;
    size_t cnt = 0;
    int rc = jaicov_neq_expansion_buffer(ENG(h), &ptr, &cnt);
    jlong v[2] = {(jlong)(intptr_t)ptr, (jlong)cnt};
    return put_longs(e, out, rc, v, 2);
}
/* the EO steps this rank back-substituted (6 per image), summed over the ranks into dx's EO entries */
JNIEXPORT jint JNICALL NAT(eoStepBuffer)(JNIEnv *e, jclass k, jlong h, jlongArray out) {
    (void)k;
    void *ptr = NULL;
    size_t cnt = 0;
    int rc = jaicov_neq_eo_step_buffer(ENG(h), &ptr, &cnt);
    jlong v[2] = {(jlong)(intptr_t)ptr, (jlong)cnt};
    return put_longs(e, out, rc, v, 2);
}
JNIEXPORT jint JNICALL NAT(abiVersion0)(JNIEnv *e, jclass k) { (void)e; (void)k; return jaicov_neq_abi_version(); }
/* small double[] out-parameters: filled through a stack copy (no array pinned across the call) */
static jint put_doubles(JNIEnv *e, jdoubleArray out, int rc, const double *v, jsize have) {
    jsize n = (*e)->GetArrayLength(e, out);
    (*e)->SetDoubleArrayRegion(e, out, 0, n < have ? n : have, v);
    return rc;
}
JNIEXPORT jint JNICALL NAT(lastTimings)(JNIEnv *e, jclass k, jlong h, jdoubleArray ms) {
    (void)k;
    double v[8] = {0};
    return put_doubles(e, ms, jaicov_neq_last_timings(ENG(h), v, 8), v, 8);
}
JNIEXPORT jint JNICALL NAT(createTimings)(JNIEnv *e, jclass k, jlong h, jdoubleArray ms) {
    (void)k;
    double v[8] = {0};
    return put_doubles(e, ms, jaicov_neq_create_timings(ENG(h), v, 8), v, 8);
}
JNIEXPORT jint JNICALL NAT(setProfiling)(JNIEnv *e, jclass k, jlong h, jboolean on) { (void)e; (void)k; return jaicov_neq_set_profiling(ENG(h), on ? 1 : 0); }
JNIEXPORT jint JNICALL NAT(kernelStats)(JNIEnv *e, jclass k, jlong h, jdoubleArray stats, jboolean reset) {
    (void)k;
    double v[16] = {0};
    return put_doubles(e, stats, jaicov_neq_kernel_stats(ENG(h), v, 16, reset ? 1 : 0), v, 16);
}
/* PDF:285-445 for the image points [begin, begin + count): w has 2 count entries, A 2 count (12 + JAICOV_MAX_DIST_PER_CAMERA) */
JNIEXPORT jint JNICALL NAT(getRows)(JNIEnv *e, jclass k, jlong h, jint begin, jint count, jdoubleArray w, jdoubleArray A) {
    (void)k;
    if (count < 0 || (*e)->GetArrayLength(e, w) < 2 * count ||
        (*e)->GetArrayLength(e, A) < 2 * count * (12 + JAICOV_MAX_DIST_PER_CAMERA)) return JAICOV_ERR_BAD_ARGUMENT;
    double *pw = (double *)(*e)->GetDoubleArrayElements(e, w, NULL);          /* small: a copy, so that only ONE array is pinned */
    if (!pw) return JAICOV_ERR_OUT_OF_MEMORY;
    double *pA = (double *)(*e)->GetPrimitiveArrayCritical(e, A, NULL);
    int rc = JAICOV_ERR_OUT_OF_MEMORY;
    if (pA) {
        rc = jaicov_neq_get_rows(ENG(h), (int32_t)begin, (int32_t)count, pw, pA);
        (*e)->ReleasePrimitiveArrayCritical(e, A, pA, 0);
    }
    (*e)->ReleaseDoubleArrayElements(e, w, pw, 0);
    return rc;
}
JNIEXPORT jint JNICALL NAT(getBlockWeight)(JNIEnv *e, jclass k, jlong h, jint block, jdoubleArray out) {
    (void)k;
    jsize n = (*e)->GetArrayLength(e, out);
    double *p = (double *)(*e)->GetPrimitiveArrayCritical(e, out, NULL);
    if (!p) return JAICOV_ERR_OUT_OF_MEMORY;
    int rc = jaicov_neq_get_block_weight(ENG(h), (int32_t)block, p, (size_t)n);
    (*e)->ReleasePrimitiveArrayCritical(e, out, p, 0);
    return rc;
}
JNIEXPORT jint JNICALL NAT(prepareInverse)(JNIEnv *e, jclass k, jlong h, jint invert) { (void)e; (void)k; return jaicov_neq_prepare_inverse(ENG(h), (int)invert); }
JNIEXPORT jint JNICALL NAT(reducedOrder)(JNIEnv *e, jclass k, jlong h) { (void)e; (void)k; return jaicov_neq_reduced_order(ENG(h)); }
JNIEXPORT jint JNICALL NAT(cofactorOrder)(JNIEnv *e, jclass k, jlong h) { (void)e; (void)k; return jaicov_neq_cofactor_order(ENG(h)); }

/* dx arrays must hold exactly U doubles: U (U + 1) / 2 == jaicov_neq_packed_length().  Checked before anything is read. */
static int has_u_entries(JNIEnv *e, jlong h, jdoubleArray dx) {
    const size_t n = (size_t)(*e)->GetArrayLength(e, dx);
    return n * (n + 1) / 2 == jaicov_neq_packed_length(ENG(h));
}
/* The calls below block on the device (a factorisation, its possible repeat, the inverse: up to seconds).  A critical region would
 * stall every garbage collection of the JVM for that long, so the vectors travel through a malloc'd copy (U doubles: 0.14 MB at
 * config 4) with Get/SetDoubleArrayRegion instead of GetPrimitiveArrayCritical. */
JNIEXPORT jint JNICALL NAT(solve)(JNIEnv *e, jclass k, jlong h, jint invert, jdoubleArray dx) {   /* invert = JAICOV_INVERT_* = MatrixInversion (BA:65-70) */
    (void)k;
    if (!has_u_entries(e, h, dx)) return JAICOV_ERR_BAD_ARGUMENT;
    const jsize n = (*e)->GetArrayLength(e, dx);
    double *p = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!p) return JAICOV_ERR_OUT_OF_MEMORY;
    int rc = jaicov_neq_solve(ENG(h), (int)invert, p);
    if (rc == JAICOV_OK) (*e)->SetDoubleArrayRegion(e, dx, 0, n, p);
    free(p);
    return rc;
}
JNIEXPORT jint JNICALL NAT(omega)(JNIEnv *e, jclass k, jlong h, jdouble s2, jdoubleArray dx, jdoubleArray out) {
    (void)k;
    double om = 0.0;
    if (!has_u_entries(e, h, dx) || (*e)->GetArrayLength(e, out) < 1) return JAICOV_ERR_BAD_ARGUMENT;
    const jsize n = (*e)->GetArrayLength(e, dx);
    double *p = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!p) return JAICOV_ERR_OUT_OF_MEMORY;
    (*e)->GetDoubleArrayRegion(e, dx, 0, n, p);
    int rc = jaicov_neq_omega(ENG(h), s2, p, &om);
    free(p);
    (*e)->SetDoubleArrayRegion(e, out, 0, 1, &om);
    return rc;
}
JNIEXPORT jint JNICALL NAT(update)(JNIEnv *e, jclass k, jlong h, jdoubleArray dx, jdoubleArray mx) {
    (void)k;
    double m = 0.0;
    if (!has_u_entries(e, h, dx) || (*e)->GetArrayLength(e, mx) < 1) return JAICOV_ERR_BAD_ARGUMENT;
    const jsize n = (*e)->GetArrayLength(e, dx);
    double *p = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!p) return JAICOV_ERR_OUT_OF_MEMORY;
    (*e)->GetDoubleArrayRegion(e, dx, 0, n, p);
    int rc = jaicov_neq_update(ENG(h), p, &m);
    free(p);
    (*e)->SetDoubleArrayRegion(e, mx, 0, 1, &m);
    return rc;
}
JNIEXPORT jint JNICALL NAT(getNormal)(JNIEnv *e, jclass k, jlong h, jdoubleArray Np, jdoubleArray nv) {
    (void)k;
    jsize len = (*e)->GetArrayLength(e, Np), U = (*e)->GetArrayLength(e, nv);
    double *pn = (double *)(*e)->GetDoubleArrayElements(e, nv, NULL);        /* small: a copy, so that only ONE array is pinned */
    if (!pn) return JAICOV_ERR_OUT_OF_MEMORY;
    double *pN = (double *)(*e)->GetPrimitiveArrayCritical(e, Np, NULL);
    int rc = JAICOV_ERR_OUT_OF_MEMORY;
    if (pN) {
        rc = jaicov_neq_get_normal(ENG(h), pN, (size_t)len, pn, (size_t)U);
        (*e)->ReleasePrimitiveArrayCritical(e, Np, pN, 0);
    }
    (*e)->ReleaseDoubleArrayElements(e, nv, pn, 0);
    return rc;
}
JNIEXPORT jint JNICALL NAT(getCofactor)(JNIEnv *e, jclass k, jlong h, jdoubleArray q) {
    (void)k;
    jsize n = (*e)->GetArrayLength(e, q);
    double *p = (double *)(*e)->GetPrimitiveArrayCritical(e, q, NULL);
    if (!p) return JAICOV_ERR_OUT_OF_MEMORY;
    int rc = jaicov_neq_get_cofactor(ENG(h), p, (size_t)n);
    (*e)->ReleasePrimitiveArrayCritical(e, q, p, 0);
    return rc;
}
static jint sub_block(JNIEnv *e, jlong h, int scaled, jdouble scale, jintArray idx, jdoubleArray out) {
    jsize n = (*e)->GetArrayLength(e, idx);
    if ((*e)->GetArrayLength(e, out) < n * n) return JAICOV_ERR_BAD_ARGUMENT;
    jint *ip = (*e)->GetIntArrayElements(e, idx, NULL);          /* jint is int32_t; a copy (small) */
    if (!ip) return JAICOV_ERR_OUT_OF_MEMORY;
    double *p = (double *)(*e)->GetPrimitiveArrayCritical(e, out, NULL);
    int rc = JAICOV_ERR_OUT_OF_MEMORY;
    if (p) {
        rc = scaled ? jaicov_neq_get_dispersion_sub(ENG(h), scale, (const int32_t *)ip, (int32_t)n, p)
                    : jaicov_neq_get_cofactor_sub(ENG(h), (const int32_t *)ip, (int32_t)n, p);
        (*e)->ReleasePrimitiveArrayCritical(e, out, p, 0);
    }
    (*e)->ReleaseIntArrayElements(e, idx, ip, JNI_ABORT);
    return rc;
}
JNIEXPORT jint JNICALL NAT(getCofactorSub)(JNIEnv *e, jclass k, jlong h, jintArray idx, jdoubleArray out) {
    (void)k;
    return sub_block(e, h, 0, 1.0, idx, out);
}
JNIEXPORT jint JNICALL NAT(getDispersionSub)(JNIEnv *e, jclass k, jlong h, jdouble scale, jintArray idx, jdoubleArray out) {
    (void)k;
    return sub_block(e, h, 1, scale, idx, out);
}
/* BA.estimateModel() on the engine.  res[0..6] = state, iterations, omega, max|dx|, final lambda, seconds total, seconds last pass */
JNIEXPORT jint JNICALL NAT(estimate)(JNIEnv *e, jclass k, jlong h, jint maxIterations, jint invert, jboolean simulation,
                                      jdouble lambda0, jdouble sigma2apriori, jdoubleArray res) {
    (void)k;
    jaicov_estimate_options o;
    jaicov_estimate_result r;
    memset(&o, 0, sizeof(o));
    memset(&r, 0, sizeof(r));
    o.struct_size = sizeof(o);
    o.max_iterations = maxIterations;
    o.invert = invert;
    o.simulation = simulation ? 1 : 0;
    o.lambda0 = lambda0;
    o.sigma2apriori = sigma2apriori;
    if ((*e)->GetArrayLength(e, res) < 7) return JAICOV_ERR_BAD_ARGUMENT;
    int rc = jaicov_neq_estimate(ENG(h), &o, &r);        /* no array is pinned while the loop runs */
    jdouble v[7] = {(jdouble)r.state, (jdouble)r.iterations, r.omega, r.max_abs_dx, r.final_lambda, r.seconds_total, r.seconds_last_pass};
    (*e)->SetDoubleArrayRegion(e, res, 0, 7, v);
    return rc;
}
/* BundleAdjustment.interrupt() (BA:1455): callable from another Java thread while estimate() runs */
JNIEXPORT jint JNICALL NAT(cancel)(JNIEnv *e, jclass k, jlong h) { (void)e; (void)k; return jaicov_neq_cancel(ENG(h)); }
