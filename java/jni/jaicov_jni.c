/*
 * JNI shim between NativeNormalEquationEngine (Java) and the C ABI of include/jaicov_neq.h.
 * NOT compiled here (no JDK / jni.h in the build image); build on a box with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include jaicov_jni.c \
 *       -L../../bundle-adjustment_amd/csrc -ljaicov_neq -o libjaicov_jni.so
 * Only two patterns occur: primitive arrays pinned with GetPrimitiveArrayCritical for the duration of one call, and the
 * one-time copy of the structure arrays in create().
 */
#include <jni.h>
#include <string.h>
#include "jaicov_neq.h"

#define CLS "org/applied_geodesy/adjustment/bundle/nativeengine/NativeNormalEquationEngine"
#define ENG(h) ((jaicov_engine *)(intptr_t)(h))

static void *pin(JNIEnv *e, jobject obj, jclass c, const char *name, const char *sig, jarray *arr) {
    *arr = (jarray)(*e)->GetObjectField(e, obj, (*e)->GetFieldID(e, c, name, sig));
    return *arr ? (*e)->GetPrimitiveArrayCritical(e, *arr, NULL) : NULL;
}

JNIEXPORT jlong JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_create(
    JNIEnv *e, jclass k, jobject d, jint device) {
    jclass c = (*e)->GetObjectClass(e, d);
    jaicov_problem_desc p;
    memset(&p, 0, sizeof(p));
    p.struct_size = sizeof(p);
    p.n_unknowns = (*e)->GetIntField(e, d, (*e)->GetFieldID(e, c, "numberOfUnknowns", "I"));
    p.rank_defect = (*e)->GetIntField(e, d, (*e)->GetFieldID(e, c, "rankDefect", "I"));
    p.datum_flags = (*e)->GetIntField(e, d, (*e)->GetFieldID(e, c, "datumFlags", "I"));
    jarray a[32]; int na = 0;
#define PIN(field, name, sig) p.field = pin(e, d, c, name, sig, &a[na]); na++
    PIN(point_col, "pointColumn", "[I");      PIN(point_datum, "pointDatum", "[B");
    PIN(io_col, "interiorColumn", "[I");      PIN(cam_r0, "cameraR0", "[D");
    PIN(cam_dist_begin, "cameraDistortionBegin", "[I");
    PIN(dist_kind, "distortionKind", "[I");   PIN(dist_order, "distortionOrder", "[I"); PIN(dist_col, "distortionColumn", "[I");
    PIN(image_camera, "imageCamera", "[I");   PIN(eo_col, "exteriorColumn", "[I");
    PIN(ip_image, "imagePointImage", "[I");   PIN(ip_point, "imagePointPoint", "[I");
    PIN(ip_x, "x", "[D"); PIN(ip_y, "y", "[D"); PIN(ip_var_x, "varianceX", "[D"); PIN(ip_var_y, "varianceY", "[D"); PIN(ip_rho, "rho", "[D");
    PIN(blk_ip_begin, "blockBegin", "[I");    PIN(blk_disp_offset, "blockDispersionOffset", "[J"); PIN(blk_disp, "blockDispersion", "[D");
    PIN(sb_point_a, "scaleBarA", "[I");       PIN(sb_point_b, "scaleBarB", "[I");
    PIN(sb_length, "scaleBarLength", "[D");   PIN(sb_var, "scaleBarVariance", "[D");
    PIN(dg_row_begin, "directRowBegin", "[I"); PIN(dg_slot, "directSlot", "[I");
    PIN(dg_obs, "directObservation", "[D");   PIN(dg_var, "directVariance", "[D");
    PIN(dg_disp_offset, "directDispersionOffset", "[J"); PIN(dg_disp, "directDispersion", "[D");
    /* the n_* counts follow from the array lengths */
    p.n_points = (*e)->GetArrayLength(e, a[0]) / 3;
    p.n_cameras = (*e)->GetArrayLength(e, a[2]) / 3;
    p.n_dist = a[5] ? (*e)->GetArrayLength(e, a[5]) : 0;
    p.n_images = (*e)->GetArrayLength(e, a[8]);
    p.n_image_points = (*e)->GetArrayLength(e, a[10]);
    p.n_image_blocks = a[17] ? (*e)->GetArrayLength(e, a[17]) - 1 : 0;
    p.n_scale_bars = a[20] ? (*e)->GetArrayLength(e, a[20]) : 0;
    p.n_direct_groups = a[24] ? (*e)->GetArrayLength(e, a[24]) - 1 : 0;
    p.n_direct_rows = a[25] ? (*e)->GetArrayLength(e, a[25]) : 0;
    jaicov_engine_options o;
    memset(&o, 0, sizeof(o));
    o.struct_size = sizeof(o); o.device = device; o.image_begin = o.image_end = -1; o.apply_shared = 1;
    jaicov_engine *eng = NULL;
    jaicov_neq_create(&p, &o, &eng);            /* copies everything; status is read back through lastError()/first call */
    for (int i = na - 1; i >= 0; i--)
        if (a[i]) (*e)->ReleasePrimitiveArrayCritical(e, a[i], NULL, JNI_ABORT);   /* address bookkeeping elided for brevity */
    return (jlong)(intptr_t)eng;
}

JNIEXPORT void JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_destroy(JNIEnv *e, jclass k, jlong h) {
    jaicov_neq_destroy(ENG(h));
}
JNIEXPORT jstring JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_lastError(JNIEnv *e, jclass k, jlong h) {
    return (*e)->NewStringUTF(e, jaicov_neq_last_error(ENG(h)));
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_build(
    JNIEnv *e, jclass k, jlong h, jdouble s2, jdouble lambda, jboolean sim) {
    return jaicov_neq_build(ENG(h), s2, lambda, sim ? 1 : 0);
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_solve(
    JNIEnv *e, jclass k, jlong h, jint invert, jdoubleArray dx) {   /* invert = JAICOV_INVERT_* = MatrixInversion (BA:65-70) */
    double *p = (*e)->GetPrimitiveArrayCritical(e, dx, NULL);
    int rc = jaicov_neq_solve(ENG(h), (int)invert, p);
    (*e)->ReleasePrimitiveArrayCritical(e, dx, p, 0);
    return rc;
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_prepareInverse(
    JNIEnv *e, jclass k, jlong h, jint invert) {
    return jaicov_neq_prepare_inverse(ENG(h), (int)invert);
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_cofactorOrder(JNIEnv *e, jclass k, jlong h) {
    return jaicov_neq_cofactor_order(ENG(h));
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_omega(
    JNIEnv *e, jclass k, jlong h, jdouble s2, jdoubleArray dx, jdoubleArray out) {
    double *p = (*e)->GetPrimitiveArrayCritical(e, dx, NULL), om = 0.0;
    int rc = jaicov_neq_omega(ENG(h), s2, p, &om);
    (*e)->ReleasePrimitiveArrayCritical(e, dx, p, JNI_ABORT);
    (*e)->SetDoubleArrayRegion(e, out, 0, 1, &om);
    return rc;
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_update(
    JNIEnv *e, jclass k, jlong h, jdoubleArray dx, jdoubleArray mx) {
    double *p = (*e)->GetPrimitiveArrayCritical(e, dx, NULL), m = 0.0;
    int rc = jaicov_neq_update(ENG(h), p, &m);
    (*e)->ReleasePrimitiveArrayCritical(e, dx, p, JNI_ABORT);
    (*e)->SetDoubleArrayRegion(e, mx, 0, 1, &m);
    return rc;
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_setParameters(
    JNIEnv *e, jclass k, jlong h, jdoubleArray s) {
    jsize n = (*e)->GetArrayLength(e, s);
    double *p = (*e)->GetPrimitiveArrayCritical(e, s, NULL);
    int rc = jaicov_neq_set_parameters(ENG(h), p, (size_t)n);
    (*e)->ReleasePrimitiveArrayCritical(e, s, p, JNI_ABORT);
    return rc;
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_getParameters(
    JNIEnv *e, jclass k, jlong h, jdoubleArray s) {
    jsize n = (*e)->GetArrayLength(e, s);
    double *p = (*e)->GetPrimitiveArrayCritical(e, s, NULL);
    int rc = jaicov_neq_get_parameters(ENG(h), p, (size_t)n);
    (*e)->ReleasePrimitiveArrayCritical(e, s, p, 0);
    return rc;
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_getCofactor(
    JNIEnv *e, jclass k, jlong h, jdoubleArray q) {
    jsize n = (*e)->GetArrayLength(e, q);
    double *p = (*e)->GetPrimitiveArrayCritical(e, q, NULL);
    int rc = jaicov_neq_get_cofactor(ENG(h), p, (size_t)n);
    (*e)->ReleasePrimitiveArrayCritical(e, q, p, 0);
    return rc;
}
JNIEXPORT jint JNICALL Java_org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_getDispersionSub(
    JNIEnv *e, jclass k, jlong h, jdouble scale, jintArray idx, jdoubleArray out) {
    jsize n = (*e)->GetArrayLength(e, idx);
    if ((*e)->GetArrayLength(e, out) < n * n) return JAICOV_ERR_BAD_ARGUMENT;
    jint *ip = (*e)->GetIntArrayElements(e, idx, NULL);          /* jint is int32_t */
    double *p = (*e)->GetPrimitiveArrayCritical(e, out, NULL);
    int rc = jaicov_neq_get_dispersion_sub(ENG(h), scale, (const int32_t *)ip, (int32_t)n, p);
    (*e)->ReleasePrimitiveArrayCritical(e, out, p, 0);
    (*e)->ReleaseIntArrayElements(e, idx, ip, JNI_ABORT);
    return rc;
}
