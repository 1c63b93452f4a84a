/*
 * Reference-side binding (NOT compiled in this repository: the build image has no JDK / jni.h).
 * A JAICOV maintainer adds this class next to org.applied_geodesy.adjustment.bundle.BundleAdjustment and replaces the
 * three call sites named in INTEGRATION.md.  It flattens the object graph once (after prepareUnknownParameters) and
 * forwards the per-iteration calls to libjaicov_neq.so through the JNI shim java/jni/jaicov_jni.c.
 */
package org.applied_geodesy.adjustment.bundle.nativeengine;

public final class NativeNormalEquationEngine implements AutoCloseable {
	static { System.loadLibrary("jaicov_jni"); }      // which in turn links libjaicov_neq.so

	private long handle;                                // jaicov_engine*

	/** Mirrors jaicov_problem_desc (include/jaicov_neq.h); every array is what BA:667-782 produced. */
	public static final class ProblemDescription {
		public int numberOfUnknowns, rankDefect, datumFlags;
		public int[] pointColumn, interiorColumn, cameraDistortionBegin, distortionKind, distortionOrder, distortionColumn;
		public int[] imageCamera, exteriorColumn, imagePointImage, imagePointPoint, blockBegin, scaleBarA, scaleBarB;
		public int[] directRowBegin, directSlot;
		public byte[] pointDatum;
		public double[] cameraR0, x, y, varianceX, varianceY, rho, blockDispersion, scaleBarLength, scaleBarVariance;
		public double[] directObservation, directVariance, directDispersion;
		public long[] blockDispersionOffset, directDispersionOffset;
	}

	public NativeNormalEquationEngine(ProblemDescription d, int device) { this.handle = create(d, device); }

	/** BA:235 createNormalEquation() */
	public void build(double sigma2apriori, double lambda, boolean simulation) { check(build(handle, sigma2apriori, lambda, simulation)); }
	/** NES.applyPrecondition + MathExtension.solve(N, n, invert) + reverse preconditioning (BA:238,270-297) */
	public void solve(int invert, double[] dx) { check(solve(handle, invert, dx)); }
	/** MatrixInversion -> JAICOV_INVERT_*: NONE 0, FULL 1, REDUCED and PRE_ELIMINATION 2 (BA:65-70, 261-271) */
	public static int invertMode(Enum<?> matrixInversion) {
		switch (matrixInversion.name()) { case "NONE": return 0; case "FULL": return 1; default: return 2; }
	}
	/** announces the invert mode of the solve after the next build (BA:250: the final pass is known before it is built) */
	public void prepareInverse(int invert) { check(prepareInverse(handle, invert)); }
	/** order of the cofactor matrix of the last inverting solve: u + d (FULL) or numRows of BA:262 (REDUCED) */
	public int cofactorOrder() { return cofactorOrder(handle); }
	/** BA:472 getOmega(dx) */
	public double omega(double sigma2apriori, double[] dx) { double[] o = new double[1]; check(omega(handle, sigma2apriori, dx, o)); return o[0]; }
	/** BA:450 updateUnknownParameters(dx); returns max|dx| */
	public double update(double[] dx) { double[] m = new double[1]; check(update(handle, dx, m)); return m[0]; }
	public void setParameters(double[] slots) { check(setParameters(handle, slots)); }
	public void getParameters(double[] slots) { check(getParameters(handle, slots)); }
	/** UpperSymmPackMatrix.getData() order (UPLO='U'), length U(U+1)/2: new UpperSymmPackMatrix(U) then copy */
	public void getCofactor(double[] packed) { check(getCofactor(handle, packed)); }
	/**
	 * scale * Qxx[indices, indices] as a dense row-major k x k block, gathered (and scaled) on the device: what
	 * DefaultResultWriter.java:139-147 (scale = sigma2apost) and MatlabResultWriter.java:210-221 (scale = 1) read
	 * element by element from the packed matrix.  The writers replace their double loop by one call.
	 */
	public double[] getDispersionSub(double scale, int[] indices) {
		double[] out = new double[indices.length * indices.length];
		check(getDispersionSub(handle, scale, indices, out));
		return out;
	}

	@Override public void close() { if (handle != 0) { destroy(handle); handle = 0; } }

	private void check(int status) {
		if (status == 0) return;
		String msg = lastError(handle);
		if (status > 0) throw new no.uib.cipr.matrix.MatrixSingularException();          // MX:350,361
		if (status == -4) throw new OutOfMemoryError(msg);                                // BA:370-375
		throw new IllegalArgumentException(msg);                                          // MX:352,363
	}

	private static native long create(ProblemDescription d, int device);
	private static native void destroy(long h);
	private static native String lastError(long h);
	private static native int setParameters(long h, double[] slots);
	private static native int getParameters(long h, double[] slots);
	private static native int build(long h, double sigma2, double lambda, boolean simulation);
	private static native int solve(long h, int invert, double[] dx);
	private static native int prepareInverse(long h, int invert);
	private static native int cofactorOrder(long h);
	private static native int omega(long h, double sigma2, double[] dx, double[] out);
	private static native int update(long h, double[] dx, double[] maxAbs);
	private static native int getCofactor(long h, double[] packed);
	private static native int getDispersionSub(long h, double scale, int[] indices, double[] out);
}
