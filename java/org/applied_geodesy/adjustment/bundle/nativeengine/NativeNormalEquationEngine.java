/*
 * Reference-side binding (not compiled in this repository: the build image has no JDK; tests/test_jni_binding.py checks it
 * against the JNI shim and the C header at the text level).
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

	/** all images on one device */
	public NativeNormalEquationEngine(ProblemDescription d, int device) { this(d, device, -1, -1, true); }
	/** one rank of a sharded run: images [imageBegin, imageEnd); applyShared on the rank that adds scale bars and directly observed groups */
	public NativeNormalEquationEngine(ProblemDescription d, int device, int imageBegin, int imageEnd, boolean applyShared) {
		this.handle = create(d, device, imageBegin, imageEnd, applyShared);   // throws on failure (status mapped as in check())
	}

	/** BA:235 createNormalEquation() */
	public void build(double sigma2apriori, double lambda, boolean simulation) { check(build(handle, sigma2apriori, lambda, simulation)); }
	/** multi-GPU form of build(): accumulate() -> all-reduce of reduceBuffer() over the ranks -> finish() */
	public void accumulate(double sigma2apriori, double lambda) { check(accumulate(handle, sigma2apriori, lambda)); }
	public void finish(double sigma2apriori, double lambda, boolean simulation) { check(finish(handle, sigma2apriori, lambda, simulation)); }
	/** {device address, count of doubles} of this rank's packed partial normal equations (ncclAllReduce, sum, double) */
	public long[] reduceBuffer() { long[] r = new long[2]; check(reduceBuffer(handle, r)); return r; }
	/** NES.applyPrecondition + MathExtension.solve(N, n, invert) + reverse preconditioning (BA:238,270-297) */
	public void solve(int invert, double[] dx) { check(solve(handle, invert, dx)); }
	/**
	 * MatrixInversion -> JAICOV_INVERT_*: NONE 0, FULL 1, REDUCED and PRE_ELIMINATION 2 (BA:65-70, 261-271).  3 = FULL_EXPANDED: the
	 * same matrix as FULL, expanded from the inverse of the EO-reduced system (faster and more accurate where the engine can
	 * pre-eliminate; served as FULL where it cannot) -- what the patched estimateModel() passes for MatrixInversion.FULL.
	 */
	public static int invertMode(Enum<?> matrixInversion) {
		switch (matrixInversion.name()) { case "NONE": return 0; case "FULL": return 1; default: return 2; }
	}
	/** announces the invert mode of the solve after the next build (BA:250: the final pass is known before it is built) */
	public void prepareInverse(int invert) { check(prepareInverse(handle, invert)); }
	/** order of the system the last build assembled: u + d, or numRows of BA:262 when the exterior orientations were pre-eliminated */
	public int reducedOrder() { return reducedOrder(handle); }
	/** order of the cofactor matrix of the last inverting solve: u + d (FULL) or numRows of BA:262 (REDUCED) */
	public int cofactorOrder() { return cofactorOrder(handle); }
	/** BA:472 getOmega(dx) */
	public double omega(double sigma2apriori, double[] dx) { double[] o = new double[1]; check(omega(handle, sigma2apriori, dx, o)); return o[0]; }
	/** BA:450 updateUnknownParameters(dx); returns max|dx| */
	public double update(double[] dx) { double[] m = new double[1]; check(update(handle, dx, m)); return m[0]; }
	public int numSlots() { return (int) numSlots(handle); }
	public long packedLength() { return packedLength(handle); }
	public void setParameters(double[] slots) { check(setParameters(handle, slots)); }
	public void getParameters(double[] slots) { check(getParameters(handle, slots)); }
	/** N (packed 'U', UpperSymmPackMatrix.getData() order) and n of the last build: BA:789 createNormalEquation()'s result */
	public void getNormal(double[] packedN, double[] n) { check(getNormal(handle, packedN, n)); }
	/** UpperSymmPackMatrix.getData() order (UPLO='U'), length order(order+1)/2 with order = cofactorOrder() */
	public void getCofactor(double[] packed) { check(getCofactor(handle, packed)); }
	/** Qxx[indices, indices] as a dense row-major k x k block gathered on the device (MatlabResultWriter.java:210-221) */
	public double[] getCofactorSub(int[] indices) {
		double[] out = new double[indices.length * indices.length];
		check(getCofactorSub(handle, indices, out));
		return out;
	}
	/**
	 * scale * Qxx[indices, indices] as a dense row-major k x k block, gathered (and scaled) on the device: what
	 * DefaultResultWriter.java:139-147 (scale = sigma2apost) reads element by element from the packed matrix.
	 */
	public double[] getDispersionSub(double scale, int[] indices) {
		double[] out = new double[indices.length * indices.length];
		check(getDispersionSub(handle, scale, indices, out));
		return out;
	}
	/**
	 * BA.estimateModel() (BA:203-387) run natively: {state (EstimationStateType id), iterations, omega, max|dx|, final lambda,
	 * seconds total, seconds of the last pass}.  interrupt() from another thread ends it with state -1 (BA:240, 320).
	 */
	public double[] estimate(int maxIterations, int invert, boolean simulation, double lambda0, double sigma2apriori) {
		double[] r = new double[7];
		check(estimate(handle, maxIterations, invert, simulation, lambda0, sigma2apriori, r));
		return r;
	}
	/** BundleAdjustment.interrupt() (BA:1455) */
	public void interrupt() { check(cancel(handle)); }

	@Override public void close() { if (handle != 0) { destroy(handle); handle = 0; } }

	private void check(int status) {
		if (status == 0) return;
		String msg = lastError(handle);
		if (status > 0) throw new no.uib.cipr.matrix.MatrixSingularException();          // MX:350,361
		if (status == -4) throw new OutOfMemoryError(msg);                                // BA:370-375
		if (status == -1) throw new IllegalArgumentException(msg);                        // MX:352,363
		throw new IllegalStateException(msg);
	}

	private static native long create(ProblemDescription d, int device, int imageBegin, int imageEnd, boolean applyShared);
	private static native void destroy(long h);
	private static native String lastError(long h);
	private static native long numSlots(long h);
	private static native long packedLength(long h);
	private static native int setParameters(long h, double[] slots);
	private static native int getParameters(long h, double[] slots);
	private static native int build(long h, double sigma2, double lambda, boolean simulation);
	private static native int accumulate(long h, double sigma2, double lambda);
	private static native int finish(long h, double sigma2, double lambda, boolean simulation);
	private static native int reduceBuffer(long h, long[] pointerAndCount);
	private static native int prepareInverse(long h, int invert);
	private static native int reducedOrder(long h);
	private static native int cofactorOrder(long h);
	private static native int solve(long h, int invert, double[] dx);
	private static native int omega(long h, double sigma2, double[] dx, double[] out);
	private static native int update(long h, double[] dx, double[] maxAbs);
	private static native int getNormal(long h, double[] packedN, double[] n);
	private static native int getCofactor(long h, double[] packed);
	private static native int getCofactorSub(long h, int[] indices, double[] out);
	private static native int getDispersionSub(long h, double scale, int[] indices, double[] out);
	private static native int estimate(long h, int maxIterations, int invert, boolean simulation, double lambda0, double sigma2apriori, double[] result);
	private static native int cancel(long h);
}
