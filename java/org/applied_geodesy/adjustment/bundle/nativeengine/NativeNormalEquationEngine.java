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

	/**
	 * Mirrors jaicov_engine_options (include/jaicov_neq.h), field for field; 0 = the engine's default everywhere.  This is the
	 * configuration surface next to the reference's setters (BundleAdjustment.java:1123-1195): what those cannot express because the
	 * pure-Java loop has no such choice (summation order, refinement of the step, which groups are pre-eliminated, the sharding).
	 */
	public static final class EngineOptions {
		public int device = 0;
		/** this engine accumulates the images [imageBegin, imageEnd) only; -1 / -1 = all images */
		public int imageBegin = -1, imageEnd = -1;
		/** the rank that adds scale bars, directly observed groups, datum, damping (rank 0 of a sharded run) */
		public boolean applyShared = true;
		/** 0 structure-aware (product), 1 dense J'WJ contraction on the matrix cores, 2 the same with fp32 accumulation (measuring mode) */
		public int assemblyMode = 0;
		public int blockSize = 0;
		/** != 0: MatrixInversion.REDUCED's last pass leaves in the EO entries of dx what BA:261-273 leaves there (SURVEY quirk Q1) */
		public int reducedReferenceQuirk = 0;
		/** 0 = default = fixed summation order (bit-reproducible, like the reference); < 0 = arrival-order sums (0.3 ms per pass faster at config 4) */
		public int deterministic = 0;
		/** steps of iterative refinement per solve: 0 = default (one), < 0 none, at most 4 */
		public int refinement = 0;
		/** < 0: ordinary ImageCoordinate groups stay outside the EO pre-elimination (rounds 1-3 path) */
		public int ordinaryGroupElimination = 0;
		/** < 0: inverse dispersions as the blocked Cholesky leaves them (no Newton-Schulz step) */
		public int dispersionRefinement = 0;
		/** != 0 on a sharded engine: the host sums expansionBuffer() over the ranks before an inverting solve with invert = 3 */
		public int expansionExchange = 0;
		/** < 0: no Newton-Schulz step on the inverse of systems of order <= 8192 (the step takes Qxx from 3e-9 to 1e-12 of the exact inverse at config 3) */
		public int inverseRefinement = 0;

		/** the options BundleAdjustment.native.patch reads from system properties, next to org.applied_geodesy.adjustment.bundle.native */
		public static EngineOptions fromSystemProperties() {
			EngineOptions o = new EngineOptions();
			o.device = Integer.getInteger("org.applied_geodesy.adjustment.bundle.native.device", 0);
			o.reducedReferenceQuirk = Boolean.getBoolean("org.applied_geodesy.adjustment.bundle.native.reducedReferenceQuirk") ? 1 : 0;
			o.deterministic = Integer.getInteger("org.applied_geodesy.adjustment.bundle.native.deterministic", 0);
			o.refinement = Integer.getInteger("org.applied_geodesy.adjustment.bundle.native.refinement", 0);
			return o;
		}
	}

	/** all images on one device */
	public NativeNormalEquationEngine(ProblemDescription d, int device) { this(d, device, -1, -1, true); }
	/** one rank of a sharded run: images [imageBegin, imageEnd); applyShared on the rank that adds scale bars and directly observed groups */
	public NativeNormalEquationEngine(ProblemDescription d, int device, int imageBegin, int imageEnd, boolean applyShared) {
		this(d, shard(device, imageBegin, imageEnd, applyShared));
	}
	/** every option of jaicov_engine_options */
	public NativeNormalEquationEngine(ProblemDescription d, EngineOptions options) {
		this.handle = create(d, options);                    // throws on failure (status mapped as in check())
	}
	private static EngineOptions shard(int device, int imageBegin, int imageEnd, boolean applyShared) {
		EngineOptions o = new EngineOptions();
		o.device = device; o.imageBegin = imageBegin; o.imageEnd = imageEnd; o.applyShared = applyShared;
		return o;
	}
	/** JAICOV_NEQ_ABI_VERSION of the loaded library */
	public static int abiVersion() { return abiVersion0(); }

	/** BA:235 createNormalEquation() */
	public void build(double sigma2apriori, double lambda, boolean simulation) { check(build(handle, sigma2apriori, lambda, simulation)); }
	/** multi-GPU form of build(): accumulate() -> all-reduce of reduceBuffer() over the ranks -> finish() */
	public void accumulate(double sigma2apriori, double lambda) { check(accumulate(handle, sigma2apriori, lambda)); }
	public void finish(double sigma2apriori, double lambda, boolean simulation) { check(finish(handle, sigma2apriori, lambda, simulation)); }
	/** {device address, count of doubles} of this rank's packed partial normal equations (ncclAllReduce, sum, double) */
	public long[] reduceBuffer() { long[] r = new long[2]; check(reduceBuffer(handle, r)); return r; }
	/**
	 * The same buffer without a host wait: {device address, count, hipStream_t}.  The buffer is complete in the order of that stream: enqueue
	 * the collective on it (ncclAllReduce(..., stream)) and finish() waits for it on the device -- the host never blocks between assembly and solve.
	 */
	public long[] reduceBufferAsync() { long[] r = new long[3]; check(reduceBufferAsync(handle, r)); return r; }
	/**
	 * {device address, count} of [F | L_E^-1] for the FINAL pass of MatrixInversion.FULL on a sharded engine (BA:268-271 per rank): this
	 * rank's images filled, zeros elsewhere.  Sum over the ranks between the all-reduce of reduceBuffer() and finish(); then solve(3, dx).
	 * Needs EngineOptions.expansionExchange and prepareInverse(3) before accumulate().
	 */
	public long[] expansionBuffer() { long[] r = new long[2]; check(expansionBuffer(handle, r)); return r; }
	/**
	 * {device address, count} of the exterior-orientation steps this engine back-substituted in the last solve (6 per image, zeros for the
	 * other ranks' images): summed over the ranks they are the EO entries of dx (BA:273 leaves them in dx on an unsharded run).
	 */
	public long[] eoStepBuffer() { long[] r = new long[2]; check(eoStepBuffer(handle, r)); return r; }
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
	/** stage times of the last pass, ms: rows, assembly, finalize, factor, solve, inverse, omega, total (jaicov_neq_last_timings) */
	public double[] lastTimings() { double[] t = new double[8]; check(lastTimings(handle, t)); return t; }
	/** stage times of the engine's creation, ms (jaicov_neq_create_timings) */
	public double[] createTimings() { double[] t = new double[8]; check(createTimings(handle, t)); return t; }
	/** per-launch HIP events around the factorisation kernel on / off (jaicov_neq_set_profiling) */
	public void setProfiling(boolean on) { check(setProfiling(handle, on)); }
	/** counters of the engine (jaicov_neq_kernel_stats): factorisation launches, ms, flops, ..., abandoned factorisations [6], ... */
	public double[] kernelStats(boolean reset) { double[] s = new double[16]; check(kernelStats(handle, s, reset)); return s; }
	/** misclosures w (2 per image point) and Jacobian rows A (2 x (12 + distortion parameters) per image point) of [begin, begin + count): PDF:285-445 */
	public void getRows(int begin, int count, double[] w, double[] A) { check(getRows(handle, begin, count, w, A)); }
	/** the weight matrix sigma0^2-free inv(D) of jointly dispersed image block `block`, row-major m x m (DOPG:82-86 getWeightMatrix) */
	public void getBlockWeight(int block, double[] out) { check(getBlockWeight(handle, block, out)); }

	@Override public void close() { if (handle != 0) { destroy(handle); handle = 0; } }

	private void check(int status) {
		if (status == 0) return;
		String msg = lastError(handle);
		if (status > 0) throw new no.uib.cipr.matrix.MatrixSingularException();          // MX:350,361
		if (status == -4) throw new OutOfMemoryError(msg);                                // BA:370-375
		if (status == -1) throw new IllegalArgumentException(msg);                        // MX:352,363
		throw new IllegalStateException(msg);
	}

	private static native long create(ProblemDescription d, EngineOptions options);
	private static native int abiVersion0();
	private static native int reduceBufferAsync(long h, long[] pointerCountStream);
	private static native int expansionBuffer(long h, long[] pointerAndCount);
	private static native int eoStepBuffer(long h, long[] pointerAndCount);
	private static native int lastTimings(long h, double[] ms);
	private static native int createTimings(long h, double[] ms);
	private static native int setProfiling(long h, boolean on);
	private static native int kernelStats(long h, double[] stats, boolean reset);
	private static native int getRows(long h, int begin, int count, double[] w, double[] A);
	private static native int getBlockWeight(long h, int block, double[] out);
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
