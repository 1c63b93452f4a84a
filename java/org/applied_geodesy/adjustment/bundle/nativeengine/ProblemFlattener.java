/*
 * Reference-side source (not compiled in this repository: no JDK in the build image; tests/test_jni_binding.py checks it at
 * the text level against NativeNormalEquationEngine.ProblemDescription and against the C++ walk it mirrors).
 *
 * flatten(): the object graph of a BundleAdjustment, AFTER prepareUnknownParameters() (BundleAdjustment.java:667-782) has
 * numbered rows and columns, as the flat arrays of jaicov_problem_desc (include/jaicov_neq.h).  The walk is the one of
 * prepareUnknownParameters() itself -- cameras -> images -> image coordinates, interior orientations and distortion models per
 * camera, exterior orientations per image, scale bars, directly observed groups -- and is the Java text of
 * host/jaicov.cpp BundleAdjustment::flatten(), which runs in the tests.
 *
 * Called from the patched BundleAdjustment (java/patch/BundleAdjustment.native.patch), which hands over its private
 * collections.  Parameter VALUES travel separately (slotValues / storeSlotValues): one double per parameter in the order
 *     [ X,Y,Z per object point | x0,y0,c per camera | distortion coefficients per camera | X0,Y0,Z0,omega,phi,kappa per image ].
 */
package org.applied_geodesy.adjustment.bundle.nativeengine;

import java.util.ArrayList;
import java.util.Collection;
import java.util.IdentityHashMap;
import java.util.List;
import java.util.Map;
import java.util.Set;

import org.applied_geodesy.adjustment.bundle.ObjectCoordinate;
import org.applied_geodesy.adjustment.bundle.ScaleBar;
import org.applied_geodesy.adjustment.bundle.camera.Camera;
import org.applied_geodesy.adjustment.bundle.camera.Image;
import org.applied_geodesy.adjustment.bundle.camera.ImageCoordinate;
import org.applied_geodesy.adjustment.bundle.camera.distortion.DistortionModel;
import org.applied_geodesy.adjustment.bundle.camera.distortion.PolynomialDistortionModel;
import org.applied_geodesy.adjustment.bundle.parameter.DirectlyObservedParameterGroup;
import org.applied_geodesy.adjustment.bundle.parameter.ObservationParameter;
import org.applied_geodesy.adjustment.bundle.parameter.PolynomialCoefficient;
import org.applied_geodesy.adjustment.bundle.parameter.UnknownParameter;
import org.applied_geodesy.adjustment.defect.RankDefect;

import no.uib.cipr.matrix.UpperSPDPackMatrix;

public final class ProblemFlattener {
	/** jaicov_dist_kind (include/jaicov_neq.h), the application order of DistortionModel.Type (DistortionModel.java:29-37) */
	private static final int AFFINITY_CX = 0, AFFINITY_CY = 1, TANGENTIAL_BX = 2, TANGENTIAL_BY = 3, TANGENTIAL_BI = 4,
			RADIAL_AI = 5, DISTANCE_DI = 6, ZERNIKE_X = 7, ZERNIKE_Y = 8, ZERNIKE_Z = 9;
	/** JAICOV_DATUM_* bits in the order of BA:523-530 */
	private static final int DATUM_TX = 1, DATUM_TY = 2, DATUM_TZ = 4, DATUM_RX = 8, DATUM_RY = 16, DATUM_RZ = 32, DATUM_SCALE = 64;

	/** the parameters behind the slot vector, in slot order: values are read before and written back after the native loop */
	private final List<UnknownParameter<?>> slotParameters = new ArrayList<UnknownParameter<?>>();
	private final List<ObservationParameter<?>> directObservations = new ArrayList<ObservationParameter<?>>();

	/** UnknownParameter.java:27: a fixed parameter carries Integer.MAX_VALUE; the engine's JAICOV_COL_FIXED is -1 */
	private static int column(UnknownParameter<?> p) {
		int c = p.getColumn();
		return (c == Integer.MAX_VALUE || c < 0) ? -1 : c;
	}

	public NativeNormalEquationEngine.ProblemDescription flatten(List<Camera> cameras, Set<ObjectCoordinate> objectCoordinates,
			Set<ScaleBar> scaleBars, Set<DirectlyObservedParameterGroup> observedParameterGroups, RankDefect rankDefect,
			int numberOfUnknownParameters) {
		NativeNormalEquationEngine.ProblemDescription d = new NativeNormalEquationEngine.ProblemDescription();
		this.slotParameters.clear();
		this.directObservations.clear();

		d.rankDefect = rankDefect.getDefect();
		d.numberOfUnknowns = numberOfUnknownParameters + d.rankDefect;          // BA:791
		d.datumFlags = (rankDefect.estimateTranslationX() ? DATUM_TX : 0) | (rankDefect.estimateTranslationY() ? DATUM_TY : 0)
				| (rankDefect.estimateTranslationZ() ? DATUM_TZ : 0) | (rankDefect.estimateRotationX() ? DATUM_RX : 0)
				| (rankDefect.estimateRotationY() ? DATUM_RY : 0) | (rankDefect.estimateRotationZ() ? DATUM_RZ : 0)
				| (rankDefect.estimateScale() ? DATUM_SCALE : 0);

		// ---- slots: object points (LinkedHashSet order of BA.objectCoordinates = the order the walk met them) ----------------
		Map<ObjectCoordinate, Integer> pointIndex = new IdentityHashMap<ObjectCoordinate, Integer>();
		Map<UnknownParameter<?>, Integer> slotOf = new IdentityHashMap<UnknownParameter<?>, Integer>();
		int P = objectCoordinates.size();
		d.pointColumn = new int[3 * P];
		d.pointDatum = new byte[P];
		int p = 0;
		for (ObjectCoordinate oc : objectCoordinates) {
			pointIndex.put(oc, p);
			UnknownParameter<?>[] xyz = { oc.getX(), oc.getY(), oc.getZ() };
			for (int b = 0; b < 3; b++) {
				slotOf.put(xyz[b], this.slotParameters.size());
				this.slotParameters.add(xyz[b]);
				d.pointColumn[3 * p + b] = column(xyz[b]);
			}
			d.pointDatum[p] = (byte) (oc.isDatum() ? 1 : 0);
			p++;
		}

		// ---- cameras: interior orientation x0, y0, c (InteriorOrientation.java:60-82) ----------------------------------------
		int C = cameras.size();
		d.interiorColumn = new int[3 * C];
		int c = 0;
		for (Camera camera : cameras) {
			int i = 0;
			for (UnknownParameter<?> up : camera.getInteriorOrientation()) {
				slotOf.put(up, this.slotParameters.size());
				this.slotParameters.add(up);
				d.interiorColumn[3 * c + i++] = column(up);
			}
			c++;
		}
		// ---- distortion coefficients per camera, models in Type order (Camera.java:50), coefficients in insertion order ---------
		List<Integer> kind = new ArrayList<Integer>(), order = new ArrayList<Integer>(), col = new ArrayList<Integer>();
		d.cameraDistortionBegin = new int[C + 1];
		d.cameraR0 = new double[C];
		c = 0;
		for (Camera camera : cameras) {
			Collection<DistortionModel> models = camera.getDistortionModels();
			for (DistortionModel model : models) {
				if (model instanceof PolynomialDistortionModel)
					d.cameraR0[c] = ((PolynomialDistortionModel) model).getR0();     // radial, distance and Zernike models carry r0
				for (UnknownParameter<?> up : model) {
					int k;
					switch (up.getParameterType()) {
					case AFFINITY_AND_SHEAR_Cx:    k = AFFINITY_CX;   break;
					case AFFINITY_AND_SHEAR_Cy:    k = AFFINITY_CY;   break;
					case TANGENTIAL_DISTORTION_Bx: k = TANGENTIAL_BX; break;
					case TANGENTIAL_DISTORTION_By: k = TANGENTIAL_BY; break;
					case TANGENTIAL_POLYNOMIAL_B:  k = TANGENTIAL_BI; break;
					case RADIAL_POLYNOMIAL_A:      k = RADIAL_AI;     break;
					case ZERNIKE_POLYNOMIAL_X:     k = ZERNIKE_X;     break;
					case ZERNIKE_POLYNOMIAL_Y:     k = ZERNIKE_Y;     break;
					case ZERNIKE_POLYNOMIAL_Z:     k = ZERNIKE_Z;     break;
					default:                       k = DISTANCE_DI;   break;
					}
					slotOf.put(up, this.slotParameters.size());
					this.slotParameters.add(up);
					kind.add(k);
					order.add(up instanceof PolynomialCoefficient ? ((PolynomialCoefficient<?>) up).getOrder() : 0);
					col.add(column(up));
				}
			}
			d.cameraDistortionBegin[++c] = kind.size();
		}
		d.distortionKind = toIntArray(kind);
		d.distortionOrder = toIntArray(order);
		d.distortionColumn = toIntArray(col);

		// ---- images: exterior orientation X0, Y0, Z0, omega, phi, kappa (ExteriorOrientation.java:37-46) ----------------------
		Map<Image, Integer> imageIndex = new IdentityHashMap<Image, Integer>();
		List<Integer> imageCamera = new ArrayList<Integer>(), eoColumn = new ArrayList<Integer>();
		c = 0;
		for (Camera camera : cameras) {
			for (Image image : camera) {
				imageIndex.put(image, imageCamera.size());
				imageCamera.add(c);
				for (UnknownParameter<?> up : image.getExteriorOrientation()) {
					slotOf.put(up, this.slotParameters.size());
					this.slotParameters.add(up);
					eoColumn.add(column(up));
				}
			}
			c++;
		}
		d.imageCamera = toIntArray(imageCamera);
		d.exteriorColumn = toIntArray(eoColumn);

		// ---- image points, image-major (rows 2k, 2k+1 of BA:670-693); images with a joint dispersion become image blocks --------
		List<Integer> ipImage = new ArrayList<Integer>(), ipPoint = new ArrayList<Integer>(), blockBegin = new ArrayList<Integer>();
		List<Double> x = new ArrayList<Double>(), y = new ArrayList<Double>(), vx = new ArrayList<Double>(), vy = new ArrayList<Double>(),
				rho = new ArrayList<Double>();
		List<Long> blockOffset = new ArrayList<Long>();
		List<double[]> blockDispersions = new ArrayList<double[]>();
		long dispersionLength = 0;
		boolean anyBlock = false, blockOpen = false;
		for (Camera camera : cameras) {
			for (Image image : camera) {
				UpperSPDPackMatrix D = image.getDispersion();             // API addition, java/patch/Image.dispersion.patch
				boolean blk = D != null;
				if (blk) {
					if (anyBlock && !blockOpen)
						throw new IllegalArgumentException("Error, images with a joint dispersion must be consecutive.");
					if (!anyBlock)
						blockBegin.add(ipImage.size());
					anyBlock = blockOpen = true;
					int m = D.numRows();                                   // 2 x number of image coordinates, rows x0, y0, x1, y1, ...
					double[] full = new double[m * m];
					for (int r = 0; r < m; r++)
						for (int s = 0; s < m; s++)
							full[r * m + s] = D.get(Math.min(r, s), Math.max(r, s));
					blockOffset.add(dispersionLength);
					blockDispersions.add(full);
					dispersionLength += full.length;
				}
				else if (blockOpen)
					blockOpen = false;
				for (ImageCoordinate ic : image) {
					ipImage.add(imageIndex.get(image));
					ipPoint.add(pointIndex.get(ic.getObjectCoordinate()));
					x.add(ic.getX().getValue());
					y.add(ic.getY().getValue());
					vx.add(ic.getX().getVariance());
					vy.add(ic.getY().getVariance());
					rho.add(ic.getCorrelationCoefficientXY());
				}
				if (blk)
					blockBegin.add(ipImage.size());
			}
		}
		if (!anyBlock)
			blockBegin.add(0);
		d.imagePointImage = toIntArray(ipImage);
		d.imagePointPoint = toIntArray(ipPoint);
		d.x = toDoubleArray(x);
		d.y = toDoubleArray(y);
		d.varianceX = toDoubleArray(vx);
		d.varianceY = toDoubleArray(vy);
		d.rho = toDoubleArray(rho);
		d.blockBegin = toIntArray(blockBegin);
		d.blockDispersionOffset = toLongArray(blockOffset);
		d.blockDispersion = concat(blockDispersions, dispersionLength);

		// ---- scale bars (ScaleBar.java, PDF:210-283) -------------------------------------------------------------------------------
		int S = scaleBars.size(), s = 0;
		d.scaleBarA = new int[S];
		d.scaleBarB = new int[S];
		d.scaleBarLength = new double[S];
		d.scaleBarVariance = new double[S];
		for (ScaleBar scaleBar : scaleBars) {
			d.scaleBarA[s] = pointIndex.get(scaleBar.getObjectCoordinateA());
			d.scaleBarB[s] = pointIndex.get(scaleBar.getObjectCoordinateB());
			d.scaleBarLength[s] = scaleBar.getLength().getValue();
			d.scaleBarVariance[s] = scaleBar.getLength().getVariance();
			s++;
		}

		// ---- directly observed parameter groups (DOPG, PDF:447-473) ------------------------------------------------------------
		List<Integer> rowBegin = new ArrayList<Integer>(), slot = new ArrayList<Integer>();
		List<Double> obs = new ArrayList<Double>(), var = new ArrayList<Double>();
		List<Long> groupOffset = new ArrayList<Long>();
		List<double[]> groupDispersions = new ArrayList<double[]>();
		long groupLength = 0;
		rowBegin.add(0);
		for (DirectlyObservedParameterGroup group : observedParameterGroups) {
			for (ObservationParameter<? extends UnknownParameter<?>> op : group) {
				slot.add(slotOf.get(op.getReference()));
				obs.add(op.getValue());
				var.add(op.getVariance());
				this.directObservations.add(op);
			}
			rowBegin.add(slot.size());
			UpperSPDPackMatrix D = group.getDispersionMatrix();          // java/patch/DirectlyObservedParameterGroup.dispersion.patch
			if (D != null) {
				int m = D.numRows();
				double[] full = new double[m * m];
				for (int r = 0; r < m; r++)
					for (int t = 0; t < m; t++)
						full[r * m + t] = D.get(Math.min(r, t), Math.max(r, t));
				groupOffset.add(groupLength);
				groupDispersions.add(full);
				groupLength += full.length;
			}
			else
				groupOffset.add(-1L);                                       // diagonal: variances only (DOPG:71-78)
		}
		d.directRowBegin = toIntArray(rowBegin);
		d.directSlot = toIntArray(slot);
		d.directObservation = toDoubleArray(obs);
		d.directVariance = toDoubleArray(var);
		d.directDispersionOffset = toLongArray(groupOffset);
		d.directDispersion = concat(groupDispersions, groupLength);
		return d;
	}

	/** Parameter.getValue() of every slot: NativeNormalEquationEngine.setParameters(slotValues()) before the loop */
	public double[] slotValues() {
		double[] v = new double[this.slotParameters.size()];
		for (int i = 0; i < v.length; i++)
			v[i] = this.slotParameters.get(i).getValue();
		return v;
	}

	/** writes the adjusted values back into the object graph (what BA:450-462 does column by column in the pure-Java loop) */
	public void storeSlotValues(double[] v) {
		for (int i = 0; i < v.length; i++)
			this.slotParameters.get(i).setValue(v[i]);
	}

	private static int[] toIntArray(List<Integer> l) { int[] a = new int[l.size()]; for (int i = 0; i < a.length; i++) a[i] = l.get(i); return a; }
	private static long[] toLongArray(List<Long> l) { long[] a = new long[l.size()]; for (int i = 0; i < a.length; i++) a[i] = l.get(i); return a; }
	private static double[] toDoubleArray(List<Double> l) { double[] a = new double[l.size()]; for (int i = 0; i < a.length; i++) a[i] = l.get(i); return a; }
	private static double[] concat(List<double[]> parts, long length) {
		double[] a = new double[(int) length];
		int o = 0;
		for (double[] part : parts) { System.arraycopy(part, 0, a, o, part.length); o += part.length; }
		return a;
	}
}
