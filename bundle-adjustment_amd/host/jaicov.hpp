// jaicov.hpp -- C++17 mirror of JAICOV's object API for the bundle-adjustment path, on top of the C ABI
// (include/jaicov_neq.h).  The reference is Java (no JDK in this image), so the host side of the boundary is written in
// C++ with the reference's class / method names, argument meaning and error behaviour:
//
//   org.applied_geodesy.adjustment.bundle.{Camera, Image, ImageCoordinate, ObjectCoordinate, ScaleBar,
//       BundleAdjustment, parameter.*, camera.distortion.*, camera.orientation.*}, adjustment.defect.RankDefect,
//       adjustment.{EstimationStateType, EstimationType}
//
// What stays on the host (integer / control work, bit-exact with the reference):
//   prepareUnknownParameters  BundleAdjustment.java:667-782     detectRankDefect  BundleAdjustment.java:836-1042
//   centroidCoordinates       BundleAdjustment.java:115-201     estimateModel loop BundleAdjustment.java:203-387
//   updateModel (LM control)  BundleAdjustment.java:389-442
// What goes to the MI355X through the C ABI: createNormalEquation, applyPrecondition, MathExtension.solve, getOmega.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/jaicov_neq.h"

namespace jaicov::host {

// parameter/ParameterType.java:27-100 (ids kept)
enum class ParameterType : int {
    PRINCIPAL_POINT_X = 111, PRINCIPAL_POINT_Y = 112, PRINCIPAL_DISTANCE = 113, RADIAL_POLYNOMIAL_A = 121,
    TANGENTIAL_POLYNOMIAL_B = 131, TANGENTIAL_DISTORTION_Bx = 132, TANGENTIAL_DISTORTION_By = 133,
    AFFINITY_AND_SHEAR_Cx = 141, AFFINITY_AND_SHEAR_Cy = 142, DISTANCE_POLYNOMIAL_D = 151,
    ZERNIKE_POLYNOMIAL_X = 161, ZERNIKE_POLYNOMIAL_Y = 162, ZERNIKE_POLYNOMIAL_Z = 163,
    CAMERA_COORDINATE_X = 251, CAMERA_COORDINATE_Y = 252, CAMERA_COORDINATE_Z = 253,
    CAMERA_OMEGA = 261, CAMERA_PHI = 262, CAMERA_KAPPA = 263,
    OBJECT_COORDINATE_X = 311, OBJECT_COORDINATE_Y = 312, OBJECT_COORDINATE_Z = 313,
    IMAGE_COORDINATE_X = 411, IMAGE_COORDINATE_Y = 412, SCALE_BAR_LENGTH = 511
};

inline const char *parameterTypeName(ParameterType t) {   // ParameterType.name()
    switch (t) {
    case ParameterType::PRINCIPAL_POINT_X: return "PRINCIPAL_POINT_X";
    case ParameterType::PRINCIPAL_POINT_Y: return "PRINCIPAL_POINT_Y";
    case ParameterType::PRINCIPAL_DISTANCE: return "PRINCIPAL_DISTANCE";
    case ParameterType::RADIAL_POLYNOMIAL_A: return "RADIAL_POLYNOMIAL_A";
    case ParameterType::TANGENTIAL_POLYNOMIAL_B: return "TANGENTIAL_POLYNOMIAL_B";
    case ParameterType::TANGENTIAL_DISTORTION_Bx: return "TANGENTIAL_DISTORTION_Bx";
    case ParameterType::TANGENTIAL_DISTORTION_By: return "TANGENTIAL_DISTORTION_By";
    case ParameterType::AFFINITY_AND_SHEAR_Cx: return "AFFINITY_AND_SHEAR_Cx";
    case ParameterType::AFFINITY_AND_SHEAR_Cy: return "AFFINITY_AND_SHEAR_Cy";
    case ParameterType::DISTANCE_POLYNOMIAL_D: return "DISTANCE_POLYNOMIAL_D";
    case ParameterType::ZERNIKE_POLYNOMIAL_X: return "ZERNIKE_POLYNOMIAL_X";
    case ParameterType::ZERNIKE_POLYNOMIAL_Y: return "ZERNIKE_POLYNOMIAL_Y";
    case ParameterType::ZERNIKE_POLYNOMIAL_Z: return "ZERNIKE_POLYNOMIAL_Z";
    case ParameterType::CAMERA_COORDINATE_X: return "CAMERA_COORDINATE_X";
    case ParameterType::CAMERA_COORDINATE_Y: return "CAMERA_COORDINATE_Y";
    case ParameterType::CAMERA_COORDINATE_Z: return "CAMERA_COORDINATE_Z";
    case ParameterType::CAMERA_OMEGA: return "CAMERA_OMEGA";
    case ParameterType::CAMERA_PHI: return "CAMERA_PHI";
    case ParameterType::CAMERA_KAPPA: return "CAMERA_KAPPA";
    case ParameterType::OBJECT_COORDINATE_X: return "OBJECT_COORDINATE_X";
    case ParameterType::OBJECT_COORDINATE_Y: return "OBJECT_COORDINATE_Y";
    case ParameterType::OBJECT_COORDINATE_Z: return "OBJECT_COORDINATE_Z";
    case ParameterType::IMAGE_COORDINATE_X: return "IMAGE_COORDINATE_X";
    case ParameterType::IMAGE_COORDINATE_Y: return "IMAGE_COORDINATE_Y";
    case ParameterType::SCALE_BAR_LENGTH: return "SCALE_BAR_LENGTH";
    }
    return "?";
}

// adjustment/EstimationStateType.java:25-42
enum class EstimationStateType : int {
    ERROR_FREE_ESTIMATION = 1, BUSY = 0, INTERRUPT = -1, SINGULAR_MATRIX = -2, ROBUST_ESTIMATION_FAILED = -3,
    NO_CONVERGENCE = -4, NOT_INITIALISED = -5, EXPORT_ADJUSTMENT_RESULTS_FAILED = -6, OUT_OF_MEMORY = -7
};
enum class EstimationType { L2NORM, SIMULATION };                       // adjustment/EstimationType.java
enum class MatrixInversion { NONE, FULL, PRE_ELIMINATION, REDUCED };    // BundleAdjustment.java:65-70

constexpr int COLUMN_NOT_SET = -1;          // UnknownParameter.java:28
constexpr int COLUMN_FIXED = INT_MAX;       // UnknownParameter.java:27 (Integer.MAX_VALUE)

class ObjectCoordinate;
class Image;
class Camera;

// parameter/UnknownParameter.java (+ PolynomialCoefficient.order)
class UnknownParameter {
public:
    UnknownParameter(ParameterType t, void *ref, int order = -1) : type_(t), ref_(ref), order_(order) {}
    ParameterType getParameterType() const { return type_; }
    double getValue() const { return value_; }
    void setValue(double v) { value_ = v; }
    int getColumn() const { return column_; }
    void setColumn(int c) { column_ = c; }
    int getOrder() const { return order_; }
    void *getReference() const { return ref_; }
    int slot = -1;      // flattening: index into the engine's value vector
private:
    ParameterType type_;
    void *ref_;
    int order_;
    double value_ = 0.0;
    int column_ = COLUMN_NOT_SET;
};

// parameter/ObservationParameter.java
class ObservationParameter {
public:
    ObservationParameter(ParameterType t, void *ref) : type_(t), ref_(ref) {}
    explicit ObservationParameter(UnknownParameter *p) : type_(p->getParameterType()), ref_(p), unknown_(p) { value_ = p->getValue(); }
    ParameterType getParameterType() const { return type_; }
    double getValue() const { return value_; }
    void setValue(double v) { value_ = v; }
    double getVariance() const { return variance_; }
    void setVariance(double v) {
        if (!(v > 0)) throw std::invalid_argument("Error, variance must be positive");   // ObservationParameter.java:55-57
        variance_ = v;
    }
    int getRow() const { return row_; }
    void setRow(int r) { row_ = r; }
    UnknownParameter *getReferenceParameter() const { return unknown_; }
private:
    ParameterType type_;
    void *ref_;
    UnknownParameter *unknown_ = nullptr;
    double value_ = 0.0, variance_ = 0.0;
    int row_ = -1;
};

// ObjectCoordinate.java
class ObjectCoordinate {
public:
    ObjectCoordinate(const ObjectCoordinate &) = delete;
    ObjectCoordinate &operator=(const ObjectCoordinate &) = delete;
    ObjectCoordinate(std::string name, double x, double y, double z)
        : name_(std::move(name)), x_(ParameterType::OBJECT_COORDINATE_X, this), y_(ParameterType::OBJECT_COORDINATE_Y, this),
          z_(ParameterType::OBJECT_COORDINATE_Z, this) {
        x_.setValue(x); y_.setValue(y); z_.setValue(z);
    }
    const std::string &getName() const { return name_; }
    UnknownParameter &getX() { return x_; }
    UnknownParameter &getY() { return y_; }
    UnknownParameter &getZ() { return z_; }
    bool isDatum() const { return datum_; }
    void setDatum(bool d) { datum_ = d; }
    int numberOfImages() const { return n_images_; }
    void addImage() { n_images_++; }
    int index = -1;     // flattening
private:
    std::string name_;
    bool datum_ = true;
    UnknownParameter x_, y_, z_;
    int n_images_ = 0;
};

// ScaleBar.java
class ScaleBar {
public:
    ScaleBar(const ScaleBar &) = delete;
    ScaleBar &operator=(const ScaleBar &) = delete;
    ScaleBar(ObjectCoordinate *a, ObjectCoordinate *b, double value, double sigma)
        : a_(a), b_(b), length_(ParameterType::SCALE_BAR_LENGTH, this) {
        length_.setValue(value);
        length_.setVariance(sigma * sigma);
    }
    ObservationParameter &getLength() { return length_; }
    ObjectCoordinate *getObjectCoordinateA() const { return a_; }
    ObjectCoordinate *getObjectCoordinateB() const { return b_; }
private:
    ObjectCoordinate *a_, *b_;
    ObservationParameter length_;
};

// camera/distortion/DistortionModel.java:29-37
class DistortionModel {
public:
    DistortionModel(const DistortionModel &) = delete;
    DistortionModel &operator=(const DistortionModel &) = delete;
    // DistortionModel.java:29-37 (ordinal = application order, Camera.java:50)
    enum class Type { AFFINITY_AND_SHEAR = 0, TANGENTIAL_DISTORTION = 1, RADIAL_DISTORTION = 2, DISTANCE_DISTORTION = 3,
                      ZERNIKE_X = 4, ZERNIKE_Y = 5, ZERNIKE_GRADIENT = 6 };
    DistortionModel(Type t, double r0) : type_(t), r0_(r0) {
        if (t == Type::AFFINITY_AND_SHEAR) {          // AffinityShearDistortionModel.java:34-40: Cx, Cy fixed by default
            params_.emplace_back(new UnknownParameter(ParameterType::AFFINITY_AND_SHEAR_Cx, this));
            params_.emplace_back(new UnknownParameter(ParameterType::AFFINITY_AND_SHEAR_Cy, this));
            params_[0]->setColumn(COLUMN_FIXED); params_[1]->setColumn(COLUMN_FIXED);
        } else if (t == Type::TANGENTIAL_DISTORTION) { // TangentialDistortionModel.java:34-42: Bx, By fixed by default
            params_.emplace_back(new UnknownParameter(ParameterType::TANGENTIAL_DISTORTION_Bx, this));
            params_.emplace_back(new UnknownParameter(ParameterType::TANGENTIAL_DISTORTION_By, this));
            params_[0]->setColumn(COLUMN_FIXED); params_[1]->setColumn(COLUMN_FIXED);
        }
    }
    Type getType() const { return type_; }
    double getR0() const { return r0_; }
    // PolynomialDistortionModel.add(order): insertion order is iteration order (LinkedHashMap, PolynomialDistortionModel.java:34,60-62)
    UnknownParameter *add(int order) {
        if (type_ == Type::AFFINITY_AND_SHEAR) throw std::invalid_argument("affinity model has no polynomial coefficients");
        if (order <= 0) throw std::invalid_argument("Error, polynomial coefficient order must be a real positive integer.");
        for (auto &p : params_)
            if (p->getOrder() == order) throw std::invalid_argument("Error, polynomial coefficient order already exists.");
        ParameterType pt = type_ == Type::TANGENTIAL_DISTORTION ? ParameterType::TANGENTIAL_POLYNOMIAL_B
                           : type_ == Type::RADIAL_DISTORTION   ? ParameterType::RADIAL_POLYNOMIAL_A
                           : type_ == Type::DISTANCE_DISTORTION ? ParameterType::DISTANCE_POLYNOMIAL_D
                           : type_ == Type::ZERNIKE_X           ? ParameterType::ZERNIKE_POLYNOMIAL_X      // ZernikeDistortionModel.java:36-38
                           : type_ == Type::ZERNIKE_Y           ? ParameterType::ZERNIKE_POLYNOMIAL_Y      // :47-49
                                                                : ParameterType::ZERNIKE_POLYNOMIAL_Z;     // :58-60
        params_.emplace_back(new UnknownParameter(pt, this, order));
        return params_.back().get();
    }
    UnknownParameter *get(int order) {
        for (auto &p : params_)
            if (p->getOrder() == order) return p.get();
        return nullptr;
    }
    UnknownParameter *getCx() { return type_ == Type::AFFINITY_AND_SHEAR ? params_[0].get() : nullptr; }
    UnknownParameter *getCy() { return type_ == Type::AFFINITY_AND_SHEAR ? params_[1].get() : nullptr; }
    UnknownParameter *getBx() { return type_ == Type::TANGENTIAL_DISTORTION ? params_[0].get() : nullptr; }
    UnknownParameter *getBy() { return type_ == Type::TANGENTIAL_DISTORTION ? params_[1].get() : nullptr; }
    std::vector<std::unique_ptr<UnknownParameter>> &parameters() { return params_; }
private:
    Type type_;
    double r0_;
    std::vector<std::unique_ptr<UnknownParameter>> params_;
};

// camera/orientation/InteriorOrientation.java:60-82 (iteration order x0, y0, c)
class InteriorOrientation {
public:
    InteriorOrientation(const InteriorOrientation &) = delete;
    InteriorOrientation &operator=(const InteriorOrientation &) = delete;
    InteriorOrientation()
        : x0_(ParameterType::PRINCIPAL_POINT_X, this), y0_(ParameterType::PRINCIPAL_POINT_Y, this), c_(ParameterType::PRINCIPAL_DISTANCE, this) {}
    UnknownParameter &getPrinciplePointX() { return x0_; }
    UnknownParameter &getPrinciplePointY() { return y0_; }
    UnknownParameter &getPrincipleDistance() { return c_; }
    UnknownParameter *at(int i) { return i == 0 ? &x0_ : (i == 1 ? &y0_ : &c_); }
private:
    UnknownParameter x0_, y0_, c_;
};

// camera/orientation/ExteriorOrientation.java:37-46 (X0, Y0, Z0, omega, phi, kappa)
class ExteriorOrientation {
public:
    ExteriorOrientation(const ExteriorOrientation &) = delete;
    ExteriorOrientation &operator=(const ExteriorOrientation &) = delete;
    ExteriorOrientation()
        : p_{UnknownParameter(ParameterType::CAMERA_COORDINATE_X, this), UnknownParameter(ParameterType::CAMERA_COORDINATE_Y, this),
             UnknownParameter(ParameterType::CAMERA_COORDINATE_Z, this), UnknownParameter(ParameterType::CAMERA_OMEGA, this),
             UnknownParameter(ParameterType::CAMERA_PHI, this), UnknownParameter(ParameterType::CAMERA_KAPPA, this)} {}
    UnknownParameter &get(ParameterType t) {
        for (auto &q : p_)
            if (q.getParameterType() == t) return q;
        throw std::invalid_argument("not an exterior orientation parameter");
    }
    UnknownParameter *at(int i) { return &p_[i]; }
private:
    UnknownParameter p_[6];
};

// camera/ImageCoordinate.java
class ImageCoordinate {
public:
    ImageCoordinate(const ImageCoordinate &) = delete;
    ImageCoordinate &operator=(const ImageCoordinate &) = delete;
    ImageCoordinate(ObjectCoordinate *oc, Image *img, double xp, double yp, double sx, double sy, double rho)
        : oc_(oc), img_(img), rho_(rho), x_(ParameterType::IMAGE_COORDINATE_X, this), y_(ParameterType::IMAGE_COORDINATE_Y, this) {
        if (std::fabs(rho) >= 1) throw std::invalid_argument("Error, correlation coefficient rho(x,y) must be in the open interval (-1 1)");
        x_.setValue(xp); y_.setValue(yp);
        x_.setVariance(sx * sx); y_.setVariance(sy * sy);
    }
    ObjectCoordinate *getObjectCoordinate() const { return oc_; }
    ObservationParameter &getX() { return x_; }
    ObservationParameter &getY() { return y_; }
    double getCorrelationCoefficientXY() const { return rho_; }
    Image *getReference() const { return img_; }
private:
    ObjectCoordinate *oc_;
    Image *img_;
    double rho_;
    ObservationParameter x_, y_;
};

// camera/Image.java (LinkedHashMap<ObjectCoordinate, ImageCoordinate>: insertion order, one observation per point)
class Image {
public:
    Image(const Image &) = delete;
    Image &operator=(const Image &) = delete;
    Image(long id, Camera *cam) : id_(id), cam_(cam) {}
    long getId() const { return id_; }
    Camera *getReference() const { return cam_; }
    ExteriorOrientation &getExteriorOrientation() { return eo_; }
    ImageCoordinate *add(ObjectCoordinate *oc, double xp, double yp, double sx, double sy, double rho = 0.0) {
        auto it = index_.find(oc);
        if (it != index_.end()) return coords_[it->second].get();       // Image.java:53-54
        coords_.emplace_back(new ImageCoordinate(oc, this, xp, yp, sx, sy, rho));
        index_[oc] = coords_.size() - 1;
        oc->addImage();
        return coords_.back().get();
    }
    int getNumberOfImageCoordinates() const { return (int)coords_.size(); }
    std::vector<std::unique_ptr<ImageCoordinate>> &coordinates() { return coords_; }
    // Joint, fully populated dispersion of ALL image coordinates of this image, rows x0,y0,x1,y1,... in insertion order
    // (the largest W the a10 contract can express; new relative to the reference).  Row-major (2m x 2m).
    void setDispersion(std::vector<double> D) {
        const size_t m = 2 * coords_.size();
        if (D.size() != m * m) throw std::invalid_argument("Error, number of observations and number of rows/columns in dispersion matrix are unequal");
        for (size_t i = 0; i < coords_.size(); i++) {                   // DOPG:53-55: variances from the diagonal
            coords_[i]->getX().setVariance(D[(2 * i) * m + 2 * i]);
            coords_[i]->getY().setVariance(D[(2 * i + 1) * m + 2 * i + 1]);
        }
        disp_ = std::move(D);
    }
    const std::vector<double> &dispersion() const { return disp_; }
    int index = -1;
private:
    long id_;
    Camera *cam_;
    ExteriorOrientation eo_;
    std::vector<std::unique_ptr<ImageCoordinate>> coords_;
    std::unordered_map<ObjectCoordinate *, size_t> index_;
    std::vector<double> disp_;
};

// camera/Camera.java:45-83: distortion models sorted by Type ordinal, images in insertion order
class Camera {
public:
    Camera(const Camera &) = delete;
    Camera &operator=(const Camera &) = delete;
    Camera(long id, double r0, std::vector<DistortionModel::Type> types) : id_(id) {
        std::sort(types.begin(), types.end());
        for (size_t i = 0; i < types.size(); i++) {
            if (i > 0 && types[i] == types[i - 1]) throw std::invalid_argument("Error, duplicate type of distortion model detected.");
            models_.emplace_back(new DistortionModel(types[i], r0));
        }
    }
    long getId() const { return id_; }
    InteriorOrientation &getInteriorOrientation() { return io_; }
    Image *add(long imageId) {
        auto it = index_.find(imageId);
        if (it != index_.end()) return images_[it->second].get();
        images_.emplace_back(new Image(imageId, this));
        index_[imageId] = images_.size() - 1;
        return images_.back().get();
    }
    int getNumberOfImages() const { return (int)images_.size(); }
    DistortionModel *getDistortionModel(DistortionModel::Type t) {
        for (auto &m : models_)
            if (m->getType() == t) return m.get();
        return nullptr;
    }
    std::vector<std::unique_ptr<DistortionModel>> &getDistortionModels() { return models_; }
    std::vector<std::unique_ptr<Image>> &images() { return images_; }
    int index = -1;
private:
    long id_;
    InteriorOrientation io_;
    std::vector<std::unique_ptr<DistortionModel>> models_;
    std::vector<std::unique_ptr<Image>> images_;
    std::unordered_map<long, size_t> index_;
};

// parameter/DirectlyObservedParameterGroup.java
class DirectlyObservedParameterGroup {
public:
    DirectlyObservedParameterGroup(const DirectlyObservedParameterGroup &) = delete;
    DirectlyObservedParameterGroup &operator=(const DirectlyObservedParameterGroup &) = delete;
    explicit DirectlyObservedParameterGroup(std::vector<ObservationParameter *> obs) : obs_(std::move(obs)) {
        std::unordered_set<ObservationParameter *> u(obs_.begin(), obs_.end());
        if (u.size() != obs_.size()) throw std::invalid_argument("Error, array contains duplicate observations.");   // DOPG:44-45
    }
    // dispersion: row-major m x m (the reference takes an UpperSPDPackMatrix, DOPG:48-61)
    DirectlyObservedParameterGroup(std::vector<double> dispersion, std::vector<ObservationParameter *> obs)
        : DirectlyObservedParameterGroup(std::move(obs)) {
        const size_t m = obs_.size();
        if (dispersion.size() != m * m) throw std::invalid_argument("Error, number of observations and number of rows/columns in dispersion matrix are unequal");
        for (size_t r = 0; r < m; r++) obs_[r]->setVariance(dispersion[r * m + r]);
        disp_ = std::move(dispersion);
    }
    bool hasFullyPopulatedWeightMatrix() const { return !disp_.empty(); }
    int getNumberOfParameters() const { return (int)obs_.size(); }
    std::vector<ObservationParameter *> &observations() { return obs_; }
    const std::vector<double> &dispersion() const { return disp_; }
private:
    std::vector<ObservationParameter *> obs_;
    std::vector<double> disp_;
};

// defect/RankDefect.java
class RankDefect {
public:
    enum class DefectType { NOT_SET, FREE, FIXED };
    void reset() { tx = ty = tz = rx = ry = rz = mxyz = DefectType::NOT_SET; }
    static DefectType norm(DefectType d) { return d == DefectType::FIXED ? DefectType::FIXED : DefectType::FREE; }
    void setScale(DefectType d) { mxyz = norm(d); }
    void setRotationX(DefectType d) { rx = norm(d); }
    void setRotationY(DefectType d) { ry = norm(d); }
    void setRotationZ(DefectType d) { rz = norm(d); }
    void setTranslationX(DefectType d) { tx = norm(d); }
    void setTranslationY(DefectType d) { ty = norm(d); }
    void setTranslationZ(DefectType d) { tz = norm(d); }
    bool estimateScale() const { return mxyz == DefectType::FREE; }
    bool estimateRotationX() const { return rx == DefectType::FREE; }
    bool estimateRotationY() const { return ry == DefectType::FREE; }
    bool estimateRotationZ() const { return rz == DefectType::FREE; }
    bool estimateTranslationX() const { return tx == DefectType::FREE; }
    bool estimateTranslationY() const { return ty == DefectType::FREE; }
    bool estimateTranslationZ() const { return tz == DefectType::FREE; }
    int getDefect() const {
        return estimateScale() + estimateRotationX() + estimateRotationY() + estimateRotationZ() + estimateTranslationX() +
               estimateTranslationY() + estimateTranslationZ();
    }
    bool allFixed() const { return getDefect() == 0; }
    int flags() const {
        return (estimateTranslationX() ? JAICOV_DATUM_TX : 0) | (estimateTranslationY() ? JAICOV_DATUM_TY : 0) |
               (estimateTranslationZ() ? JAICOV_DATUM_TZ : 0) | (estimateRotationX() ? JAICOV_DATUM_RX : 0) |
               (estimateRotationY() ? JAICOV_DATUM_RY : 0) | (estimateRotationZ() ? JAICOV_DATUM_RZ : 0) |
               (estimateScale() ? JAICOV_DATUM_SCALE : 0);
    }
private:
    DefectType rx = DefectType::NOT_SET, ry = DefectType::NOT_SET, rz = DefectType::NOT_SET, tx = DefectType::NOT_SET,
               ty = DefectType::NOT_SET, tz = DefectType::NOT_SET, mxyz = DefectType::NOT_SET;
};

class BundleAdjustment;
class AdjustmentResultWritable {                                   // util/io/writer/AdjustmentResultWritable.java:36
public:
    virtual ~AdjustmentResultWritable() = default;
    virtual void exportResults(BundleAdjustment &bundleAdjustment) = 0;      // `export` is a C++ keyword
    virtual std::string toString() const = 0;
};

// BundleAdjustment.java
class BundleAdjustment {
public:
    using Listener = std::function<void(const std::string &name, double oldValue, double newValue)>;   // PropertyChangeListener

    BundleAdjustment() = default;
    ~BundleAdjustment() { if (engine_) jaicov_neq_destroy(engine_); }
    BundleAdjustment(const BundleAdjustment &) = delete;

    void add(Camera *c) { cameras_.push_back(c); }                                      // BA:652-655
    void add(ScaleBar *s) { if (std::find(scaleBars_.begin(), scaleBars_.end(), s) == scaleBars_.end()) scaleBars_.push_back(s); }
    void add(DirectlyObservedParameterGroup *g) { if (std::find(groups_.begin(), groups_.end(), g) == groups_.end()) groups_.push_back(g); }
    void addPropertyChangeListener(Listener l) { listeners_.push_back(std::move(l)); }

    void setEstimationType(EstimationType t) { estimationType_ = t; }                   // BA:1132
    void setInvertNormalEquation(MatrixInversion m) { inversion_ = m; }                 // BA:1146
    MatrixInversion getInvertNormalEquation() const { return inversion_; }
    void useCentroidedCoordinates(bool b) { centroided_ = b; }                          // BA:1181
    void centroidCoordinates(bool invert);    // BA:115-201 (private in the reference; public here so that the parity tests can call it)
    void applyAposterioriVarianceOfUnitWeight(bool b) { applyAposteriori_ = b; }        // BA:1185
    void setLevenbergMarquardtDampingValue(double l) { damping_ = std::fabs(l); }       // BA:1189
    double getLevenbergMarquardtDampingValue() const { return damping_; }
    void setMaximalNumberOfIterations(int n) { maxIter_ = n; }
    void setDevice(int d) { device_ = d; }
    void setAdjustmentResultWriter(AdjustmentResultWritable *w) { resultWriter_ = w; }   // BA:1123 (not owned)
    void interrupt() { interrupt_ = true; }                                             // BA:1455

    int getNumberOfObservations() const { return numberOfObservations_; }
    int getNumberOfUnknownParameters() const { return numberOfUnknownParameters_; }
    int getNumberOfDatumConditions() const { return rankDefect_.getDefect(); }
    int getDegreeOfFreedom() const { return numberOfObservations_ - numberOfUnknownParameters_ + rankDefect_.getDefect(); }   // BA:1080
    double getVarianceFactorApriori() const { return sigma2apriori_; }
    double getVarianceFactorAposteriori() const {                                       // BA:1090-1093
        const int dof = getDegreeOfFreedom();
        return dof > 0 && omega_ > 0 && estimationType_ != EstimationType::SIMULATION && applyAposteriori_ ? std::fabs(omega_ / (double)dof)
                                                                                                            : sigma2apriori_;
    }
    double getOmega() const { return omega_; }
    int getIterations() const { return iterationStep_; }
    const RankDefect &getRankDefect() const { return rankDefect_; }
    std::vector<ObjectCoordinate *> &getObjectCoordinates() { return objectCoordinates_; }
    std::vector<Camera *> &getCameras() { return cameras_; }
    // packed UPLO='U' cofactor matrix, order u + d (UpperSymmPackMatrix.getData()); empty for MatrixInversion.NONE (BA:1177)
    // The host copy is made on first use: the writers and most callers need a few hundred entries, which cofactorSub
    // gathers on the device, not the 1.3 GB packed array.
    const std::vector<double> &getCofactorMatrix() const { fetchCofactor(); return Qxx_; }
    double cofactor(int r, int c) const {
        fetchCofactor();
        if (r > c) std::swap(r, c);
        return Qxx_.at((size_t)r + (size_t)c * (c + 1) / 2);
    }
    // what the writers test: cofactor != null && numRows >= u + d (MatlabResultWriter.java:72, DefaultResultWriter.java:128)
    bool hasCofactorMatrix() const { return inversion_ != MatrixInversion::NONE && (qxxOnDevice_ || !Qxx_.empty()); }
    // scale * Qxx[idx, idx], row-major k x k, gathered on the device (jaicov_neq_get_dispersion_sub)
    std::vector<double> cofactorSub(const std::vector<int32_t> &idx, double scale = 1.0) const;
    const std::string &lastError() const { return lastError_; }

    // ---- index contract -----------------------------------------------------------------------------------
    void prepareUnknownParameters();          // BA:667-782
    void flatten();                           // object graph -> jaicov_problem_desc arrays
    EstimationStateType estimateModel();      // BA:203-387

    // flattened arrays (kept public for the parity tests)
    struct Flat {
        std::vector<int32_t> point_col, io_col, cam_dist_begin, dist_kind, dist_order, dist_col, image_camera, eo_col, ip_image,
            ip_point, blk_ip_begin, sb_a, sb_b, dg_row_begin, dg_slot;
        std::vector<uint8_t> point_datum;
        std::vector<double> cam_r0, ip_x, ip_y, ip_var_x, ip_var_y, ip_rho, blk_disp, sb_len, sb_var, dg_obs, dg_var, dg_disp, values;
        std::vector<int64_t> blk_disp_offset, dg_disp_offset;
        std::vector<UnknownParameter *> slot_param;
    } flat;

private:
    void fire(const std::string &n, double a, double b) { for (auto &l : listeners_) l(n, a, b); }
    void addUnknownParameter(UnknownParameter *p) {                                    // BA:645-650
        if (p->getColumn() == COLUMN_NOT_SET && !unknownSet_.count(p)) {
            p->setColumn(numberOfUnknownParameters_++);
            unknownSet_.insert(p);
            unknownParameters_.push_back(p);
        }
    }
    void addObjectCoordinate(ObjectCoordinate *oc) {
        if (ocSet_.insert(oc).second) objectCoordinates_.push_back(oc);
    }
    void noteVariance(double v) { sigma2apriori_ = std::min(sigma2apriori_, v); }      // BA:641
    void detectRankDefect();                  // BA:836-1042
    void pushValues();                        // objects -> flat.values
    void pullValues(const std::vector<double> &v);

    std::vector<Camera *> cameras_;
    std::vector<ScaleBar *> scaleBars_;
    std::vector<DirectlyObservedParameterGroup *> groups_;
    std::vector<ObjectCoordinate *> objectCoordinates_;
    std::unordered_set<ObjectCoordinate *> ocSet_;
    std::vector<UnknownParameter *> unknownParameters_;
    std::unordered_set<UnknownParameter *> unknownSet_;
    std::vector<Listener> listeners_;
    RankDefect rankDefect_;
    EstimationType estimationType_ = EstimationType::L2NORM;
    MatrixInversion inversion_ = MatrixInversion::FULL;
    int maxIter_ = 5000, iterationStep_ = 0, numberOfUnknownParameters_ = 0, numberOfObservations_ = 0, device_ = 0;
    bool interrupt_ = false, applyAposteriori_ = true, centroided_ = true, prepared_ = false;
    double damping_ = 0.0, omega_ = 0.0, sigma2apriori_ = 1.0, maxAbsDx_ = 0.0;
    double centroid_[3] = {0, 0, 0};
    mutable std::vector<double> Qxx_;
    mutable bool qxxOnDevice_ = false;    // an inverse is on the device and Qxx_ has not been fetched yet
    void fetchCofactor() const;
    AdjustmentResultWritable *resultWriter_ = nullptr;
    std::string lastError_;
    jaicov_engine *engine_ = nullptr;
};

}  // namespace jaicov::host
