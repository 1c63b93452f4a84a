// jaicov.cpp -- host-side driver logic of the JAICOV mirror (see jaicov.hpp).  Control and integer work only; all
// per-iteration arithmetic goes through the C ABI (include/jaicov_neq.h) to the MI355X.
#include "jaicov.hpp"

#include <cstring>

namespace jaicov::host {

static const double SQRT_EPS = std::sqrt(1.1102230246251565e-16);   // sqrt(Constant.EPS), BundleAdjustment.java:77

// ---------------------------------------------------------------------------------------------------------------
// BundleAdjustment.prepareUnknownParameters (BundleAdjustment.java:667-782), loop for loop
// ---------------------------------------------------------------------------------------------------------------
void BundleAdjustment::prepareUnknownParameters() {
    if (prepared_) throw std::logic_error("estimateModel() is single-shot (BundleAdjustment.java:80-83,776-781)");
    prepared_ = true;
    for (Camera *camera : cameras_)
        for (auto &image : camera->images())
            for (auto &ic : image->coordinates()) {
                ic->getX().setRow(numberOfObservations_++);
                ic->getY().setRow(numberOfObservations_++);
                ObjectCoordinate *oc = ic->getObjectCoordinate();
                addObjectCoordinate(oc);
                addUnknownParameter(&oc->getX());
                addUnknownParameter(&oc->getY());
                addUnknownParameter(&oc->getZ());
                noteVariance(ic->getX().getVariance());
                noteVariance(ic->getY().getVariance());
            }
    for (Camera *camera : cameras_) {
        for (int i = 0; i < 3; i++) addUnknownParameter(camera->getInteriorOrientation().at(i));
        for (auto &model : camera->getDistortionModels())
            for (auto &p : model->parameters()) addUnknownParameter(p.get());
    }
    for (Camera *camera : cameras_)
        for (auto &image : camera->images())
            for (int i = 0; i < 6; i++) addUnknownParameter(image->getExteriorOrientation().at(i));
    for (ScaleBar *sb : scaleBars_) {
        sb->getLength().setRow(numberOfObservations_++);
        ObjectCoordinate *a = sb->getObjectCoordinateA(), *b = sb->getObjectCoordinateB();
        addObjectCoordinate(a);
        addObjectCoordinate(b);
        addUnknownParameter(&a->getX()); addUnknownParameter(&a->getY()); addUnknownParameter(&a->getZ());
        addUnknownParameter(&b->getX()); addUnknownParameter(&b->getY()); addUnknownParameter(&b->getZ());
        noteVariance(sb->getLength().getVariance());
    }
    for (DirectlyObservedParameterGroup *g : groups_) {
        for (ObservationParameter *op : g->observations()) {
            UnknownParameter *up = op->getReferenceParameter();
            switch (up->getParameterType()) {
            case ParameterType::OBJECT_COORDINATE_X:
            case ParameterType::OBJECT_COORDINATE_Y:
            case ParameterType::OBJECT_COORDINATE_Z:
                addObjectCoordinate(static_cast<ObjectCoordinate *>(up->getReference()));
                break;
            default:
                break;
            }
            addUnknownParameter(up);
            op->setRow(numberOfObservations_++);
            noteVariance(op->getVariance());
        }
    }
    detectRankDefect();
    const int d = rankDefect_.getDefect();
    if (d > 0)
        for (UnknownParameter *p : unknownParameters_) p->setColumn(p->getColumn() + d);
}

// BundleAdjustment.detectRankDefect (BundleAdjustment.java:836-1042), literal
void BundleAdjustment::detectRankDefect() {
    using DT = RankDefect::DefectType;
    const bool hasScaleBars = !scaleBars_.empty();
    RankDefect &rd = rankDefect_;
    rd.reset();
    rd.setTranslationX(DT::FREE); rd.setTranslationY(DT::FREE); rd.setTranslationZ(DT::FREE);
    rd.setRotationX(DT::FREE); rd.setRotationY(DT::FREE); rd.setRotationZ(DT::FREE);
    rd.setScale(hasScaleBars ? DT::FIXED : DT::FREE);
    int kx = 0, ky = 0, kz = 0;
    if (rd.allFixed()) return;
    for (auto *g : groups_) {
        for (ObservationParameter *op : g->observations()) {
            switch (op->getParameterType()) {
            case ParameterType::CAMERA_OMEGA: rd.setRotationX(DT::FIXED); break;
            case ParameterType::CAMERA_PHI: rd.setRotationY(DT::FIXED); break;
            case ParameterType::CAMERA_KAPPA: rd.setRotationZ(DT::FIXED); break;
            default: break;
            }
            if (!rd.estimateRotationX() && !rd.estimateRotationY() && !rd.estimateRotationZ()) break;
        }
    }
    auto theory = [&]() {
        if (rd.estimateTranslationX() && kx > 0) rd.setTranslationX(DT::FIXED);
        if (rd.estimateTranslationY() && ky > 0) rd.setTranslationY(DT::FIXED);
        if (rd.estimateTranslationZ() && kz > 0) rd.setTranslationZ(DT::FIXED);
        if (!hasScaleBars && (kx >= 2 || ky >= 2 || kz >= 2)) rd.setScale(DT::FIXED);
        if (rd.estimateRotationX() && ky >= 2 && kz >= 2) rd.setRotationX(DT::FIXED);
        if (rd.estimateRotationY() && kx >= 2 && kz >= 2) rd.setRotationY(DT::FIXED);
        if (rd.estimateRotationZ() && kx >= 2 && ky >= 2) rd.setRotationZ(DT::FIXED);
        if (kx > 0 && ky > 0 && kz > 0 && ((hasScaleBars && kx + ky + kz >= 6) || (!hasScaleBars && kx + ky + kz >= 7))) {
            rd.setRotationX(DT::FIXED); rd.setRotationY(DT::FIXED); rd.setRotationZ(DT::FIXED);
        }
    };
    for (auto *g : groups_) {
        for (ObservationParameter *op : g->observations()) {
            switch (op->getParameterType()) {
            case ParameterType::CAMERA_COORDINATE_X: case ParameterType::OBJECT_COORDINATE_X: kx++; break;
            case ParameterType::CAMERA_COORDINATE_Y: case ParameterType::OBJECT_COORDINATE_Y: ky++; break;
            case ParameterType::CAMERA_COORDINATE_Z: case ParameterType::OBJECT_COORDINATE_Z: kz++; break;
            case ParameterType::CAMERA_OMEGA: rd.setRotationX(DT::FIXED); break;
            case ParameterType::CAMERA_PHI: rd.setRotationY(DT::FIXED); break;
            case ParameterType::CAMERA_KAPPA: rd.setRotationZ(DT::FIXED); break;
            default: break;
            }
            theory();
            if (rd.allFixed()) break;
        }
    }
    for (ObjectCoordinate *oc : objectCoordinates_) {
        kx += oc->getX().getColumn() == COLUMN_FIXED ? 1 : 0;
        ky += oc->getY().getColumn() == COLUMN_FIXED ? 1 : 0;
        kz += oc->getZ().getColumn() == COLUMN_FIXED ? 1 : 0;
        theory();
        if (rd.allFixed()) break;
    }
    if (rd.allFixed()) return;
    for (Camera *camera : cameras_) {
        for (auto &image : camera->images()) {
            ExteriorOrientation &eo = image->getExteriorOrientation();
            if (rd.estimateRotationX() && eo.get(ParameterType::CAMERA_OMEGA).getColumn() == COLUMN_FIXED) rd.setRotationX(DT::FIXED);
            if (rd.estimateRotationY() && eo.get(ParameterType::CAMERA_PHI).getColumn() == COLUMN_FIXED) rd.setRotationY(DT::FIXED);
            if (rd.estimateRotationZ() && eo.get(ParameterType::CAMERA_KAPPA).getColumn() == COLUMN_FIXED) rd.setRotationZ(DT::FIXED);
            kx += eo.get(ParameterType::CAMERA_COORDINATE_X).getColumn() == COLUMN_FIXED ? 1 : 0;
            ky += eo.get(ParameterType::CAMERA_COORDINATE_Y).getColumn() == COLUMN_FIXED ? 1 : 0;
            kz += eo.get(ParameterType::CAMERA_COORDINATE_Z).getColumn() == COLUMN_FIXED ? 1 : 0;
            theory();
            if (rd.allFixed()) break;
        }
        if (rd.allFixed()) break;
    }
}

// BundleAdjustment.centroidCoordinates (BundleAdjustment.java:115-201)
void BundleAdjustment::centroidCoordinates(bool invert) {
    auto axis = [](ParameterType t) {
        switch (t) {
        case ParameterType::CAMERA_COORDINATE_X: case ParameterType::OBJECT_COORDINATE_X: return 0;
        case ParameterType::CAMERA_COORDINATE_Y: case ParameterType::OBJECT_COORDINATE_Y: return 1;
        case ParameterType::CAMERA_COORDINATE_Z: case ParameterType::OBJECT_COORDINATE_Z: return 2;
        default: return -1;
        }
    };
    if (!invert) {
        double s[3] = {0, 0, 0};
        int cnt[3] = {0, 0, 0};
        for (UnknownParameter *p : unknownParameters_) {
            const int a = axis(p->getParameterType());
            if (a >= 0) { s[a] += p->getValue(); cnt[a]++; }
        }
        if (cnt[0] == cnt[1] && cnt[0] == cnt[2] && cnt[0] > 0) {
            for (int a = 0; a < 3; a++) centroid_[a] = s[a] / cnt[a];
        } else
            throw std::logic_error("Error, the numbers of coordinate components are un-equal or zero (BundleAdjustment.java:151)");
    }
    const double sign = invert ? 1.0 : -1.0;
    const double c[3] = {sign * centroid_[0], sign * centroid_[1], sign * centroid_[2]};
    for (UnknownParameter *p : unknownParameters_) {
        const int a = axis(p->getParameterType());
        if (a >= 0) p->setValue(p->getValue() + c[a]);
    }
    for (auto *g : groups_)
        for (ObservationParameter *op : g->observations()) {
            const int a = axis(op->getParameterType());
            if (a >= 0) op->setValue(op->getValue() + c[a]);
        }
}

// ---------------------------------------------------------------------------------------------------------------
// object graph -> flat arrays of jaicov_problem_desc
// ---------------------------------------------------------------------------------------------------------------
static int32_t flat_col(const UnknownParameter *p) {
    const int c = p->getColumn();
    return (c == COLUMN_FIXED || c < 0) ? JAICOV_COL_FIXED : c;
}

void BundleAdjustment::flatten() {
    Flat &f = flat;
    f = Flat();
    const int P = (int)objectCoordinates_.size();
    for (int i = 0; i < P; i++) objectCoordinates_[i]->index = i;
    int nimg = 0, ncam = 0;
    for (Camera *c : cameras_) {
        c->index = ncam++;
        for (auto &im : c->images()) im->index = nimg++;
    }
    // slots: points | io | dist | eo
    f.slot_param.clear();
    for (ObjectCoordinate *oc : objectCoordinates_) {
        UnknownParameter *q[3] = {&oc->getX(), &oc->getY(), &oc->getZ()};
        for (auto *p : q) { p->slot = (int)f.slot_param.size(); f.slot_param.push_back(p); f.point_col.push_back(flat_col(p)); }
        f.point_datum.push_back(oc->isDatum() ? 1 : 0);
    }
    for (Camera *c : cameras_)
        for (int i = 0; i < 3; i++) {
            UnknownParameter *p = c->getInteriorOrientation().at(i);
            p->slot = (int)f.slot_param.size(); f.slot_param.push_back(p); f.io_col.push_back(flat_col(p));
        }
    f.cam_dist_begin.push_back(0);
    for (Camera *c : cameras_) {
        double r0 = 0.0;
        for (auto &m : c->getDistortionModels()) {
            if (m->getType() >= DistortionModel::Type::RADIAL_DISTORTION) r0 = m->getR0();   // radial, distance, Zernike models carry r0
            for (auto &p : m->parameters()) {
                int kind;
                switch (p->getParameterType()) {
                case ParameterType::AFFINITY_AND_SHEAR_Cx: kind = JAICOV_DIST_AFFINITY_CX; break;
                case ParameterType::AFFINITY_AND_SHEAR_Cy: kind = JAICOV_DIST_AFFINITY_CY; break;
                case ParameterType::TANGENTIAL_DISTORTION_Bx: kind = JAICOV_DIST_TANGENTIAL_BX; break;
                case ParameterType::TANGENTIAL_DISTORTION_By: kind = JAICOV_DIST_TANGENTIAL_BY; break;
                case ParameterType::TANGENTIAL_POLYNOMIAL_B: kind = JAICOV_DIST_TANGENTIAL_BI; break;
                case ParameterType::RADIAL_POLYNOMIAL_A: kind = JAICOV_DIST_RADIAL_AI; break;
                case ParameterType::ZERNIKE_POLYNOMIAL_X: kind = JAICOV_DIST_ZERNIKE_X; break;
                case ParameterType::ZERNIKE_POLYNOMIAL_Y: kind = JAICOV_DIST_ZERNIKE_Y; break;
                case ParameterType::ZERNIKE_POLYNOMIAL_Z: kind = JAICOV_DIST_ZERNIKE_Z; break;
                default: kind = JAICOV_DIST_DISTANCE_DI; break;
                }
                p->slot = (int)f.slot_param.size(); f.slot_param.push_back(p.get());
                f.dist_kind.push_back(kind); f.dist_order.push_back(p->getOrder() > 0 ? p->getOrder() : 0); f.dist_col.push_back(flat_col(p.get()));
            }
        }
        f.cam_r0.push_back(r0);
        f.cam_dist_begin.push_back((int32_t)f.dist_kind.size());
    }
    for (Camera *c : cameras_)
        for (auto &im : c->images()) {
            f.image_camera.push_back(c->index);
            for (int i = 0; i < 6; i++) {
                UnknownParameter *p = im->getExteriorOrientation().at(i);
                p->slot = (int)f.slot_param.size(); f.slot_param.push_back(p); f.eo_col.push_back(flat_col(p));
            }
        }
    // image points, image-major; images with a joint dispersion become image blocks
    f.blk_ip_begin.clear();
    bool any_block = false, block_open = false;
    for (Camera *c : cameras_)
        for (auto &im : c->images()) {
            const bool blk = !im->dispersion().empty();
            if (blk) {
                if (any_block && !block_open) throw std::invalid_argument("images with a joint dispersion must be consecutive");
                if (!any_block) f.blk_ip_begin.push_back((int32_t)f.ip_image.size());
                any_block = block_open = true;
                f.blk_disp_offset.push_back((int64_t)f.blk_disp.size());
                f.blk_disp.insert(f.blk_disp.end(), im->dispersion().begin(), im->dispersion().end());
            } else if (block_open) block_open = false;
            for (auto &ic : im->coordinates()) {
                f.ip_image.push_back(im->index);
                f.ip_point.push_back(ic->getObjectCoordinate()->index);
                f.ip_x.push_back(ic->getX().getValue()); f.ip_y.push_back(ic->getY().getValue());
                f.ip_var_x.push_back(ic->getX().getVariance()); f.ip_var_y.push_back(ic->getY().getVariance());
                f.ip_rho.push_back(ic->getCorrelationCoefficientXY());
            }
            if (blk) f.blk_ip_begin.push_back((int32_t)f.ip_image.size());
        }
    if (!any_block) f.blk_ip_begin.push_back(0);
    for (ScaleBar *sb : scaleBars_) {
        f.sb_a.push_back(sb->getObjectCoordinateA()->index); f.sb_b.push_back(sb->getObjectCoordinateB()->index);
        f.sb_len.push_back(sb->getLength().getValue()); f.sb_var.push_back(sb->getLength().getVariance());
    }
    f.dg_row_begin.push_back(0);
    for (auto *g : groups_) {
        for (ObservationParameter *op : g->observations()) {
            f.dg_slot.push_back(op->getReferenceParameter()->slot);
            f.dg_obs.push_back(op->getValue());
            f.dg_var.push_back(op->getVariance());
        }
        f.dg_row_begin.push_back((int32_t)f.dg_slot.size());
        if (g->hasFullyPopulatedWeightMatrix()) {
            f.dg_disp_offset.push_back((int64_t)f.dg_disp.size());
            f.dg_disp.insert(f.dg_disp.end(), g->dispersion().begin(), g->dispersion().end());
        } else
            f.dg_disp_offset.push_back(-1);
    }
    pushValues();
}

void BundleAdjustment::pushValues() {
    flat.values.resize(flat.slot_param.size());
    for (size_t s = 0; s < flat.slot_param.size(); s++) flat.values[s] = flat.slot_param[s]->getValue();
    // observed values of directly observed groups move with the centroid (BA:180-200)
    size_t r = 0;
    for (auto *g : groups_)
        for (ObservationParameter *op : g->observations()) flat.dg_obs[r++] = op->getValue();
}

void BundleAdjustment::pullValues(const std::vector<double> &v) {
    for (size_t s = 0; s < flat.slot_param.size(); s++) flat.slot_param[s]->setValue(v[s]);
}

// ---------------------------------------------------------------------------------------------------------------
// BundleAdjustment.estimateModel (BundleAdjustment.java:203-387) with updateModel (BA:389-442): the loop stays on the
// host exactly as in the reference; BA:235 (createNormalEquation), BA:238/270-297 (precondition + MathExtension.solve)
// and BA:397/430 (getOmega) are the calls that cross the C ABI.
// ---------------------------------------------------------------------------------------------------------------
EstimationStateType BundleAdjustment::estimateModel() {
    fire("BUSY", 0, 1);
    bool deriveFirst = damping_ > 0;
    double adapted = 0.0, lastValid = 0.0;
    maxAbsDx_ = 0.0;
    int runs = maxIter_ - 1;
    bool isEstimated = false, complete = false, isConverge = true;
    if (maxIter_ == 0) { complete = isEstimated = true; adapted = 0; }
    sigma2apriori_ = sigma2apriori_ > 0 ? sigma2apriori_ : 1.0;
    try {
        prepareUnknownParameters();
        if (centroided_) centroidCoordinates(false);
        flatten();
    } catch (const std::exception &ex) {
        lastError_ = ex.what();
        return EstimationStateType::NOT_INITIALISED;
    }
    const Flat &f = flat;
    jaicov_problem_desc d;
    std::memset(&d, 0, sizeof(d));
    d.struct_size = sizeof(d);
    d.n_unknowns = numberOfUnknownParameters_ + rankDefect_.getDefect();
    d.rank_defect = rankDefect_.getDefect();
    d.datum_flags = rankDefect_.flags();
    d.n_points = (int)objectCoordinates_.size(); d.n_cameras = (int)cameras_.size(); d.n_images = (int)f.image_camera.size();
    d.n_dist = (int)f.dist_kind.size(); d.n_image_points = (int)f.ip_image.size(); d.n_image_blocks = (int)f.blk_ip_begin.size() - 1;
    d.n_scale_bars = (int)f.sb_a.size(); d.n_direct_groups = (int)f.dg_row_begin.size() - 1; d.n_direct_rows = (int)f.dg_slot.size();
    d.point_col = f.point_col.data(); d.point_datum = f.point_datum.data(); d.io_col = f.io_col.data(); d.cam_r0 = f.cam_r0.data();
    d.cam_dist_begin = f.cam_dist_begin.data(); d.dist_kind = f.dist_kind.data(); d.dist_order = f.dist_order.data(); d.dist_col = f.dist_col.data();
    d.image_camera = f.image_camera.data(); d.eo_col = f.eo_col.data(); d.ip_image = f.ip_image.data(); d.ip_point = f.ip_point.data();
    d.ip_x = f.ip_x.data(); d.ip_y = f.ip_y.data(); d.ip_var_x = f.ip_var_x.data(); d.ip_var_y = f.ip_var_y.data(); d.ip_rho = f.ip_rho.data();
    d.blk_ip_begin = f.blk_ip_begin.data(); d.blk_disp_offset = f.blk_disp_offset.data(); d.blk_disp = f.blk_disp.data();
    d.sb_point_a = f.sb_a.data(); d.sb_point_b = f.sb_b.data(); d.sb_length = f.sb_len.data(); d.sb_var = f.sb_var.data();
    d.dg_row_begin = f.dg_row_begin.data(); d.dg_slot = f.dg_slot.data(); d.dg_obs = f.dg_obs.data(); d.dg_var = f.dg_var.data();
    d.dg_disp_offset = f.dg_disp_offset.data(); d.dg_disp = f.dg_disp.data();

    jaicov_engine_options eo;
    std::memset(&eo, 0, sizeof(eo));
    eo.struct_size = sizeof(eo); eo.device = device_; eo.image_begin = eo.image_end = -1; eo.apply_shared = 1;
    int rc = jaicov_neq_create(&d, &eo, &engine_);
    auto fail = [&](int code) {
        lastError_ = engine_ ? jaicov_neq_last_error(engine_) : "engine creation failed";
        if (code == JAICOV_ERR_OUT_OF_MEMORY) return EstimationStateType::OUT_OF_MEMORY;        // BA:370-375
        if (code > 0 || code == JAICOV_ERR_BAD_ARGUMENT) return EstimationStateType::SINGULAR_MATRIX;   // BA:304-309
        return EstimationStateType::INTERRUPT;                                                     // BA:310-315
    };
    if (rc != JAICOV_OK) return fail(rc);
    if ((rc = jaicov_neq_set_parameters(engine_, f.values.data(), f.values.size())) != JAICOV_OK) return fail(rc);
    const int U = d.n_unknowns;
    const bool simulation = estimationType_ == EstimationType::SIMULATION;
    // REDUCED / PRE_ELIMINATION (BA:261-267, 283-291, 1197-1453): the final pass inverts the system from which the
    // exterior orientations were eliminated, so Qxx holds the block of the datum border, the points, the interior
    // orientation and the distortion parameters only.  The engine eliminates the EO blocks itself in every pass whenever
    // the problem allows it (jaicov_neq_reduced_order() < U), whatever the mode; where it cannot, REDUCED is served by the
    // full inverse (a superset of what the reference leaves in N).
    const bool wantInverse = inversion_ != MatrixInversion::NONE;
    const int invertMode = inversion_ == MatrixInversion::NONE ? JAICOV_INVERT_NONE
                         : inversion_ == MatrixInversion::FULL ? JAICOV_INVERT_FULL_EXPANDED : JAICOV_INVERT_REDUCED;   // FULL: all of Qxx,
    // expanded from the inverse of the EO-reduced system where the engine can pre-eliminate (jaicov_neq.h), else the plain full inverse
    std::vector<double> dx((size_t)std::max(U, 1));
    EstimationStateType status = EstimationStateType::BUSY;
    do {
        maxAbsDx_ = 0.0;
        iterationStep_ = maxIter_ - runs;
        fire("ITERATE", maxIter_, iterationStep_);
        if (deriveFirst) { adapted = damping_; deriveFirst = false; }                         // BA:801-812
        jaicov_neq_prepare_inverse(engine_, isEstimated ? invertMode : JAICOV_INVERT_NONE);   // BA:250: the final pass is known before it is built
        if ((rc = jaicov_neq_build(engine_, sigma2apriori_, adapted, simulation ? 1 : 0)) != JAICOV_OK) return fail(rc);   // BA:235
        if (interrupt_) { interrupt_ = false; return EstimationStateType::INTERRUPT; }         // BA:240-245
        complete = isEstimated;
        if (complete && wantInverse) fire("INVERT_NORMAL_EQUATION_MATRIX", 0, 1);
        if ((rc = jaicov_neq_solve(engine_, complete ? invertMode : JAICOV_INVERT_NONE, dx.data())) != JAICOV_OK) return fail(rc);     // BA:264,270,294
        // ---- updateModel (BA:389-442) ----
        bool rejected = false;
        if (adapted > 0) {
            double alpha = 0.25 * std::pow(adapted, -0.05);
            alpha = std::min(alpha, 0.75);
            for (auto &v : dx) v *= alpha;
            double prevOmega = omega_, curOmega = 0.0;
            if ((rc = jaicov_neq_omega(engine_, sigma2apriori_, dx.data(), &curOmega)) != JAICOV_OK) return fail(rc);
            prevOmega = prevOmega <= 0 ? 1.7976931348623157e308 : prevOmega;
            const bool lmaConverge = prevOmega >= curOmega;
            omega_ = curOmega;
            const double last = adapted;
            if (lmaConverge) adapted *= 0.2;
            else {
                adapted *= 5.0;
                if (adapted > 1.0 / SQRT_EPS) { adapted = 1.0 / SQRT_EPS; omega_ = 0.0; }
            }
            fire("LEVENBERG_MARQUARDT_STEP", last, adapted);
            if (!lmaConverge) { maxAbsDx_ = lastValid; rejected = true; }
        }
        if (!rejected) {
            if (complete) {
                if (simulation) omega_ = 0.0;
                else if ((rc = jaicov_neq_omega(engine_, sigma2apriori_, dx.data(), &omega_)) != JAICOV_OK) return fail(rc);
            }
            if ((rc = jaicov_neq_update(engine_, dx.data(), &maxAbsDx_)) != JAICOV_OK) return fail(rc);   // BA:450-462
            lastValid = maxAbsDx_;
        }
        if (interrupt_) { interrupt_ = false; return EstimationStateType::INTERRUPT; }
        // ---- BA:327-353 ----
        if (std::isinf(maxAbsDx_) || std::isnan(maxAbsDx_)) return EstimationStateType::SINGULAR_MATRIX;
        else if (maxAbsDx_ <= SQRT_EPS && runs > 0 && adapted == 0) {
            isEstimated = true;
            fire("CONVERGENCE", SQRT_EPS, maxAbsDx_);
        } else if (runs-- <= 1) {
            if (complete) { fire("NO_CONVERGENCE", SQRT_EPS, maxAbsDx_); isConverge = false; }
            isEstimated = true;
        } else
            fire("CONVERGENCE", SQRT_EPS, maxAbsDx_);
        if (isEstimated || adapted <= SQRT_EPS || runs < maxIter_ * 0.5 + 1) adapted = 0.0;
    } while (!complete);

    std::vector<double> v(f.values.size());
    if ((rc = jaicov_neq_get_parameters(engine_, v.data(), v.size())) != JAICOV_OK) return fail(rc);
    pullValues(v);
    Qxx_.clear();
    qxxOnDevice_ = wantInverse;      // fetched by getCofactorMatrix() / gathered by cofactorSub() on demand
    if (centroided_) centroidCoordinates(true);      // BA:357-358
    if (resultWriter_) {                             // BA:360-368, exportAdjustmentResults BA:1164-1171
        fire("EXPORT_ADJUSTMENT_RESULTS", 0.0, 0.0);
        try {
            resultWriter_->exportResults(*this);
        } catch (const std::exception &ex) {
            lastError_ = ex.what();
            fire("EXPORT_ADJUSTMENT_RESULTS_FAILED", 0.0, 1.0);
            return EstimationStateType::EXPORT_ADJUSTMENT_RESULTS_FAILED;
        }
    }
    status = isConverge ? EstimationStateType::ERROR_FREE_ESTIMATION : EstimationStateType::NO_CONVERGENCE;   // BA:377-384
    fire(isConverge ? "ERROR_FREE_ESTIMATION" : "NO_CONVERGENCE", SQRT_EPS, maxAbsDx_);
    return status;
}

void BundleAdjustment::fetchCofactor() const {
    if (!qxxOnDevice_ || !engine_) return;
    // packed 'U' (column-major upper): the leading k x k block is the leading k(k+1)/2 entries, so the reduced
    // cofactor matrix lands where the reference's in-place solve(N, n, numRows, true) leaves it (BA:264,274)
    const size_t k = (size_t)jaicov_neq_cofactor_order(engine_);
    Qxx_.assign(jaicov_neq_packed_length(engine_), 0.0);
    const int rc = jaicov_neq_get_cofactor(engine_, Qxx_.data(), k * (k + 1) / 2);
    if (rc != JAICOV_OK) { Qxx_.clear(); throw std::runtime_error(std::string("jaicov_neq_get_cofactor: ") + jaicov_neq_last_error(engine_)); }
    qxxOnDevice_ = false;
}

std::vector<double> BundleAdjustment::cofactorSub(const std::vector<int32_t> &idx, double scale) const {
    const size_t k = idx.size();
    std::vector<double> out(k * k);
    if (k == 0) return out;
    if (engine_ && (qxxOnDevice_ || !Qxx_.empty())) {
        const int rc = jaicov_neq_get_dispersion_sub(engine_, scale, idx.data(), (int32_t)k, out.data());
        if (rc != JAICOV_OK) throw std::runtime_error(std::string("jaicov_neq_get_dispersion_sub: ") + jaicov_neq_last_error(engine_));
        return out;
    }
    if (Qxx_.empty()) throw std::runtime_error("no cofactor matrix (MatrixInversion.NONE or not estimated)");
    for (size_t r = 0; r < k; r++)
        for (size_t c = 0; c < k; c++) {
            int a = idx[r], b = idx[c];
            if (a > b) std::swap(a, b);
            out[r * k + c] = scale * Qxx_.at((size_t)a + (size_t)b * (b + 1) / 2);
        }
    return out;
}

}  // namespace jaicov::host
