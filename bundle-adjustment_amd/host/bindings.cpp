// pybind11 module _jaicov_host: exposes the C++ mirror of JAICOV's object API (jaicov.hpp) to the Python tests,
// examples and loaders.  Names follow the reference's Java API.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "aicon_reader.hpp"
#include "result_writer.hpp"
#include "jaicov.hpp"

namespace py = pybind11;
using namespace jaicov::host;

PYBIND11_MODULE(_jaicov_host, m) {
    m.doc() = "C++ mirror of JAICOV's Camera/Image/ObjectCoordinate/BundleAdjustment API on the MI355X engine";
    m.attr("COLUMN_FIXED") = COLUMN_FIXED;
    m.attr("COLUMN_NOT_SET") = COLUMN_NOT_SET;

    py::enum_<ParameterType>(m, "ParameterType")
        .value("PRINCIPAL_POINT_X", ParameterType::PRINCIPAL_POINT_X).value("PRINCIPAL_POINT_Y", ParameterType::PRINCIPAL_POINT_Y)
        .value("PRINCIPAL_DISTANCE", ParameterType::PRINCIPAL_DISTANCE).value("RADIAL_POLYNOMIAL_A", ParameterType::RADIAL_POLYNOMIAL_A)
        .value("TANGENTIAL_POLYNOMIAL_B", ParameterType::TANGENTIAL_POLYNOMIAL_B)
        .value("TANGENTIAL_DISTORTION_Bx", ParameterType::TANGENTIAL_DISTORTION_Bx).value("TANGENTIAL_DISTORTION_By", ParameterType::TANGENTIAL_DISTORTION_By)
        .value("AFFINITY_AND_SHEAR_Cx", ParameterType::AFFINITY_AND_SHEAR_Cx).value("AFFINITY_AND_SHEAR_Cy", ParameterType::AFFINITY_AND_SHEAR_Cy)
        .value("DISTANCE_POLYNOMIAL_D", ParameterType::DISTANCE_POLYNOMIAL_D)
        .value("ZERNIKE_POLYNOMIAL_X", ParameterType::ZERNIKE_POLYNOMIAL_X).value("ZERNIKE_POLYNOMIAL_Y", ParameterType::ZERNIKE_POLYNOMIAL_Y)
        .value("ZERNIKE_POLYNOMIAL_Z", ParameterType::ZERNIKE_POLYNOMIAL_Z)
        .value("CAMERA_COORDINATE_X", ParameterType::CAMERA_COORDINATE_X).value("CAMERA_COORDINATE_Y", ParameterType::CAMERA_COORDINATE_Y)
        .value("CAMERA_COORDINATE_Z", ParameterType::CAMERA_COORDINATE_Z).value("CAMERA_OMEGA", ParameterType::CAMERA_OMEGA)
        .value("CAMERA_PHI", ParameterType::CAMERA_PHI).value("CAMERA_KAPPA", ParameterType::CAMERA_KAPPA)
        .value("OBJECT_COORDINATE_X", ParameterType::OBJECT_COORDINATE_X).value("OBJECT_COORDINATE_Y", ParameterType::OBJECT_COORDINATE_Y)
        .value("OBJECT_COORDINATE_Z", ParameterType::OBJECT_COORDINATE_Z);
    py::enum_<EstimationStateType>(m, "EstimationStateType")
        .value("ERROR_FREE_ESTIMATION", EstimationStateType::ERROR_FREE_ESTIMATION).value("BUSY", EstimationStateType::BUSY)
        .value("INTERRUPT", EstimationStateType::INTERRUPT).value("SINGULAR_MATRIX", EstimationStateType::SINGULAR_MATRIX)
        .value("NO_CONVERGENCE", EstimationStateType::NO_CONVERGENCE).value("NOT_INITIALISED", EstimationStateType::NOT_INITIALISED)
        .value("ROBUST_ESTIMATION_FAILED", EstimationStateType::ROBUST_ESTIMATION_FAILED)
        .value("EXPORT_ADJUSTMENT_RESULTS_FAILED", EstimationStateType::EXPORT_ADJUSTMENT_RESULTS_FAILED)
        .value("OUT_OF_MEMORY", EstimationStateType::OUT_OF_MEMORY);
    py::enum_<EstimationType>(m, "EstimationType").value("L2NORM", EstimationType::L2NORM).value("SIMULATION", EstimationType::SIMULATION);
    py::enum_<MatrixInversion>(m, "MatrixInversion")
        .value("NONE", MatrixInversion::NONE).value("FULL", MatrixInversion::FULL)
        .value("PRE_ELIMINATION", MatrixInversion::PRE_ELIMINATION).value("REDUCED", MatrixInversion::REDUCED);
    py::enum_<DistortionModel::Type>(m, "DistortionModelType")
        .value("AFFINITY_AND_SHEAR", DistortionModel::Type::AFFINITY_AND_SHEAR).value("TANGENTIAL_DISTORTION", DistortionModel::Type::TANGENTIAL_DISTORTION)
        .value("RADIAL_DISTORTION", DistortionModel::Type::RADIAL_DISTORTION).value("DISTANCE_DISTORTION", DistortionModel::Type::DISTANCE_DISTORTION)
        .value("ZERNIKE_X", DistortionModel::Type::ZERNIKE_X).value("ZERNIKE_Y", DistortionModel::Type::ZERNIKE_Y)
        .value("ZERNIKE_GRADIENT", DistortionModel::Type::ZERNIKE_GRADIENT);

    py::class_<UnknownParameter>(m, "UnknownParameter")
        .def("getParameterType", &UnknownParameter::getParameterType)
        .def("getValue", &UnknownParameter::getValue).def("setValue", &UnknownParameter::setValue)
        .def("getColumn", &UnknownParameter::getColumn).def("setColumn", &UnknownParameter::setColumn)
        .def("getOrder", &UnknownParameter::getOrder);
    py::class_<ObservationParameter>(m, "ObservationParameter")
        .def(py::init<UnknownParameter *>(), py::keep_alive<1, 2>())
        .def("getValue", &ObservationParameter::getValue).def("setValue", &ObservationParameter::setValue)
        .def("getVariance", &ObservationParameter::getVariance).def("setVariance", &ObservationParameter::setVariance)
        .def("getRow", &ObservationParameter::getRow);
    py::class_<ObjectCoordinate>(m, "ObjectCoordinate")
        .def(py::init<std::string, double, double, double>())
        .def("getName", &ObjectCoordinate::getName)
        .def("getX", &ObjectCoordinate::getX, py::return_value_policy::reference_internal)
        .def("getY", &ObjectCoordinate::getY, py::return_value_policy::reference_internal)
        .def("getZ", &ObjectCoordinate::getZ, py::return_value_policy::reference_internal)
        .def("isDatum", &ObjectCoordinate::isDatum).def("setDatum", &ObjectCoordinate::setDatum);
    py::class_<ScaleBar>(m, "ScaleBar")
        .def(py::init<ObjectCoordinate *, ObjectCoordinate *, double, double>(), py::keep_alive<1, 2>(), py::keep_alive<1, 3>())
        .def("getLength", &ScaleBar::getLength, py::return_value_policy::reference_internal);
    py::class_<DistortionModel>(m, "DistortionModel")
        .def("getType", &DistortionModel::getType).def("getR0", &DistortionModel::getR0)
        .def("add", &DistortionModel::add, py::return_value_policy::reference_internal)
        .def("get", &DistortionModel::get, py::return_value_policy::reference_internal)
        .def("getCx", &DistortionModel::getCx, py::return_value_policy::reference_internal)
        .def("getCy", &DistortionModel::getCy, py::return_value_policy::reference_internal)
        .def("getBx", &DistortionModel::getBx, py::return_value_policy::reference_internal)
        .def("getBy", &DistortionModel::getBy, py::return_value_policy::reference_internal)
        .def("parameters", [](DistortionModel &d) {
            std::vector<UnknownParameter *> v;
            for (auto &p : d.parameters()) v.push_back(p.get());
            return v;
        }, py::return_value_policy::reference_internal);
    py::class_<InteriorOrientation>(m, "InteriorOrientation")
        .def("getPrinciplePointX", &InteriorOrientation::getPrinciplePointX, py::return_value_policy::reference_internal)
        .def("getPrinciplePointY", &InteriorOrientation::getPrinciplePointY, py::return_value_policy::reference_internal)
        .def("getPrincipleDistance", &InteriorOrientation::getPrincipleDistance, py::return_value_policy::reference_internal);
    py::class_<ExteriorOrientation>(m, "ExteriorOrientation")
        .def("get", &ExteriorOrientation::get, py::return_value_policy::reference_internal);
    py::class_<ImageCoordinate>(m, "ImageCoordinate")
        .def("getObjectCoordinate", &ImageCoordinate::getObjectCoordinate, py::return_value_policy::reference)
        .def("getX", &ImageCoordinate::getX, py::return_value_policy::reference_internal)
        .def("getY", &ImageCoordinate::getY, py::return_value_policy::reference_internal)
        .def("getCorrelationCoefficientXY", &ImageCoordinate::getCorrelationCoefficientXY);
    py::class_<Image>(m, "Image")
        .def("getId", &Image::getId)
        .def("getExteriorOrientation", &Image::getExteriorOrientation, py::return_value_policy::reference_internal)
        .def("add", &Image::add, py::return_value_policy::reference_internal, py::keep_alive<1, 2>(), py::arg("objectCoordinate"),
             py::arg("xp"), py::arg("yp"), py::arg("sigmax"), py::arg("sigmay"), py::arg("corrCoefXY") = 0.0)
        .def("getNumberOfImageCoordinates", &Image::getNumberOfImageCoordinates)
        .def("setDispersion", [](Image &im, py::array_t<double, py::array::c_style | py::array::forcecast> D) {
            im.setDispersion(std::vector<double>(D.data(), D.data() + D.size()));
        })
        .def("coordinates", [](Image &im) {
            std::vector<ImageCoordinate *> v;
            for (auto &c : im.coordinates()) v.push_back(c.get());
            return v;
        }, py::return_value_policy::reference_internal);
    py::class_<Camera>(m, "Camera")
        .def(py::init<long, double, std::vector<DistortionModel::Type>>())
        .def("getId", &Camera::getId)
        .def("getInteriorOrientation", &Camera::getInteriorOrientation, py::return_value_policy::reference_internal)
        .def("add", &Camera::add, py::return_value_policy::reference_internal)
        .def("getNumberOfImages", &Camera::getNumberOfImages)
        .def("getDistortionModel", &Camera::getDistortionModel, py::return_value_policy::reference_internal)
        .def("images", [](Camera &c) {
            std::vector<Image *> v;
            for (auto &im : c.images()) v.push_back(im.get());
            return v;
        }, py::return_value_policy::reference_internal);
    py::class_<DirectlyObservedParameterGroup>(m, "DirectlyObservedParameterGroup")
        .def(py::init<std::vector<ObservationParameter *>>(), py::keep_alive<1, 2>())
        .def(py::init([](py::array_t<double, py::array::c_style | py::array::forcecast> D, std::vector<ObservationParameter *> obs) {
                 return new DirectlyObservedParameterGroup(std::vector<double>(D.data(), D.data() + D.size()), std::move(obs));
             }), py::keep_alive<1, 3>())
        .def("hasFullyPopulatedWeightMatrix", &DirectlyObservedParameterGroup::hasFullyPopulatedWeightMatrix)
        .def("getNumberOfParameters", &DirectlyObservedParameterGroup::getNumberOfParameters);

    py::class_<BundleAdjustment>(m, "BundleAdjustment")
        .def(py::init<>())
        .def("add", py::overload_cast<Camera *>(&BundleAdjustment::add), py::keep_alive<1, 2>())
        .def("add", py::overload_cast<ScaleBar *>(&BundleAdjustment::add), py::keep_alive<1, 2>())
        .def("add", py::overload_cast<DirectlyObservedParameterGroup *>(&BundleAdjustment::add), py::keep_alive<1, 2>())
        .def("addPropertyChangeListener", &BundleAdjustment::addPropertyChangeListener)
        .def("setEstimationType", &BundleAdjustment::setEstimationType)
        .def("setInvertNormalEquation", &BundleAdjustment::setInvertNormalEquation)
        .def("setAdjustmentResultWriter", &BundleAdjustment::setAdjustmentResultWriter, py::keep_alive<1, 2>())
        .def("hasCofactorMatrix", &BundleAdjustment::hasCofactorMatrix)
        .def("cofactorSub", [](BundleAdjustment &b, std::vector<int32_t> idx, double scale) {
            std::vector<double> v = b.cofactorSub(idx, scale);
            return py::array_t<double>({idx.size(), idx.size()}, v.data());
        }, py::arg("indices"), py::arg("scale") = 1.0)
        .def("useCentroidedCoordinates", &BundleAdjustment::useCentroidedCoordinates)
        .def("centroidCoordinates", &BundleAdjustment::centroidCoordinates)
        .def("applyAposterioriVarianceOfUnitWeight", &BundleAdjustment::applyAposterioriVarianceOfUnitWeight)
        .def("setLevenbergMarquardtDampingValue", &BundleAdjustment::setLevenbergMarquardtDampingValue)
        .def("setMaximalNumberOfIterations", &BundleAdjustment::setMaximalNumberOfIterations)
        .def("setDevice", &BundleAdjustment::setDevice)
        .def("prepareUnknownParameters", &BundleAdjustment::prepareUnknownParameters)
        .def("flatten", &BundleAdjustment::flatten)
        .def("estimateModel", &BundleAdjustment::estimateModel, py::call_guard<py::gil_scoped_release>())
        .def("getNumberOfObservations", &BundleAdjustment::getNumberOfObservations)
        .def("getNumberOfUnknownParameters", &BundleAdjustment::getNumberOfUnknownParameters)
        .def("getNumberOfDatumConditions", &BundleAdjustment::getNumberOfDatumConditions)
        .def("getDegreeOfFreedom", &BundleAdjustment::getDegreeOfFreedom)
        .def("getVarianceFactorApriori", &BundleAdjustment::getVarianceFactorApriori)
        .def("getVarianceFactorAposteriori", &BundleAdjustment::getVarianceFactorAposteriori)
        .def("getOmega", &BundleAdjustment::getOmega)
        .def("getIterations", &BundleAdjustment::getIterations)
        .def("getDatumFlags", [](BundleAdjustment &b) { return b.getRankDefect().flags(); })
        .def("lastError", &BundleAdjustment::lastError)
        .def("getObjectCoordinates", [](BundleAdjustment &b) { return b.getObjectCoordinates(); }, py::return_value_policy::reference_internal)
        .def("getCofactorMatrix", [](BundleAdjustment &b) {
            const auto &q = b.getCofactorMatrix();
            return py::array_t<double>(q.size(), q.data());
        })
        .def("flat", [](BundleAdjustment &b) {
            // the flattened arrays as a dict of numpy arrays (parity tests feed them to the oracle)
            py::dict d;
            const auto &f = b.flat;
            auto I = [](const std::vector<int32_t> &v) { return py::array_t<int32_t>(v.size(), v.data()); };
            auto D = [](const std::vector<double> &v) { return py::array_t<double>(v.size(), v.data()); };
            auto L = [](const std::vector<int64_t> &v) { return py::array_t<int64_t>(v.size(), v.data()); };
            d["point_col"] = I(f.point_col); d["io_col"] = I(f.io_col); d["cam_dist_begin"] = I(f.cam_dist_begin);
            d["dist_kind"] = I(f.dist_kind); d["dist_order"] = I(f.dist_order); d["dist_col"] = I(f.dist_col);
            d["image_camera"] = I(f.image_camera); d["eo_col"] = I(f.eo_col); d["ip_image"] = I(f.ip_image); d["ip_point"] = I(f.ip_point);
            d["blk_ip_begin"] = I(f.blk_ip_begin); d["sb_point_a"] = I(f.sb_a); d["sb_point_b"] = I(f.sb_b);
            d["dg_row_begin"] = I(f.dg_row_begin); d["dg_slot"] = I(f.dg_slot);
            d["point_datum"] = py::array_t<uint8_t>(f.point_datum.size(), f.point_datum.data());
            d["cam_r0"] = D(f.cam_r0); d["ip_x"] = D(f.ip_x); d["ip_y"] = D(f.ip_y); d["ip_var_x"] = D(f.ip_var_x); d["ip_var_y"] = D(f.ip_var_y);
            d["ip_rho"] = D(f.ip_rho); d["blk_disp"] = D(f.blk_disp); d["sb_length"] = D(f.sb_len); d["sb_var"] = D(f.sb_var);
            d["dg_obs"] = D(f.dg_obs); d["dg_var"] = D(f.dg_var); d["dg_disp"] = D(f.dg_disp); d["values"] = D(f.values);
            d["blk_disp_offset"] = L(f.blk_disp_offset); d["dg_disp_offset"] = L(f.dg_disp_offset);
            d["n_unknowns"] = b.getNumberOfUnknownParameters() + b.getNumberOfDatumConditions();
            d["rank_defect"] = b.getNumberOfDatumConditions();
            d["datum_flags"] = b.getRankDefect().flags();
            d["sigma2apriori"] = b.getVarianceFactorApriori();
            return d;
        });

    py::class_<AdjustmentResultWritable>(m, "AdjustmentResultWritable")
        .def("export", &AdjustmentResultWritable::exportResults);
    py::class_<DefaultResultWriter, AdjustmentResultWritable>(m, "DefaultResultWriter")
        .def(py::init<std::string>())
        .def("toString", &DefaultResultWriter::toString);
    py::class_<MatlabResultWriter, AdjustmentResultWritable>(m, "MatlabResultWriter")
        .def(py::init<std::string>())
        .def("toString", &MatlabResultWriter::toString);
    m.def("java_fixed", &java_fixed, "java.util.Formatter %[+].<prec>f");

    py::class_<AiconProject>(m, "AiconProject")
        .def_property_readonly("camera", [](AiconProject &p) { return p.camera ? p.camera.get() : (p.cameras.empty() ? nullptr : p.cameras[0].get()); },
                               py::return_value_policy::reference_internal)
        .def("cameras", [](AiconProject &p) {
            std::vector<Camera *> v;
            if (p.camera) v.push_back(p.camera.get());
            for (auto &q : p.cameras) v.push_back(q.get());
            return v;
        }, py::return_value_policy::reference_internal)
        .def("points", [](AiconProject &p) {
            std::vector<ObjectCoordinate *> v;
            for (auto &q : p.points) v.push_back(q.get());
            return v;
        }, py::return_value_policy::reference_internal)
        .def("point", [](AiconProject &p, const std::string &n) { return p.byName.at(n); }, py::return_value_policy::reference_internal)
        .def("scaleBars", [](AiconProject &p) {
            std::vector<ScaleBar *> v;
            for (auto &q : p.scaleBars) v.push_back(q.get());
            return v;
        }, py::return_value_policy::reference_internal);
    m.def("read_aicon_flat", [](const std::string &base, std::vector<DistortionModel::Type> extra) { return read_aicon_flat(base, extra).release(); },
          py::arg("base"), py::arg("extra") = std::vector<DistortionModel::Type>{}, py::return_value_policy::take_ownership,
          "AICON flat files <base>.{obc,ior,scale,eor,phc} -> object graph (ExampleFlatFiles.java:76-103)");
    m.def("read_aicon_report", [](const std::string &path) { return read_aicon_report(path).release(); }, py::return_value_policy::take_ownership,
          "AICON 3D Studio adjustment report (.htm) -> object graph (AICONReportFileReader.java:117-390)");
}
