// Native counterpart of org.applied_geodesy.adjustment.bundle.example.ExampleDistortionModel (ExampleDistortionModel.java:58-186) on the
// MI355X engine: the bundled block with the camera's distortion described by Zernike polynomials instead of the radial set -- the
// gradient model with the radially symmetric polynomials Z_2^0 .. Z_10^0 (single indices 4, 12, 24, 40, 60; :95-99), the principal
// distance fixed at 28 (:80-82: c correlates with Z(4)), every radial coefficient fixed at 0 (:85-89), MatrixInversion.REDUCED (:121).
// Prints the listing the Java example prints.  No Python, no oracle: C++ host mirror + libjaicov_neq.so only.
//   usage: example_distortion_model <base path> [FULL|REDUCED|PRE_ELIMINATION|NONE]
#include <chrono>
#include <cstdio>
#include <cstring>

#include "aicon_reader.hpp"

using namespace jaicov::host;

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <base path of the .obc/.scale/.ior/.eor/.phc files> [FULL|REDUCED|PRE_ELIMINATION|NONE]\n", argv[0]);
        return 2;
    }
    const auto t0 = std::chrono::steady_clock::now();
    MatrixInversion inv = MatrixInversion::REDUCED;                      // ExampleDistortionModel.java:121
    if (argc > 2) {
        if (!std::strcmp(argv[2], "FULL")) inv = MatrixInversion::FULL;
        else if (!std::strcmp(argv[2], "PRE_ELIMINATION")) inv = MatrixInversion::PRE_ELIMINATION;
        else if (!std::strcmp(argv[2], "NONE")) inv = MatrixInversion::NONE;
    }
    try {
        using T = DistortionModel::Type;
        std::unique_ptr<AiconProject> pr = read_aicon_flat(argv[1], {T::ZERNIKE_GRADIENT, T::ZERNIKE_X, T::ZERNIKE_Y});   // :77
        Camera &cam = *pr->camera;
        cam.getInteriorOrientation().getPrincipleDistance().setValue(28);                                               // :81
        cam.getInteriorOrientation().getPrincipleDistance().setColumn(COLUMN_FIXED);                                    // :82
        for (auto &p : cam.getDistortionModel(T::RADIAL_DISTORTION)->parameters()) {                                     // :85-89
            p->setValue(0);
            p->setColumn(COLUMN_FIXED);
        }
        DistortionModel *zernike = cam.getDistortionModel(T::ZERNIKE_GRADIENT);
        for (int i = 1, order = 0; i < 6; i++) {                                                                         // :95-98
            order += i * 4;
            zernike->add(order);
        }
        BundleAdjustment ba;
        ba.add(&cam);
        for (auto &s : pr->scaleBars) ba.add(s.get());
        ba.addPropertyChangeListener([](const std::string &name, double a, double b) {                                   // :55
            std::printf("Info: %s %g --> %g\n", name.c_str(), a, b);
        });
        ba.setInvertNormalEquation(inv);
        const EstimationStateType state = ba.estimateModel();
        if (state != EstimationStateType::ERROR_FREE_ESTIMATION) {
            std::fprintf(stderr, "Error, bundle adjustment fails... (state %d%s%s)\n", (int)state, ba.lastError().empty() ? "" : ": ", ba.lastError().c_str());
            return 1;
        }
        std::printf("Bundle adjustment finished successfully...\n");
        const bool haveD = inv != MatrixInversion::NONE;
        const double s2 = ba.getVarianceFactorAposteriori();
        // :138-162 object points with their uncertainties ('o' = coded target that took part in the datum, 'n' = new point)
        for (ObjectCoordinate *p : ba.getObjectCoordinates()) {
            const int cx = p->getX().getColumn(), cy = p->getY().getColumn(), cz = p->getZ().getColumn();
            double ux = 0, uy = 0, uz = 0;
            if (haveD && cx >= 0 && cy >= 0 && cz >= 0 && cx != COLUMN_FIXED && cy != COLUMN_FIXED && cz != COLUMN_FIXED) {
                ux = std::sqrt(std::fabs(s2 * ba.cofactor(cx, cx)));
                uy = std::sqrt(std::fabs(s2 * ba.cofactor(cy, cy)));
                uz = std::sqrt(std::fabs(s2 * ba.cofactor(cz, cz)));
            }
            std::printf("%10s\t%+16.5f\t%+16.5f\t%+16.5f\t%+12.5f\t%+12.5f\t%+12.5f\t%c\n", p->getName().c_str(), p->getX().getValue(), p->getY().getValue(),
                        p->getZ().getValue(), ux, uy, uz, p->getName().size() > 3 ? 'n' : 'o');
        }
        std::printf("\n");
        // :165-175 interior orientation and every distortion parameter, "fixed" where the column is Integer.MAX_VALUE
        auto &io = cam.getInteriorOrientation();
        for (UnknownParameter *u : {&io.getPrinciplePointX(), &io.getPrinciplePointY(), &io.getPrincipleDistance()})
            std::printf("%-27s = %+15.10f %s\n", parameterTypeName(u->getParameterType()), u->getValue(), u->getColumn() == COLUMN_FIXED ? "fixed" : "");
        for (auto &model : cam.getDistortionModels())
            for (auto &u : model->parameters()) {
                std::string name = parameterTypeName(u->getParameterType());
                if (u->getOrder() >= 0) name += "(" + std::to_string(u->getOrder()) + ")";
                std::printf("%-27s = %+15.10f %s", name.c_str(), u->getValue(), u->getColumn() == COLUMN_FIXED ? "fixed" : "");
                if (haveD && u->getColumn() != COLUMN_FIXED && u->getColumn() >= 0)      // interior orientation and distortion sit in the leading numRows of Qxx (BA:262)
                    std::printf("   +/- %.10f", std::sqrt(std::fabs(s2 * ba.cofactor(u->getColumn(), u->getColumn()))));   // (not in the Java listing)
                std::printf("\n");
            }
        std::printf("\n");
        // :179-184
        std::printf("Number of observations:           %d\n", ba.getNumberOfObservations());
        std::printf("Number of unknown parameters:     %d\n", ba.getNumberOfUnknownParameters());
        std::printf("Degree of freedom:                %d\n", ba.getDegreeOfFreedom());
        std::printf("Variances of unit weight:         1.0 : %.15g\n", ba.getVarianceFactorAposteriori() / ba.getVarianceFactorApriori());
        std::printf("Variances of unit weight (ratio): %.15g : %.15g\n", ba.getVarianceFactorApriori(), ba.getVarianceFactorAposteriori());
        std::printf("Iterations:                       %d\n", ba.getIterations());
        std::printf("Estimation time:                  %.3f sec\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        return 0;
    } catch (const std::exception &ex) {
        std::fprintf(stderr, "error: %s\n", ex.what());
        return 3;
    }
}
