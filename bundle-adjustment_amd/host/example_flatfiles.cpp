// Native counterpart of org.applied_geodesy.adjustment.bundle.example.ExampleFlatFiles (ExampleFlatFiles.java:73-230) on
// the MI355X engine: reads the AICON flat files <base>.obc/.scale/.ior/.eor/.phc, fixes A3, Cx, Cy as the example does
// (ExampleFlatFiles.java:89-95), defines the datum on the points with short names, runs estimateModel() through the C ABI
// and prints what the Java example prints.  No Python, no oracle: C++ host mirror + libjaicov_neq.so only.
//   usage: example_flatfiles <base path> [FULL|REDUCED|PRE_ELIMINATION|NONE]
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
    MatrixInversion inv = MatrixInversion::FULL;
    if (argc > 2) {
        if (!std::strcmp(argv[2], "REDUCED")) inv = MatrixInversion::REDUCED;
        else if (!std::strcmp(argv[2], "PRE_ELIMINATION")) inv = MatrixInversion::PRE_ELIMINATION;
        else if (!std::strcmp(argv[2], "NONE")) inv = MatrixInversion::NONE;
    }
    try {
        std::unique_ptr<AiconProject> pr = read_aicon_flat(argv[1]);
        Camera &cam = *pr->camera;
        cam.getDistortionModel(DistortionModel::Type::RADIAL_DISTORTION)->get(3)->setColumn(COLUMN_FIXED);
        cam.getDistortionModel(DistortionModel::Type::AFFINITY_AND_SHEAR)->getCx()->setColumn(COLUMN_FIXED);
        cam.getDistortionModel(DistortionModel::Type::AFFINITY_AND_SHEAR)->getCy()->setColumn(COLUMN_FIXED);
        for (auto &p : pr->points)
            if (p->getName().size() > 3) p->setDatum(false);      // coded targets carry the datum (ExampleReport.java:71-82)
        BundleAdjustment ba;
        ba.add(&cam);
        for (auto &s : pr->scaleBars) ba.add(s.get());
        ba.setInvertNormalEquation(inv);
        ba.addPropertyChangeListener([](const std::string &name, double a, double b) {
            if (name == "CONVERGENCE") std::printf("  max|dx| = %.3e (threshold %.3e)\n", b, a);
        });
        const EstimationStateType state = ba.estimateModel();
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("state                      %d%s\n", (int)state, state == EstimationStateType::ERROR_FREE_ESTIMATION ? " (ERROR_FREE_ESTIMATION)" : "");
        if (!ba.lastError().empty()) std::printf("engine                     %s\n", ba.lastError().c_str());
        std::printf("observations               %d\n", ba.getNumberOfObservations());
        std::printf("unknown parameters         %d\n", ba.getNumberOfUnknownParameters());
        std::printf("datum conditions           %d\n", ba.getNumberOfDatumConditions());
        std::printf("degree of freedom          %d\n", ba.getDegreeOfFreedom());
        std::printf("iterations                 %d\n", ba.getIterations());
        std::printf("omega                      %.10e\n", ba.getOmega());
        std::printf("sigma0 a-posteriori        %.9f\n", std::sqrt(ba.getVarianceFactorAposteriori()));
        auto &io = cam.getInteriorOrientation();
        std::printf("c, x0, y0                  %.6f %.6f %.6f\n", io.getPrincipleDistance().getValue(), io.getPrinciplePointX().getValue(),
                    io.getPrinciplePointY().getValue());
        if (inv != MatrixInversion::NONE && !ba.getObjectCoordinates().empty()) {
            ObjectCoordinate *p = ba.getObjectCoordinates().front();
            const double s2 = ba.getVarianceFactorAposteriori();
            std::printf("point %-8s            %.5f %.5f %.5f  +/- %.5f %.5f %.5f\n", p->getName().c_str(), p->getX().getValue(),
                        p->getY().getValue(), p->getZ().getValue(), std::sqrt(s2 * ba.cofactor(p->getX().getColumn(), p->getX().getColumn())),
                        std::sqrt(s2 * ba.cofactor(p->getY().getColumn(), p->getY().getColumn())),
                        std::sqrt(s2 * ba.cofactor(p->getZ().getColumn(), p->getZ().getColumn())));
        }
        std::printf("Estimation time: %.3f sec\n", secs);       // ExampleFlatFiles.java:230
        return state == EstimationStateType::ERROR_FREE_ESTIMATION ? 0 : 1;
    } catch (const std::exception &ex) {
        std::fprintf(stderr, "error: %s\n", ex.what());
        return 3;
    }
}
