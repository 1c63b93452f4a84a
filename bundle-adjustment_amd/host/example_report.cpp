// Native counterpart of org.applied_geodesy.adjustment.bundle.example.ExampleReport (ExampleReport.java:52-172) on the
// MI355X engine: reads an AICON 3D Studio adjustment report (.htm) with the report reader, takes the datum from the points
// with short names (ExampleReport.java:71-82), adjusts with MatrixInversion.REDUCED (:89) through the C ABI and prints
// the listing of the Java example.  No Python, no oracle: C++ host mirror + libjaicov_neq.so only.
//   usage: example_report <report.htm> [REDUCED|FULL|PRE_ELIMINATION|NONE]
#include <chrono>
#include <cstdio>
#include <cstring>

#include "aicon_reader.hpp"

using namespace jaicov::host;

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <report.htm> [REDUCED|FULL|PRE_ELIMINATION|NONE]\n", argv[0]);
        return 2;
    }
    const auto t0 = std::chrono::steady_clock::now();
    MatrixInversion inv = MatrixInversion::REDUCED;
    if (argc > 2) {
        if (!std::strcmp(argv[2], "FULL")) inv = MatrixInversion::FULL;
        else if (!std::strcmp(argv[2], "PRE_ELIMINATION")) inv = MatrixInversion::PRE_ELIMINATION;
        else if (!std::strcmp(argv[2], "NONE")) inv = MatrixInversion::NONE;
    }
    try {
        std::unique_ptr<AiconProject> pr = read_aicon_report(argv[1]);
        if (pr->cameras.empty()) throw std::runtime_error("no interior orientation in the report");
        for (auto &cam : pr->cameras)
            for (auto &im : cam->images())
                for (auto &ic : im->coordinates())
                    if (ic->getObjectCoordinate()->getName().size() > 3) ic->getObjectCoordinate()->setDatum(false);
        BundleAdjustment ba;
        import_report(ba, *pr);
        ba.setEstimationType(EstimationType::L2NORM);
        ba.setInvertNormalEquation(inv);
        ba.addPropertyChangeListener([](const std::string &name, double a, double b) {
            if (name == "CONVERGENCE") std::printf("  max|dx| = %.3e (threshold %.3e)\n", b, a);
        });
        const EstimationStateType state = ba.estimateModel();
        if (state != EstimationStateType::ERROR_FREE_ESTIMATION) {
            std::fprintf(stderr, "Error, bundle adjustment fails... (state %d) %s\n", (int)state, ba.lastError().c_str());
            return 1;
        }
        std::printf("Bundle adjustment finished successfully...\n");
        const double s2 = ba.getVarianceFactorAposteriori();
        const bool haveD = inv != MatrixInversion::NONE;
        for (ObjectCoordinate *p : ba.getObjectCoordinates()) {
            double u[3] = {0, 0, 0};
            UnknownParameter *q[3] = {&p->getX(), &p->getY(), &p->getZ()};
            bool est = haveD;
            for (int i = 0; i < 3; i++) est = est && q[i]->getColumn() >= 0 && q[i]->getColumn() != COLUMN_FIXED;
            if (est)
                for (int i = 0; i < 3; i++) u[i] = std::sqrt(std::fabs(s2 * ba.cofactor(q[i]->getColumn(), q[i]->getColumn())));
            std::printf("%10s\t%+16.5f\t%+16.5f\t%+16.5f\t%+12.5f\t%+12.5f\t%+12.5f\t%c\n", p->getName().c_str(), q[0]->getValue(),
                        q[1]->getValue(), q[2]->getValue(), u[0], u[1], u[2], p->isDatum() ? 'd' : 'o');
        }
        std::printf("\n");
        for (auto &cam : pr->cameras) {
            auto &io = cam->getInteriorOrientation();
            for (int i = 0; i < 3; i++)
                std::printf("%-27s = %+15.10f %s\n", parameterTypeName(io.at(i)->getParameterType()), io.at(i)->getValue(),
                            io.at(i)->getColumn() == COLUMN_FIXED ? "fixed" : "");
            std::printf("\n");
        }
        for (auto &cam : pr->cameras) {
            for (auto &m : cam->getDistortionModels())
                for (auto &up : m->parameters()) {
                    char nm[64];
                    if (up->getOrder() < 0) std::snprintf(nm, sizeof nm, "%s", parameterTypeName(up->getParameterType()));
                    else std::snprintf(nm, sizeof nm, "%s(%d)", parameterTypeName(up->getParameterType()), up->getOrder());
                    std::printf("%-27s = %+15.10f %s\n", nm, up->getValue(), up->getColumn() == COLUMN_FIXED ? "fixed" : "");
                }
            std::printf("\n");
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("Number of observations:           %d\n", ba.getNumberOfObservations());
        std::printf("Number of unknown parameters:     %d\n", ba.getNumberOfUnknownParameters());
        std::printf("Number of datum conditions:       %d\n", ba.getNumberOfDatumConditions());
        std::printf("Degree of freedom:                %d\n", ba.getDegreeOfFreedom());
        std::printf("Iterations:                       %d\n", ba.getIterations());
        std::printf("Variances of unit weight:         1.0 : %.15g\n", s2 / ba.getVarianceFactorApriori());
        std::printf("Variances of unit weight (ratio): %.15g : %.15g\n", ba.getVarianceFactorApriori(), s2);
        std::printf("Estimation time:                  %.3f sec\n", secs);
        return 0;
    } catch (const std::exception &ex) {
        std::fprintf(stderr, "error: %s\n", ex.what());
        return 3;
    }
}
