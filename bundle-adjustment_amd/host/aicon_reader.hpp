// aicon_reader.hpp -- readers for the AICON 3D Studio flat files of JAICOV/example (mirror of
// org.applied_geodesy.util.io.reader.aicon.{IORFileReader, EORFileReader, OBCFileReader, PHCFileReader, ScaleFileReader}).
// Line handling follows LockFileReader.java:69-103: UTF-8 BOM stripped, blank lines and lines starting with '#' skipped,
// fields split on white space; malformed lines are dropped (the Java readers print the exception and continue).
#pragma once
#include <fstream>
#include <map>
#include <sstream>

#include "jaicov.hpp"

namespace jaicov::host {

inline std::vector<std::string> split_ws(const std::string &line) {
    std::istringstream is(line);
    std::vector<std::string> out;
    std::string t;
    while (is >> t) out.push_back(t);
    return out;
}

template <typename F>
inline void for_each_line(const std::string &path, F &&fn) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Error, could not find source file: " + path);
    std::string line;
    bool first = true;
    while (std::getline(in, line)) {
        if (first && line.size() >= 3 && (unsigned char)line[0] == 0xEF && (unsigned char)line[1] == 0xBB && (unsigned char)line[2] == 0xBF)
            line = line.substr(3);
        first = false;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.find_first_not_of(" \t") == std::string::npos) continue;
        if (line[0] == '#') continue;
        try { fn(line); } catch (const std::exception &) { /* the Java readers print the stack trace and go on */ }
    }
}

// Owns everything that was read; one camera (the AICON flat format carries one interior orientation)
struct AiconProject {
    AiconProject() = default;
    AiconProject(const AiconProject &) = delete;
    std::unique_ptr<Camera> camera;
    std::vector<std::unique_ptr<ObjectCoordinate>> points;
    std::map<std::string, ObjectCoordinate *> byName;
    std::vector<std::unique_ptr<ScaleBar>> scaleBars;
    std::vector<std::unique_ptr<ObservationParameter>> observations;
    std::vector<std::unique_ptr<DirectlyObservedParameterGroup>> groups;
};

// IORFileReader.java:95-206: camera id, -ck, xh, yh, A1, A2, R0 / A3 / B1 B2 / C1 C2 / sensor; every parameter set FREE
inline void read_ior(AiconProject &pr, const std::string &path, std::vector<DistortionModel::Type> extra = {}) {
    std::vector<DistortionModel::Type> types = {DistortionModel::Type::RADIAL_DISTORTION, DistortionModel::Type::TANGENTIAL_DISTORTION,
                                                DistortionModel::Type::AFFINITY_AND_SHEAR};
    for (auto t : extra)
        if (std::find(types.begin(), types.end(), t) == types.end()) types.push_back(t);
    static const size_t LINE_LENGTHS[5] = {8, 1, 2, 2, 4};
    int lineCounter = 0;
    for_each_line(path, [&](const std::string &line) {
        auto col = split_ws(line);
        if (lineCounter >= 5 || col.size() < LINE_LENGTHS[lineCounter] || (lineCounter > 0 && !pr.camera)) return;
        switch (lineCounter++) {
        case 0: {
            const long camid = std::stol(col[0]);
            const double c = std::stod(col[2]), x0 = std::stod(col[3]), y0 = std::stod(col[4]);
            const double A1 = std::stod(col[5]), A2 = std::stod(col[6]), r0 = std::stod(col[7]);
            pr.camera.reset(new Camera(camid, r0, types));
            auto &io = pr.camera->getInteriorOrientation();
            io.getPrincipleDistance().setValue(-c); io.getPrincipleDistance().setColumn(COLUMN_NOT_SET);
            io.getPrinciplePointX().setValue(x0); io.getPrinciplePointX().setColumn(COLUMN_NOT_SET);
            io.getPrinciplePointY().setValue(y0); io.getPrinciplePointY().setColumn(COLUMN_NOT_SET);
            auto *rad = pr.camera->getDistortionModel(DistortionModel::Type::RADIAL_DISTORTION);
            auto *a1 = rad->add(1); a1->setValue(A1); a1->setColumn(COLUMN_NOT_SET);
            auto *a2 = rad->add(2); a2->setValue(A2); a2->setColumn(COLUMN_NOT_SET);
            break;
        }
        case 1: {
            auto *a3 = pr.camera->getDistortionModel(DistortionModel::Type::RADIAL_DISTORTION)->add(3);
            a3->setValue(std::stod(col[0])); a3->setColumn(COLUMN_NOT_SET);
            break;
        }
        case 2: {
            auto *t = pr.camera->getDistortionModel(DistortionModel::Type::TANGENTIAL_DISTORTION);
            t->getBx()->setValue(std::stod(col[0])); t->getBx()->setColumn(COLUMN_NOT_SET);
            t->getBy()->setValue(std::stod(col[1])); t->getBy()->setColumn(COLUMN_NOT_SET);
            break;
        }
        case 3: {
            auto *a = pr.camera->getDistortionModel(DistortionModel::Type::AFFINITY_AND_SHEAR);
            a->getCx()->setValue(std::stod(col[0])); a->getCx()->setColumn(COLUMN_NOT_SET);
            a->getCy()->setValue(std::stod(col[1])); a->getCy()->setColumn(COLUMN_NOT_SET);
            break;
        }
        default: break;
        }
    });
    if (!pr.camera) throw std::runtime_error("no interior orientation in " + path);
}

// EORFileReader.java:70-128: image, camera, X0 Y0 Z0, omega phi kappa, rotation order (0 = CAP), status, orientation state
inline void read_eor(AiconProject &pr, const std::string &path) {
    for_each_line(path, [&](const std::string &line) {
        auto col = split_ws(line);
        if (col.size() < 11) return;
        const long camid = std::stol(col[1]);
        const bool capRotation = col[8] == "0", enable = col[9] != "0", orient = col[10] != "1";
        if (!enable || !capRotation || !orient || camid != pr.camera->getId()) return;
        Image *im = pr.camera->add(std::stol(col[0]));
        auto &eo = im->getExteriorOrientation();
        for (int i = 0; i < 6; i++) eo.at(i)->setValue(std::stod(col[2 + i]));
    });
}

// OBCFileReader.java:73-111: name X Y Z sx sy sz rays status new datum
inline void read_obc(AiconProject &pr, const std::string &path) {
    for_each_line(path, [&](const std::string &line) {
        auto col = split_ws(line);
        if (col.size() < 4) return;
        const bool enable = col.size() < 11 || col[8] != "0";
        if (!enable) return;
        auto *oc = new ObjectCoordinate(col[0], std::stod(col[1]), std::stod(col[2]), std::stod(col[3]));
        pr.points.emplace_back(oc);
        pr.byName[col[0]] = oc;
    });
}

// PHCFileReader.java:74-117: image, point, x, y, sx, sy, vx, vy, method, status, internal
inline void read_phc(AiconProject &pr, const std::string &path) {
    for_each_line(path, [&](const std::string &line) {
        auto col = split_ws(line);
        if (col.size() < 11) return;
        if (!(std::stoi(col[9]) > 0)) return;
        const long imgid = std::stol(col[0]);
        const double xp = std::stod(col[2]), yp = std::stod(col[3]), sx = std::stod(col[4]), sy = std::stod(col[5]);
        Image *im = pr.camera->add(imgid);
        auto it = pr.byName.find(col[1]);
        if (it != pr.byName.end()) im->add(it->second, xp, yp, sx, sy);
    });
}

// ScaleFileReader.java:77-110:  0 "Scalebar" 506 507 1389.6880 0.0100 1
inline void read_scale(AiconProject &pr, const std::string &path) {
    for_each_line(path, [&](const std::string &raw) {
        std::string line = raw;
        const size_t pos = line.rfind('"');
        if (pos != std::string::npos) line = line.substr(pos + 1);
        auto col = split_ws(line);
        if (col.size() < 5) return;
        const bool enable = col[4] != "0";
        auto a = pr.byName.find(col[0]), b = pr.byName.find(col[1]);
        if (!enable || a == pr.byName.end() || b == pr.byName.end()) return;
        pr.scaleBars.emplace_back(new ScaleBar(a->second, b->second, std::stod(col[2]), std::stod(col[3])));
    });
}

// ExampleFlatFiles.java:76-103 order: obc, scale, ior, eor, phc
inline std::unique_ptr<AiconProject> read_aicon_flat(const std::string &basepath) {
    std::unique_ptr<AiconProject> pr(new AiconProject());
    read_obc(*pr, basepath + ".obc");
    read_ior(*pr, basepath + ".ior");
    read_scale(*pr, basepath + ".scale");
    read_eor(*pr, basepath + ".eor");
    read_phc(*pr, basepath + ".phc");
    return pr;
}

}  // namespace jaicov::host
