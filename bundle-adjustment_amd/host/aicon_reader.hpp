// aicon_reader.hpp -- readers for the AICON 3D Studio flat files of JAICOV/example (mirror of
// org.applied_geodesy.util.io.reader.aicon.{IORFileReader, EORFileReader, OBCFileReader, PHCFileReader, ScaleFileReader}).
// Line handling follows LockFileReader.java:69-103: UTF-8 BOM stripped, blank lines and lines starting with '#' skipped,
// fields split on white space; malformed lines are dropped (the Java readers print the exception and continue).
#pragma once
#include <fstream>
#include <map>
#include <regex>
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
inline void for_each_line(const std::string &path, F &&fn, bool skip_hash = true) {
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
        if (skip_hash && line[0] == '#') continue;
        try { fn(line); } catch (const std::exception &) { /* the Java readers print the stack trace and go on */ }
    }
}

// Owns everything that was read; one camera (the AICON flat format carries one interior orientation)
struct AiconProject {
    AiconProject() = default;
    AiconProject(const AiconProject &) = delete;
    std::unique_ptr<Camera> camera;                              // flat files: the one interior orientation
    std::vector<std::unique_ptr<Camera>> cameras;                // report file: every "Kamera/R0" block, in file order
    std::map<long, Camera *> cameraById;
    std::map<long, Image *> imageById;
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
// `extra`: distortion model types the camera carries beside the three of the .ior file (IORFileReader's varargs constructor,
// ExampleDistortionModel.java:77: ZERNIKE_GRADIENT, ZERNIKE_X, ZERNIKE_Y)
inline std::unique_ptr<AiconProject> read_aicon_flat(const std::string &basepath, std::vector<DistortionModel::Type> extra = {}) {
    std::unique_ptr<AiconProject> pr(new AiconProject());
    read_obc(*pr, basepath + ".obc");
    read_ior(*pr, basepath + ".ior", extra);
    read_scale(*pr, basepath + ".scale");
    read_eor(*pr, basepath + ".eor");
    read_phc(*pr, basepath + ".phc");
    return pr;
}

// ---- AICON 3D Studio adjustment report (.htm), mirror of AICONReportFileReader.java ------------------------------------
// One pass over the lines (LockFileReader.java:69-103 without a comment prefix); a line first switches the section
// (AICONReportFileReader.java:134-153), then goes to that section's parser; a line that fails to parse is dropped
// (:174-178).  Numbers are parsed strictly (whole token), as Double.parseDouble / Integer.parseInt do.
inline double parse_double_strict(const std::string &t) {
    size_t pos = 0;
    const double v = std::stod(t, &pos);
    if (pos != t.size()) throw std::invalid_argument("NumberFormatException: " + t);
    return v;
}
inline long parse_int_strict(const std::string &t) {
    size_t pos = 0;
    const long v = std::stol(t, &pos);
    if (pos != t.size()) throw std::invalid_argument("NumberFormatException: " + t);
    return v;
}
inline std::string trim_java(const std::string &s) {   // String.trim(): code points <= ' ' off both ends
    size_t a = 0, b = s.size();
    while (a < b && (unsigned char)s[a] <= ' ') a++;
    while (b > a && (unsigned char)s[b - 1] <= ' ') b--;
    return s.substr(a, b - a);
}

// datumCoordinates: optional replacement objects by point name (AICONReportFileReader.java:93-99,262-267): without it every
// point is a datum point, with it the listed names are replaced by the caller's objects and the others are not datum.
inline std::unique_ptr<AiconProject> read_aicon_report(const std::string &path,
                                                       const std::map<std::string, ObjectCoordinate *> &datumCoordinates = {}) {
    std::unique_ptr<AiconProject> owner(new AiconProject());
    AiconProject &pr = *owner;
    enum class Content { INTERIOR_ORIENTATION, EXTERIOR_ORIENTATION, OBJECT_COORDINATES, IMAGE_COORDINATES, SCALE_BARS, UNDEFINED };
    Content content = Content::UNDEFINED;
    Camera *camera = nullptr;
    Image *image = nullptr;
    static const std::string NUM = "[\\d\\.+-]+", UNS = "[\\d\\.]+", WS = "\\s+";
    static const std::regex reScale("^\\w+\\s+\\w+\\s+[\\d\\.+-]+.+");                                               // :185
    static const std::regex reImage("^\\w+\\s+\\d+" + (WS + NUM) + (WS + NUM) + (WS + NUM) + (WS + NUM) + (WS + UNS) + (WS + UNS) +
                                    (WS + UNS) + (WS + UNS) + (WS + UNS) + (WS + UNS));                                // :215
    static const std::regex reObject("^\\w+" + (WS + NUM) + (WS + NUM) + (WS + NUM) + (WS + UNS) + (WS + UNS) + (WS + UNS) + "\\s+\\d+\\s+\\d+");   // :246
    static const std::regex reEoXYZ("^\\d+\\s+\\d+" + (WS + NUM) + (WS + NUM) + (WS + NUM) + (WS + UNS) + (WS + UNS) + (WS + UNS) + "\\s+\\d+");      // :274
    static const std::regex reEoAngle("^air\\s+rad" + (WS + NUM) + (WS + NUM) + (WS + NUM) + (WS + UNS) + (WS + UNS) + (WS + UNS) + (WS + UNS) + (WS + UNS));   // :276
    auto has = [](const std::string &l, const char *what) { return l.find(what) != std::string::npos; };
    const auto COL = [](bool fixed) { return fixed ? COLUMN_FIXED : COLUMN_NOT_SET; };

    auto interior = [&](const std::string &line) {                                                                     // :298-390
        if (!has(line, ":")) return;
        std::vector<std::string> col;   // split("[:\\s]+")
        std::string t;
        for (char ch : line) {
            if (ch == ':' || std::isspace((unsigned char)ch)) { if (!t.empty() || col.empty()) { col.push_back(t); t.clear(); } }
            else t.push_back(ch);
        }
        if (!t.empty()) col.push_back(t);
        while (!col.empty() && col.back().empty()) col.pop_back();   // Java drops trailing empty strings
        if (col.size() != 3) return;
        const std::string &type = col[0];
        if (type.size() >= 3 && type.compare(type.size() - 3, 3, "/R0") == 0) {
            const long camId = parse_int_strict(col[1]);
            const double r0 = parse_double_strict(col[2]);
            auto it = pr.cameraById.find(camId);
            Camera *c = new Camera(camId, r0, {DistortionModel::Type::RADIAL_DISTORTION, DistortionModel::Type::TANGENTIAL_DISTORTION,
                                               DistortionModel::Type::AFFINITY_AND_SHEAR, DistortionModel::Type::DISTANCE_DISTORTION});
            if (it != pr.cameraById.end()) {   // LinkedHashMap.put on an existing key keeps its position
                for (auto &u : pr.cameras) if (u.get() == it->second) { u.reset(c); break; }
            } else pr.cameras.emplace_back(c);
            pr.cameraById[camId] = c;
            camera = c;
        }
        if (!camera) return;
        const double value = parse_double_strict(col[1]);
        const bool fixed = std::regex_match(col[2], std::regex("\\w+"));
        auto &io = camera->getInteriorOrientation();
        auto *rad = camera->getDistortionModel(DistortionModel::Type::RADIAL_DISTORTION);
        auto *tan = camera->getDistortionModel(DistortionModel::Type::TANGENTIAL_DISTORTION);
        auto *aff = camera->getDistortionModel(DistortionModel::Type::AFFINITY_AND_SHEAR);
        auto *dis = camera->getDistortionModel(DistortionModel::Type::DISTANCE_DISTORTION);
        auto set = [&](UnknownParameter *u, double v) { u->setValue(v); u->setColumn(COL(fixed)); };
        if (type == "Ck") set(&io.getPrincipleDistance(), -value);
        else if (type == "Xh") set(&io.getPrinciplePointX(), value);
        else if (type == "Yh") set(&io.getPrinciplePointY(), value);
        else if (type == "A1" || type == "A2" || type == "A3") set(rad->add(type[1] - '0'), value);
        else if (type == "B1") set(tan->getBx(), value);
        else if (type == "B2") set(tan->getBy(), value);
        else if (type == "C1") set(aff->getCx(), value);
        else if (type == "C2") set(aff->getCy(), value);
        else if (type == "AZ1" || type == "AZ2" || type == "AZ3") set(dis->add(type[2] - '0'), value);
    };
    auto exterior = [&](const std::string &line) {                                                                     // :272-296
        if (std::regex_match(line, reEoXYZ)) {
            auto col = split_ws(line);
            auto it = pr.cameraById.find(parse_int_strict(col[1]));
            if (it == pr.cameraById.end()) return;
            const long imgId = parse_int_strict(col[0]);
            image = it->second->add(imgId);
            auto &eo = image->getExteriorOrientation();
            for (int i = 0; i < 3; i++) eo.at(i)->setValue(parse_double_strict(col[2 + i]));
            pr.imageById[imgId] = image;
        } else if (image && std::regex_match(line, reEoAngle)) {
            auto col = split_ws(line);
            auto &eo = image->getExteriorOrientation();
            for (int i = 0; i < 3; i++) eo.at(3 + i)->setValue(parse_double_strict(col[2 + i]));
        }
    };
    auto object = [&](const std::string &line) {                                                                       // :241-270
        if (!std::regex_match(line, reObject)) return;
        auto col = split_ws(line);
        if (col.size() != 9) return;
        const std::string &name = col[0];
        const double x = parse_double_strict(col[1]), y = parse_double_strict(col[2]), z = parse_double_strict(col[3]);
        ObjectCoordinate *oc = nullptr;
        auto dit = datumCoordinates.find(name);
        if (!datumCoordinates.empty() && dit != datumCoordinates.end()) oc = dit->second;
        else {
            oc = new ObjectCoordinate(name, x, y, z);
            oc->setDatum(datumCoordinates.empty());
            pr.points.emplace_back(oc);
        }
        pr.byName[name] = oc;
    };
    auto imagecoord = [&](const std::string &line) {                                                                   // :205-239
        if (line.size() >= 3 && line.compare(line.size() - 3, 3, "***") == 0) return;
        if (!std::regex_match(line, reImage)) return;
        auto col = split_ws(line);
        if (col.size() != 12) return;
        const long imgId = parse_int_strict(col[1]);
        auto p = pr.byName.find(col[0]);
        auto im = pr.imageById.find(imgId);
        if (p == pr.byName.end() || im == pr.imageById.end()) return;
        im->second->add(p->second, parse_double_strict(col[2]), parse_double_strict(col[3]), parse_double_strict(col[6]),
                        parse_double_strict(col[7]));
    };
    auto scalebar = [&](const std::string &line) {                                                                     // :180-203
        if (!std::regex_match(line, reScale)) return;
        auto col = split_ws(line);
        if (col.size() < 7) return;
        auto a = pr.byName.find(col[0]), b = pr.byName.find(col[1]);
        if (a == pr.byName.end() || b == pr.byName.end() || col[0] == col[1]) return;
        pr.scaleBars.emplace_back(new ScaleBar(a->second, b->second, parse_double_strict(col[2]), parse_double_strict(col[5])));
    };

    for_each_line(path, [&](const std::string &raw) {
        const std::string line = trim_java(raw);
        if (has(line, "#Start") || has(line, "zum Anfang")) content = Content::UNDEFINED;
        if (has(line, "name=\"interior_orientations\"") || has(line, "*** Innere Orientierungen ***")) content = Content::INTERIOR_ORIENTATION;
        if (has(line, "name=\"exterior_orientations\"") || has(line, "ussere Orientierungen ***")) content = Content::EXTERIOR_ORIENTATION;
        if (has(line, "name=\"object_points\"") || has(line, "*** Objektpunkte ***")) content = Content::OBJECT_COORDINATES;
        if (has(line, "name=\"image_coordinates\"") || has(line, "*** Bildkoordinaten ***")) content = Content::IMAGE_COORDINATES;
        if (has(line, "name=\"distances\"") || has(line, "*** Strecken ***")) content = Content::SCALE_BARS;
        switch (content) {
        case Content::INTERIOR_ORIENTATION: interior(line); break;
        case Content::EXTERIOR_ORIENTATION: exterior(line); break;
        case Content::OBJECT_COORDINATES: object(line); break;
        case Content::IMAGE_COORDINATES: imagecoord(line); break;
        case Content::SCALE_BARS: scalebar(line); break;
        default: break;
        }
    }, false);
    return owner;
}

// readAndImport(): every camera and every scale bar of the report added to a new adjustment (AICONReportFileReader.java:117-128)
inline void import_report(BundleAdjustment &ba, AiconProject &pr) {
    for (auto &c : pr.cameras) ba.add(c.get());
    for (auto &s : pr.scaleBars) ba.add(s.get());
}

}  // namespace jaicov::host
