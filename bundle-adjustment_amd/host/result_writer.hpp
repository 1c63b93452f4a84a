// result_writer.hpp -- the reference's result writers on the engine: mirror of
// org.applied_geodesy.util.io.writer.{AdjustmentResultWritable, BundleAdjustmentResultWriter, DefaultResultWriter,
// MatlabResultWriter}.  The Java writers read the packed 1.3 GB cofactor matrix element by element
// (`cofactor.get(row, column)` over an index list, DefaultResultWriter.java:139-147, MatlabResultWriter.java:210-221); here
// the index list goes to the device, which gathers (and for the text writer scales) the k x k block
// (jaicov_neq_get_cofactor_sub / jaicov_neq_get_dispersion_sub), so the full matrix never crosses PCIe.
//   DefaultResultWriter: <base>.info ("%25s\t%5s\t%35.15f\t%10d%n") and <base>.cxx ("%+35.15f  " per element, sigma2apost * q)
//   MatlabResultWriter : <base>.mat, MAT-file level 5 (uncompressed), the variables, classes and field names of
//                        MatlabResultWriter.java:92-222 (note: its `dispersion` is the UNSCALED cofactor block, :213-217)
#pragma once
#include <charconv>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "jaicov.hpp"

namespace jaicov::host {

class BundleAdjustmentResultWriter : public AdjustmentResultWritable {       // BundleAdjustmentResultWriter.java:24-42
public:
    explicit BundleAdjustmentResultWriter(std::string base) : base_(std::move(base)) {}
    void setExportPathAndFileBaseName(std::string b) { base_ = std::move(b); }
    const std::string &getExportPathAndFileBaseName() const { return base_; }
    std::string toString() const override { return base_; }
protected:
    void requireBase() const {
        if (base_.empty()) throw std::invalid_argument("Error, export path cannot be null!");
    }
    std::string base_;
};

// java.util.Formatter "%.<prec>f" (Locale.ENGLISH): the digits of the shortest decimal that round-trips (Double.toString),
// rounded HALF_UP to prec decimals and zero padded -- NOT the exact binary expansion printf would continue with.
inline std::string java_fixed(double v, int prec, bool plus) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? (plus ? "+Infinity" : "Infinity") : "-Infinity";
    const bool neg = std::signbit(v);
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, std::fabs(v), std::chars_format::scientific);
    std::string sci(buf, r.ptr);                                 // d[.ddd]e[+-]xx
    const size_t epos = sci.find('e');
    int exp10 = std::stoi(sci.substr(epos + 1));
    std::string digits;
    for (size_t i = 0; i < epos; i++)
        if (sci[i] != '.') digits.push_back(sci[i]);
    // value = 0.d1d2d3... * 10^(exp10 + 1)
    int point = exp10 + 1;                                       // number of digits before the decimal point (may be <= 0)
    std::string ip, fp;
    if (point <= 0) { ip = "0"; fp = std::string((size_t)(-point), '0') + digits; }
    else if ((size_t)point >= digits.size()) { ip = digits + std::string((size_t)point - digits.size(), '0'); fp = ""; }
    else { ip = digits.substr(0, (size_t)point); fp = digits.substr((size_t)point); }
    if ((int)fp.size() > prec) {                                 // HALF_UP on the decimal digits
        const bool up = fp[(size_t)prec] >= '5';
        fp.resize((size_t)prec);
        if (up) {
            std::string all = ip + fp;
            int i = (int)all.size() - 1;
            while (i >= 0 && all[(size_t)i] == '9') all[(size_t)i--] = '0';
            if (i >= 0) all[(size_t)i]++;
            else all.insert(all.begin(), '1');
            ip = all.substr(0, all.size() - (size_t)prec);
            fp = all.substr(all.size() - (size_t)prec);
        }
    } else fp += std::string((size_t)prec - fp.size(), '0');
    std::string out = neg ? "-" : (plus ? "+" : "");
    out += ip;
    if (prec > 0) { out += '.'; out += fp; }
    return out;
}
inline std::string pad_left(const std::string &s, size_t width) { return s.size() >= width ? s : std::string(width - s.size(), ' ') + s; }

// ---- DefaultResultWriter.java:40-156 -------------------------------------------------------------------------------------
class DefaultResultWriter : public BundleAdjustmentResultWriter {
public:
    using BundleAdjustmentResultWriter::BundleAdjustmentResultWriter;
    void exportResults(BundleAdjustment &ba) override {
        requireBase();
        const std::vector<int32_t> indices = exportCovarianceInformation(ba, base_ + ".info");
        exportCovarianceMatrix(ba, indices, base_ + ".cxx");
    }

private:
    static std::vector<int32_t> exportCovarianceInformation(BundleAdjustment &ba, const std::string &file) {   // :62-117
        std::ofstream pw(file, std::ios::binary);
        if (!pw) throw std::runtime_error("IOException: " + file);
        std::vector<int32_t> indices;
        int columnIndex = 0;
        for (ObjectCoordinate *oc : ba.getObjectCoordinates()) {
            UnknownParameter *q[3] = {&oc->getX(), &oc->getY(), &oc->getZ()};
            const char type[3] = {'X', 'Y', 'Z'};
            int col[3];
            for (int i = 0; i < 3; i++) {
                col[i] = q[i]->getColumn();
                if (col[i] >= 0 && col[i] < COLUMN_FIXED) { indices.push_back(col[i]); col[i] = columnIndex++; }
                else col[i] = -1;
            }
            for (int i = 0; i < 3; i++)     // "%25s\t%5s\t%35.15f\t%10d%n"
                pw << pad_left(oc->getName(), 25) << '\t' << pad_left(std::string(1, type[i]), 5) << '\t'
                   << pad_left(java_fixed(q[i]->getValue(), 15, false), 35) << '\t' << pad_left(std::to_string(col[i]), 10) << '\n';
        }
        return indices;
    }
    static void exportCovarianceMatrix(BundleAdjustment &ba, const std::vector<int32_t> &indices, const std::string &file) {   // :124-155
        if (!ba.hasCofactorMatrix()) return;      // cofactor == null || numRows < u + d
        const double sigma2apost = ba.getVarianceFactorAposteriori();
        const std::vector<double> D = ba.cofactorSub(indices, sigma2apost);      // sigma2apost * Qxx[idx, idx] from the device
        std::ofstream pw(file, std::ios::binary);
        if (!pw) throw std::runtime_error("IOException: " + file);
        const size_t k = indices.size();
        std::string line;
        for (size_t r = 0; r < k; r++) {
            line.clear();
            for (size_t c = 0; c < k; c++) { line += pad_left(java_fixed(D[r * k + c], 15, true), 35); line += "  "; }
            line += '\n';
            pw << line;
        }
    }
};

// ---- MAT-file level 5 (uncompressed, little endian) ---------------------------------------------------------------------
namespace mat5 {
enum : uint32_t { miINT8 = 1, miUINT16 = 4, miINT32 = 5, miUINT32 = 6, miDOUBLE = 9, miINT64 = 12, miMATRIX = 14 };
enum : uint32_t { mxSTRUCT = 2, mxCHAR = 4, mxDOUBLE = 6, mxINT32 = 12, mxINT64 = 14 };

struct Buf {
    std::string b;
    void u32(uint32_t v) { b.append(reinterpret_cast<const char *>(&v), 4); }
    void raw(const void *p, size_t n) { b.append(static_cast<const char *>(p), n); }
    void pad8() { while (b.size() % 8) b.push_back('\0'); }
    void element(uint32_t type, const void *p, size_t n) { u32(type); u32((uint32_t)n); raw(p, n); pad8(); }
};
inline void header(Buf &o, uint32_t cls, const std::vector<int32_t> &dims, const std::string &name) {
    uint32_t flags[2] = {cls, 0};
    o.element(miUINT32, flags, 8);
    o.element(miINT32, dims.data(), dims.size() * 4);
    o.element(miINT8, name.data(), name.size());
}
inline std::string wrap(const Buf &body) {            // miMATRIX tag around the sub-elements
    if (body.b.size() > 0xFFFFFFFFull) throw std::length_error("MAT-file level 5: array larger than 4 GB");
    Buf o;
    o.u32(miMATRIX); o.u32((uint32_t)body.b.size());
    o.b += body.b;
    return o.b;
}
template <typename T>
inline std::string numeric(const std::string &name, uint32_t cls, uint32_t mi, const T *data, int32_t rows, int32_t cols) {
    Buf o;
    header(o, cls, {rows, cols}, name);
    o.element(mi, data, sizeof(T) * (size_t)rows * (size_t)cols);
    return wrap(o);
}
inline std::string scalarDouble(const std::string &n, double v) { return numeric(n, mxDOUBLE, miDOUBLE, &v, 1, 1); }
inline std::string scalarInt32(const std::string &n, int32_t v) { return numeric(n, mxINT32, miINT32, &v, 1, 1); }
inline std::string scalarInt64(const std::string &n, int64_t v) { return numeric(n, mxINT64, miINT64, &v, 1, 1); }
inline std::string chars(const std::string &name, const std::string &utf8) {      // 1 x n char row vector, UTF-16 code units
    std::vector<uint16_t> u;
    for (size_t i = 0; i < utf8.size();) {
        uint32_t cp = (unsigned char)utf8[i];
        int extra = cp >= 0xF0 ? 3 : cp >= 0xE0 ? 2 : cp >= 0xC0 ? 1 : 0;
        if (extra) cp &= (0x3F >> extra);
        i++;
        for (int j = 0; j < extra && i < utf8.size(); j++, i++) cp = (cp << 6) | ((unsigned char)utf8[i] & 0x3F);
        if (cp >= 0x10000) { cp -= 0x10000; u.push_back((uint16_t)(0xD800 + (cp >> 10))); u.push_back((uint16_t)(0xDC00 + (cp & 0x3FF))); }
        else u.push_back((uint16_t)cp);
    }
    Buf o;
    header(o, mxCHAR, {u.empty() ? 0 : 1, (int32_t)u.size()}, name);
    o.element(miUINT16, u.data(), u.size() * 2);
    return wrap(o);
}
// 1 x n struct array; cells[element][field] are complete miMATRIX elements with empty names
inline std::string structArray(const std::string &name, const std::vector<std::string> &fields,
                               const std::vector<std::vector<std::string>> &cells) {
    Buf o;
    header(o, mxSTRUCT, {1, (int32_t)cells.size()}, name);
    const int32_t flen = 32;                                    // field name length incl. terminator (MATLAB's own value)
    o.u32((4u << 16) | miINT32); o.raw(&flen, 4);               // small data element format
    std::string names((size_t)flen * fields.size(), '\0');
    for (size_t f = 0; f < fields.size(); f++) {
        if (fields[f].size() >= (size_t)flen) throw std::length_error("field name too long");
        std::memcpy(&names[f * flen], fields[f].data(), fields[f].size());
    }
    o.element(miINT8, names.data(), names.size());
    for (const auto &el : cells)
        for (const auto &c : el) o.b += c;
    return wrap(o);
}
// arrays: complete miMATRIX elements; the (possibly multi-GB) double matrix `big` is streamed after them without a copy
inline void writeFile(const std::string &path, const std::vector<std::string> &arrays, const std::string &bigName = "",
                      const double *big = nullptr, int32_t bigRows = 0, int32_t bigCols = 0) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("IOException: " + path);
    char head[128];
    std::memset(head, ' ', 116);
    const char *text = "MATLAB 5.0 MAT-file, Platform: MI355X, Created by: jaicov_neq host (MatlabResultWriter)";
    std::memcpy(head, text, std::strlen(text));
    std::memset(head + 116, 0, 8);
    const uint16_t version = 0x0100, endian = 0x4D49;           // "MI" read back as little endian
    std::memcpy(head + 124, &version, 2);
    std::memcpy(head + 126, &endian, 2);
    f.write(head, 128);
    for (const auto &a : arrays) f.write(a.data(), (std::streamsize)a.size());
    if (big) {
        Buf h;
        header(h, mxDOUBLE, {bigRows, bigCols}, bigName);
        const uint64_t nbytes = 8ull * (uint64_t)bigRows * (uint64_t)bigCols, total = h.b.size() + 8 + nbytes;
        if (total > 0xFFFFFFFFull) throw std::length_error("MAT-file level 5: array larger than 4 GB");
        Buf t;
        t.u32(miMATRIX); t.u32((uint32_t)total);
        t.b += h.b;
        t.u32(miDOUBLE); t.u32((uint32_t)nbytes);
        f.write(t.b.data(), (std::streamsize)t.b.size());
        f.write(reinterpret_cast<const char *>(big), (std::streamsize)nbytes);      // multiples of 8: no padding
    }
    f.flush();
    if (!f) throw std::runtime_error("IOException: " + path);
}
}  // namespace mat5

// ---- MatlabResultWriter.java:46-245 ---------------------------------------------------------------------------------------
class MatlabResultWriter : public BundleAdjustmentResultWriter {
public:
    using BundleAdjustmentResultWriter::BundleAdjustmentResultWriter;
    void exportResults(BundleAdjustment &ba) override {
        requireBase();
        using namespace mat5;
        const bool exportDispersionMatrix = ba.hasCofactorMatrix();                                  // :72
        const int order = ba.getNumberOfUnknownParameters() + ba.getNumberOfDatumConditions();      // cofactor.numColumns()
        std::vector<int32_t> indices;
        int columnIndex = 1;                                                                        // :94 (MATLAB indices)
        auto lower = [](std::string s) { for (auto &c : s) c = (char)std::tolower((unsigned char)c); return s; };

        std::vector<std::vector<std::string>> coords;                                               // :96-142
        for (ObjectCoordinate *oc : ba.getObjectCoordinates()) {
            UnknownParameter *q[3] = {&oc->getX(), &oc->getY(), &oc->getZ()};
            int col[3];
            for (int i = 0; i < 3; i++) {
                col[i] = q[i]->getColumn();
                if (col[i] >= 0 && col[i] < COLUMN_FIXED) { if (exportDispersionMatrix) indices.push_back(col[i]); col[i] = columnIndex++; }
                else col[i] = -1;
            }
            coords.push_back({chars("", oc->getName()), scalarDouble("", q[0]->getValue()), scalarDouble("", q[1]->getValue()),
                              scalarDouble("", q[2]->getValue()), scalarInt32("", col[0]), scalarInt32("", col[1]), scalarInt32("", col[2])});
        }
        auto covIndex = [&](UnknownParameter *up) {                                                 // :150-158, :178-186
            int column = up->getColumn();
            if (column >= 0 && column < order) { indices.push_back(column); return columnIndex++; }
            return -1;
        };
        std::vector<std::vector<std::string>> ios, dist;
        for (Camera *cam : ba.getCameras()) {                                                       // :144-164
            auto &io = cam->getInteriorOrientation();
            for (int i = 0; i < 3; i++) {
                UnknownParameter *up = io.at(i);
                std::vector<std::string> el = {scalarInt64("", cam->getId()), chars("", lower(parameterTypeName(up->getParameterType()))),
                                               scalarDouble("", up->getValue())};
                if (exportDispersionMatrix) el.push_back(scalarInt32("", covIndex(up)));
                ios.push_back(std::move(el));
            }
        }
        for (Camera *cam : ba.getCameras())                                                         // :166-194
            for (auto &m : cam->getDistortionModels())
                for (auto &up : m->parameters()) {
                    std::vector<std::string> el = {scalarInt64("", cam->getId()), chars("", lower(parameterTypeName(up->getParameterType()))),
                                                   scalarDouble("", up->getValue()), scalarInt32("", up->getOrder())};
                    if (exportDispersionMatrix) el.push_back(scalarInt32("", covIndex(up.get())));
                    dist.push_back(std::move(el));
                }

        std::vector<std::string> arrays;                                                            // :197-208
        arrays.push_back(scalarDouble("variance_of_unit_weight_prio", ba.getVarianceFactorApriori()));
        arrays.push_back(scalarDouble("variance_of_unit_weight_post", ba.getVarianceFactorAposteriori()));
        arrays.push_back(scalarInt32("degree_of_freedom", ba.getDegreeOfFreedom()));
        arrays.push_back(scalarInt32("number_of_observations", ba.getNumberOfObservations()));
        arrays.push_back(scalarInt32("number_of_unknowns", ba.getNumberOfUnknownParameters()));
        arrays.push_back(structArray("coordinates", coords.empty() ? std::vector<std::string>{}
                                                                   : std::vector<std::string>{"name", "X", "Y", "Z", "covx", "covy", "covz"}, coords));
        std::vector<std::string> iof = {"cam_id", "name", "value"}, dif = {"cam_id", "name", "value", "order"};
        if (exportDispersionMatrix) { iof.push_back("cov"); dif.push_back("cov"); }
        arrays.push_back(structArray("interior_orientations", ios.empty() ? std::vector<std::string>{} : iof, ios));
        arrays.push_back(structArray("distortion_parameters", dist.empty() ? std::vector<std::string>{} : dif, dist));
        std::vector<double> D;
        if (exportDispersionMatrix) D = ba.cofactorSub(indices, 1.0);                               // :210-222, unscaled; symmetric
        writeFile(base_ + ".mat", arrays, "dispersion", exportDispersionMatrix ? D.data() : nullptr, (int32_t)indices.size(),
                  (int32_t)indices.size());
    }
};

}  // namespace jaicov::host
