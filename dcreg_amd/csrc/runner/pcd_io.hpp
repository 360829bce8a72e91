// PCD v0.7 reader / writer (stands for pcl::io::loadPCDFile<PointXYZI> / savePCDFileBinary,
// DCReg/src/icp_test_runner.cpp:160-164, 369-373).  Reads ascii and binary (not binary_compressed) files with
// float32 x y z [intensity]; writes binary "x y z intensity".
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace pcdio {

struct Cloud {
    std::vector<float> xyzi;   // 4 floats per point
    size_t size() const { return xyzi.size() / 4; }
    bool empty() const { return xyzi.empty(); }
};

inline bool load(const std::string &path, Cloud &out, std::string *err = nullptr) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::vector<std::string> fields, types;
    std::vector<int> sizes, counts;
    size_t npts = 0;
    std::string data_kind, line;
    while (std::getline(f, line)) {
        std::stringstream ss(line);
        std::string tag; ss >> tag;
        std::string tok;
        if (tag == "FIELDS") while (ss >> tok) fields.push_back(tok);
        else if (tag == "SIZE") while (ss >> tok) sizes.push_back(std::stoi(tok));
        else if (tag == "TYPE") while (ss >> tok) types.push_back(tok);
        else if (tag == "COUNT") while (ss >> tok) counts.push_back(std::stoi(tok));
        else if (tag == "POINTS") ss >> npts;
        else if (tag == "DATA") { ss >> data_kind; break; }
    }
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (fields.empty() || sizes.size() != fields.size()) { if (err) *err = "bad PCD header in " + path; return false; }
    int off[4] = {-1, -1, -1, -1};
    int col[4] = {-1, -1, -1, -1};
    size_t rec = 0; int c = 0;
    for (size_t i = 0; i < fields.size(); ++i) {
        const char *names[4] = {"x", "y", "z", "intensity"};
        for (int k = 0; k < 4; ++k) if (fields[i] == names[k]) { off[k] = (int)rec; col[k] = c; }
        rec += (size_t)sizes[i] * counts[i]; c += counts[i];
    }
    if (off[0] < 0 || off[1] < 0 || off[2] < 0) { if (err) *err = "PCD has no x y z fields: " + path; return false; }
    if (data_kind == "binary") {
        // the 4-byte copies below are only meaningful for float32 fields (pcl::PointXYZI): check TYPE F / SIZE 4 / COUNT 1
        for (size_t i = 0; i < fields.size(); ++i) {
            const bool used = fields[i] == "x" || fields[i] == "y" || fields[i] == "z" || fields[i] == "intensity";
            if (!used) continue;
            const bool is_float = types.size() != fields.size() || types[i] == "F";
            if (sizes[i] != 4 || counts[i] != 1 || !is_float) {
                if (err) *err = "PCD field '" + fields[i] + "' is not a float32 scalar (TYPE F, SIZE 4, COUNT 1): " + path;
                return false;
            }
        }
        // the header's POINTS must fit the file: no allocation from an unchecked count
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streamoff remaining = f.tellg() - here;
        f.seekg(here);
        if (rec == 0 || remaining < 0 || (unsigned long long)npts > (unsigned long long)remaining / rec) {
            if (err) *err = "truncated PCD " + path + " (POINTS exceeds the file size)";
            return false;
        }
    } else if (npts > ((size_t)1 << 31)) { if (err) *err = "PCD POINTS out of range in " + path; return false; }
    out.xyzi.assign(npts * 4, 0.f);
    if (data_kind == "binary") {
        std::vector<char> buf(rec * npts);
        f.read(buf.data(), (std::streamsize)buf.size());
        if ((size_t)f.gcount() != buf.size()) { if (err) *err = "truncated PCD " + path; return false; }
        for (size_t i = 0; i < npts; ++i)
            for (int k = 0; k < 4; ++k) if (off[k] >= 0) std::memcpy(&out.xyzi[i * 4 + k], &buf[i * rec + off[k]], 4);
    } else if (data_kind == "ascii") {
        for (size_t i = 0; i < npts; ++i) {
            if (!std::getline(f, line)) { if (err) *err = "truncated PCD " + path; return false; }
            std::stringstream ss(line);
            std::vector<double> v; double d;
            while (ss >> d) v.push_back(d);
            for (int k = 0; k < 4; ++k) if (col[k] >= 0 && col[k] < (int)v.size()) out.xyzi[i * 4 + k] = (float)v[col[k]];
        }
    } else { if (err) *err = "unsupported PCD DATA kind '" + data_kind + "' in " + path; return false; }
    return true;
}

inline bool save_binary(const std::string &path, const float *xyzi, size_t n) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
      << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    f.write(reinterpret_cast<const char *>(xyzi), (std::streamsize)(n * 4 * sizeof(float)));
    return (bool)f;
}

// pcl::PointXYZRGB as pcl::io::savePCDFileBinary writes it: x y z + packed 0xAARRGGBB (alpha 255) in one uint32 field
inline bool save_binary_rgb(const std::string &path, const std::vector<float> &xyz, const std::vector<uint32_t> &rgba) {
    const size_t n = rgba.size();
    std::ofstream f(path, std::ios::binary);
    if (!f || xyz.size() != 3 * n) return false;
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
      << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    for (size_t i = 0; i < n; ++i) {
        f.write(reinterpret_cast<const char *>(&xyz[3 * i]), 3 * sizeof(float));
        f.write(reinterpret_cast<const char *>(&rgba[i]), sizeof(uint32_t));
    }
    return (bool)f;
}

inline uint32_t pack_rgb(unsigned r, unsigned g, unsigned b) { return 0xFF000000u | (r << 16) | (g << 8) | b; }

}  // namespace pcdio
