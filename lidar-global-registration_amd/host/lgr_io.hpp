// lgr_io.hpp -- the on-disk formats either side of the hot path for the C++ host, header-only and
// std-only, on the shim's types (lgr_compat.hpp).  Same names, argument meaning and error behaviour as the reference's:
//   include/io.h:6-20             loadPLYFile<PointT>(file_name, cloud, fields, offset)   (pcl::PLYReader + fromPCLPointCloud2)
//   include/common.h:465-480      pointCloudHasNormals<PointT>(fields)
//   pcl::io::savePLYFileBinary / savePLYFileASCII   (src/common.cpp:768, 1050: how the reference writes clouds)
//   include/csv_parser.h:11-22    CSVRow, operator>>                                     (src/csv_parser.cpp:5-29)
//   src/utils.cpp:13-24           split
//   src/common.cpp:83-104         getTransformation(csv_path, src_filename, tgt_filename)
//   src/common.cpp:106-125        getTransformation(csv_path, transformation_name)
//   src/common.cpp:127-153        saveTransformation
//   src/common.cpp:1223-1244      readCorrespondencesFromCSV
//   src/common.cpp:1246-1266      saveCorrespondencesToCSV
// lgr_amd/formats.py is the same set for the Python host; tests/test_host_io.py checks the two against each other byte
// for byte (and the CSV tokeniser against the reference's own csv_parser.cpp compiled into oracle/_ref).
//
// PLY: ascii / binary_little_endian / binary_big_endian, any scalar property types, non-vertex elements and unknown
// properties skipped; x y z normal_x|nx normal_y|ny normal_z|nz intensity|scalar_intensity curvature go into the
// 48-byte PointXYZINormal layout, everything else keeps the point type's defaults (PCL's fromPCLPointCloud2 does the same:
// fields the point type has but the file lacks stay as constructed).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>

#include "lgr_compat.hpp"

namespace lgr {

// pcl::PCLPointField (name / offset in the point / PCL datatype code / count): what loadPLYFile hands back in `fields`
struct PCLPointField {
    std::string name;
    std::uint32_t offset = 0;
    std::uint8_t datatype = 7;   // pcl::PCLPointField::FLOAT32
    std::uint32_t count = 1;
};

namespace io_detail {

struct Prop { std::string name; int type = -1; bool list = false; int count_type = -1, item_type = -1; };
struct Element { std::string name; std::size_t count = 0; std::vector<Prop> props; };

// type codes: 0 i8, 1 u8, 2 i16, 3 u16, 4 i32, 5 u32, 6 f32, 7 f64
inline int ply_type(const std::string& t) {
    static const char* names[8][2] = {{"char", "int8"}, {"uchar", "uint8"}, {"short", "int16"}, {"ushort", "uint16"},
                                      {"int", "int32"}, {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"}};
    for (int i = 0; i < 8; ++i) if (t == names[i][0] || t == names[i][1]) return i;
    return -1;
}
inline std::size_t type_size(int t) { static const std::size_t s[8] = {1, 1, 2, 2, 4, 4, 4, 8}; return s[t]; }

inline bool host_little_endian() { const std::uint16_t one = 1; return *reinterpret_cast<const unsigned char*>(&one) == 1; }

// one scalar of PLY type t at p (file byte order `little`) as a double
inline double scalar(const unsigned char* p, int t, bool little) {
    unsigned char b[8];
    const std::size_t n = type_size(t);
    if (little == host_little_endian()) std::memcpy(b, p, n);
    else for (std::size_t i = 0; i < n; ++i) b[i] = p[n - 1 - i];
    switch (t) {
        case 0: { std::int8_t v; std::memcpy(&v, b, 1); return v; }
        case 1: { std::uint8_t v; std::memcpy(&v, b, 1); return v; }
        case 2: { std::int16_t v; std::memcpy(&v, b, 2); return v; }
        case 3: { std::uint16_t v; std::memcpy(&v, b, 2); return v; }
        case 4: { std::int32_t v; std::memcpy(&v, b, 4); return v; }
        case 5: { std::uint32_t v; std::memcpy(&v, b, 4); return v; }
        case 6: { float v; std::memcpy(&v, b, 4); return v; }
        default: { double v; std::memcpy(&v, b, 8); return v; }
    }
}

// float slot of the 12-float PointXYZINormal a vertex property lands in, or -1
inline int field_slot(const std::string& name) {
    static const struct { const char* n; int slot; } map[] = {
        {"x", 0}, {"y", 1}, {"z", 2}, {"normal_x", 4}, {"nx", 4}, {"normal_y", 5}, {"ny", 5}, {"normal_z", 6}, {"nz", 6},
        {"intensity", 8}, {"scalar_intensity", 8}, {"curvature", 9}};
    for (const auto& e : map) if (name == e.n) return e.slot;
    return -1;
}
inline const char* canonical_name(int slot) {
    static const char* names[12] = {"x", "y", "z", "", "normal_x", "normal_y", "normal_z", "", "intensity", "curvature", "", ""};
    return names[slot];
}

inline std::vector<std::string> words(const std::string& line) {
    std::istringstream ss(line);
    std::vector<std::string> w;
    for (std::string t; ss >> t;) w.push_back(t);
    return w;
}

}  // namespace io_detail

// include/io.h:6-20.  Returns 0, or a negative value when the file cannot be read as a PLY point cloud (the reference's
// callers test `< 0`).  `offset`: byte offset of the PLY data in the file, as in pcl::PLYReader::read.
template <typename PointT>
inline int loadPLYFile(const std::string& file_name, Cloud<PointT>& cloud, std::vector<PCLPointField>& fields, const int offset = 0) {
    using namespace io_detail;
    static_assert(sizeof(PointT) == 48, "loadPLYFile maps into the 48-byte PointXYZINormal layout");
    std::ifstream f(file_name, std::ios::binary);
    if (!f.is_open()) return -1;
    if (offset > 0) f.seekg(offset);
    std::string line;
    if (!std::getline(f, line) || words(line).empty() || words(line)[0] != "ply") return -1;
    std::string fmt;
    std::vector<Element> elements;
    bool ended = false;
    while (std::getline(f, line)) {
        std::vector<std::string> w = words(line);
        if (w.empty() || w[0] == "comment" || w[0] == "obj_info") continue;
        if (w[0] == "format" && w.size() >= 2) fmt = w[1];
        else if (w[0] == "element" && w.size() >= 3) { Element e; e.name = w[1]; e.count = std::strtoull(w[2].c_str(), nullptr, 10); elements.push_back(e); }
        else if (w[0] == "property" && !elements.empty()) {
            Prop p;
            if (w.size() >= 5 && w[1] == "list") { p.list = true; p.count_type = ply_type(w[2]); p.item_type = ply_type(w[3]); p.name = w[4]; if (p.count_type < 0 || p.item_type < 0) return -1; }
            else if (w.size() >= 3) { p.type = ply_type(w[1]); p.name = w[2]; if (p.type < 0) return -1; }
            else return -1;
            elements.back().props.push_back(p);
        } else if (w[0] == "end_header") { ended = true; break; }
    }
    if (!ended) return -1;
    const bool ascii = fmt == "ascii", little = fmt == "binary_little_endian";
    if (!ascii && !little && fmt != "binary_big_endian") return -1;

    bool have_vertex = false;
    fields.clear();
    for (const Element& el : elements) {
        bool has_list = false;
        std::size_t rec = 0;
        for (const Prop& p : el.props) { has_list = has_list || p.list; if (!p.list) rec += type_size(p.type); }
        if (el.name != "vertex") {
            if (ascii) { for (std::size_t i = 0; i < el.count; ++i) if (!std::getline(f, line)) return -1; }
            else if (!has_list) f.seekg(static_cast<std::streamoff>(el.count * rec), std::ios::cur);
            else {
                unsigned char b[8];
                for (std::size_t i = 0; i < el.count; ++i)
                    for (const Prop& p : el.props) {
                        if (!p.list) { f.seekg(static_cast<std::streamoff>(type_size(p.type)), std::ios::cur); continue; }
                        if (!f.read(reinterpret_cast<char*>(b), static_cast<std::streamsize>(type_size(p.count_type)))) return -1;
                        f.seekg(static_cast<std::streamoff>(scalar(b, p.count_type, little) * type_size(p.item_type)), std::ios::cur);
                    }
            }
            continue;
        }
        if (has_list) return -1;
        have_vertex = true;
        std::vector<int> slot(el.props.size());
        for (std::size_t k = 0; k < el.props.size(); ++k) {
            slot[k] = field_slot(el.props[k].name);
            if (slot[k] >= 0) fields.push_back(PCLPointField{canonical_name(slot[k]), static_cast<std::uint32_t>(4 * slot[k]), 7, 1});
        }
        cloud.points.assign(el.count, PointT());
        if (ascii) {
            for (std::size_t i = 0; i < el.count; ++i) {
                if (!std::getline(f, line)) return -1;
                float* dst = reinterpret_cast<float*>(&cloud.points[i]);
                const char* s = line.c_str();
                for (std::size_t k = 0; k < el.props.size(); ++k) {
                    char* end = nullptr;
                    const double v = std::strtod(s, &end);
                    if (end == s) return -1;
                    s = end;
                    if (slot[k] >= 0) dst[slot[k]] = static_cast<float>(v);
                }
            }
        } else {
            const std::size_t chunk = 65536;
            std::vector<unsigned char> buf(chunk * rec);
            for (std::size_t base = 0; base < el.count; base += chunk) {
                const std::size_t n = std::min(chunk, el.count - base);
                if (!f.read(reinterpret_cast<char*>(buf.data()), static_cast<std::streamsize>(n * rec))) return -1;
                for (std::size_t i = 0; i < n; ++i) {
                    float* dst = reinterpret_cast<float*>(&cloud.points[base + i]);
                    const unsigned char* p = buf.data() + i * rec;
                    for (std::size_t k = 0; k < el.props.size(); ++k) {
                        if (slot[k] >= 0) dst[slot[k]] = static_cast<float>(scalar(p, el.props[k].type, little));
                        p += type_size(el.props[k].type);
                    }
                }
            }
        }
    }
    if (!have_vertex) return -1;
    cloud.width = static_cast<unsigned>(cloud.points.size());
    cloud.height = 1;
    cloud.is_dense = true;
    return 0;
}

// include/common.h:465-480, including what it actually tests: the z flag is set by a normal_x field too, so a cloud "has
// normals" as soon as normal_x and normal_y are present.
template <typename PointT>
inline bool pointCloudHasNormals(const std::vector<PCLPointField>& fields) {
    bool normal_x = false, normal_y = false, normal_z = false;
    for (const auto& field : fields) {
        if (field.name == "normal_x") { normal_x = true; normal_z = true; }
        if (field.name == "normal_y") normal_y = true;
    }
    return normal_x && normal_y && normal_z;
}

namespace io_detail {
inline int save_ply(const std::string& file_name, const PointNCloud& cloud, bool binary) {
    std::ofstream f(file_name, std::ios::binary);
    if (!f.is_open()) return -1;
    static const int slots[8] = {0, 1, 2, 4, 5, 6, 8, 9};
    f << "ply\nformat " << (binary ? (host_little_endian() ? "binary_little_endian" : "binary_big_endian") : "ascii") << " 1.0\n";
    f << "element vertex " << cloud.points.size() << "\n";
    for (int s : slots) f << "property float " << canonical_name(s) << "\n";
    f << "end_header\n";
    if (binary) {
        std::vector<float> row(8 * 4096);
        for (std::size_t base = 0; base < cloud.points.size(); base += 4096) {
            const std::size_t n = std::min<std::size_t>(4096, cloud.points.size() - base);
            for (std::size_t i = 0; i < n; ++i) {
                const float* src = reinterpret_cast<const float*>(&cloud.points[base + i]);
                for (int k = 0; k < 8; ++k) row[8 * i + k] = src[slots[k]];
            }
            f.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(n * 8 * sizeof(float)));
        }
    } else {
        char buf[64];
        for (const PointN& pt : cloud.points) {
            const float* src = reinterpret_cast<const float*>(&pt);
            for (int k = 0; k < 8; ++k) {
                std::snprintf(buf, sizeof buf, "%.9g", static_cast<double>(src[slots[k]]));   // shortest text that reads back to the same float
                f << (k ? " " : "") << buf;
            }
            f << "\n";
        }
    }
    return f.good() ? 0 : -1;
}
}  // namespace io_detail

// pcl::io::savePLYFileBinary / savePLYFileASCII for PointN clouds: x y z normal_x normal_y normal_z intensity curvature
inline int savePLYFileBinary(const std::string& file_name, const PointNCloud& cloud) { return io_detail::save_ply(file_name, cloud, true); }
inline int savePLYFileASCII(const std::string& file_name, const PointNCloud& cloud) { return io_detail::save_ply(file_name, cloud, false); }

// src/utils.cpp:13-24: split at every occurrence of the delimiter; an empty last piece is dropped, empty inner pieces stay
inline void split(const std::string& str, std::vector<std::string>& tokens, const std::string& delimiter) {
    tokens.clear();
    std::size_t from = 0, pos;
    while ((pos = str.find(delimiter, from)) != std::string::npos) {
        tokens.push_back(str.substr(from, pos - from));
        from = pos + delimiter.length();
    }
    if (from < str.size()) tokens.push_back(str.substr(from));
}

// include/csv_parser.h:11-22, src/csv_parser.cpp:5-29: one line, cut at every ',', no quoting; a trailing comma yields a
// last empty field, a '\r' stays in the last field
class CSVRow {
public:
    std::string operator[](std::size_t index) const { return m_line.substr(m_start[index], m_start[index + 1] - m_start[index] - 1); }
    std::size_t size() const { return m_start.size() - 1; }
    void readNextRow(std::istream& str) {
        std::getline(str, m_line);
        m_start.assign(1, 0);
        for (std::size_t pos = 0; (pos = m_line.find(',', pos)) != std::string::npos; ++pos) m_start.push_back(pos + 1);
        m_start.push_back(m_line.size() + 1);
    }
private:
    std::string m_line;
    std::vector<std::size_t> m_start{0, 1};
};
inline std::istream& operator>>(std::istream& str, CSVRow& data) { data.readNextRow(str); return str; }

namespace io_detail {
inline Matrix4f row_matrix(const CSVRow& row) {
    Matrix4f T;
    for (int i = 0; i < 16; ++i) T(i / 4, i % 4) = std::stof(row[i + 1]);
    return T;
}
inline Matrix4f multiply(const Matrix4f& a, const Matrix4f& b) {
    Matrix4f c;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a(i, k) * b(k, j);
            c(i, j) = s;
        }
    return c;
}
// general 4x4 inverse by cofactors (what a fixed-size Eigen inverse computes, in float)
inline Matrix4f inverse(const Matrix4f& m) {
    float a[16], inv[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a[4 * i + j] = m(i, j);
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    Matrix4f r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r(i, j) = inv[4 * i + j] / det;
    return r;
}
}  // namespace io_detail

// src/common.cpp:83-104: relative pose of two scans from a ground-truth table of absolute poses, tgt^-1 * src; nullopt when
// either row is missing (a later row with the same key replaces an earlier one)
inline std::optional<Matrix4f> getTransformation(const std::string& csv_path, const std::string& src_filename, const std::string& tgt_filename) {
    std::ifstream file(csv_path);
    Matrix4f src_position = Matrix4f::Identity(), tgt_position = Matrix4f::Identity();
    CSVRow row;
    bool success_src = false, success_tgt = false;
    while (file >> row) {
        if (row[0] == src_filename) { src_position = io_detail::row_matrix(row); success_src = true; }
        if (row[0] == tgt_filename) { tgt_position = io_detail::row_matrix(row); success_tgt = true; }
    }
    if (!(success_src && success_tgt)) return std::nullopt;
    return io_detail::multiply(io_detail::inverse(tgt_position), src_position);
}

// src/common.cpp:106-125: the first row whose key matches; a missing key ends the program like the reference (exit 1)
inline Matrix4f getTransformation(const std::string& csv_path, const std::string& transformation_name) {
    std::ifstream file(csv_path);
    CSVRow row;
    while (file >> row)
        if (row[0] == transformation_name) return io_detail::row_matrix(row);
    std::fprintf(stderr, "Failed to get transformation %s!\n", transformation_name.c_str());
    std::exit(1);
}

// src/common.cpp:127-153: append one row (default ostream float formatting); a file that does not exist yet gets the header
inline void saveTransformation(const std::string& csv_path, const std::string& transformation_name, const Matrix4f& transformation) {
    const bool fresh = !std::filesystem::exists(csv_path);
    std::ofstream out(csv_path, fresh ? std::ios::out : std::ios::app);
    if (!out.is_open()) { perror(("error while opening file " + csv_path).c_str()); return; }
    if (fresh) {
        out << "reading";
        for (int i = 0; i < 16; ++i) out << ",gT" << i / 4 << i % 4;
        out << "\n";
    }
    out << transformation_name;
    for (int i = 0; i < 16; ++i) out << "," << transformation(i / 4, i % 4);
    out << "\n";
}

// src/common.cpp:1223-1244: index_query, index_match, distance, threshold from the first four columns of every line after
// the header; `success` is only ever set to true (a missing file leaves it untouched and returns an empty list)
inline CorrespondencesPtr readCorrespondencesFromCSV(const std::string& filepath, bool& success) {
    auto correspondences = std::make_shared<Correspondences>();
    if (!std::filesystem::exists(filepath)) return correspondences;
    std::ifstream fin(filepath);
    if (!fin.is_open()) { perror(("error while opening file " + filepath).c_str()); return correspondences; }
    std::string line;
    std::vector<std::string> tok;
    std::getline(fin, line);
    while (std::getline(fin, line)) {
        split(line, tok, ",");
        correspondences->emplace_back(std::stoi(tok.at(0)), std::stoi(tok.at(1)), std::stof(tok.at(2)), std::stof(tok.at(3)));
    }
    success = true;
    return correspondences;
}

// src/common.cpp:1246-1266: one line per correspondence with the coordinates of its two points
inline void saveCorrespondencesToCSV(const std::string& filepath, const PointNCloud::ConstPtr& src, const PointNCloud::ConstPtr& tgt,
                                     const CorrespondencesConstPtr& correspondences) {
    std::ofstream out(filepath);
    if (!out.is_open()) { perror(("error while opening file " + filepath).c_str()); return; }
    out << "query_idx,match_idx,distance,threshold,x_s,y_s,z_s,x_t,y_t,z_t\n";
    for (const Correspondence& c : *correspondences) {
        const PointN& s = src->points[c.index_query];
        const PointN& t = tgt->points[c.index_match];
        out << c.index_query << ',' << c.index_match << ',' << c.distance << ',' << c.threshold << ','
            << s.x << ',' << s.y << ',' << s.z << ',' << t.x << ',' << t.y << ',' << t.z << '\n';
    }
}

}  // namespace lgr
