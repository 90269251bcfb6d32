// Host-side formats through the C++ header (lgr_io.hpp), driven by tests/test_host_io.py, which wrote the inputs with
// lgr_amd/formats.py and compares what this program writes byte for byte.  No GPU, no library call.
//   io_roundtrip <dir>
//     <dir>/py_bin.ply, py_ascii.ply, foreign_ascii.ply, foreign_be.ply   -> re-saved as cpp_*.ply (binary and ascii)
//     <dir>/py_t.csv (poses a.ply, b.ply)    -> cpp_t.csv with the same rows re-saved + the relative pose row "rel"
//     <dir>/py_corr.csv                      -> cpp_corr.csv (coordinates from py_bin.ply for both clouds)
//     <dir>/lines.txt                        -> tokens of every line as CSVRow cuts them, one "n|tok|tok|..." line each
#include <cstdio>
#include <fstream>

#include "../../lidar-global-registration_amd/host/lgr_io.hpp"

using namespace lgr;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string d = std::string(argv[1]) + "/";
    std::vector<PCLPointField> fields;
    auto cloud = std::make_shared<PointNCloud>();
    const char* names[4] = {"py_bin", "py_ascii", "foreign_ascii", "foreign_be"};
    for (const char* n : names) {
        PointNCloud c;
        if (loadPLYFile<PointN>(d + n + ".ply", c, fields) < 0) { std::printf("failed to load %s\n", n); return 1; }
        std::printf("%s: %zu points, normals=%d, fields=", n, c.size(), (int) pointCloudHasNormals<PointN>(fields));
        for (const auto& f : fields) std::printf("%s ", f.name.c_str());
        std::printf("\n");
        if (savePLYFileBinary(d + "cpp_" + n + "_bin.ply", c) < 0 || savePLYFileASCII(d + "cpp_" + n + "_ascii.ply", c) < 0) return 1;
        if (std::string(n) == "py_bin") *cloud = c;
    }
    PointNCloud none;
    std::printf("missing file -> %d, not a ply -> %d\n", loadPLYFile<PointN>(d + "nope.ply", none, fields), loadPLYFile<PointN>(d + "lines.txt", none, fields));

    Matrix4f A = getTransformation(d + "py_t.csv", "a.ply"), B = getTransformation(d + "py_t.csv", "b.ply");
    saveTransformation(d + "cpp_t.csv", "a.ply", A);
    saveTransformation(d + "cpp_t.csv", "b.ply", B);
    auto rel = getTransformation(d + "py_t.csv", "a.ply", "b.ply");
    auto missing = getTransformation(d + "py_t.csv", "a.ply", "zzz.ply");
    if (!rel.has_value() || missing.has_value()) return 1;
    saveTransformation(d + "cpp_t.csv", "rel", *rel);

    bool ok = false, untouched = false;
    CorrespondencesPtr corr = readCorrespondencesFromCSV(d + "py_corr.csv", ok);
    CorrespondencesPtr nothing = readCorrespondencesFromCSV(d + "nope.csv", untouched);
    if (!ok || untouched || !nothing->empty()) return 1;
    saveCorrespondencesToCSV(d + "cpp_corr.csv", cloud, cloud, corr);

    std::ifstream lines(d + "lines.txt");
    std::ofstream out(d + "cpp_tokens.txt");
    CSVRow row;
    while (lines >> row) {
        out << row.size();
        for (std::size_t i = 0; i < row.size(); ++i) out << "|" << row[i];
        out << "\n";
    }
    std::vector<std::string> tok;
    split("a,,b,", tok, ",");
    std::printf("split: %zu\n", tok.size());
    return tok.size() == 3 ? 0 : 1;
}
