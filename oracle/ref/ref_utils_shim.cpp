// oracle/_ref/liblgr_ref_utils.so -- the REFERENCE's own std-only sources, compiled where they lie under /root/reference
// (src/utils.cpp, src/csv_parser.cpp, include/utils.h, include/csv_parser.h; recipe: oracle/Makefile target `ref`), behind
// this extern "C" shim so the tests can call them through ctypes.  Test infrastructure: it pins the oracle's restatements of
//   UniformRandIntGenerator          include/utils.h:13-26   (RANSAC sample stream, src/sac_prerejective_omp.cpp:192-199)
//   calculateCombinationOrMax<int>   include/utils.h:34-43   (iteration cap, src/sac_prerejective_omp.cpp:130)
//   combineHash<int> / <float>       include/utils.h:28-32   (HashEigen voxel order include/common.h:212-223, PointHash :202-210)
//   quantile / mean / stddev         include/utils.h:45-90
//   split, saveVector (ostream << float formatting), rassert, CSVRow (src/csv_parser.cpp)
// Nothing of the reference is copied into this repository: this file only CALLS the reference's functions.  The rest of the
// reference (PCL / OpenCV / Eigen / FLANN / yaml-cpp) is not buildable in this image.
#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "utils.h"
#include "csv_parser.h"

namespace {
int put(const std::vector<std::string>& tok, char* out, int cap) {   // tokens joined by '\x1f'; returns the token count or -1
    std::string s;
    for (size_t i = 0; i < tok.size(); ++i) { if (i) s += '\x1f'; s += tok[i]; }
    if ((int) s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int) tok.size();
}
}  // namespace

extern "C" {
void ref_rng_stream(int lo, int hi, unsigned seed, int n, int* out) {
    UniformRandIntGenerator rand(lo, hi, seed);
    for (int i = 0; i < n; ++i) out[i] = rand();
}
int ref_comb_or_max_int(int n, int k) { return calculateCombinationOrMax<int>(n, k); }
unsigned long long ref_combine_hash_int(unsigned long long seed, int v) { std::size_t s = seed; combineHash(s, v); return s; }
unsigned long long ref_combine_hash_float(unsigned long long seed, float v) { std::size_t s = seed; combineHash(s, v); return s; }
float ref_quantile_float(double q, const float* v, int n) { return quantile<float>(q, std::vector<float>(v, v + n)); }
float ref_mean_float(const float* v, int n) { return calculateMean<float>(std::vector<float>(v, v + n)); }
float ref_stddev_float(const float* v, int n) { return calculateStandardDeviation<float>(std::vector<float>(v, v + n)); }
int ref_split(const char* str, const char* delim, char* out, int cap) {
    std::vector<std::string> tok;
    split(str, tok, delim);
    return put(tok, out, cap);
}
// every row of `text` through `stream >> CSVRow` exactly as the reference's readers loop (src/common.cpp:90-101): rows are
// separated by '\x1e' in the output, fields by '\x1f'; returns the number of rows
int ref_csv_rows(const char* text, char* out, int cap) {
    std::istringstream in(text);
    CSVRow row;
    std::string all;
    int rows = 0;
    while (in >> row) {
        if (rows) all += '\x1e';
        for (std::size_t i = 0; i < row.size(); ++i) { if (i) all += '\x1f'; all += row[i]; }
        ++rows;
    }
    if ((int) all.size() + 1 > cap) return -1;
    std::memcpy(out, all.c_str(), all.size() + 1);
    return rows;
}
void ref_save_vector_float(const float* v, int n, const char* path) { saveVector<float>(std::vector<float>(v, v + n), path); }
void ref_save_vector_double(const double* v, int n, const char* path) { saveVector<double>(std::vector<double>(v, v + n), path); }
// 0 = the assertion held, 1 = it threw std::runtime_error (message copied to `msg`)
int ref_rassert(int condition, char* msg, int cap) {
    try { rassert(condition, 42); } catch (const std::runtime_error& e) { std::strncpy(msg, e.what(), cap - 1); msg[cap - 1] = 0; return 1; }
    return 0;
}
}
