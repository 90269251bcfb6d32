// orc_features.cpp -- ORACLE (test infrastructure): bounding box, voxel downsample, k-NN normals, FPFH.
// Each function cites the reference file:line it restates (paths relative to /root/reference).
#include <omp.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <unordered_map>
#include <vector>

#include "../lgr_oracle.h"
#include "orc_grid.h"
#include "orc_math.h"
#include "orc_libm.h"

using namespace orc;

// Arithmetic mode of the third-party pieces (normals eigen-solver, pair-feature swap test / atan2, FPFH weighting) -- a bit mask
// (lgr_oracle.h), so that every piece can be switched alone and its effect on the north-star observables measured
// (tools/pcl_order_report.py --by-piece -> profiles/r5_pcl_order_by_piece*.json):
//   ORC_ARITH_CANONICAL (default; EIGEN33 | ACOS | ATAN2): what the HIP library restates bit for bit in its default mode.  Since round 5
//     the normals and the pair features ARE PCL 1.12.1's own sequences: pcl::eigen33's closed form and the acosf / atan2f / cosf / sinf
//     of glibc 2.35, restated op for op in orc_libm.h and pinned against the running libm (tests/test_oracle_libm.py).  Only the FPFH
//     weighting keeps its own order (one fused chain per bin in grid order = what v_mfma_f32_16x16x4_f32 computes).
//   ORC_ARITH_PCL (all bits): also the weighting as PCL writes it (neighbours by ascending distance, val = hist * w rounded then added,
//     double block sums of the vals) = lgr_ctx_options.arithmetic LGR_ARITH_PCL; tests/test_gpu_pcl_arith.py compares the two bit for bit.
//   ORC_ARITH_ROUND4 (0): the canonical orders of rounds 1-4 (Jacobi normals, comparison of the arguments, own atan2 polynomial); measurement only.
static int g_arith_mode = ORC_ARITH_CANONICAL;
extern "C" void orc_set_arith_mode(int mode) { g_arith_mode = mode & ORC_ARITH_PCL; }   // bit mask: single deviations can be switched alone
extern "C" int orc_arith_mode(void) { return g_arith_mode; }

extern "C" int orc_num_threads(void) { return omp_get_max_threads(); }
extern "C" void orc_set_num_threads(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); }

// include/common.h:266-280 calculateBoundingBox<PointT>: min corner starts at FLT_MAX, max corner starts at
// numeric_limits<float>::min() (smallest POSITIVE float -- a quirk of the reference, reproduced), std::min/max.
extern "C" int orc_bbox(const float* pts, int n, float* mn3, float* mx3) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float mx[3] = {FLT_MIN, FLT_MIN, FLT_MIN};
    for (int i = 0; i < n; ++i) {
        const float* p = pts + 12 * (size_t) i;
        for (int a = 0; a < 3; ++a) {
            mn[a] = (p[a] < mn[a]) ? p[a] : mn[a];  // std::min(mn, p)
            mx[a] = (mx[a] < p[a]) ? p[a] : mx[a];  // std::max(mx, p)
        }
    }
    for (int a = 0; a < 3; ++a) { mn3[a] = mn[a]; mx3[a] = mx[a]; }
    return 0;
}

namespace {
struct Key3 { int x, y, z; bool operator==(const Key3& o) const { return x == o.x && y == o.y && z == o.z; } };
// include/common.h:212-223 HashEigen<Eigen::Vector3i>; std::hash<int> is the identity in libstdc++.
struct HashKey3 {
    std::size_t operator()(const Key3& k) const {
        std::size_t seed = 0;
        const int e[3] = {k.x, k.y, k.z};
        for (int i = 0; i < 3; ++i) seed ^= std::hash<int>()(e[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};
}  // namespace
extern "C" uint64_t orc_voxel_hash(int ix, int iy, int iz) { return (uint64_t) HashKey3()(Key3{ix, iy, iz}); }   // pinned against oracle/_ref
namespace {
// include/downsample.h:6-30 AccumulatedPoint
struct Acc {
    float x = 0, y = 0, z = 0, w = 0, nx = 0, ny = 0, nz = 0;
    void add(const float* p) {
        float weight = p[8];
        x += weight * p[0]; y += weight * p[1]; z += weight * p[2];
        w += weight;
        nx += weight * p[4]; ny += weight * p[5]; nz += weight * p[6];
    }
    void avg(float* o) const {
        float weight = w;
        float ax = nx / weight, ay = ny / weight, az = nz / weight;
        float norm = std::sqrt(ax * ax + ay * ay + az * az);
        norm = norm < 1e-5 ? 1.f : norm;  // double literal compare as in the reference
        o[0] = x / weight; o[1] = y / weight; o[2] = z / weight; o[3] = 1.f;
        o[4] = ax / norm; o[5] = ay / norm; o[6] = az / norm; o[7] = 0.f;
        o[8] = weight; o[9] = 0.f; o[10] = 0.f; o[11] = 0.f;
    }
};
}  // namespace

// src/downsample.cpp:5-41.  order = ORC_ORDER_LIBSTDCXX reproduces the reference output order literally (iteration
// order of std::unordered_map with HashEigen under this container's libstdc++); ORC_ORDER_CANONICAL emits voxels
// sorted by (iz, iy, ix) -- the order the HIP path produces; sums inside a voxel are in input order in both.
extern "C" int orc_downsample(const float* pts, int n, float voxel, int order, float* out, int* n_out) {
    if (!(voxel > 0.f)) return -1;
    float mn[3], mx[3];
    orc_bbox(pts, n, mn, mx);
    float bound[3];
    for (int a = 0; a < 3; ++a) bound[a] = mn[a] - voxel * 0.5f;
    auto key_of = [&](const float* p) {
        Key3 k;
        k.x = (int) std::floor((p[0] - bound[0]) / voxel);
        k.y = (int) std::floor((p[1] - bound[1]) / voxel);
        k.z = (int) std::floor((p[2] - bound[2]) / voxel);
        return k;
    };
    int cnt = 0;
    if (order == ORC_ORDER_LIBSTDCXX) {
        std::unordered_map<Key3, Acc, HashKey3> m;
        for (int i = 0; i < n; ++i) {
            const float* p = pts + 12 * (size_t) i;
            if (!finite3(p)) continue;  // DefaultPointRepresentation<PointN>::isValid: first 3 floats finite
            m[key_of(p)].add(p);
        }
        for (const auto& kv : m) kv.second.avg(out + 12 * (size_t) cnt++);
    } else {
        struct Less {
            bool operator()(const Key3& a, const Key3& b) const {
                if (a.z != b.z) return a.z < b.z;
                if (a.y != b.y) return a.y < b.y;
                return a.x < b.x;
            }
        };
        std::map<Key3, Acc, Less> m;
        for (int i = 0; i < n; ++i) {
            const float* p = pts + 12 * (size_t) i;
            if (!finite3(p)) continue;
            m[key_of(p)].add(p);
        }
        for (const auto& kv : m) kv.second.avg(out + 12 * (size_t) cnt++);
    }
    *n_out = cnt;
    return 0;
}

// exact k-NN table
extern "C" int orc_knn(const float* qpts, int nq, const float* pts, int n, int k, int* idx, float* d2) {
    Grid g;
    g.build(pts, n, auto_cell(pts, n, 4.f));
#pragma omp parallel
    {
        std::vector<Grid::Cand> c(k);
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < nq; ++i) {
            int f = g.knn(qpts + 12 * (size_t) i, k, c.data());
            for (int j = 0; j < k; ++j) {
                idx[(size_t) i * k + j] = j < f ? c[j].idx : -1;
                d2[(size_t) i * k + j] = j < f ? c[j].d2 : INFINITY;
            }
        }
    }
    return 0;
}

// src/common.cpp:644-655 estimateNormalsPoints + :593-628 postprocessNormals, around pcl::NormalEstimationOMP with
// setKSearch(k) [3P, PCL 1.12.1 features/impl/normal_3d_omp.hpp + common/impl/centroid.hpp]:
//   neighbours = k nearest in the surface, sorted by (d2, index);
//   covariance: single pass, shifted by the first neighbour K, float accumulators, accu /= n,
//               cov = E[(p-K)(p-K)^T] - E[p-K]E[p-K]^T                    (computeMeanAndCovarianceMatrix)
//   normal    : eigenvector of the smallest eigenvalue.  PCL uses pcl::eigen33 (closed form with trig); the oracle
//               substitutes the canonical one-sided Jacobi SVD of the symmetric matrix (orc_math.h) -- DEVIATION, ~1e-6
//   curvature : |lambda_min / trace(cov)| (0 if trace == 0)                 (solvePlaneParameters)
//   flip      : if dot(vp - p, n) < 0 negate                                 (flipNormalTowardsViewpoint)
//   fewer than 3 neighbours or non-finite query -> NaN normal and curvature.
//   postprocessNormals: normals_available orients by the stored normal (compares the point with itself in the
//   reference, src/common.cpp:597-598, i.e. a no-op unless NaN); then unit-normalise finite normals.
namespace {
// pcl::eigen33(mat, eigenvalue, eigenvector) -- smallest eigenvalue and its eigenvector of a symmetric 3x3 [3P, PCL 1.12.1
// common/impl/eigen.hpp: computeRoots / computeRoots2 / detail::getLargest3x3Eigenvector], all in float with libm, as
// pcl::solvePlaneParameters calls it.  C row-major (symmetric).
inline void pcl_roots2(float b, float c, float r[3]) {
    r[0] = 0.f;
    float d = (float) (b * b - 4.0 * c);
    if (d < 0.0) d = 0.0;
    float sd = std::sqrt(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}
inline void pcl_roots(const float m[9], float r[3]) {
    float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    float c2 = m[0] + m[4] + m[8];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) { pcl_roots2(c2, c1, r); return; }
    const float s_inv3 = (float) (1.0 / 3.0);
    const float s_sqrt3 = std::sqrt(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    float rho = std::sqrt(-a_over_3);
    float theta = glibc235::atan2f_(std::sqrt(-q), half_b) * s_inv3;   // std::atan2 / std::cos / std::sin on floats: glibc 2.35's routines (orc_libm.h)
    float cos_theta = glibc235::cosf_(theta), sin_theta = glibc235::sinf_(theta);
    r[0] = c2_over_3 + 2.f * rho * cos_theta;
    r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    if (r[0] >= r[1]) std::swap(r[0], r[1]);
    if (r[1] >= r[2]) { std::swap(r[1], r[2]); if (r[0] >= r[1]) std::swap(r[0], r[1]); }
    if (r[0] <= 0) pcl_roots2(c2, c1, r);
}
inline void pcl_eigen33(const float C[9], float& eigenvalue, float v[3]) {
    float scale = 0.f;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(C[i]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.f;
    float m[9];
    for (int i = 0; i < 9; ++i) m[i] = C[i] / scale;
    float r[3];
    pcl_roots(m, r);
    eigenvalue = r[0] * scale;
    m[0] -= r[0]; m[4] -= r[0]; m[8] -= r[0];
    auto cross = [](const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    float cp[3][3];
    cross(m, m + 3, cp[0]); cross(m, m + 6, cp[1]); cross(m + 3, m + 6, cp[2]);
    int best = 0;
    float len = -1.f;
    for (int i = 0; i < 3; ++i) {
        float l = std::sqrt(cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1] + cp[i][2] * cp[i][2]);
        if (l > len) { len = l; best = i; }      // maxCoeff: first maximum
    }
    for (int i = 0; i < 3; ++i) v[i] = cp[best][i] / len;
}
}  // namespace

extern "C" int orc_normals_knn(float* pts, int n, const float* surf, int ns, int k, const float* vp, int normals_available) {
    std::vector<float> copy;
    const float* S = surf;
    if (!S) { copy.assign(pts, pts + 12 * (size_t) n); S = copy.data(); ns = n; }
    Grid g;
    g.build(S, ns, auto_cell(S, ns, 4.f));
    float vpx = vp ? vp[0] : 0.f, vpy = vp ? vp[1] : 0.f, vpz = vp ? vp[2] : 0.f;
    (void) normals_available;  // no-op in the reference (point compared with itself)
#pragma omp parallel
    {
        std::vector<Grid::Cand> c(k);
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < n; ++i) {
            float* p = pts + 12 * (size_t) i;
            const float nanv = std::numeric_limits<float>::quiet_NaN();
            int f = finite3(p) ? g.knn(p, k, c.data()) : 0;
            if (f < 3) { p[4] = p[5] = p[6] = nanv; p[9] = nanv; continue; }
            const float* K = S + 12 * (size_t) c[0].idx;
            float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < f; ++j) {
                const float* q = S + 12 * (size_t) c[j].idx;
                float x = q[0] - K[0], y = q[1] - K[1], z = q[2] - K[2];
                a[0] += x * x; a[1] += x * y; a[2] += x * z; a[3] += y * y; a[4] += y * z; a[5] += z * z;
                a[6] += x; a[7] += y; a[8] += z;
            }
            float fn = (float) f;
            for (int j = 0; j < 9; ++j) a[j] /= fn;
            float C[9];
            C[0] = a[0] - a[6] * a[6]; C[1] = a[1] - a[6] * a[7]; C[2] = a[2] - a[6] * a[8];
            C[4] = a[3] - a[7] * a[7]; C[5] = a[4] - a[7] * a[8]; C[8] = a[5] - a[8] * a[8];
            C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
            float nx, ny, nz, lambda_min;
            if (g_arith_mode & ORC_ARITH_PCL_EIGEN33) {
                float v[3];
                pcl_eigen33(C, lambda_min, v);   // (atan2f / cosf / sinf inside: orc_libm.h)
                nx = v[0]; ny = v[1]; nz = v[2];
            } else {
                float U[9], Sg[3], V[9];
                c_svd3(C, U, Sg, V);
                nx = V[2]; ny = V[5]; nz = V[8];  // column 2 = smallest singular value
                lambda_min = Sg[2];
            }
            float eig_sum = C[0] + C[4] + C[8];
            float curv = (eig_sum != 0.f) ? std::fabs(lambda_min / eig_sum) : 0.f;
            float dx = vpx - p[0], dy = vpy - p[1], dz = vpz - p[2];
            float cos_theta = (dx * nx + dy * ny + dz * nz);
            if (cos_theta < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
            // postprocessNormals: renormalise
            if (std::isfinite(nx) && std::isfinite(ny) && std::isfinite(nz)) {
                float norm = std::sqrt(nx * nx + ny * ny + nz * nz);
                nx /= norm; ny /= norm; nz /= norm;
            }
            p[4] = nx; p[5] = ny; p[6] = nz; p[9] = curv;
        }
    }
    return 0;
}

// ---------------------------------------------------------------- FPFH
namespace {
// pcl::computePairFeatures [3P, PCL 1.12.1 features/src/pfh.cpp]; see SURVEY.md A.1.  p1/n1 = query point of the
// SPFH row, p2/n2 = neighbour.  Returns false when the pair is skipped.  All float; Eigen 4-vector dot/norm on SSE
// reduce as (a0+a2)+(a1+a3) with a3 = 0  ->  (x*x + z*z) + y*y  (documented choice, see DESIGN.md "orders").
inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[2] * b[2]) + a[1] * b[1]; }
inline bool pair_features(const float* p1, const float* n1, const float* p2, const float* n2, bool libm,
                          float& f1, float& f2, float& f3) {
    float d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    float f4 = std::sqrt(dot3(d, d));
    if (f4 == 0.0f) return false;
    float angle1 = dot3(n1, d) / f4;
    float angle2 = dot3(n2, d) / f4;
    const float *u = n1, *m2 = n2;
    float a1 = std::fabs(angle1), a2 = std::fabs(angle2);
    // reference: if (acos(|angle1|) > acos(|angle2|)) swap.  acos is decreasing on [0,1] and NaN outside, so this is
    // restated as (|a1| <= 1 && |a2| <= 1 && |a1| < |a2|)  (no libm call; differs from libm only when two distinct
    // arguments round to the same acosf value).
    const bool swap = (g_arith_mode & ORC_ARITH_PCL_ACOS) ? (glibc235::acosf_(a1) > glibc235::acosf_(a2))   // acosf on floats, as PCL writes it (orc_libm.h)
                                                      : (a1 <= 1.0f && a2 <= 1.0f && a1 < a2);
    if (swap) {
        u = n2; m2 = n1;
        d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2];
        f3 = -angle2;
    } else {
        f3 = angle1;
    }
    // v = d x u
    float v[3] = {d[1] * u[2] - d[2] * u[1], d[2] * u[0] - d[0] * u[2], d[0] * u[1] - d[1] * u[0]};
    float v_norm = std::sqrt(dot3(v, v));
    if (v_norm == 0.0f) return false;
    v[0] /= v_norm; v[1] /= v_norm; v[2] /= v_norm;   // PCL: v /= v_norm; Eigen 3.3+ divides every component (no reciprocal)
    // w = u x v
    float w[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    f2 = dot3(v, m2);
    float yy = dot3(w, m2), xx = dot3(u, m2);
    f1 = (g_arith_mode & ORC_ARITH_PCL_ATAN2) ? glibc235::atan2f_(yy, xx) : libm ? std::atan2(yy, xx) : c_atan2f(yy, xx);
    return true;
}

// bin index computed in double exactly as pcl::FPFHEstimation::computePointSPFHSignature does
// ((f + M_PI) * d_pi_ with d_pi_ a float; (f + 1.0) * 0.5); NaN -> x86 cvttsd2si gives INT_MIN -> clamped to 0.
inline int bin11(double t) {
    double v = std::floor(11 * t);
    if (!(v == v)) return 0;
    if (v < 0) return 0;
    if (v >= 11) return 10;
    return (int) v;
}

void spfh_row(const Grid& g, const float* S, int p, float r2, bool libm, float* row /*33*/) {
    const float* P = S + 12 * (size_t) p;
    int cnt[33];
    for (int b = 0; b < 33; ++b) cnt[b] = 0;
    int k = 0;
    if (finite3(P)) g.visit27(P, [&](int q) { if (dist2(P, S + 12 * (size_t) q) < r2) ++k; });
    for (int b = 0; b < 33; ++b) row[b] = 0.f;
    if (k == 0) return;
    const float d_pi = 1.0f / (2.0f * static_cast<float>(M_PI));
    g.visit27(P, [&](int q) {
        if (q == p) return;
        const float* Q = S + 12 * (size_t) q;
        if (!(dist2(P, Q) < r2)) return;
        float f1, f2, f3;
        if (!pair_features(P, P + 4, Q, Q + 4, libm, f1, f2, f3)) return;
        cnt[bin11((f1 + M_PI) * d_pi)]++;
        cnt[11 + bin11((f2 + 1.0) * 0.5)]++;
        cnt[22 + bin11((f3 + 1.0) * 0.5)]++;
    });
    // hist[b] += hist_incr, repeated cnt[b] times: every increment is the same float, so the value depends on the
    // count only (order-free); it is the sequential float sum of cnt copies of hist_incr.
    float incr = 100.0f / static_cast<float>(k - 1);
    for (int b = 0; b < 33; ++b) { float s = 0.f; for (int t = 0; t < cnt[b]; ++t) s += incr; row[b] = s; }
}
}  // namespace

extern "C" int orc_spfh(const float* surf, int n, float radius, float* out33, int libm_mode) {
    Grid g;
    g.build(surf, n, radius * 1.001f);
    float r2 = radius * radius;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) spfh_row(g, surf, i, r2, libm_mode != 0, out33 + 33 * (size_t) i);
    return 0;
}

// include/common.h:322-332 estimateFeatures<FPFH> -> pcl::FPFHEstimationOMP::computeFeature [3P, PCL 1.12.1
// features/impl/fpfh_omp.hpp + fpfh.hpp]; SURVEY.md A.1.
//   radius search: strict d2 < r*r (FLANN RadiusResultSet), d2 = ((dx*dx)+dy*dy)+dz*dz.
//   neighbour visiting order: PCL visits by ascending distance; the oracle visits by (grid cell z,y,x; index) with
//   cell = 1.001*r anchored at the surface AABB min -- DEVIATION in float summation order only (documented).
//   weight = 1.0f / d2 (squared distance, IEEE division).  CANONICAL weighting (DEVIATION from PCL in float rounding only, see
//   DESIGN.md section 4): PCL rounds val = hist * weight, adds it to the float bin and to a double block sum, in its
//   ascending-distance neighbour order.  Here every bin is ONE fused chain over the accepted neighbours in grid order,
//       fp[b] = fmaf(weight, hist[b], fp[b])            (one rounding per neighbour)
//   -- which is bit for bit what the gfx950 f32 MFMA computes (v_mfma_f32_16x16x4_f32 = k-ordered fmaf chain), so the HIP
//   path can run the weighting W(keypoints x neighbours) . SPFH(neighbours x 33) on the matrix cores -- and the block
//   normaliser is taken from the finished bins: sum_j = sum over the block's 11 bins, ascending, in double;
//   fpfh[b] = float(double(fp[b]) * (100.0 / sum_j)).
extern "C" int orc_fpfh(const float* kps, int m, const float* surf, int n, float radius, float* out33, int libm_mode) {
    Grid g;
    g.build(surf, n, radius * 1.001f);
    float r2 = radius * radius;
    // SPFH only for surface points that are within r of some keypoint (same set as PCL's spfh_indices)
    std::vector<uint8_t> need(n, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < m; ++i) {
        const float* P = kps + 12 * (size_t) i;
        if (!finite3(P)) continue;
        g.visit27(P, [&](int q) { if (dist2(P, surf + 12 * (size_t) q) < r2) need[q] = 1; });
    }
    std::vector<float> spfh((size_t) n * 33, 0.f);
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i)
        if (need[i]) spfh_row(g, surf, i, r2, libm_mode != 0, spfh.data() + 33 * (size_t) i);
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < m; ++i) {
        const float* P = kps + 12 * (size_t) i;
        float* o = out33 + 33 * (size_t) i;
        int found = 0;
        float fp[33];
        for (int b = 0; b < 33; ++b) fp[b] = 0.f;
        if (g_arith_mode & ORC_ARITH_PCL_WEIGHTING) {
            // pcl::FPFHEstimation::weightPointSPFHSignature [3P, PCL 1.12.1 features/impl/fpfh.hpp]: radiusSearch returns the neighbours
            // by ascending squared distance (sorted results; ties: index); per neighbour and bin val = hist * weight (rounded to float),
            // sum_f += val in double, fpfh[bin] += val in float; finally fpfh[bin] * (100.0 / sum_f) in double, stored as float.
            // The three pieces (W_ORDER: the visiting order, W_ROUND: rounded product then add instead of one fused chain, W_NORM: the
            // normaliser from the running double sum of the vals instead of the finished bins) can be switched alone.
            const bool w_order = g_arith_mode & ORC_ARITH_PCL_W_ORDER, w_round = g_arith_mode & ORC_ARITH_PCL_W_ROUND, w_norm = g_arith_mode & ORC_ARITH_PCL_W_NORM;
            std::vector<std::pair<float, int>> nb;
            if (finite3(P)) g.visit27(P, [&](int q) { float d2 = dist2(P, surf + 12 * (size_t) q); if (d2 < r2) nb.emplace_back(d2, q); });
            if (nb.empty()) { for (int b = 0; b < 33; ++b) o[b] = std::numeric_limits<float>::quiet_NaN(); continue; }
            if (w_order) std::sort(nb.begin(), nb.end());
            double sum[3] = {0, 0, 0};
            for (const auto& e : nb) {
                if (e.first == 0.f) continue;
                float weight = 1.0f / e.first;
                const float* h = spfh.data() + 33 * (size_t) e.second;
                for (int b = 0; b < 33; ++b) {
                    float val = h[b] * weight;
                    sum[b / 11] += val;
                    fp[b] = w_round ? fp[b] + val : std::fmaf(weight, h[b], fp[b]);
                }
            }
            if (!w_norm) { sum[0] = sum[1] = sum[2] = 0; for (int b = 0; b < 33; ++b) sum[b / 11] += (double) fp[b]; }
            for (int s3 = 0; s3 < 3; ++s3) if (sum[s3] != 0) sum[s3] = 100.0 / sum[s3];
            for (int b = 0; b < 33; ++b) o[b] = (float) ((double) fp[b] * sum[b / 11]);
            continue;
        }
        if (finite3(P)) {
            g.visit27(P, [&](int q) {
                float d2 = dist2(P, surf + 12 * (size_t) q);
                if (!(d2 < r2)) return;
                ++found;
                if (d2 == 0.f) return;
                float weight = 1.0f / d2;
                const float* h = spfh.data() + 33 * (size_t) q;
                for (int b = 0; b < 33; ++b) fp[b] = std::fmaf(weight, h[b], fp[b]);
            });
        }
        if (found == 0) {
            for (int b = 0; b < 33; ++b) o[b] = std::numeric_limits<float>::quiet_NaN();
            continue;
        }
        double sum[3] = {0, 0, 0};
        for (int b = 0; b < 33; ++b) sum[b / 11] += (double) fp[b];
        for (int s = 0; s < 3; ++s) if (sum[s] != 0) sum[s] = 100.0 / sum[s];
        for (int b = 0; b < 33; ++b) o[b] = (float) ((double) fp[b] * sum[b / 11]);
    }
    return 0;
}

// ---- orc_libm.h against the libm this process runs on (tests/test_oracle_libm.py) and element-wise evaluation (tests compare the HIP
// library's restatement, lgr_selfcheck_libm, with these)
extern "C" int orc_libm_eval(int fn, const float* a, const float* b, long n, float* out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = glibc235::acosf_(a[i]); break;
            case 1: out[i] = glibc235::atanf_(a[i]); break;
            case 2: out[i] = glibc235::atan2f_(a[i], b[i]); break;
            case 3: out[i] = glibc235::sinf_(a[i]); break;
            default: out[i] = glibc235::cosf_(a[i]); break;
        }
    }
    return 0;
}
// every float with bits in [lo_bits, hi_bits] (fn 0 acosf, 1 atanf, 3 sinf, 4 cosf) against std:: of the running libm: returns the number of
// results that differ in any bit (two NaNs count as equal)
extern "C" long orc_libm_check_range(int fn, uint32_t lo_bits, uint32_t hi_bits) {
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long bb = lo_bits; bb <= (long long) hi_bits; ++bb) {
        const float x = glibc235::wfu((uint32_t) bb);
        float r, l;
        switch (fn) {
            case 0: r = glibc235::acosf_(x); l = std::acos(x); break;
            case 1: r = glibc235::atanf_(x); l = std::atan(x); break;
            case 3: r = glibc235::sinf_(x); l = std::sin(x); break;
            default: r = glibc235::cosf_(x); l = std::cos(x); break;
        }
        if (glibc235::fw(r) != glibc235::fw(l) && !(r != r && l != l)) ++bad;
    }
    return bad;
}
extern "C" long orc_libm_check_atan2(const float* y, const float* x, long n) {
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long i = 0; i < n; ++i) {
        const float r = glibc235::atan2f_(y[i], x[i]), l = std::atan2(y[i], x[i]);
        if (glibc235::fw(r) != glibc235::fw(l) && !(r != r && l != l)) ++bad;
    }
    return bad;
}
