// orc_gror.cpp -- ORACLE (test infrastructure): GROR initial alignment (BASELINE config 5).
// Restates include/gror/ia_gror.hpp + ia_gror.h (vendored WHU GROR, driven by alignGror, src/alignment.cpp:21-35:
// resolution = distance_thr, K_optimal = 800, best_count = 3).  Reference paths relative to /root/reference.
//
// Canonical choices where the reference is implementation defined (DEVIATIONS, documented in DESIGN.md):
//   * the three std::sort calls (node degrees ia_gror.hpp:176, graph rows :202, interval ends :568) have no tie
//     order in the standard; the oracle uses std::stable_sort (ties keep the original order);
//   * clearReduentPoints (:29-78) only renumbers points; indices are kept (no observable effect on the result);
//   * Transform::rotation() (:479) is the 3x3 block of the matrix (the reference runs an SVD-based polar
//     decomposition on a matrix that is already a rotation);
//   * pcl::umeyama (:314) with sequential sums in correspondence order and the canonical Jacobi SVD (orc_math.h);
//   * an empty graph leaves the reference's matrices uninitialised (UB); the oracle returns the refit of identity.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "../lgr_oracle.h"
#include "orc_math.h"

using namespace orc;

namespace {
struct V3 { float x, y, z; };
inline V3 ld(const float* pts, int i) { const float* p = pts + 12 * (size_t) i; return V3{p[0], p[1], p[2]}; }
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
inline float dist(V3 a, V3 b) { return norm(sub(a, b)); }   // pcl::geometry::distance
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct M3 { float m[3][3]; };
inline M3 ident() { M3 r{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = i == j ? 1.f : 0.f; return r; }
inline V3 mul(const M3& R, V3 v) { return V3{(R.m[0][0] * v.x + R.m[0][1] * v.y) + R.m[0][2] * v.z, (R.m[1][0] * v.x + R.m[1][1] * v.y) + R.m[1][2] * v.z, (R.m[2][0] * v.x + R.m[2][1] * v.y) + R.m[2][2] * v.z}; }
inline M3 mul(const M3& A, const M3& B) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = (A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j]) + A.m[i][2] * B.m[2][j]; return r; }
inline M3 transpose(const M3& A) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = A.m[j][i]; return r; }

// ia_gror.hpp:450-478 SkewSymmetric + twoVectorsAlign
inline M3 two_vectors_align(V3 a, V3 b) {
    V3 v = cross(a, b);
    float c = dot(a, b);
    M3 K{};
    K.m[0][1] = -1.0f * v.z; K.m[0][2] = v.y; K.m[1][0] = v.z; K.m[1][2] = -1.0f * v.x; K.m[2][0] = -1.0f * v.y; K.m[2][1] = v.x;
    M3 K2 = mul(K, K);
    float f = 1.0f / (1.0f + c);
    M3 R = ident();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R.m[i][j] = (R.m[i][j] + K.m[i][j]) + K2.m[i][j] * f;
    return R;
}
struct Rigid { M3 R; V3 t; };
// pcl::transformPointCloudWithNormals -> pcl::detail::Transformer<float>::se3 on SSE (PCL 1.12.1 common/impl/transforms.hpp):
// p0 + (p1 + (p2 + c3)) with p_k = coord_k * column_k
inline V3 apply(const Rigid& T, V3 p) {
    return V3{T.R.m[0][0] * p.x + (T.R.m[0][1] * p.y + (T.R.m[0][2] * p.z + T.t.x)), T.R.m[1][0] * p.x + (T.R.m[1][1] * p.y + (T.R.m[1][2] * p.z + T.t.y)),
              T.R.m[2][0] * p.x + (T.R.m[2][1] * p.y + (T.R.m[2][2] * p.z + T.t.z))};
}
inline Rigid compose(const Rigid& A, const Rigid& B) {   // A * B
    Rigid r; r.R = mul(A.R, B.R); V3 t = mul(A.R, B.t); r.t = V3{t.x + A.t.x, t.y + A.t.y, t.z + A.t.z}; return r;
}
inline V3 normalized(V3 v) { float n = norm(v); return n > 0.f ? V3{v.x / n, v.y / n, v.z / n} : v; }   // Eigen normalized(): z > 0 guard

// ia_gror.h:291-311
inline float vl_fast_atan2_f(float y, float x) {
    float angle, r;
    float const c3 = 0.1821F, c1 = 0.9675F;
    float abs_y = std::abs(y);
    if (x >= 0) { r = (x - abs_y) / (x + abs_y); angle = (float) (3.1415926f / 4); }
    else { r = (x + abs_y) / (abs_y - x); angle = (float) (3 * 3.1415926f / 4); }
    angle += (c3 * r * r - c1) * r;
    return (y < 0) ? -angle : angle;
}
// :517-552
inline double circle_intersection(double R, double d, double r) {
    if (d <= 1e-12) return M_PI;
    double x = (d * d - r * r + R * R) / (2 * d);
    double rat = x / R;
    if (rat <= -1.0) return M_PI;
    return std::acos(rat);
}
struct IntervalEnd { double location; bool is_start; int corr_idx; };

// :555-617 intervalStab, one_to_one = true (ACTab has no influence on the outputs in that branch)
void interval_stab(std::vector<IntervalEnd>& ia, double& out_angle, int& out_upbnd) {
    int curr_upbnd = 0, NOEnd = 0;
    out_upbnd = 0;
    std::stable_sort(ia.begin(), ia.end(), [](const IntervalEnd& a, const IntervalEnd& b) { return a.location < b.location; });
    double currLoc = 0;
    for (size_t i = 0; i < ia.size(); i++) {
        if (ia[i].is_start) {
            curr_upbnd++;
            if (curr_upbnd > out_upbnd) { out_upbnd = curr_upbnd; out_angle = ia[i].location; }
        } else NOEnd++;
        if (ia[i].location > currLoc) { curr_upbnd -= NOEnd; NOEnd = 0; currLoc = ia[i].location; }
    }
}
}  // namespace

// ia_gror.hpp:126-170 node degrees: for every unordered pair, |dist_s - dist_t| < 2.0 * resolution (double compare)
extern "C" int orc_gror_node_degree(const float* src, const float* tgt, const lgr_orc_corr* corr, int c, float resolution, int* degree) {
    std::vector<V3> S(c), T(c);
    for (int i = 0; i < c; ++i) { S[i] = ld(src, corr[i].query); T[i] = ld(tgt, corr[i].match); }
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < c; ++i) {
        int d = 0;
        for (int j = 0; j < c; ++j) {
            if (j == i) continue;
            float delta = std::abs(dist(S[i], S[j]) - dist(T[i], T[j]));
            if (delta < 2.0 * resolution) ++d;
        }
        degree[i] = d;
    }
    return 0;
}

// computeTransformation (:367-415).  T16 column-major.  diag (optional, 8 ints): [0] K used [1] best_count
// [2] rows evaluated in TCFS [3] n_inliers of the refine step
extern "C" int orc_gror(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
                        float resolution, int K_optimal, float T16[16], int* diag, float* best_angle_out) {
    (void) ns; (void) nt;
    std::vector<int> degree(c);
    orc_gror_node_degree(src, tgt, corr, c, resolution, degree.data());
    std::vector<int> order(c);
    for (int i = 0; i < c; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return degree[a] > degree[b]; });   // sortByVoteNumber
    int K = c >= K_optimal ? K_optimal : c;
    std::vector<int> sel(K);
    if (c >= K_optimal) for (int i = 0; i < K; ++i) sel[i] = order[i];
    else for (int i = 0; i < K; ++i) sel[i] = i;                   // output = input (original order) :183-185
    std::vector<V3> S(K), T(K);
    for (int i = 0; i < K; ++i) { S[i] = ld(src, corr[sel[i]].query); T[i] = ld(tgt, corr[sel[i]].match); }
    // enumeratePairOfCorrespondence :82-124
    std::vector<std::vector<int>> graph(K);   // graph[i] = list of j > i
    for (int i = 0; i < K; ++i)
        for (int j = i + 1; j < K; ++j) {
            float delta = std::abs(dist(S[i], S[j]) - dist(T[i], T[j]));
            if (delta < (2.0 * resolution)) graph[i].push_back(j);
        }
    std::vector<int> rows(K);
    for (int i = 0; i < K; ++i) rows[i] = i;
    std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return graph[a].size() > graph[b].size(); });   // sortByNumber
    // obtainMaximumConsistentSetBasedOnEdgeReliability :199-259
    int best_count = 3;                      // GRORInitialAlignment() : best_count_(3)
    Rigid best_T{ident(), V3{0, 0, 0}};
    V3 best_axis{0, 0, 1}, best_origin{0, 0, 0};
    float best_angle = 0.f;
    int tcfs_rows = 0;
    const float two_res = 2 * resolution;    // float (ia_gror.hpp:492)
    for (int ri = 0; ri < K; ++ri) {
        int i = rows[ri];
        if (graph[i].size() < 10) continue;
        int j = graph[i][0];
        V3 first_s = S[i], first_t = T[i], second_s = S[j], second_t = T[j];
        // twoPairPointsAlign(first_t, first_s, second_t, second_s) :417-448
        V3 vec_sour = normalized(sub(first_s, second_s)), vec_tart = normalized(sub(first_t, second_t));
        V3 rot_axis = vec_tart, rot_origin = first_t;
        Rigid mat;
        mat.R = two_vectors_align(vec_sour, vec_tart);
        V3 r1 = mul(mat.R, first_s), r2 = mul(mat.R, second_s);
        V3 tv1 = sub(first_t, r1), tv2 = sub(second_t, r2);
        mat.t = V3{0.5f * (tv1.x + tv2.x), 0.5f * (tv1.y + tv2.y), 0.5f * (tv1.z + tv2.z)};
        // calEdgeReliabilityInRCFS :472-501
        V3 rot_axis_s = mul(transpose(mat.R), rot_axis);
        int der_rcfs = 0;
        for (int k = 0; k < K; ++k) {
            V3 dt = sub(T[k], first_t), dsv = sub(S[k], first_s);
            float dis_t = norm(dt), dis_s = norm(dsv);
            if (std::abs(dis_t - dis_s) < two_res && std::abs(dot(dt, rot_axis) - dot(dsv, rot_axis_s)) < two_res) der_rcfs++;
        }
        if (der_rcfs <= best_count) continue;
        // calEdgeReliabilityInTCFS :620-747
        ++tcfs_rows;
        Rigid tm_t;                                   // IdM_2 * IdM_1
        tm_t.R = two_vectors_align(rot_axis, V3{0, 0, 1});
        { V3 o = V3{(float) (-1.0 * rot_origin.x), (float) (-1.0 * rot_origin.y), (float) (-1.0 * rot_origin.z)}; tm_t.t = mul(tm_t.R, o); }
        Rigid tm_s = compose(tm_t, mat);
        std::vector<IntervalEnd> ia;
        double threshold = 2.0 * resolution;
        float TWOPI = 2.0 * M_PI;
        for (int k = 0; k < K; ++k) {
            V3 ps = apply(tm_s, S[k]), pt = apply(tm_t, T[k]);
            float M_z = ps.z, M_len = std::sqrt(ps.x * ps.x + ps.y * ps.y), M_azi = vl_fast_atan2_f(ps.y, ps.x);
            float B_z = pt.z, B_len = std::sqrt(pt.x * pt.x + pt.y * pt.y), B_azi = vl_fast_atan2_f(pt.y, pt.x);
            double dz = B_z - M_z, d = B_len - M_len;
            double thMz = threshold * threshold - dz * dz;
            if (d * d <= thMz) {
                double rth = std::sqrt(thMz);
                auto ins = [&](double b, double e) { ia.push_back({b, true, k}); ia.push_back({e, false, k}); };
                if (M_len <= 1e-12) ins(0, TWOPI);
                else {
                    double dev = circle_intersection(M_len, B_len, rth);
                    if (std::fabs(dev - M_PI) <= 1e-12) ins(0, TWOPI);
                    else {
                        double beg = std::fmod(B_azi - dev - M_azi, TWOPI);
                        if (beg < 0) beg += TWOPI;
                        double end = std::fmod(B_azi + dev - M_azi, TWOPI);
                        if (end < 0) end += TWOPI;
                        if (end >= beg) ins(beg, end);
                        else { ins(beg, TWOPI); ins(0, end); }
                    }
                }
            }
        }
        double out_angle = 0; int out_count = 0;
        interval_stab(ia, out_angle, out_count);
        float angle = (float) out_angle;
        if (out_count > best_count) { best_count = out_count; best_T = mat; best_axis = rot_axis; best_origin = rot_origin; best_angle = angle; }
    }
    // gr_tran_mat = IdM_3 * IdM_2 * IdM_1 * two_point_tran_mat, rot = AngleAxisf(best_angle, axis) (:404-413)
    M3 rot;
    {
        float sn = std::sin(best_angle), cs = std::cos(best_angle);
        V3 ax = best_axis, sa = V3{sn * ax.x, sn * ax.y, sn * ax.z}, c1 = V3{(1 - cs) * ax.x, (1 - cs) * ax.y, (1 - cs) * ax.z};
        float tmp;
        tmp = c1.x * ax.y; rot.m[0][1] = tmp - sa.z; rot.m[1][0] = tmp + sa.z;
        tmp = c1.x * ax.z; rot.m[0][2] = tmp + sa.y; rot.m[2][0] = tmp - sa.y;
        tmp = c1.y * ax.z; rot.m[1][2] = tmp - sa.x; rot.m[2][1] = tmp + sa.x;
        rot.m[0][0] = c1.x * ax.x + cs; rot.m[1][1] = c1.y * ax.y + cs; rot.m[2][2] = c1.z * ax.z + cs;
    }
    Rigid I1{ident(), V3{(float) (-1.0 * best_origin.x), (float) (-1.0 * best_origin.y), (float) (-1.0 * best_origin.z)}};
    Rigid I2{rot, V3{0, 0, 0}}, I3{ident(), best_origin};
    Rigid G = compose(compose(compose(I3, I2), I1), best_T);
    // refineTransformationMatrix :261-316: inliers of ALL input correspondences under G, then umeyama (no scaling)
    std::vector<int> inl;
    for (int i = 0; i < c; ++i) {
        V3 s = ld(src, corr[i].query), t = ld(tgt, corr[i].match);
        V3 sp = apply(G, s);   // transformPointCloudWithNormals (:265)
        if (dist(t, sp) < two_res) inl.push_back(i);
    }
    int N = (int) inl.size();
    float sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
    for (int i : inl) { V3 s = ld(src, corr[i].query), t = ld(tgt, corr[i].match); sm[0] += s.x; sm[1] += s.y; sm[2] += s.z; dm[0] += t.x; dm[1] += t.y; dm[2] += t.z; }
    const float one_over_n = 1.0f / (float) N;
    for (int a = 0; a < 3; ++a) { sm[a] *= one_over_n; dm[a] *= one_over_n; }
    float sigma[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : inl) {
        V3 s = ld(src, corr[i].query), t = ld(tgt, corr[i].match);
        float sd[3] = {s.x - sm[0], s.y - sm[1], s.z - sm[2]}, dd[3] = {t.x - dm[0], t.y - dm[1], t.z - dm[2]};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) sigma[3 * a + b] += dd[a] * sd[b];
    }
    for (int k = 0; k < 9; ++k) sigma[k] *= one_over_n;
    float U[9], Sg[3], V[9];
    c_svd3(sigma, U, Sg, V);
    float sgn = (c_det3(U) * c_det3(V) < 0.f) ? -1.f : 1.f;
    float R[9], t[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        R[3 * i + j] = (U[3 * i + 0] * V[3 * j + 0] + U[3 * i + 1] * V[3 * j + 1]) + (U[3 * i + 2] * sgn) * V[3 * j + 2];
    for (int i = 0; i < 3; ++i) t[i] = dm[i] - ((R[3 * i + 0] * sm[0] + R[3 * i + 1] * sm[1]) + R[3 * i + 2] * sm[2]);
    for (int i = 0; i < 16; ++i) T16[i] = 0.f;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T16[4 * j + i] = R[3 * i + j]; T16[12 + i] = t[i]; }
    T16[15] = 1.f;
    if (diag) { diag[0] = K; diag[1] = best_count; diag[2] = tcfs_rows; diag[3] = N; }
    if (best_angle_out) *best_angle_out = best_angle;
    return 0;
}
