// orc_iss.cpp -- ORACLE (test infrastructure): ISS key points.
// Restates detectKeyPoints (src/common.cpp:657-691): ISSKeypoint3DDebug = pcl::ISSKeypoint3D (PCL 1.12.1
// keypoints/impl/iss_3d.hpp) with salient radius = non-maxima radius = iss_radius, gamma21 = gamma32 = 0.975,
// min_neighbors = 4, border estimation off (border_radius 0); key point indices ascending (fix_seed sorts them, and
// PCL emits them in index order anyway).
//
//   scatter(i)   = sum over radius neighbours q (strict d2 < r^2, the point itself included) of (q - p)(q - p)^T, in double
//                  (getScatterMatrix; zero matrix when fewer than min_neighbors neighbours)
//   e1 >= e2 >= e3 eigenvalues; skipped when not finite or e3 < 0
//   third(i)     = e3 when e2/e1 < gamma21 and e3/e2 < gamma32, else 0
//   key point    <=> third(i) > 0, at least min_neighbors radius neighbours, and no neighbour with a larger third
//
// Canonical choices (parity unpinned: PCL / Eigen / FLANN are not in this image):
//   * neighbour order = grid cells (z, y, x) ascending (cell = 1.001 r), then index (FLANN's order is unspecified;
//     only the double summation order depends on it);
//   * eigenvalues by cyclic Jacobi in double, 10 fixed sweeps (Eigen::SelfAdjointEigenSolver is iterative QL).
#include <cmath>
#include <vector>

#include "../lgr_oracle.h"
#include "orc_grid.h"

using namespace orc;

// eigenvalues of a symmetric 3x3 (a00 a01 a02 a11 a12 a22), ascending; same sequence as lgr_eigvals3d (lgr_math.cuh)
static void c_eigvals3d(const double S[6], double ev[3]) {
    double a[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
    for (int sweep = 0; sweep < 10; ++sweep) {
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                int r = 3 - p - q;
                double app = a[p][p], aqq = a[q][q], arp = a[r][p], arq = a[r][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
            }
    }
    double x = a[0][0], y = a[1][1], z = a[2][2], tmp;
    if (x > y) { tmp = x; x = y; y = tmp; }
    if (y > z) { tmp = y; y = z; z = tmp; }
    if (x > y) { tmp = x; x = y; y = tmp; }
    ev[0] = x; ev[1] = y; ev[2] = z;
}

extern "C" void orc_eigvals3d(const double S6[6], double ev3[3]) { c_eigvals3d(S6, ev3); }

// out_idx must hold n ints; third (optional) n doubles
extern "C" int orc_iss_keypoints(const float* pts, int n, float radius, float gamma21, float gamma32, int min_neighbors,
                                 int* out_idx, int* n_out, double* third_out) {
    if (!(radius > 0.f) || !(gamma21 > 0.f) || !(gamma32 > 0.f) || min_neighbors <= 0) return -2;
    Grid g;
    g.build(pts, n, radius * 1.001f);
    const float r2 = radius * radius;
    std::vector<double> third(n, 0.0);
    std::vector<int> nnb(n, 0);
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) {
        const float* P = pts + 12 * (size_t) i;
        if (!finite3(P)) continue;
        double S[6] = {0, 0, 0, 0, 0, 0};
        int k = 0;
        g.visit27(P, [&](int q) {
            const float* Q = pts + 12 * (size_t) q;
            if (!(dist2(P, Q) < r2)) return;
            ++k;
            double dx = (double) Q[0] - (double) P[0], dy = (double) Q[1] - (double) P[1], dz = (double) Q[2] - (double) P[2];
            S[0] += dx * dx; S[1] += dx * dy; S[2] += dx * dz; S[3] += dy * dy; S[4] += dy * dz; S[5] += dz * dz;
        });
        nnb[i] = k;
        if (k < min_neighbors) continue;   // zero scatter: e2/e1 = NaN, never below gamma
        double ev[3];
        c_eigvals3d(S, ev);
        double e1 = ev[2], e2 = ev[1], e3 = ev[0];
        if (!std::isfinite(e1) || !std::isfinite(e2) || !std::isfinite(e3)) continue;
        if (e3 < 0) continue;
        if (e2 / e1 < (double) gamma21 && e3 / e2 < (double) gamma32) third[i] = e3;
    }
    std::vector<unsigned char> is_kp(n, 0);
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) {
        if (!(third[i] > 0.0) || nnb[i] < min_neighbors) continue;
        const float* P = pts + 12 * (size_t) i;
        bool is_max = true;
        g.visit27(P, [&](int q) {
            if (!(dist2(P, pts + 12 * (size_t) q) < r2)) return;
            if (third[i] < third[q]) is_max = false;
        });
        is_kp[i] = is_max ? 1 : 0;
    }
    int m = 0;
    for (int i = 0; i < n; ++i) if (is_kp[i]) out_idx[m++] = i;
    *n_out = m;
    if (third_out) for (int i = 0; i < n; ++i) third_out[i] = third[i];
    return 0;
}
