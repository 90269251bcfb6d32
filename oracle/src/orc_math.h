// orc_math.h -- canonical elementary functions of the ORACLE (test infrastructure).
//
// The reference calls libm / Eigen for these (std::atan2 in pcl::computePairFeatures, std::log / std::cbrt in
// src/analysis.cpp:95-130, std::exp in src/metric.cpp:72, Eigen::JacobiSVD in src/transformation.cpp:27 and in
// pcl::umeyama).  libm results are not reproducible bit-for-bit on a GPU, so the oracle DEFINES each of them as a
// fixed sequence of IEEE-754 binary32 operations (+ - * / sqrt, no FMA contraction: build with -ffp-contract=off).
// The HIP kernels restate the same sequences independently; tests compare the two bit-for-bit and compare this file
// against libm (tests/test_oracle_math.py) to bound the deviation from the reference (<= 2 ulp).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// atan2f: Cephes-style atanf on a = min/max in [0,1], two ranges, then octant fix-up.
static inline float c_atan2f(float y, float x) {
    float ax = std::fabs(x), ay = std::fabs(y);
    float mx = (ax > ay) ? ax : ay;
    float mn = (ax > ay) ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx;
    float base = 0.0f;
    float z = a;
    if (a > 0.41421356237f) { base = 0.78539816339f; z = (a - 1.0f) / (a + 1.0f); }
    float z2 = z * z;
    float p = 8.05374449538e-2f * z2 - 1.38776856032e-1f;
    p = p * z2 + 1.99777106478e-1f;
    p = p * z2 - 3.33329491539e-1f;
    p = p * z2 * z + z;
    float r = base + p;
    if (ay > ax) r = 1.57079632679f - r;
    if (x < 0.0f) r = 3.14159265359f - r;
    if (y < 0.0f) r = -r;
    return r;
}

// logf for x > 0 (normal numbers): x = m * 2^e, m in (sqrt(1/2), sqrt(2)], log m = 2 atanh((m-1)/(m+1)).
static inline float c_logf(float x) {
    uint32_t b = f2u(x);
    int e = (int) ((b >> 23) & 0xff) - 127;
    float m = u2f((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e = e + 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float q = 0.111111111111f;             // 1/9
    q = q * z + 0.142857142857f;           // 1/7
    q = q * z + 0.2f;                      // 1/5
    q = q * z + 0.333333333333f;           // 1/3
    q = q * z;
    float r = 2.0f * s + 2.0f * s * q;
    float fe = (float) e;
    return fe * 0.693359375f + (r + fe * (-2.12194440e-4f));
}

// cbrtf for x >= 0: bit seed + 4 Newton steps.
static inline float c_cbrtf(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? 0.0f : x;   // 0 -> 0, NaN/negative -> itself (never used for negatives)
    uint32_t b = f2u(x);
    float y = u2f(b / 3u + 0x2a5137a0u);
    for (int i = 0; i < 4; ++i) y = (y + y + x / (y * y)) / 3.0f;
    return y;
}

// expf (Cephes polynomial), argument clamped to [-87, 88].
static inline float c_expf(float x) {
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    float fn = std::floor(x * 1.44269504089f + 0.5f);
    float r = x - fn * 0.693359375f;
    r = r - fn * (-2.12194440e-4f);
    float z = r * r;
    float p = 1.9875691500e-4f * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * z + r + 1.0f;
    int n = (int) fn;
    return p * u2f((uint32_t) (n + 127) << 23);
}

// 3x3 SVD by one-sided (Hestenes) Jacobi. A row-major; A = U diag(S) V^T, S descending, U/V row-major, orthonormal.
// Fixed 12 sweeps over (0,1),(0,2),(1,2); a pair is skipped when gamma^2 <= 1e-14 * alpha * beta.
static inline void c_svd3(const float A[9], float U[9], float S[3], float V[9]) {
    float W[3][3], Vm[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { W[i][j] = A[3 * i + j]; Vm[i][j] = (i == j) ? 1.0f : 0.0f; }
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        for (int k = 0; k < 3; ++k) {
            int p = PQ[k][0], q = PQ[k][1];
            float alpha = (W[0][p] * W[0][p] + W[1][p] * W[1][p]) + W[2][p] * W[2][p];
            float beta = (W[0][q] * W[0][q] + W[1][q] * W[1][q]) + W[2][q] * W[2][q];
            float gamma = (W[0][p] * W[0][q] + W[1][p] * W[1][q]) + W[2][p] * W[2][q];
            if (gamma * gamma <= 1e-14f * alpha * beta) continue;
            float zeta = (beta - alpha) / (2.0f * gamma);
            float az = std::fabs(zeta);
            float t = 1.0f / (az + std::sqrt(1.0f + zeta * zeta));
            if (zeta < 0.0f) t = -t;
            float c = 1.0f / std::sqrt(1.0f + t * t);
            float s = c * t;
            for (int i = 0; i < 3; ++i) {
                float wp = W[i][p], wq = W[i][q];
                W[i][p] = c * wp - s * wq;
                W[i][q] = s * wp + c * wq;
                float vp = Vm[i][p], vq = Vm[i][q];
                Vm[i][p] = c * vp - s * vq;
                Vm[i][q] = s * vp + c * vq;
            }
        }
    }
    float sg[3];
    for (int j = 0; j < 3; ++j) sg[j] = std::sqrt((W[0][j] * W[0][j] + W[1][j] * W[1][j]) + W[2][j] * W[2][j]);
    // sort columns by descending sigma (stable selection: ties keep lower column first)
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (sg[ord[b]] > sg[ord[a]]) { int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
    float Uc[3][3];  // Uc[j] = j-th column
    for (int j = 0; j < 3; ++j) {
        int o = ord[j];
        S[j] = sg[o];
        for (int i = 0; i < 3; ++i) { V[3 * i + j] = Vm[i][o]; Uc[j][i] = W[i][o]; }
    }
    float tiny = 1e-5f * S[0];
    if (!(S[0] > 0.0f)) {  // zero matrix
        for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0f : 0.0f;
        return;
    }
    for (int i = 0; i < 3; ++i) Uc[0][i] = Uc[0][i] / S[0];
    if (S[1] > tiny) {
        for (int i = 0; i < 3; ++i) Uc[1][i] = Uc[1][i] / S[1];
    } else {
        // rank 1: any unit vector orthogonal to u0: cross with the axis of smallest |component|
        float a0 = std::fabs(Uc[0][0]), a1 = std::fabs(Uc[0][1]), a2 = std::fabs(Uc[0][2]);
        int ax = 0;
        if (a1 < a0) { ax = 1; a0 = a1; }
        if (a2 < a0) { ax = 2; }
        float e[3] = {0.f, 0.f, 0.f};
        e[ax] = 1.0f;
        float cx = Uc[0][1] * e[2] - Uc[0][2] * e[1];
        float cy = Uc[0][2] * e[0] - Uc[0][0] * e[2];
        float cz = Uc[0][0] * e[1] - Uc[0][1] * e[0];
        float nn = std::sqrt((cx * cx + cy * cy) + cz * cz);
        Uc[1][0] = cx / nn; Uc[1][1] = cy / nn; Uc[1][2] = cz / nn;
    }
    if (S[2] > tiny) {
        for (int i = 0; i < 3; ++i) Uc[2][i] = Uc[2][i] / S[2];
    } else {
        Uc[2][0] = Uc[0][1] * Uc[1][2] - Uc[0][2] * Uc[1][1];
        Uc[2][1] = Uc[0][2] * Uc[1][0] - Uc[0][0] * Uc[1][2];
        Uc[2][2] = Uc[0][0] * Uc[1][1] - Uc[0][1] * Uc[1][0];
    }
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) U[3 * i + j] = Uc[j][i];
}

static inline float c_det3(const float M[9]) {
    return (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

}  // namespace orc
