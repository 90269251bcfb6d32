// lgr_libm.cuh -- the float routines of ONE named libm as fixed IEEE-754 operation sequences on the device (gfx950).
//
// PCL 1.12.1 calls libm on the hot path in pcl::computePairFeatures (acos of |angle| for the source / target swap, atan2 for f1: call
// site include/common.h:326-331) and in pcl::computeRoots under pcl::eigen33 (atan2, cos, sin: call site src/common.cpp:646-653), all on
// floats.  Named libm: GNU libc 2.35, x86-64 (sysdeps/ieee754/flt-32: e_acosf.c, s_atanf.c, e_atan2f.c -- Sun fdlibm's float ports -- and
// s_sinf.c / s_cosf.c / s_sincosf.h -- double polynomials rounded to float once).  The CPU oracle states the same sequences
// (oracle/src/orc_libm.h) and pins them against the running libm.so.6 for every float of the ranges used (tests/test_oracle_libm.py);
// lgr_selfcheck_libm evaluates THIS file on the device and tests/test_gpu_pcl_arith.py compares the two bit for bit.
// Compiled with -ffp-contract=off: every a * b + c below is an IEEE multiply followed by an IEEE add.  Branches of the C sources are
// written as selects where both sides are cheap; each path still performs exactly its own operations.
#pragma once
#include <hip/hip_runtime.h>

namespace lgr_glibc {

__device__ __forceinline__ float uf(unsigned u) { return __uint_as_float(u); }

// e_acosf.c (|x| <= 1 -> [0, pi]; |x| > 1 -> NaN)
__device__ __forceinline__ float acosf_(float x) {
    const float pi = uf(0x40490fdau), pio2_hi = uf(0x3fc90fdau), pio2_lo = uf(0x33a22168u);
    const float pS0 = uf(0x3e2aaaabu), pS1 = uf(0xbea6b090u), pS2 = uf(0x3e4e0aa8u), pS3 = uf(0xbd241146u), pS4 = uf(0x3a4f7f04u), pS5 = uf(0x3811ef08u);
    const float qS1 = uf(0xc019d139u), qS2 = uf(0x4001572du), qS3 = uf(0xbf303361u), qS4 = uf(0x3d9dc62eu);
    const int hx = __float_as_int(x), ix = hx & 0x7fffffff;
    if (ix >= 0x3f800000) {
        if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
        return (x - x) / (x - x);
    }
    const bool small = ix < 0x3f000000;                    // |x| < 0.5
    if (small && ix <= 0x32800000) return pio2_hi + pio2_lo;
    const float z = small ? x * x : (hx < 0 ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    if (small) return pio2_hi - (x - (pio2_lo - x * r));
    const float s = __builtin_sqrtf(z);
    if (hx < 0) {
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float df = __int_as_float(__float_as_int(s) & (int) 0xfffff000);
    const float c = (z - df * df) / (s + df);
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// s_atanf.c
__device__ __forceinline__ float atanf_(float x) {
    const float hi3 = uf(0x3fc90fdau), lo3 = uf(0x33a22168u);
    const int hx = __float_as_int(x), ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {                                // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x31000000) return x;                         // |x| < 2^-29
    const float ax = fabsf(x);
    // argument reduction: id -1 (|x| < 7/16: x itself), 0 [7/16, 11/16), 1 [11/16, 19/16), 2 [19/16, 39/16), 3 above; ONE division, operands by range
    const int id = ix < 0x3ee00000 ? -1 : ix < 0x3f300000 ? 0 : ix < 0x3f980000 ? 1 : ix < 0x401c0000 ? 2 : 3;
    float num = x, den = 1.0f, ahi = 0.0f, alo = 0.0f;
    if (id == 0) { num = 2.0f * ax - 1.0f; den = 2.0f + ax; ahi = uf(0x3eed6338u); alo = uf(0x31ac3769u); }
    if (id == 1) { num = ax - 1.0f; den = ax + 1.0f; ahi = uf(0x3f490fdau); alo = uf(0x33222168u); }
    if (id == 2) { num = ax - 1.5f; den = 1.0f + 1.5f * ax; ahi = uf(0x3f7b985eu); alo = uf(0x33140fb4u); }
    if (id == 3) { num = -1.0f; den = ax; ahi = hi3; alo = lo3; }
    const float t = id < 0 ? x : num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (uf(0x3eaaaaabu) + w * (uf(0x3e124925u) + w * (uf(0x3dba2e6eu) + w * (uf(0x3d886b35u) + w * (uf(0x3d4bda59u) + w * uf(0x3c8569d7u))))));
    const float s2 = w * (uf(0xbe4ccccdu) + w * (uf(0xbde38e38u) + w * (uf(0xbd9d8795u) + w * (uf(0xbd6ef16bu) + w * uf(0xbd15a221u)))));
    if (id < 0) return t - t * (s1 + s2);
    const float r = ahi - ((t * (s1 + s2) - alo) - t);
    return hx < 0 ? -r : r;
}

// e_atan2f.c
__device__ __forceinline__ float atan2f_(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = uf(0x3f490fdbu), pi_o_2 = uf(0x3fc90fdbu), pi = uf(0x40490fdbu), pi_lo = uf(0xb3bbbd2eu);
    const int hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// s_sincosf.h / s_sinf.c / s_cosf.c for |y| < 120 (the callers' arguments lie in [0, pi / 3]): double arithmetic, rounded to float once
__device__ __forceinline__ float sincos_poly(double x, double x2, bool neg, int n) {
    // table 0 / table 1 (neg): the cosine coefficients change sign, the sine coefficients do not
    const double c0 = neg ? -0x1p0 : 0x1p0, c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2, c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
    const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10, c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = s2c + x2 * s3c;
        const double x7 = x3 * x2;
        const double s = x + x3 * s1c;
        return (float) (s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double cc2 = c3 + x2 * c4;
    const double cc1 = c1 + x2 * c2;
    const double x6 = x4 * x2;
    const double c = c0 + x2 * cc1;
    return (float) (c + x6 * cc2);
}
__device__ __forceinline__ unsigned abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
template <bool COS>
__device__ __forceinline__ float sincosf_(float y) {
    double x = (double) y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {          // |y| < pi / 4
        if (abstop12(y) < abstop12(0x1p-12f)) return COS ? 1.0f : y;
        return sincos_poly(x, x * x, false, COS ? 1 : 0);
    }
    const double r = x * 0x1.45F306DC9C883p+23;            // reduce_fast: n = round(x * 2 / pi)
    const int n = ((int) r + 0x800000) >> 24;
    x = x - (double) n * 0x1.921FB54442D18p0;
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, (n & 2) != 0, COS ? (n ^ 1) : n);
}
__device__ __forceinline__ float sinf_(float y) { return sincosf_<false>(y); }
__device__ __forceinline__ float cosf_(float y) { return sincosf_<true>(y); }

}  // namespace lgr_glibc

// pcl::eigen33(mat, eigenvalue, eigenvector) [3P, PCL 1.12.1 common/impl/eigen.hpp: computeRoots / computeRoots2 /
// detail::getLargest3x3Eigenvector], Scalar = float as pcl::solvePlaneParameters instantiates it: the smallest eigenvalue of the
// symmetric 3 x 3 matrix C (row-major) and its eigenvector.  Restates oracle/src/orc_features.cpp pcl_eigen33 op for op.
__device__ __forceinline__ void lgr_pcl_roots2(float b, float c, float& r0, float& r1, float& r2) {
    r0 = 0.f;
    float d = (float) ((double) (b * b) - 4.0 * (double) c);   // Scalar (b * b - 4.0 * c): the product in float, the rest in double
    if ((double) d < 0.0) d = 0.f;
    const float sd = __builtin_sqrtf(d);
    r2 = 0.5f * (b + sd);
    r1 = 0.5f * (b - sd);
}
__device__ __forceinline__ void lgr_pcl_eigen33(const float C[9], float& eigenvalue, float& vx, float& vy, float& vz) {
    float scale = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(C[i]));
    if (scale <= 1.17549435e-38f) scale = 1.f;
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = C[i] / scale;
    const float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    float r0, r1, r2;
    if (fabsf(c0) < 1.1920929e-07f) {
        lgr_pcl_roots2(c2, c1, r0, r1, r2);
    } else {
        const float s_inv3 = (float) (1.0 / 3.0);
        const float s_sqrt3 = 1.73205078f;   // std::sqrt(3.0f)
        const float c2_over_3 = c2 * s_inv3;
        float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
        if (a_over_3 > 0.f) a_over_3 = 0.f;
        const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
        float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
        if (q > 0.f) q = 0.f;
        const float rho = __builtin_sqrtf(-a_over_3);
        const float theta = lgr_glibc::atan2f_(__builtin_sqrtf(-q), half_b) * s_inv3;
        const float cos_theta = lgr_glibc::cosf_(theta), sin_theta = lgr_glibc::sinf_(theta);
        r0 = c2_over_3 + 2.f * rho * cos_theta;
        r1 = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
        r2 = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
        float t;
        if (r0 >= r1) { t = r0; r0 = r1; r1 = t; }
        if (r1 >= r2) {
            t = r1; r1 = r2; r2 = t;
            if (r0 >= r1) { t = r0; r0 = r1; r1 = t; }
        }
        if (r0 <= 0.f) lgr_pcl_roots2(c2, c1, r0, r1, r2);
    }
    eigenvalue = r0 * scale;
    m[0] -= r0; m[4] -= r0; m[8] -= r0;
    // rows 0 x 1, 0 x 2, 1 x 2; the longest (first maximum) normalised
    const float ax = m[1] * m[5] - m[2] * m[4], ay = m[2] * m[3] - m[0] * m[5], az = m[0] * m[4] - m[1] * m[3];
    const float bx = m[1] * m[8] - m[2] * m[7], by = m[2] * m[6] - m[0] * m[8], bz = m[0] * m[7] - m[1] * m[6];
    const float cx = m[4] * m[8] - m[5] * m[7], cy = m[5] * m[6] - m[3] * m[8], cz = m[3] * m[7] - m[4] * m[6];
    const float la = __builtin_sqrtf(ax * ax + ay * ay + az * az), lb = __builtin_sqrtf(bx * bx + by * by + bz * bz), lc = __builtin_sqrtf(cx * cx + cy * cy + cz * cz);
    float len = -1.f;
    vx = ax; vy = ay; vz = az;   // (all three lengths NaN: the first row product over -1, as the oracle's loop leaves it)
    if (la > len) { len = la; vx = ax; vy = ay; vz = az; }
    if (lb > len) { len = lb; vx = bx; vy = by; vz = bz; }
    if (lc > len) { len = lc; vx = cx; vy = cy; vz = cz; }
    vx = vx / len; vy = vy / len; vz = vz / len;
}
