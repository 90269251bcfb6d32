// orc_libm.h -- ORACLE (test infrastructure): the float routines of ONE named libm, restated op for op.
//
// PCL 1.12.1 calls libm in two places of the hot path [3P]: pcl::computePairFeatures (std::acos(std::fabs(angle)) on floats for the
// source/target swap, std::atan2 on floats for f1) and pcl::computeRoots inside pcl::eigen33 (std::atan2, std::cos, std::sin on
// floats).  A libm is a build, not a specification, so "PCL's own arithmetic" is only defined once the libm is named.  Named here:
//
//     GNU libc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.x, this image), x86-64 --
//       acosf   sysdeps/ieee754/flt-32/e_acosf.c   (Sun fdlibm, float port; no multiarch variant)
//       atanf   sysdeps/ieee754/flt-32/s_atanf.c   (Sun fdlibm, float port; no multiarch variant)
//       atan2f  sysdeps/ieee754/flt-32/e_atan2f.c  (Sun fdlibm, float port; calls atanf)
//       sinf / cosf  sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h (ARM optimized-routines: double polynomial, results
//                    rounded to float once).  x86-64 selects an FMA build of these by ifunc on CPUs that have FMA; the restatement
//                    is the plain build (no contraction) -- the two agree wherever the double value is not within ~1e-16 relative
//                    of a float rounding boundary; tests/test_oracle_libm.py counts the differences against the running libm.
//   (Ubuntu 20.04's glibc 2.31, the reference's CI image, carries the same sources for these five routines.)
//
// Every function below is a fixed sequence of IEEE-754 operations (build with -ffp-contract=off); the HIP library restates the same
// sequences (csrc/lgr_libm.cuh) and tests/test_oracle_libm.py pins THIS file against the libm.so.6 it runs on: acosf for every
// float of [0, 1], atan2f / sinf / cosf on the argument ranges the path uses.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {
namespace glibc235 {

static inline int32_t fw(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
static inline float wf(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline float wfu(uint32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

// ---- e_acosf.c
static inline float acosf_(float x) {
    const float one = 1.0f, pi = wfu(0x40490fdau), pio2_hi = wfu(0x3fc90fdau), pio2_lo = wfu(0x33a22168u);
    const float pS0 = wfu(0x3e2aaaabu), pS1 = wfu(0xbea6b090u), pS2 = wfu(0x3e4e0aa8u), pS3 = wfu(0xbd241146u), pS4 = wfu(0x3a4f7f04u),
                pS5 = wfu(0x3811ef08u), qS1 = wfu(0xc019d139u), qS2 = wfu(0x4001572du), qS3 = wfu(0xbf303361u), qS4 = wfu(0x3d9dc62eu);
    float z, p, q, r, w, s, c, df;
    const int32_t hx = fw(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) {            // |x| == 1
        if (hx > 0) return 0.0f;
        return pi + 2.0f * pio2_lo;
    } else if (ix > 0x3f800000) {
        return (x - x) / (x - x);      // NaN
    }
    if (ix < 0x3f000000) {             // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;   // |x| <= 2^-26
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {               // x < -0.5
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = std::sqrt(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {                           // x > 0.5
        z = (one - x) * 0.5f;
        s = std::sqrt(z);
        df = wf(fw(s) & (int32_t) 0xfffff000);
        c = (z - df * df) / (s + df);
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        w = r * s + c;
        return 2.0f * (df + w);
    }
}

// ---- s_atanf.c
static inline float atanf_(float x) {
    const float atanhi[4] = {wfu(0x3eed6338u), wfu(0x3f490fdau), wfu(0x3f7b985eu), wfu(0x3fc90fdau)};
    const float atanlo[4] = {wfu(0x31ac3769u), wfu(0x33222168u), wfu(0x33140fb4u), wfu(0x33a22168u)};
    const float aT[11] = {wfu(0x3eaaaaabu), wfu(0xbe4ccccdu), wfu(0x3e124925u), wfu(0xbde38e38u), wfu(0x3dba2e6eu), wfu(0xbd9d8795u),
                          wfu(0x3d886b35u), wfu(0xbd6ef16bu), wfu(0x3d4bda59u), wfu(0xbd15a221u), wfu(0x3c8569d7u)};
    const float one = 1.0f;
    float w, s1, s2, z;
    int id;
    const int32_t hx = fw(x), ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {            // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {             // |x| < 0.4375
        if (ix < 0x31000000) return x; // |x| < 2^-29
        id = -1;
    } else {
        x = std::fabs(x);
        if (ix < 0x3f980000) {         // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }   // 7/16 <= |x| < 11/16
            else { id = 1; x = (x - one) / (x + one); }                           // 11/16 <= |x| < 19/16
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }   // |x| < 2.4375
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

// ---- e_atan2f.c
static inline float atan2f_(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = wfu(0x3f490fdbu), pi_o_2 = wfu(0x3fc90fdbu), pi = wfu(0x40490fdbu), pi_lo = wfu(0xb3bbbd2eu);
    float z;
    const int32_t hx = fw(x), ix = hx & 0x7fffffff, hy = fw(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;    // NaN
    if (hx == 0x3f800000) return atanf_(y);                  // x == 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);       // 2 * sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                   // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;                    // |y| / x < -2^60
    else z = atanf_(std::fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return wfu((uint32_t) fw(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// ---- s_sincosf.h / s_sinf.c / s_cosf.c for |x| < 120 (reduce_fast); the path's arguments lie in [0, pi / 3]
struct sincos_t { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
static inline const sincos_t& sincos_table(int i) {
    static const sincos_t t[2] = {
        {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
         -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
        {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
         0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
    return t[i];
}
static inline uint32_t abstop12(float x) { return ((uint32_t) fw(x) >> 20) & 0x7ff; }
static inline float sinf_poly(double x, double x2, const sincos_t& p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = p.s2 + x2 * p.s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * p.s1;
        return (float) (s + x7 * s1);
    } else {
        const double x4 = x2 * x2;
        const double c2 = p.c3 + x2 * p.c4;
        const double c1 = p.c1 + x2 * p.c2;
        const double x6 = x4 * x2;
        const double c = p.c0 + x2 * c1;
        return (float) (c + x6 * c2);
    }
}
static inline double reduce_fast(double x, const sincos_t& p, int* np) {
    const double r = x * p.hpi_inv;
    const int n = ((int32_t) r + 0x800000) >> 24;
    *np = n;
    return x - n * p.hpi;
}
// NOTE: valid for |y| < 120 (the callers' arguments are in [0, pi/3]); larger arguments take reduce_large in glibc, not restated
static inline float sinf_(float y) {
    double x = y;
    int n;
    const sincos_t* p = &sincos_table(0);
    if (abstop12(y) < abstop12(0x1.921FB6p-1f /* pi/4 */)) {
        const double s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sinf_poly(x, s, *p, 0);
    }
    x = reduce_fast(x, *p, &n);
    const double s = p->sign[n & 3];
    if (n & 2) p = &sincos_table(1);
    return sinf_poly(x * s, x * x, *p, n);
}
static inline float cosf_(float y) {
    double x = y;
    int n;
    const sincos_t* p = &sincos_table(0);
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        const double s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sinf_poly(x, s, *p, 1);
    }
    x = reduce_fast(x, *p, &n);
    const double s = p->sign[n & 3];
    if (n & 2) p = &sincos_table(1);
    return sinf_poly(x * s, x * x, *p, n ^ 1);
}

}  // namespace glibc235
}  // namespace orc
