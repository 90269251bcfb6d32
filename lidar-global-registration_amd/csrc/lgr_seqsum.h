// lgr_seqsum.h -- the float a loop `v = 0; repeat n times: v += x;` ends with, without running the loop.
//
// pcl::FPFHEstimation::computePointSPFHSignature adds hist_incr = 100 / (k - 1) to a bin once per neighbour that falls into it
// (SURVEY A.1), so a finished SPFH bin is the SEQUENTIAL float sum of `count` copies of the increment -- which is what the oracle
// defines and what this function returns bit for bit.  Run as a loop on the device that sum was 0.66 ms of the 2.4 ms spfh_tile_kernel
// takes per 1M-point cloud (a wave iterates as long as its largest count: a few hundred dependent adds, thirty-three times per point).
//
// Inside one binade [2^e, 2^(e+1)) every partial sum is a multiple of ulp_e, so fl(v + x) = v + c with ONE constant c (x rounded to a
// multiple of ulp_e) for as long as v + x stays below 2^(e+1); when x lies exactly half way between two multiples the tie goes to the
// even neighbour, which makes the FIRST step of a binade depend on the parity of the value it starts from and every later one constant
// again.  So per binade: two real additions (the second one's difference to the first is c), then the number of further steps
// that stay below the top, counted in units of ulp_e with integer arithmetic, at once; then on to the next binade through a real addition.  At most ~3 real additions
// per binade, <= 9 binades between an increment of 100 / k and a sum of 100.  tests/cpp/seqsum_test.cpp compares it with the loop for
// every (k - 1, count) up to 3000 and for random increments over 40 binades.
#pragma once
#ifdef __HIPCC__
#define LGR_HD __host__ __device__ __forceinline__
#else
#define LGR_HD inline
#endif
#include <math.h>
#include <string.h>

LGR_HD float lgr_seqsum(float x, int n) {
    if (n <= 0) return 0.0f;
    unsigned xb;
    memcpy(&xb, &x, 4);
    const int ex = (int) ((xb >> 23) & 0xffu);
    if ((xb >> 31) || ex == 0 || ex >= 200) {   // zero, denormal, negative, huge, inf, NaN: the plain loop (never on the FPFH path: x = 100 / (k - 1))
        float v = 0.0f;
        for (int i = 0; i < n; ++i) v += x;
        return v;
    }
    const unsigned mx = (xb & 0x7fffffu) | 0x800000u;   // x = mx 2^(ex - 150)
    float v = x;   // 0 + x
    int rem = n - 1;
    while (rem > 0) {
        unsigned vb;
        memcpy(&vb, &v, 4);
        const unsigned eb = vb & 0x7f800000u;   // v in [2^e, 2^(e+1)), ulp_e = 2^(e - 150) in the biased notation
        const unsigned tb = eb + 0x00800000u;
        float top;
        memcpy(&top, &tb, 4);
        const float v1 = v + x;
        --rem;
        if (rem == 0 || !(v1 < top)) { v = v1; continue; }
        const float v2 = v1 + x;
        --rem;
        if (rem == 0 || !(v2 < top)) { v = v2; continue; }
        // everything from here on in units of ulp_e: V = mantissa with its hidden bit (2^23 <= V < 2^24), top = 2^24
        unsigned b1, b2;
        memcpy(&b1, &v1, 4); memcpy(&b2, &v2, 4);
        const unsigned V2 = (b2 & 0x7fffffu) | 0x800000u;
        const unsigned C = V2 - ((b1 & 0x7fffffu) | 0x800000u);   // the constant step (exact)
        if (C == 0u) { v = v2; break; }   // x below half an ulp: every further addition leaves the sum where it is
        // a further step from V is allowed while V ulp_e + x < top in exact arithmetic, i.e. V < 2^24 - X with X = x / ulp_e = mx / 2^sh;
        // V is an integer, so that is V <= 2^24 - floor(X) - 1 whether or not X is one.  Steps j = 1 .. m start from V2 + (j - 1) C.
        const int sh = (int) (eb >> 23) - ex;   // >= 0: the sum is never below x
        const unsigned FX = sh >= 24 ? 0u : (mx >> sh);
        const int R = (int) (0x1000000u - V2 - FX) - 1;
        int m = R < 0 ? 0 : (int) ((unsigned) R / C) + 1;
        if (m > rem) m = rem;
        const unsigned V = V2 + (unsigned) m * C;   // <= 2^24 (a last step may round up to the top itself)
        const unsigned ob = V >= 0x1000000u ? tb : (eb | (V & 0x7fffffu));
        memcpy(&v, &ob, 4);
        rem -= m;
    }
    return v;
}
