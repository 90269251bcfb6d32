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
// again.  So per binade: two real additions (the second one's difference to the first is c), then n = the number of further steps
// that stay below the top in exact arithmetic at once, then on to the next binade through a real addition.  At most ~3 real additions
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
    if (!(x > 0.0f) || !(x < 3.0e38f)) {   // zero, negative, inf, NaN: the plain loop (never on the FPFH path: x = 100 / (k - 1), k >= 2 where n > 0)
        float v = 0.0f;
        for (int i = 0; i < n; ++i) v += x;
        return v;
    }
    float v = x;   // 0 + x
    int rem = n - 1;
    while (rem > 0) {
        unsigned b;
        memcpy(&b, &v, 4);
        b = (b & 0x7f800000u) + 0x00800000u;
        float top;
        memcpy(&top, &b, 4);   // 2^(e+1): the end of v's binade (inf in the last one: the comparisons below then never end the binade)
        const float v1 = v + x;
        --rem;
        if (rem == 0 || !(v1 < top)) { v = v1; continue; }
        const float v2 = v1 + x;
        --rem;
        if (rem == 0 || !(v2 < top)) { v = v2; continue; }
        const float c = v2 - v1;   // exact: both are multiples of ulp_e
        if (!(c > 0.0f)) { v = v2; break; }   // x below half an ulp: every further addition leaves the sum where it is
        // further steps j = 1 .. m take v2 + (j - 1) c to v2 + j c as long as (v2 + (j - 1) c) + x < top in exact arithmetic (doubles hold
        // every term exactly: multiples of 2^-60 below 2^128 do not occur here -- the FPFH range is 2^-20 .. 2^7)
        const double room = (double) top - (double) x - (double) v2;
        long long m = room > 0.0 ? (long long) ceil(room / (double) c) : 0;
        if (m > rem) m = rem;
        while (m > 0 && !(((double) v2 + (double) (m - 1) * (double) c) + (double) x < (double) top)) --m;            // (guards of the division's rounding:
        while (m < rem && (((double) v2 + (double) m * (double) c) + (double) x < (double) top)) ++m;                    //  they never fire in the tests)
        v = (float) ((double) v2 + (double) m * (double) c);
        rem -= (int) m;
    }
    return v;
}
