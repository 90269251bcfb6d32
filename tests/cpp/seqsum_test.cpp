// lgr_seqsum (csrc/lgr_seqsum.h) against the loop it replaces: every FPFH increment 100 / (k - 1), k - 1 = 1 .. KMAX, with every count
// 0 .. k - 1 (a bin cannot hold more than the point's neighbours), and random increments over 40 binades with counts up to 5000.
// Built with -ffp-contract=off like the library.  Prints "ok <cases>" or the first mismatch.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../lidar-global-registration_amd/csrc/lgr_seqsum.h"

static float loop_sum(float x, int n) {
    volatile float v = 0.0f;
    for (int i = 0; i < n; ++i) v = v + x;
    return v;
}

int main(int argc, char** argv) {
    const int KMAX = argc > 1 ? atoi(argv[1]) : 3000;
    long long cases = 0;
    for (int k1 = 1; k1 <= KMAX; ++k1) {
        const float x = 100.0f / (float) k1;
        float v = 0.0f;
        for (int n = 0; n <= k1; ++n) {
            if (n > 0) { volatile float t = v + x; v = t; }
            const float got = lgr_seqsum(x, n);
            uint32_t a, b;
            memcpy(&a, &v, 4); memcpy(&b, &got, 4);
            if (a != b) { printf("MISMATCH k-1 %d n %d: loop %.9g (%08x) seqsum %.9g (%08x)\n", k1, n, v, a, got, b); return 1; }
            ++cases;
        }
    }
    uint64_t s = 88172645463325252ull;
    for (int t = 0; t < 200000; ++t) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t bits = (uint32_t) (((s >> 11) & 0x7fffffu) | ((uint32_t) (100 + (s >> 40) % 40) << 23));   // exponents 2^-27 .. 2^12
        if (t % 7 == 0) bits &= 0xffffff00u;   // short mantissas: increments that hit exact ties
        if (t % 11 == 0) bits &= 0xffff0000u;
        float x;
        memcpy(&x, &bits, 4);
        const int n = (int) ((s >> 20) % 5000);
        const float want = loop_sum(x, n), got = lgr_seqsum(x, n);
        uint32_t a, b;
        memcpy(&a, &want, 4); memcpy(&b, &got, 4);
        if (a != b) { printf("MISMATCH x %.9g (%08x) n %d: loop %.9g seqsum %.9g\n", x, bits, n, want, got); return 1; }
        ++cases;
    }
    // degenerate increments take the plain loop
    if (lgr_seqsum(0.0f, 5) != 0.0f || lgr_seqsum(1.0f, 0) != 0.0f) { printf("MISMATCH degenerate\n"); return 1; }
    printf("ok %lld\n", cases);
    return 0;
}
