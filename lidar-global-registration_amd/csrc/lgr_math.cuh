// lgr_math.cuh -- device elementary functions with a FIXED binary32 operation sequence (gfx950).
//
// libm results (std::log / std::cbrt in src/analysis.cpp:95-130, std::exp in src/metric.cpp:72; the acos / atan2 / cos / sin PCL calls are
// restated from ONE named libm in lgr_libm.cuh) and Eigen::JacobiSVD (src/transformation.cpp:27, pcl::umeyama) are not reproducible bit for
// bit across CPU and GPU math libraries.  The parity contract therefore fixes each of them as a sequence of IEEE
// + - * / sqrt operations (documented in DESIGN.md "canonical elementary functions"); this header is the device
// statement of those sequences and must be compiled with -ffp-contract=off.  Accuracy vs libm: <= 2 ulp.
#pragma once
#include <hip/hip_runtime.h>

// Philox4x32-10 (Salmon et al., SC'11), key = (seed_lo, seed_hi): the RANSAC sampler's stream (counter = (iteration, 0, 0, 0)) and the
// closest-plane metric's sparse subsets (lgr_plane.hip).  Pinned by Random123's three known-answer vectors through the full counter
// (lgr_selfcheck_philox, tests/test_gpu_ransac.py; the oracle's copy in tests/test_oracle_golden.py).
__device__ __forceinline__ void lgr_philox4(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned out[4]) {
    unsigned k0 = (unsigned) seed, k1 = (unsigned) (seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        unsigned n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// log for normal x > 0: x = m 2^e with m in (sqrt(1/2), sqrt(2)], log m = 2 atanh((m-1)/(m+1)), 4-term series.
__device__ __forceinline__ float lgr_logf(float x) {
    unsigned b = __float_as_uint(x);
    int e = (int) ((b >> 23) & 0xff) - 127;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e = e + 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float q = 0.111111111111f;
    q = q * z + 0.142857142857f;
    q = q * z + 0.2f;
    q = q * z + 0.333333333333f;
    q = q * z;
    float r = 2.0f * s + 2.0f * s * q;
    float fe = (float) e;
    return fe * 0.693359375f + (r + fe * (-2.12194440e-4f));
}

// cube root for x >= 0: exponent/3 bit seed, four Newton steps.
__device__ __forceinline__ float lgr_cbrtf(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? 0.0f : x;
    unsigned b = __float_as_uint(x);
    float y = __uint_as_float(b / 3u + 0x2a5137a0u);
#pragma unroll
    for (int i = 0; i < 4; ++i) y = (y + y + x / (y * y)) / 3.0f;
    return y;
}

// exp (Cephes polynomial), argument clamped to [-87, 88].
__device__ __forceinline__ float lgr_expf(float x) {
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    float fn = floorf(x * 1.44269504089f + 0.5f);
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
    return p * __uint_as_float((unsigned) (n + 127) << 23);
}

__device__ __forceinline__ float lgr_det3(const float* M) {
    return (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// One-sided (Hestenes) Jacobi SVD of a row-major 3x3: A = U diag(S) V^T, S descending.  12 fixed sweeps over the
// column pairs (0,1),(0,2),(1,2); a pair is skipped when gamma^2 <= 1e-14 alpha beta.  Rank-deficient completion:
// sigma_j <= 1e-5 sigma_0 -> u_j from cross products.  Op for op the sequence of c_svd3 (oracle/src/orc_math.h).
//
// Shape of the code (round 4): NO lane-divergent control flow and NO memory.  A lane that skips a pair keeps its values through
// selects (the rotation is computed and discarded); the only branch is wave-uniform (no lane of the wave rotates -> nothing is
// issued).  The rank-deficient completion computes both alternatives and selects.  Every local array is indexed by compile-time
// constants only, so nothing is demoted to scratch (tools/isa_hazards.py --gate checks the code object: 0 bytes of scratch in every
// caller); callers that need only V and S instantiate WANT_U = false and pass U = nullptr.
template <bool WANT_U = true>
__device__ __forceinline__ void lgr_svd3(const float* A, float* U, float* S, float* V) {
    float W[3][3], Vm[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { W[i][j] = A[3 * i + j]; Vm[i][j] = (i == j) ? 1.0f : 0.0f; }
    for (int sweep = 0; sweep < 12; ++sweep) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = (k == 2) ? 1 : 0, q = (k == 0) ? 1 : 2;
            const float alpha = (W[0][p] * W[0][p] + W[1][p] * W[1][p]) + W[2][p] * W[2][p];
            const float beta = (W[0][q] * W[0][q] + W[1][q] * W[1][q]) + W[2][q] * W[2][q];
            const float gamma = (W[0][p] * W[0][q] + W[1][p] * W[1][q]) + W[2][p] * W[2][q];
            const bool rot = !(gamma * gamma <= 1e-14f * alpha * beta);
#ifdef LGR_EXP_SVD_DIVERGENT   // experiment only (tools/exp_svd_variants.sh): the round-3 control flow, a lane-divergent skip
            if (!rot) continue;
#else
            if (__builtin_amdgcn_ballot_w64(rot) == 0ull) continue;   // wave-uniform
#endif
            const float zeta = (beta - alpha) / (2.0f * gamma);       // (inf / nan in a lane that does not rotate: discarded below)
            const float az = fabsf(zeta);
            float t = 1.0f / (az + __builtin_sqrtf(1.0f + zeta * zeta));
            t = (zeta < 0.0f) ? -t : t;
            const float c = 1.0f / __builtin_sqrtf(1.0f + t * t);
            const float s = c * t;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float wp = W[i][p], wq = W[i][q];
                const float nwp = c * wp - s * wq, nwq = s * wp + c * wq;
                W[i][p] = rot ? nwp : wp;
                W[i][q] = rot ? nwq : wq;
                const float vp = Vm[i][p], vq = Vm[i][q];
                const float nvp = c * vp - s * vq, nvq = s * vp + c * vq;
                Vm[i][p] = rot ? nvp : vp;
                Vm[i][q] = rot ? nvq : vq;
            }
        }
    }
    const float sg0 = __builtin_sqrtf((W[0][0] * W[0][0] + W[1][0] * W[1][0]) + W[2][0] * W[2][0]);
    const float sg1 = __builtin_sqrtf((W[0][1] * W[0][1] + W[1][1] * W[1][1]) + W[2][1] * W[2][1]);
    const float sg2 = __builtin_sqrtf((W[0][2] * W[0][2] + W[1][2] * W[1][2]) + W[2][2] * W[2][2]);
    // column order by descending sigma; ties keep the lower column first (same selection network as the oracle)
    auto sgv = [&](int o) { return (o == 0) ? sg0 : ((o == 1) ? sg1 : sg2); };
    int o0 = 0, o1 = 1, o2 = 2;
    if (sgv(o1) > sgv(o0)) { int t = o0; o0 = o1; o1 = t; }
    if (sgv(o2) > sgv(o0)) { int t = o0; o0 = o2; o2 = t; }
    if (sgv(o2) > sgv(o1)) { int t = o1; o1 = o2; o2 = t; }
    auto col = [&](const float (&M)[3][3], int i, int o) { return (o == 0) ? M[i][0] : ((o == 1) ? M[i][1] : M[i][2]); };
    const float s0 = sgv(o0), s1 = sgv(o1), s2 = sgv(o2);
    S[0] = s0; S[1] = s1; S[2] = s2;
#pragma unroll
    for (int i = 0; i < 3; ++i) { V[3 * i + 0] = col(Vm, i, o0); V[3 * i + 1] = col(Vm, i, o1); V[3 * i + 2] = col(Vm, i, o2); }
    if constexpr (WANT_U) {
        const float tiny = 1e-5f * s0;
        const bool zero = !(s0 > 0.0f);
        // u0 = w_o0 / s0
        const float u00 = col(W, 0, o0) / s0, u01 = col(W, 1, o0) / s0, u02 = col(W, 2, o0) / s0;
        // u1 = w_o1 / s1, or (rank 1) a unit vector orthogonal to u0: cross with the axis of u0's smallest |component|
        const float d10 = col(W, 0, o1) / s1, d11 = col(W, 1, o1) / s1, d12 = col(W, 2, o1) / s1;
        float a0 = fabsf(u00);
        const float a1 = fabsf(u01), a2 = fabsf(u02);
        int ax = 0;
        if (a1 < a0) { ax = 1; a0 = a1; }
        if (a2 < a0) { ax = 2; }
        const float e0 = ax == 0 ? 1.0f : 0.0f, e1 = ax == 1 ? 1.0f : 0.0f, e2 = ax == 2 ? 1.0f : 0.0f;
        const float cx = u01 * e2 - u02 * e1;
        const float cy = u02 * e0 - u00 * e2;
        const float cz = u00 * e1 - u01 * e0;
        const float nn = __builtin_sqrtf((cx * cx + cy * cy) + cz * cz);
        const bool full1 = s1 > tiny;
        const float u10 = full1 ? d10 : cx / nn, u11 = full1 ? d11 : cy / nn, u12 = full1 ? d12 : cz / nn;
        // u2 = w_o2 / s2, or u0 x u1
        const bool full2 = s2 > tiny;
        const float u20 = full2 ? col(W, 0, o2) / s2 : u01 * u12 - u02 * u11;
        const float u21 = full2 ? col(W, 1, o2) / s2 : u02 * u10 - u00 * u12;
        const float u22 = full2 ? col(W, 2, o2) / s2 : u00 * u11 - u01 * u10;
        U[0] = zero ? 1.0f : u00; U[1] = zero ? 0.0f : u10; U[2] = zero ? 0.0f : u20;
        U[3] = zero ? 0.0f : u01; U[4] = zero ? 1.0f : u11; U[5] = zero ? 0.0f : u21;
        U[6] = zero ? 0.0f : u02; U[7] = zero ? 0.0f : u12; U[8] = zero ? 1.0f : u22;
    }
}

// eigenvalues of a symmetric 3x3 in double (a00 a01 a02 a11 a12 a22), ascending: cyclic Jacobi, 10 fixed sweeps.
// Op for op the sequence of c_eigvals3d (oracle/src/orc_iss.cpp).
__device__ __forceinline__ void lgr_eigvals3d(const double* S, double* ev) {
    double a[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
    for (int sweep = 0; sweep < 10; ++sweep) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = 1.0 / (fabs(theta) + __builtin_sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                double c = 1.0 / __builtin_sqrt(t * t + 1.0), s = t * c;
                const int r = 3 - p - q;
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

