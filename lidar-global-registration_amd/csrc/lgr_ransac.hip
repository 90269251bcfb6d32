// lgr_ransac.hip -- prerejective RANSAC on gfx950: on-device Philox sampling, polygon prerejection, 3-point Umeyama,
// batched hypothesis verification, uniformity / score metrics, adaptive bound, final SVD refit.
//
// Replaces src/sac_prerejective_omp.cpp:115-314 (SampleConsensusPrerejectiveOMP::align), src/metric.cpp:55-179,
// src/analysis.cpp:95-130 (uniformity), src/transformation.cpp:4-38 and the PCL pieces called from there
// (CorrespondenceRejectorPoly::thresholdPolygon, TransformationEstimationSVD -> umeyama).
//
// Schedule (deterministic, order-independent; the oracle's ORC_RNG_PHILOX mode states the same thing on the CPU):
//   iteration i draws Philox4x32-10(counter = i, key = seed); iterations are processed in batches; after a batch
//   the record inlier set (largest count, ties -> lowest i) tightens the bound through estimateMaxIterations and the
//   best hypothesis is the maximum metric (strict '>', ties -> lowest i).
// Float sequences restate the oracle op for op (oracle/src/orc_ransac.cpp); compiled with -ffp-contract=off.
// Round 4: for the uniformity / correspondences metrics the loop itself runs from the device (RState, ransac_device_schedule below):
// the host enqueues two rounds and the final block blind and reads one record -- one synchronisation per alignment; the closest-plane
// metrics keep the host-driven rounds (their sparse subsets are keyed by batch).
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <vector>

#include "lgr_internal.h"
#include "lgr_math.cuh"

namespace {

constexpr int MIN_NR_INLIERS = 10;         // src/sac_prerejective_omp.cpp:8
constexpr int MIN_NR_FINAL_INLIERS = 20;   // :9
constexpr double MIN_INLIER_RATE = 0.15;   // :10

// ---------------------------------------------------------------------------------------------------- sampling
__device__ __forceinline__ void philox4x32(unsigned long long seed, unsigned iter, unsigned out[4]) { lgr_philox4(seed, iter, 0u, 0u, 0u, out); }

// src/sac_prerejective_omp.cpp:33-77 selectCorrespondences (control flow kept literally), NS = AlignmentParameters::n_samples
// The reference's loops (for i < NS: draw, for j < i: bump / wrap / insert-and-break) unrolled at compile time so that sample[] stays in
// registers (with run-time indices it lived in scratch memory).  `step` is one pass of the j loop's body at position j for the value x
// being placed: returns true for `continue` (x was bumped and stays the candidate for the next j), false for "insert x at j".
template <int NS>
__device__ __forceinline__ void select_n(const int (&r)[NS], int n_corr, int (&sample)[NS]) {
    auto step = [&](int& x, int sj) {
        if (x >= sj) {
            if (x < n_corr - 1) { x++; return true; }
            else if (sj == 0) { x = 1; return true; }
            else { x = 0; }
        }
        return false;
    };
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        int x = r[i] % n_corr;
        bool placed = false;
#pragma unroll
        for (int j = 0; j < i; ++j) {
            if (!placed && !step(x, sample[j])) {
#pragma unroll
                for (int k = i; k > j; --k) sample[k] = sample[k - 1];
                sample[j] = x;
                placed = true;
            }
        }
        if (!placed) sample[i] = x;
    }
}

// the raw draws of iteration `iter`: draw j = word j % 4 of Philox(seed; counter (iter, j / 4, 0, 0)), top 31 bits
template <int NS>
__device__ __forceinline__ void draws_n(unsigned long long seed, unsigned iter, int (&r)[NS]) {
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        if ((j & 3) == 0) lgr_philox4(seed, iter, (unsigned) (j >> 2), 0u, 0u, w);
        r[j] = (int) (w[j & 3] >> 1);
    }
}

template <int NS>
__global__ void samples_kernel(unsigned long long seed, int first, int n, int n_corr, int32_t* __restrict__ tuples) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    int r[NS], s[NS];
    draws_n<NS>(seed, (unsigned) (first + b), r);
    select_n<NS>(r, n_corr, s);
#pragma unroll
    for (int j = 0; j < NS; ++j) tuples[(size_t) NS * b + j] = s[j];
}

// ---------------------------------------------------------------------------------------------------- hypotheses
struct P3 { float x, y, z; };
__device__ __forceinline__ P3 ldp(const float* pts, int i) { const float* p = pts + (size_t) i * 12; return P3{p[0], p[1], p[2]}; }
__device__ __forceinline__ float p3c(const P3& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

// pcl::registration::CorrespondenceRejectorPoly::thresholdPolygon: every edge i -> (i + 1) % NS
template <int NS>
__device__ __forceinline__ bool poly_ok(const P3 (&s)[NS], const P3 (&t)[NS], float thr2) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int j = (i + 1) % NS;
        float dx = s[i].x - s[j].x, dy = s[i].y - s[j].y, dz = s[i].z - s[j].z;
        float ds = dx * dx + dy * dy + dz * dz;
        dx = t[i].x - t[j].x; dy = t[i].y - t[j].y; dz = t[i].z - t[j].z;
        float dt = dx * dx + dy * dy + dz * dz;
        float sim = ds < dt ? ds / dt : dt / ds;
        if (!(sim >= thr2)) return false;
    }
    return true;
}

// pcl::umeyama (no scaling) on NS pairs; T column-major.  Means and the entries of sigma are left-to-right sums over the points.
template <int NS>
__device__ __forceinline__ void umeyama_n(const P3 (&s)[NS], const P3 (&d)[NS], float* T) {
    const float one_over_n = 1.0f / (float) NS;
    float sm[3], dm[3], S[3][NS], D[3][NS];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float ss = p3c(s[0], a), ds = p3c(d[0], a);
#pragma unroll
        for (int j = 1; j < NS; ++j) { ss += p3c(s[j], a); ds += p3c(d[j], a); }
        sm[a] = ss * one_over_n;
        dm[a] = ds * one_over_n;
#pragma unroll
        for (int j = 0; j < NS; ++j) { S[a][j] = p3c(s[j], a) - sm[a]; D[a][j] = p3c(d[j], a) - dm[a]; }
    }
    float sigma[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = D[i][0] * S[j][0];
#pragma unroll
            for (int k = 1; k < NS; ++k) acc += D[i][k] * S[j][k];
            sigma[3 * i + j] = one_over_n * acc;
        }
    float U[9], Sg[3], V[9];
    lgr_svd3(sigma, U, Sg, V);
    float sgn = (lgr_det3(U) * lgr_det3(V) < 0.f) ? -1.f : 1.f;
    float R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = (U[3 * i + 0] * V[3 * j + 0] + U[3 * i + 1] * V[3 * j + 1]) + (U[3 * i + 2] * sgn) * V[3 * j + 2];
    float t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = dm[i] - ((R[3 * i + 0] * sm[0] + R[3 * i + 1] * sm[1]) + R[3 * i + 2] * sm[2]);
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) T[4 * j + i] = R[3 * i + j];
        T[12 + i] = t[i];
    }
    T[15] = 1.f;
}

// one thread per iteration of the batch: sample (or replay a given tuple) -> prerejection -> NS-point transform
template <int NS>
__global__ void hypotheses_kernel(const float* __restrict__ src, const float* __restrict__ tgt, const lgr_corr* __restrict__ corr,
                                  int c, unsigned long long seed, int first, int n, const int32_t* __restrict__ tuples,
                                  float edge_thr, float* __restrict__ Ts, int* __restrict__ ok) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    int smp[NS];
    if (tuples) {
#pragma unroll
        for (int j = 0; j < NS; ++j) smp[j] = tuples[(size_t) NS * b + j];
    } else {
        int r[NS];
        draws_n<NS>(seed, (unsigned) (first + b), r);
        select_n<NS>(r, c, smp);
    }
    P3 s[NS], t[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) { lgr_corr cr = corr[smp[j]]; s[j] = ldp(src, cr.index_query); t[j] = ldp(tgt, cr.index_match); }   // buildIndices :17-31
    bool good = poly_ok<NS>(s, t, edge_thr * edge_thr);
    float T[16];
    if (good) umeyama_n<NS>(s, t, T);
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
    }
    float4* o = reinterpret_cast<float4*>(Ts + (size_t) b * 16);
    o[0] = make_float4(T[0], T[1], T[2], T[3]); o[1] = make_float4(T[4], T[5], T[6], T[7]);
    o[2] = make_float4(T[8], T[9], T[10], T[11]); o[3] = make_float4(T[12], T[13], T[14], T[15]);
    ok[b] = good ? 1 : 0;
}

// n_samples the kernels are instantiated for (the reference's sampler, polygon test and Umeyama are generic in it; every shipped config uses 3)
constexpr int LGR_MIN_SAMPLES = 3, LGR_MAX_SAMPLES = 8;
#define LGR_NS_DISPATCH(ns, CALL)                                                                                      \
    switch (ns) {                                                                                                      \
        case 3: { constexpr int NS = 3; CALL; break; }                                                                 \
        case 4: { constexpr int NS = 4; CALL; break; }                                                                 \
        case 5: { constexpr int NS = 5; CALL; break; }                                                                 \
        case 6: { constexpr int NS = 6; CALL; break; }                                                                 \
        case 7: { constexpr int NS = 7; CALL; break; }                                                                 \
        default: { constexpr int NS = 8; CALL; break; }                                                                \
    }

// ---------------------------------------------------------------------------------------------------- packing
// pack[i] = {sx, sy, sz, thr | tx, ty, tz, bins}; sstar[i] = smallest float s with sqrt_rn(s) >= thr, so that the
// inlier test `sqrtf(s) < thr` (src/metric.cpp:141-144) is exactly `s < sstar` without a square root per pair.
__device__ __forceinline__ float next_up(float x) { return __uint_as_float(__float_as_uint(x) + 1u); }
__device__ __forceinline__ float next_down(float x) { return __uint_as_float(__float_as_uint(x) - 1u); }

// PP (count_kernel's operand): one 64-byte record per TWO correspondences, {sx sy sz | qx qy qz | s* | band slope} as 2-vectors, padded
// with never-inlier fillers to a multiple of 64 correspondences; pstats = bit patterns of max |source coordinate|, max |target
// coordinate|, max finite s* (float max through integer atomics: all values >= 0).
constexpr int CP_FLOATS = 16;
__global__ void pack_kernel(const float* __restrict__ src, const float* __restrict__ tgt, const lgr_corr* __restrict__ corr, int c,
                            const unsigned* __restrict__ bbk /* lgr_bbox_launch's keys: [6..8] the reference's min, [9..11] its max of the source cloud */,
                            float4* __restrict__ P0, float4* __restrict__ P1, float* __restrict__ sstar,
                            float* __restrict__ PP, unsigned* __restrict__ pstats, int cpad) {
    // (bbk == nullptr: the caller does not use the uniformity bins -- unit box)
    const float mnx = bbk ? lgr_bbox_key_inv(bbk[6]) : 0.f, mny = bbk ? lgr_bbox_key_inv(bbk[7]) : 0.f, mnz = bbk ? lgr_bbox_key_inv(bbk[8]) : 0.f;
    const float mxx = bbk ? lgr_bbox_key_inv(bbk[9]) : 1.f, mxy = bbk ? lgr_bbox_key_inv(bbk[10]) : 1.f, mxz = bbk ? lgr_bbox_key_inv(bbk[11]) : 1.f;
    unsigned m0 = 0u, m1 = 0u, m2 = 0u;   // this lane's contribution to pstats (padding lanes: the neutral 0)
  for (int base = blockIdx.x * blockDim.x; base < cpad; base += gridDim.x * blockDim.x) {   // (a few hundred workgroups: their statistics meet in 3 atomics each)
    const int i = base + threadIdx.x;
    if (i >= c) {
        if (PP && i < cpad) {
            float* r = PP + (size_t) (i >> 1) * CP_FLOATS + (i & 1);
#pragma unroll
            for (int f = 0; f < 8; ++f) r[2 * f] = 0.f;   // s* = 0: d2 < 0 never holds
        }
    } else {
    lgr_corr cr = corr[i];
    P3 s = ldp(src, cr.index_query), t = ldp(tgt, cr.index_match);
    float thr = cr.threshold;
    // bins of calculateCorrespondenceUniformity (src/analysis.cpp:108-112); NaN/negative pinned to 0 like the oracle
    float f0 = floorf((s.x - mnx) / (mxx - mnx) * 100), f1 = floorf((s.y - mny) / (mxy - mny) * 100), f2 = floorf((s.z - mnz) / (mxz - mnz) * 100);
    f0 = (99.f < f0) ? 99.f : f0; f1 = (99.f < f1) ? 99.f : f1; f2 = (99.f < f2) ? 99.f : f2;   // std::min(f, 99.f)
    int b0 = (f0 >= 0.f) ? (int) f0 : 0, b1 = (f1 >= 0.f) ? (int) f1 : 0, b2 = (f2 >= 0.f) ? (int) f2 : 0;
    P0[i] = make_float4(s.x, s.y, s.z, thr);
    P1[i] = make_float4(t.x, t.y, t.z, __int_as_float(b0 | (b1 << 8) | (b2 << 16)));
    float ss;
    if (!(thr > 0.f)) ss = 0.f;                           // nothing is < thr (NaN thr: nothing either)
    else if (!(thr < 3.4028234663852886e38f)) ss = thr;   // inf: every finite s qualifies, s < inf
    else {
        float g = thr * thr;
        if (!(g < 3.4028234663852886e38f)) g = 3.4028234663852886e38f;
        if (g < 1.17549435e-38f) g = 1.17549435e-38f;
        // walk to the boundary: smallest g with sqrt(g) >= thr
        for (int it = 0; it < 8 && __builtin_sqrtf(g) >= thr && g > 0.f; ++it) g = next_down(g);
        for (int it = 0; it < 16 && __builtin_sqrtf(g) < thr; ++it) g = next_up(g);
        ss = g;
    }
    sstar[i] = ss;
    if (PP) {
        float* r = PP + (size_t) (i >> 1) * CP_FLOATS + (i & 1);
        r[0] = s.x; r[2] = s.y; r[4] = s.z; r[6] = t.x; r[8] = t.y; r[10] = t.z; r[12] = ss;
        // slope of the decision band of count_kernel's fused evaluation: 28 sqrt(s*), rounded up (inf for an infinite threshold)
        r[14] = (ss < 3.4028234663852886e38f) ? next_up(28.f * __builtin_sqrtf(ss)) * 1.000001f : __uint_as_float(0x7f800000u);
        const float sm = fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fabsf(s.z)), qm = fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fabsf(t.z));
        // NaN coordinates: the integer max of the bit pattern keeps them (a NaN pattern is above every finite one) -> the band becomes NaN
        // -> every chunk takes the exact path
        m0 = max(m0, __float_as_uint(sm)); m1 = max(m1, __float_as_uint(qm)); m2 = max(m2, (ss < 3.4028234663852886e38f) ? __float_as_uint(ss) : 0u);
    }
    }
  }
    // one atomic per wave and statistic (as one per thread: 3 x 282 k updates of the same three words on the bench pair -- even one per wave of a
    // thread-per-correspondence grid was 13 k same-address atomics, most of the kernel's 0.16 ms); every lane of the wave is here
    if (PP) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m0 = max(m0, (unsigned) __shfl_xor((int) m0, o)); m1 = max(m1, (unsigned) __shfl_xor((int) m1, o)); m2 = max(m2, (unsigned) __shfl_xor((int) m2, o));
        }
        if ((threadIdx.x & 63) == 0) {
            if (m0) atomicMax(&pstats[0], m0);
            if (m1) atomicMax(&pstats[1], m1);
            if (m2) atomicMax(&pstats[2], m2);
        }
    }
}

// T (column-major) applied as Eigen's Matrix4f * Vector4f on SSE: ((c0*x + c1*y) + c2*z) + c3
#define LGR_APPLY(T, sx, sy, sz, ox, oy, oz)                       \
    float ox = ((T[0] * sx + T[4] * sy) + T[8] * sz) + T[12];      \
    float oy = ((T[1] * sx + T[5] * sy) + T[9] * sz) + T[13];      \
    float oz = ((T[2] * sx + T[6] * sy) + T[10] * sz) + T[14];

// ---------------------------------------------------------------------------------------------------- phase 1
// lane = hypothesis (T in registers), loop over a chunk of correspondences broadcast from LDS.
// counts[h] = {inliers (4-norm rule, src/metric.cpp:141), support (3-norm rule, src/metric.cpp:111)}
constexpr int CB = 64;        // hypotheses per workgroup (one wave)
constexpr int CCH = 2048;     // correspondences per workgroup (fewer when there are few hypotheses: cch of count_kernel)
// The O(H x C) verification.  Round 3: per (hypothesis, correspondence) pair the reference's expressions (LGR_APPLY, the Eigen
// 4-vector and 3-vector norms: ~25 unfused multiply / add instructions per pair) are evaluated only where they can decide something.
// A FUSED evaluation -- e_k = fma(c_k0, x, fma(c_k1, y, fma(c_k2, z, c_k3 - q_k))), d2~ = fma(e_z, e_z, fma(e_y, e_y, e_x e_x)): 15 packed
// instructions per two correspondences -- differs from both reference values by at most
//     err(x) = 7u x + 3.5 eta sqrt(x) + 3 eta^2,   eta = 12u (max_k sum_j |c_kj| * max|s| + max_k |c_k3| + max|q|),  u = 2^-24
// (4 roundings per component in either order, 3 in either sum of squares), so the sign of d2~ - s* IS the
// reference's decision whenever |d2~ - s*| > 8 err(s*) + 16 eta^2.  Per correspondence the kernel keeps the sign bit (one v_alignbit);
// a pair of correspondences for which ANY lane of the wave comes within that band (hypotheses that survive the prerejection are good
// enough that ~1e-3 of the pairs do: ~10 % of the iterations) is re-evaluated on the spot with the reference's own expressions, for
// the inlier (4-norm) and the support (3-norm) rule.  The correspondences are wave-uniform: they arrive through scalar loads
// (s_load_dwordx16 per pair record), not through LDS.
typedef float v2f_c __attribute__((ext_vector_type(2)));
struct CPair { v2f_c sx, sy, sz, qx, qy, qz, ss, rs; };
static_assert(sizeof(CPair) == CP_FLOATS * 4, "pack_kernel writes this layout");

// Inlier bit masks, hypothesis-major (round 5): row h = the mask of survivor h, mask_pitch(c) words -- a whole number of 128-correspondence groups,
// the unit count_item stores (one 16-byte store per lane and group).  The metric phase reads a candidate's row front to back; with the rows
// word-major ([word][hypothesis], coalesced stores) every word of a candidate was a cache line of its own: 14.9 GB of fetches per 1M cluster-filter
// alignment, 80 % of metric_kernel's wave cycles parked.
__host__ __device__ inline size_t mask_pitch(int c) { return (size_t) ((c + 127) >> 7) * 4; }
__device__ __forceinline__ void count_item(const int bx /* block of CB hypotheses */, const int by /* chunk of cch correspondences */,
                                           const float* Ts, const int* list, int nh,
                                           const CPair* __restrict__ PP, const unsigned* __restrict__ pstats, int c, int2* counts,
                                           unsigned* maskT /* [nh][mask_pitch(c)] inlier bits, or nullptr */, int cch, const int lane) {
    const int h = bx * CB + lane;
    const bool act = h < nh;
    float T[16];
    {
        const float4* tp = reinterpret_cast<const float4*>(Ts + (size_t) (act ? list[h] : 0) * 16);
        float4 a = tp[0], b = tp[1], cc = tp[2], d = tp[3];
        T[0] = a.x; T[1] = a.y; T[2] = a.z; T[3] = a.w; T[4] = b.x; T[5] = b.y; T[6] = b.z; T[7] = b.w;
        T[8] = cc.x; T[9] = cc.y; T[10] = cc.z; T[11] = cc.w; T[12] = d.x; T[13] = d.y; T[14] = d.z; T[15] = d.w;
    }
    // decision band of this hypothesis (see above); anything non-finite -> kh = +inf: every pair is evaluated with the reference's expressions
    const float smax = __uint_as_float(pstats[0]), qmax = __uint_as_float(pstats[1]), ssmax = __uint_as_float(pstats[2]);
    const float rowl1 = fmaxf(fmaxf(fabsf(T[0]) + fabsf(T[4]) + fabsf(T[8]), fabsf(T[1]) + fabsf(T[5]) + fabsf(T[9])), fabsf(T[2]) + fabsf(T[6]) + fabsf(T[10]));
    const float Ah = rowl1 * smax + fmaxf(fmaxf(fabsf(T[12]), fabsf(T[13])), fabsf(T[14])) + qmax;
    float eta = 7.152557373046875e-7f * Ah;                                  // 12 u
    float kh = (3.814697265625e-6f * ssmax + 56.f * eta * eta) * 1.0001f;    // 64 u s*max + 56 eta^2 >= 56 u s* + 40 eta^2
    if (!(kh < 3.4028234663852886e38f) || !(eta < 3.4028234663852886e38f)) { kh = __uint_as_float(0x7f800000u); eta = 0.f; }
    const float neg_eta = -eta;
    int ninl = 0, nsup = 0;
    uint4 wq = make_uint4(0u, 0u, 0u, 0u);   // the 128-correspondence group being assembled
    const int c0 = by * cch, c1 = min(c, c0 + cch);
    for (int base = c0; base < c1; base += 64) {
        const CPair* __restrict__ pp = PP + (base >> 1);   // wave-uniform: scalar loads
        unsigned w[2], sd[2] = {0u, 0u};                   // inlier bits; support bits that differ from them (borderline pairs only)
        CPair nxt = pp[0], nxt2 = pp[1];                   // two pair records are in flight while the current one is evaluated
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned bits = 0u;
#pragma unroll 4
            for (int jj = 0; jj < 16; ++jj) {
                const CPair p = nxt;
                nxt = nxt2;
                nxt2 = pp[min(half * 16 + jj + 2, 31)];
                v2f_c ex = T[12] - p.qx, ey = T[13] - p.qy, ez = T[14] - p.qz;
                ex = __builtin_elementwise_fma(v2f_c{T[8], T[8]}, p.sz, ex); ey = __builtin_elementwise_fma(v2f_c{T[9], T[9]}, p.sz, ey); ez = __builtin_elementwise_fma(v2f_c{T[10], T[10]}, p.sz, ez);
                ex = __builtin_elementwise_fma(v2f_c{T[4], T[4]}, p.sy, ex); ey = __builtin_elementwise_fma(v2f_c{T[5], T[5]}, p.sy, ey); ez = __builtin_elementwise_fma(v2f_c{T[6], T[6]}, p.sy, ez);
                ex = __builtin_elementwise_fma(v2f_c{T[0], T[0]}, p.sx, ex); ey = __builtin_elementwise_fma(v2f_c{T[1], T[1]}, p.sx, ey); ez = __builtin_elementwise_fma(v2f_c{T[2], T[2]}, p.sx, ez);
                v2f_c d2 = ex * ex;
                d2 = __builtin_elementwise_fma(ey, ey, d2);
                d2 = __builtin_elementwise_fma(ez, ez, d2);
                const v2f_c u = d2 - p.ss;
                // sign bit of u = "d2~ < s*" (u = -0 cannot occur: x - x is +0); correspondence 2 jj (+1) ends up at bit 31 - 2 jj (- 1)
                bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(u.x), 31);
                bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(u.y), 31);
                const float t0 = __builtin_fmaf(neg_eta, p.rs.x, fabsf(u.x)), t1 = __builtin_fmaf(neg_eta, p.rs.y, fabsf(u.y));
                if (__any(!(t0 > kh) || !(t1 > kh))) {   // wave-uniform; NaN -> taken
                    const v2f_c ox = ((T[0] * p.sx + T[4] * p.sy) + T[8] * p.sz) + T[12];     // LGR_APPLY, elementwise
                    const v2f_c oy = ((T[1] * p.sx + T[5] * p.sy) + T[9] * p.sz) + T[13];
                    const v2f_c oz = ((T[2] * p.sx + T[6] * p.sy) + T[10] * p.sz) + T[14];
                    const v2f_c dx = ox - p.qx, dy = oy - p.qy, dz = oz - p.qz;
                    const v2f_c xx = dx * dx, yy = dy * dy, zz = dz * dz;
                    const v2f_c d4 = (xx + zz) + (yy + 0.f);   // Eigen 4-vector squaredNorm reduction
                    const v2f_c d3 = (xx + yy) + zz;           // 3-vector block norm
                    const unsigned in0 = d4.x < p.ss.x ? 1u : 0u, in1 = d4.y < p.ss.y ? 1u : 0u;
                    const unsigned s0 = d3.x < p.ss.x ? 1u : 0u, s1 = d3.y < p.ss.y ? 1u : 0u;
                    bits = (bits & ~3u) | (in0 << 1) | in1;
                    sd[half] |= ((in0 ^ s0) | ((in1 ^ s1) << 1)) << (2 * jj);
                }
            }
            w[half] = __builtin_bitreverse32(bits);
        }
        ninl += __popc(w[0]) + __popc(w[1]);
        nsup += __popc(w[0] ^ sd[0]) + __popc(w[1] ^ sd[1]);
        // inlier bits of this hypothesis for the correspondences [base, base + 64): word-major, so the lanes (consecutive
        // hypotheses) store consecutive words; phase 2 walks the set bits instead of testing every correspondence again
        // (chunks are whole groups: cch is a multiple of 128; the last group of the table may end behind c: its padding pairs are never inliers)
        if (maskT) {
            if (((base - c0) & 64) == 0) { wq.x = w[0]; wq.y = w[1]; wq.z = 0u; wq.w = 0u; }
            else { wq.z = w[0]; wq.w = w[1]; }
            if (act && ((((base - c0) & 64) != 0) || base + 64 >= c1))
                *reinterpret_cast<uint4*>(maskT + (size_t) h * mask_pitch(c) + ((size_t) ((base & ~127) >> 5))) = wq;
        }
    }
    if (act) { atomicAdd(&counts[h].x, ninl); atomicAdd(&counts[h].y, nsup); }
}
// correspondences per work item: shorter chunks when there are few hypotheses (the first round, the lr filter), so that the launch still
// has a few thousand waves
__host__ __device__ inline int count_chunk(int nh, int c) {
    const long long hb = (nh + CB - 1) / CB;
    return (hb * ((c + CCH - 1) / CCH) >= 4096) ? CCH : ((hb * ((c + 511) / 512) >= 4096) ? 512 : 128);
}
__global__ __launch_bounds__(CB) void count_kernel(const float* __restrict__ Ts, const int* __restrict__ list, int nh,
                                                    const CPair* __restrict__ PP, const unsigned* __restrict__ pstats, int c, int2* __restrict__ counts,
                                                    unsigned* __restrict__ maskT, int cch) {
    count_item(blockIdx.x, blockIdx.y, Ts, list, nh, PP, pstats, c, counts, maskT, cch, threadIdx.x);
}
// the same over a work list whose size only the device knows (device-driven schedule, lgr_ransac_dev): nh = *nh_dev hypotheses, a fixed
// grid of single-wave workgroups strides over the (hypothesis block, chunk) items, hypothesis blocks fastest (neighbouring workgroups
// read the same correspondences)
__global__ __launch_bounds__(CB) void count_list_kernel(const float* __restrict__ Ts, const int* __restrict__ list, const int* __restrict__ nh_dev,
                                                         const CPair* __restrict__ PP, const unsigned* __restrict__ pstats, int c, int2* __restrict__ counts,
                                                         unsigned* __restrict__ maskT, int mask_cap /* hypotheses maskT has room for */) {
    const int nh = *nh_dev;
    if (nh <= 0) return;
    const int hb_n = (nh + CB - 1) / CB, cch = count_chunk(nh, c);
    const long long items = (long long) hb_n * ((c + cch - 1) / cch);
    unsigned* const mt = nh <= mask_cap ? maskT : nullptr;
    for (long long it = blockIdx.x; it < items; it += gridDim.x)
        count_item((int) (it % hb_n), (int) (it / hb_n), Ts, list, nh, PP, pstats, c, counts, mt, cch, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------- phase 2
// one workgroup per hypothesis with >= MIN_NR_INLIERS inliers: metric in the reference's exact summation order.
//   uniformity      : 3 x 100 x 100 int histogram of inlier source points (order-free), then
//                     entropy_k = -(sum_b p log p) in bin order, /log(1e4), cbrt of the product (src/analysis.cpp:114-129)
//   correspondences : score = sequential float sum over inliers in correspondence order (src/metric.cpp:55-81), /C
// mask (optional) receives the inlier flags; rmse_out (optional) the rmse of src/metric.cpp:147-155.
constexpr int MB = 1024;
#ifndef LGR_METRIC_GATHERS
#define LGR_METRIC_GATHERS 4   // (4 / 8 / 16 measured equal: the gathers are not what a candidate waits for)
#endif
__device__ __forceinline__ int block_excl_scan_1024(int v, int* scan /* [MB] */, int tid, int* total) {
    // wave-level inclusive scan by shuffles, then a scan over the 16 wave totals
    int lane = tid & 63, w = tid >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) scan[w] = x;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int i = 0; i < MB / 64; ++i) { int t = scan[i]; scan[i] = acc; acc += t; } scan[MB / 64] = acc; }
    __syncthreads();
    int base = scan[w];
    *total = scan[MB / 64];
    __syncthreads();
    return base + x - v;
}

// Single-transform evaluation of the uniformity metric, first half: inlier test of every correspondence (the same expressions as
// metric_kernel), inlier mask, the three projection histograms and the inlier count -- integer counts, so any order gives the
// same numbers -- spread over the whole device; metric_kernel then does the entropy part from the finished histogram.  (One
// workgroup walking 3e5 correspondences for ONE hypothesis cost 0.19 ms per evaluation, two evaluations per alignment.)
__global__ __launch_bounds__(256) void inlier_hist_kernel(const float* __restrict__ T16, const float4* __restrict__ P0, const float4* __restrict__ P1,
                                                          const float* __restrict__ sstar, int c, uint8_t* __restrict__ mask, int* __restrict__ ghist /* [30000 + 1], zeroed */) {
    __shared__ float T[16];
    if (threadIdx.x < 16) T[threadIdx.x] = T16[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (i < c) {
        float4 a = P0[i], b = P1[i];
        LGR_APPLY(T, a.x, a.y, a.z, ox, oy, oz)
        float dx = ox - b.x, dy = oy - b.y, dz = oz - b.z;
        float d4 = (dx * dx + dz * dz) + (dy * dy + 0.f);
        in = d4 < sstar[i];
        if (mask) mask[i] = in ? 1 : 0;
        if (in) {
            const int bins = __float_as_int(b.w);
            const int b0 = bins & 0xff, b1 = (bins >> 8) & 0xff, b2 = (bins >> 16) & 0xff;
            atomicAdd(&ghist[(0 * 100 + b1) * 100 + b2], 1);
            atomicAdd(&ghist[(1 * 100 + b2) * 100 + b0], 1);
            atomicAdd(&ghist[(2 * 100 + b0) * 100 + b1], 1);
        }
    }
    const unsigned long long m = __ballot(in);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&ghist[30000], __popcll(m));
}

// (the body: one workgroup of MB threads = workgroup `wg` of `n_wg`; the resident RANSAC kernel runs it as one of its phases.  No __restrict__
// on what another phase of that kernel writes.)
__device__ __forceinline__ void metric_body(const int tid, const int wg, const int n_wg, const float* Ts, const int* list2, int nh2,
                                            const float4* __restrict__ P0, const float4* __restrict__ P1,
                                            const float* __restrict__ sstar, int c, int metric_id, int score_id,
                                            float* metric_out, int* ninl_out, float* rmse_out, uint8_t* mask,
                                            float2* scratch, const unsigned* maskT, const int* hpos, int mask_nh,
                                            const int* ghist, const int* nh2_dev, const int* mask_nh_dev) {
    extern __shared__ int hist[];   // 30000 ints (uniformity) + 64 ints scan scratch
    __shared__ float T[16];
    __shared__ int s_count;
    __shared__ int s_nnz[3];
    __shared__ float ent[3];
    if (nh2_dev) {
        nh2 = *nh2_dev;
        const int cols = *mask_nh_dev;   // hypotheses count_list_kernel wrote mask columns for (its stride)
        if (cols > mask_nh) maskT = nullptr;
        mask_nh = cols;
    }
  for (int hb = wg; hb < nh2; hb += n_wg) {
    __syncthreads();   // the previous candidate of this workgroup is finished with the shared arrays
    int hyp = list2 ? list2[hb] : hb;
    if (tid < 16) T[tid] = Ts[(size_t) hyp * 16 + tid];
    const bool uni = metric_id == LGR_METRIC_UNIFORMITY;
    if (uni) for (int i = tid; i < 30000; i += MB) hist[i] = 0;
    if (tid == 0) s_count = 0;
    int* scan = hist + 30000;
    float2* lst = scratch ? scratch + (size_t) wg * c : nullptr;   // one list per workgroup
    __syncthreads();
    const bool from_hist = ghist && uni && !lst;
    if (from_hist) {
        for (int i = tid; i < 30000; i += MB) hist[i] = ghist[i];
        if (tid == 0) s_count = ghist[30000];
    }
    const bool from_bits = !from_hist && maskT && uni && !lst && !mask;
    if (from_bits) {
        // uniformity needs the inlier SET only: walk the set bits of the masks the counting phase left (a candidate has a
        // few thousand inliers among hundreds of thousands of correspondences)
        // (a row is contiguous: consecutive lanes read consecutive words, four words per lane in flight; the bins of up to MG inliers are requested
        //  before the first of them is counted: the loop is a chain of dependent gathers, MG deep instead of one)
        constexpr int MG = LGR_METRIC_GATHERS;
        const unsigned* row = maskT + (size_t) hpos[hb] * mask_pitch(c);
        const int n_words = (c + 31) >> 5;
        int cnt = 0;
        for (int w0 = tid; w0 < n_words; w0 += 4 * MB) {
            unsigned mw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) mw[k] = w0 + k * MB < n_words ? row[w0 + k * MB] : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned m = mw[k];
                cnt += __popc(m);
                const int i0 = ((w0 + k * MB) << 5) - 1;
                while (m) {
                    int bins[MG];
                    bool on[MG];
#pragma unroll
                    for (int u = 0; u < MG; ++u) {
                        on[u] = m != 0u;
                        const int i = i0 + (on[u] ? __ffs((int) m) : 1);
                        m &= m - 1u;   // (0 stays 0)
                        bins[u] = on[u] ? __float_as_int(P1[i].w) : 0;
                    }
#pragma unroll
                    for (int u = 0; u < MG; ++u) {
                        if (!on[u]) continue;
                        const int b0 = bins[u] & 0xff, b1 = (bins[u] >> 8) & 0xff, b2 = (bins[u] >> 16) & 0xff;
                        atomicAdd(&hist[(0 * 100 + b1) * 100 + b2], 1);
                        atomicAdd(&hist[(1 * 100 + b2) * 100 + b0], 1);
                        atomicAdd(&hist[(2 * 100 + b0) * 100 + b1], 1);
                    }
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if ((tid & 63) == 0 && cnt) atomicAdd(&s_count, cnt);
    }
    for (int base = 0; base < c && !from_bits && !from_hist; base += MB) {
        int i = base + tid;
        bool in = false;
        float dist = 0.f, thr = 0.f;
        int bins = 0;
        if (i < c) {
            float4 a = P0[i], b = P1[i];
            LGR_APPLY(T, a.x, a.y, a.z, ox, oy, oz)
            float dx = ox - b.x, dy = oy - b.y, dz = oz - b.z;
            float d4 = (dx * dx + dz * dz) + (dy * dy + 0.f);
            in = d4 < sstar[i];
            thr = a.w; bins = __float_as_int(b.w);
            if (in && lst) dist = __builtin_sqrtf(d4);
            if (mask) mask[i] = in ? 1 : 0;
        }
        if (uni && in) {
            int b0 = bins & 0xff, b1 = (bins >> 8) & 0xff, b2 = (bins >> 16) & 0xff;
            atomicAdd(&hist[(0 * 100 + b1) * 100 + b2], 1);   // count[k][bin[(k+1)%3]][bin[(k+2)%3]]
            atomicAdd(&hist[(1 * 100 + b2) * 100 + b0], 1);
            atomicAdd(&hist[(2 * 100 + b0) * 100 + b1], 1);
        }
        if (!lst) {
            // uniformity without an ordered list: only the inlier count is needed (order-free)
            unsigned long long m = __ballot(in);
            if ((tid & 63) == 0 && m) atomicAdd(&s_count, __popcll(m));
        } else {
            // ordered compaction of the inliers of this tile
            int tot;
            int pos = block_excl_scan_1024(in ? 1 : 0, scan, tid, &tot);
            if (in) lst[s_count + pos] = make_float2(dist, thr);
            __syncthreads();
            if (tid == 0) s_count += tot;
            __syncthreads();
        }
    }
    __syncthreads();
    int n_inl = s_count;
    if (uni) {
        // entropy_k = -(sum over bins in ascending order of p log p) / log(1e4)   (src/analysis.cpp:114-127).
        // The terms are computed in parallel, compacted IN BIN ORDER over the (dead) histogram slab, then summed
        // sequentially by one lane per projection: the reference's summation order, without 10^4 serial steps.
        float n = (float) n_inl;
        for (int k = 0; k < 3; ++k) {
            int* hk = hist + k * 10000;
            int cntv[10];
            int nz = 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                cntv[j] = tid < 1000 ? hk[tid * 10 + j] : 0;
                float p = (float) cntv[j] / n;
                nz += (cntv[j] != 0 && p != 0.f) ? 1 : 0;
            }
            int tot;
            int pos = block_excl_scan_1024(nz, scan, tid, &tot);   // its barriers retire every read of hk before the writes below
            float* tk = reinterpret_cast<float*>(hk);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                float p = (float) cntv[j] / n;
                if (cntv[j] != 0 && p != 0.f) { tk[pos] = p * lgr_logf(p); ++pos; }
            }
            if (tid == 0) s_nnz[k] = tot;
            __syncthreads();
        }
        if (tid == 0 || tid == 64 || tid == 128) {
            int k = tid >> 6;
            const float* tk = reinterpret_cast<const float*>(hist + k * 10000);
            float e = 0.f;
            int nn = s_nnz[k];
            // (the reference's order: one dependent subtraction per non-empty bin, up to 10 000 of them; the reads run ahead of the chain in blocks
            //  of sixteen -- as a plain loop every term waited for its own LDS read: 100 of the 140 us a candidate of 80 000 inliers took)
            int j = 0;
            for (; j + 16 <= nn; j += 16) {
                const float4 t0 = *reinterpret_cast<const float4*>(tk + j), t1 = *reinterpret_cast<const float4*>(tk + j + 4);
                const float4 t2 = *reinterpret_cast<const float4*>(tk + j + 8), t3 = *reinterpret_cast<const float4*>(tk + j + 12);
                e -= t0.x; e -= t0.y; e -= t0.z; e -= t0.w; e -= t1.x; e -= t1.y; e -= t1.z; e -= t1.w;
                e -= t2.x; e -= t2.y; e -= t2.z; e -= t2.w; e -= t3.x; e -= t3.y; e -= t3.z; e -= t3.w;
            }
            for (; j < nn; ++j) e -= tk[j];
            e /= 9.210340371976184f;
            ent[k] = e;
        }
        __syncthreads();
        if (tid == 0) {
            float m = n_inl == 0 ? 0.f : lgr_cbrtf(ent[0] * ent[1] * ent[2]);
            metric_out[hb] = m;
            ninl_out[hb] = n_inl;
        }
    }
    // The score and the rmse are sequential float sums over the inliers in correspondence order (src/metric.cpp:55-81, 147-155); their TERMS are not:
    // the whole workgroup turns the list's (distance, threshold) entries into (distance^2, score term) in place -- the same expressions, by another
    // thread -- and one lane adds them up with its reads running sixteen terms ahead of the two dependent chains.  (Round 5: as one loop on one lane,
    // every term waited for its own global load and its division.)
    if (lst && (!uni || rmse_out)) {   // (workgroup uniform)
        for (int j = tid; j < n_inl; j += MB) {
            const float2 dt = lst[j];
            const float d = dt.x, t = dt.y;
            float value = 1.f;
            if (score_id == LGR_SCORE_MAE) value = fabsf(d - t) / t;
            else if (score_id == LGR_SCORE_MSE) value = (d - t) * (d - t) / (t * t);
            else if (score_id == LGR_SCORE_EXP) value = lgr_expf(-d * d / (2 * t * t));
            lst[j] = make_float2(d * d, value);
        }
        __syncthreads();
    }
    if (tid == 0 && (!uni || rmse_out)) {
        float score = 0.f, rm = 0.f;
        if (lst) {
            int j = 0;
            for (; j + 8 <= n_inl; j += 8) {
                float2 e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = lst[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) { rm += e[u].x; score += e[u].y; }
            }
            for (; j < n_inl; ++j) { const float2 e = lst[j]; rm += e.x; score += e.y; }
        }
        if (!uni) { metric_out[hb] = score / (float) c; ninl_out[hb] = n_inl; }
        if (rmse_out) rmse_out[hb] = n_inl ? __builtin_sqrtf(rm / (float) n_inl) : 3.4028234663852886e38f;
    }
  }
}
__global__ __launch_bounds__(MB) void metric_kernel(const float* __restrict__ Ts, const int* __restrict__ list2, int nh2,
                                                     const float4* __restrict__ P0, const float4* __restrict__ P1,
                                                     const float* __restrict__ sstar, int c, int metric_id, int score_id,
                                                     float* __restrict__ metric_out, int* __restrict__ ninl_out,
                                                     float* __restrict__ rmse_out, uint8_t* __restrict__ mask,
                                                     float2* __restrict__ scratch /* [gridDim.x][c] inlier (dist, thr) lists */,
                                                     const unsigned* __restrict__ maskT /* count_kernel's inlier bits [mask_nh][mask_pitch(c)], or nullptr */,
                                                     const int* __restrict__ hpos /* candidate -> row of maskT */, int mask_nh,
                                                     const int* __restrict__ ghist = nullptr /* [30000 + 1]: the uniformity histogram and the inlier count of the ONE
                                                        hypothesis, already counted by inlier_hist_kernel (single-transform evaluations) */,
                                                     const int* __restrict__ nh2_dev = nullptr /* device-driven schedule: the number of candidates lives on the
                                                        device and the grid strides over them; maskT is used when mask_nh_dev[0] <= mask_nh */,
                                                     const int* __restrict__ mask_nh_dev = nullptr) {
    metric_body(threadIdx.x, blockIdx.x, gridDim.x, Ts, list2, nh2, P0, P1, sstar, c, metric_id, score_id, metric_out, ninl_out, rmse_out, mask, scratch, maskT, hpos, mask_nh,
                ghist, nh2_dev, mask_nh_dev);
}

// ---------------------------------------------------------------------------------------------------- batch reduce
constexpr int MAX_ROUND_BATCHES = 16;   // batches of the schedule evaluated per round of launches
struct BatchStats {
    unsigned long long best_key;   // (metric bits << 32) | (0xffffffff - batch offset); 0 = none
    unsigned long long rec_key;    // (n_inl << 32) | (0xffffffff - batch offset); 0 = none
    int n_ok, n_cand, rec_support, pad;
};

__global__ void flag_ge_kernel(const int2* __restrict__ counts, int nh, int min_inliers, int* __restrict__ flags) {
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < nh) flags[h] = counts[h].x >= min_inliers ? 1 : 0;
}
// closest-plane metric plumbing: inlier counts of the plane test replace the correspondence counts (the estimator's
// `inliers` are the plane pairs, src/metric.cpp:187-199); candidates pick up their plane metric; combination multiplies
__global__ void plane_counts_kernel(const int* __restrict__ cnt, int nh, int2* __restrict__ counts) {
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < nh) counts[h].x = cnt[h];
}
__global__ void plane_pick_kernel(const int* __restrict__ flags, const int* __restrict__ pos, int nh, const int* __restrict__ cnt,
                                  const float* __restrict__ cp, float* __restrict__ metric, int* __restrict__ ninl) {
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < nh && flags[h]) { metric[pos[h]] = cp[h]; ninl[pos[h]] = cnt[h]; }
}
__global__ void plane_mul_kernel(float* __restrict__ metric, const float* __restrict__ cp, int n) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) metric[j] = metric[j] * cp[j];   // metric_cs * metric_cp (src/metric.cpp:248)
}
__global__ void plane_pack_kernel(const float* __restrict__ src, const float* __restrict__ tgt, const int2* __restrict__ pairs, int n,
                                  float4* __restrict__ P0, float4* __restrict__ P1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* s = src + (size_t) pairs[i].x * 12;
    const float* t = tgt + (size_t) pairs[i].y * 12;
    P0[i] = make_float4(s[0], s[1], s[2], 0.f);
    P1[i] = make_float4(t[0], t[1], t[2], 0.f);
}
__global__ void compact_kernel(const int* __restrict__ flags, const int* __restrict__ pos, int n, const int* __restrict__ map,
                               int* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) out[pos[i]] = map ? map[i] : i;
}
// Batch statistics, one BatchStats per sub-batch of `sub` iterations (several batches of the reference schedule are
// evaluated by one round of launches; the host replays them in order, see lgr_ransac_dev).
__global__ void reduce_kernel(const int* __restrict__ list2, int nh2, const float* __restrict__ metric, const int* __restrict__ ninl,
                              int sub, BatchStats* __restrict__ st) {
    // list2[j] = offset of candidate j in the round; metric[j], ninl[j] its phase-2 results
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nh2) return;
    unsigned off = (unsigned) list2[j];
    BatchStats* t = st + off / (unsigned) sub;
    atomicMax(&t->best_key, ((unsigned long long) __float_as_uint(metric[j]) << 32) | (0xffffffffu - off));
    atomicMax(&t->rec_key, ((unsigned long long) (unsigned) ninl[j] << 32) | (0xffffffffu - off));
}
__global__ void support_kernel(const int* __restrict__ list, const int2* __restrict__ counts, int nh, int sub, BatchStats* __restrict__ st) {
    // per sub-batch: number of prerejection survivors, and the support count of the record holder
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nh) return;
    int off = list[h];
    BatchStats* t = st + off / sub;
    atomicAdd(&t->n_ok, 1);
    unsigned long long rk = t->rec_key;
    if (rk != 0 && (int) (0xffffffffu - (unsigned) (rk & 0xffffffffu)) == off) t->rec_support = counts[h].y;
}

// ordered compaction of the inlier pairs (mask -> flags -> exclusive scan -> scatter) ahead of the sequential refit
__global__ void mask_flags_kernel(const uint8_t* __restrict__ mask, int c, int* __restrict__ flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) flags[i] = mask[i] ? 1 : 0;
}
__global__ void compact_pairs_kernel(const float4* __restrict__ P0, const float4* __restrict__ P1, const int* __restrict__ flags,
                                     const int* __restrict__ pos, int c, float4* __restrict__ Q0, float4* __restrict__ Q1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c && flags[i]) { Q0[pos[i]] = P0[i]; Q1[pos[i]] = P1[i]; }
}

// ---------------------------------------------------------------------------------------------------- refit
// src/transformation.cpp:4-38: sequential float sums over the inliers in correspondence order.  Lanes 0..5 own the
// six centroid accumulators, then lanes 0..8 the nine entries of H; the SVD and R, t follow on lane 0.
constexpr int RCH = 2048;   // pairs staged per chunk
__global__ __launch_bounds__(256) void refit_kernel(const float4* __restrict__ P0, const float4* __restrict__ P1, const uint8_t* __restrict__ mask, int c,
                                                    float* __restrict__ Tout, const int* __restrict__ n_a = nullptr, const int* __restrict__ n_b = nullptr) {
    if (n_a) c = n_a[0] + n_b[0];   // device-driven schedule: the number of compacted pairs = last exclusive-scan entry + last flag
    // the sums are sequential by definition; the pairs are staged through LDS by the whole block (coalesced loads), so the
    // summing lanes walk LDS instead of waiting on one global load per term
    // (round 5: the staged chunk is COMPONENT-major, so a summing lane reads consecutive words; in the second pass the whole block also forms the
    //  nine products per pair -- the same two subtractions and one multiplication, by another thread -- and the summing lanes are left with one LDS
    //  read and the one dependent addition per term: 80 000 inliers 2.0 -> ~1 ms, the lanes were bound by instruction issue, not by the chain)
    constexpr int RCH2 = 1024;        // pairs per chunk of the second pass (nine products per pair in the same array)
    __shared__ float sp[RCH * 8];     // pass 1: [6 components][RCH]; pass 2: [9 products][RCH2]
    __shared__ uint8_t sm[RCH];
    __shared__ float cen[6];
    __shared__ float Hs[9];
    __shared__ int sn;
    const int l = threadIdx.x;
    {
        float acc = 0.f;
        int n = 0;
        for (int i0 = 0; i0 < c; i0 += RCH) {
            __syncthreads();
            for (int i = l; i < RCH && i0 + i < c; i += blockDim.x) {
                const float4 p = P0[i0 + i], q = P1[i0 + i];
                sp[0 * RCH + i] = p.x; sp[1 * RCH + i] = p.y; sp[2 * RCH + i] = p.z;
                sp[3 * RCH + i] = q.x; sp[4 * RCH + i] = q.y; sp[5 * RCH + i] = q.z;
                sm[i] = mask ? mask[i0 + i] : (uint8_t) 1;
            }
            __syncthreads();
            if (l < 6) {
                const int m = min(RCH, c - i0);
                const float* col = sp + l * RCH;
                if (!mask) {   // (every caller compacts first: no per-element branch, so the LDS reads run ahead of the dependent adds)
#pragma unroll 16
                    for (int i = 0; i < m; ++i) acc += col[i];
                    n += m;
                } else {
                    for (int i = 0; i < m; ++i) {
                        if (!sm[i]) continue;
                        acc += col[i];
                        ++n;
                    }
                }
            }
        }
        if (l < 6) cen[l] = acc / (float) n;
        if (l == 0) sn = n;
    }
    {
        float acc = 0.f;
        for (int i0 = 0; i0 < c; i0 += RCH2) {
            __syncthreads();   // (the first one also publishes cen[])
            float cc[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) cc[k] = cen[k];
            for (int i = l; i < RCH2 && i0 + i < c; i += blockDim.x) {
                const float4 p = P0[i0 + i], q = P1[i0 + i];
                const float da[3] = {p.x - cc[0], p.y - cc[1], p.z - cc[2]}, db[3] = {q.x - cc[3], q.y - cc[4], q.z - cc[5]};
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) sp[(3 * a + b) * RCH2 + i] = da[a] * db[b];
                sm[i] = mask ? mask[i0 + i] : (uint8_t) 1;
            }
            __syncthreads();
            if (l < 9) {
                const int m = min(RCH2, c - i0);
                const float* col = sp + l * RCH2;
                if (!mask) {
#pragma unroll 16
                    for (int i = 0; i < m; ++i) acc += col[i];
                } else {
                    for (int i = 0; i < m; ++i) {
                        if (!sm[i]) continue;
                        acc += col[i];
                    }
                }
            }
        }
        if (l < 9) Hs[l] = acc;
    }
    __syncthreads();
    if (l == 0) {
        float T[16];
        if (sn == 0) {
_Pragma("unroll")
            for (int i = 0; i < 16; ++i) T[i] = __uint_as_float(0x7fc00000u);   // 0/0 centroids in the reference
        } else {
            float H[9], U[9], Sg[3], V[9], R[9];
            _Pragma("unroll") for (int i = 0; i < 9; ++i) H[i] = Hs[i];
            lgr_svd3(H, U, Sg, V);
            _Pragma("unroll") for (int i = 0; i < 3; ++i)
                _Pragma("unroll") for (int j = 0; j < 3; ++j)
                    R[3 * i + j] = (V[3 * i + 0] * U[3 * j + 0] + V[3 * i + 1] * U[3 * j + 1]) + V[3 * i + 2] * U[3 * j + 2];
            if (lgr_det3(R) < 0.f) {
                V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8];
                _Pragma("unroll") for (int i = 0; i < 3; ++i)
                    _Pragma("unroll") for (int j = 0; j < 3; ++j)
                        R[3 * i + j] = (V[3 * i + 0] * U[3 * j + 0] + V[3 * i + 1] * U[3 * j + 1]) + V[3 * i + 2] * U[3 * j + 2];
            }
            float t[3];
            _Pragma("unroll") for (int i = 0; i < 3; ++i) t[i] = cen[3 + i] - ((R[3 * i + 0] * cen[0] + R[3 * i + 1] * cen[1]) + R[3 * i + 2] * cen[2]);
            _Pragma("unroll") for (int i = 0; i < 16; ++i) T[i] = 0.f;
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {
                _Pragma("unroll") for (int j = 0; j < 3; ++j) T[4 * j + i] = R[3 * i + j];
                T[12 + i] = t[i];
            }
            T[15] = 1.f;
        }
_Pragma("unroll")
        for (int i = 0; i < 16; ++i) Tout[i] = T[i];
    }
}

// ---------------------------------------------------------------------------------------------------- device-driven schedule
// The whole loop of SampleConsensusPrerejectiveOMP::align (src/sac_prerejective_omp.cpp:156-257) without the host in it (round 4; the
// closest-plane metrics keep the host-driven rounds).  The schedule's state -- iterations done, the adaptive bound, the record inlier set,
// the best metric and transform -- lives in an RState on the device.  A ROUND (the first: one batch; then up to MAX_ROUND_BATCHES) is six
// launches whose sizes are upper bounds and whose real extents are read from the RState:
//   rs_begin    the round's iteration range, the candidate gate, counters and BatchStats cleared
//   rs_hyp      sample -> prerejection -> 3-point transform; survivors appended to a list (one atomic per wave; ANY list order gives the
//               same results: every later reduction carries the iteration number as its tie-break)
//   count_list  the O(H x C) verification over (hypothesis block, chunk) items, a fixed grid striding over them
//   rs_cand     hypotheses with enough inliers for the gate -> candidate list
//   metric      the metric of every candidate, a fixed grid striding over them
//   rs_replay   per-batch statistics, then ONE lane replays the round's batches in schedule order exactly as the host did: best hypothesis
//               (strict >, ties -> lowest iteration), record inlier set -> estimateMaxIterations (src/metric.cpp:103-123, in double) ->
//               bound, batches behind the end of the loop discarded
// and every kernel returns at once when the loop has ended.  The host enqueues two rounds and the final block (evaluation of the best
// transform, refit over its inliers, evaluation of the refit: also sized on the device) blind and then reads ONE record; when the loop
// has not ended by then (max_iterations far above two rounds and no record yet) it repeats.  The adaptive bound is evaluated with the
// device's double-precision log / pow, the oracle with libm's: both are accurate to an ulp, the bound is the integer part of a quotient of
// the two, so a difference needs a quotient within ~1e-15 of an integer.
struct RState {
    int done, bound, max_iterations, batch, round_cap;
    int largest, num_rejections, best_iter;
    float final_metric;
    int min_inliers;
    int round_first, round_nb, round_batches;
    int n_ok, n_cand;
    int stop;
    int rounds;
    int metric_id, c, nr_samples;
    float confidence;
    unsigned bar_count, bar_gen; int abort;                   // the resident kernel's grid barrier (arrivals, generation) and its time-out flag
    float best_T[16];
    float Tn[16];
    float e_metric; int e_ninl; float e_rmse; int pad1;       // evaluation of best_T (the final block, :265-296)
    float e2_metric; int e2_ninl; float e2_rmse; int pad2;    // evaluation of the refit
    int int_max, tot_ok, tot_cand, pad3;                       // (int_max: a constant the plane gate reads where the records are not plane counts; tot_*: survivors / candidates over all rounds, for LGR_RANSAC_DEBUG)
    unsigned long long phase_ticks[8];                         // resident kernel: 100 MHz ticks workgroup 0 spent per phase incl. its barrier (begin/replay, hyp, count, cand, metric)
    unsigned long long busy_max[8], busy_sum[8];               // ... and the workgroups' own work per phase (without the barrier): maximum and sum over the workgroups
};

__device__ __noinline__ int est_from_support_dev(int count, int c, float confidence, int nr_samples) {   // = est_from_support below
    // (a real call: inlined, its double-precision log / pow bring ~100 registers of constants that the resident kernel's loop would carry)
    float frac = (float) count / (float) c;
    frac /= 4.f;
    if (frac <= 0.0 || log(1.0 - pow((double) frac, (double) nr_samples)) >= 0.0) return INT_MAX;
    const double iterations = log(1.0 - (double) confidence) / log(1.0 - pow((double) frac, (double) nr_samples));
    return (int) fmin((double) INT_MAX, iterations);
}

__device__ __noinline__ int rs_gate_dev(float final_metric, int metric_id, int c) {   // (a real call for the same reason as est_from_support_dev)
    int mi = MIN_NR_INLIERS;
    if (final_metric > 0.f && metric_id == LGR_METRIC_UNIFORMITY) mi = max(mi, (int) floor(pow(10000.0, (double) final_metric / 1.001)) - 1);
    else if (final_metric > 0.f && metric_id == LGR_METRIC_CORRESPONDENCES) mi = max(mi, (int) floor((double) final_metric * (double) c / 1.001) - 1);
    return mi;
}
__device__ __forceinline__ void rs_begin_body(RState* S, BatchStats* st, int first_round) {
    if (threadIdx.x < MAX_ROUND_BATCHES) { BatchStats z{}; st[threadIdx.x] = z; }
    if (threadIdx.x != 0) return;
    S->n_ok = 0; S->n_cand = 0; S->round_nb = 0; S->round_batches = 0;
    if (S->stop) return;
    if (S->done >= S->bound || S->done >= S->max_iterations) { S->stop = 1; return; }
    const long long want = (long long) min(S->bound, S->max_iterations) - S->done;
    int n_batches = first_round ? 1 : (int) min((long long) S->round_cap, (want + S->batch - 1) / S->batch);
    const int nb = (int) min((long long) n_batches * S->batch, (long long) S->max_iterations - S->done);
    n_batches = (nb + S->batch - 1) / S->batch;
    // candidate gate (see lgr_ransac_dev): a hypothesis whose metric cannot reach the best one so far is not scored
    int mi = rs_gate_dev(S->final_metric, S->metric_id, S->c);
    // ... and it must not hide a RECORD inlier set (:224-228 feed the adaptive bound from every hypothesis with >= MIN_NR_INLIERS): only counts
    // up to the record so far are safe to drop.  When the best metric is a loop hypothesis's, its own count already is >= the gate and <= the
    // record, so this changes nothing; a GUESS (:134-147) sets the metric to beat without ever entering the record.
    mi = min(mi, max(MIN_NR_INLIERS, S->largest + 1));
    S->min_inliers = mi;
    S->round_first = S->done; S->round_nb = nb; S->round_batches = n_batches;
    S->rounds += 1;
}
__global__ void rs_begin_kernel(RState* __restrict__ S, BatchStats* __restrict__ st, int first_round) { rs_begin_body(S, st, first_round); }

// iteration `b` of the round (b - lane is wave-uniform; a wave wholly behind the round's end does nothing)
// (round 5: the sampled pairs come from the packed correspondences -- P0[i].xyz / P1[i].xyz ARE the source / target point of correspondence i, 32
// contiguous bytes per sample in a 9 MB array instead of a 16-byte record and two 12-byte gathers out of the 48 MB clouds)
template <int NS>
__device__ __forceinline__ void rs_hyp_item(const int b, const float4* __restrict__ P0, const float4* __restrict__ P1, int c,
                                            unsigned long long seed, RState* S, float edge_thr, float* Ts, int* list, int* posmap, int2* counts) {
    const int nb = S->round_nb;
    bool good = false;
    if (b < nb) {
        counts[b] = make_int2(0, 0);   // (list positions are < the survivors' number <= nb)
        int r[NS], smp[NS];
        draws_n<NS>(seed, (unsigned) (S->round_first + b), r);
        select_n<NS>(r, c, smp);
        P3 s[NS], t[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) { const float4 a = P0[smp[j]], q = P1[smp[j]]; s[j] = P3{a.x, a.y, a.z}; t[j] = P3{q.x, q.y, q.z}; }
        good = poly_ok<NS>(s, t, edge_thr * edge_thr);
        if (good) {
            float T[16];
            umeyama_n<NS>(s, t, T);
            float4* o = reinterpret_cast<float4*>(Ts + (size_t) b * 16);
            o[0] = make_float4(T[0], T[1], T[2], T[3]); o[1] = make_float4(T[4], T[5], T[6], T[7]);
            o[2] = make_float4(T[8], T[9], T[10], T[11]); o[3] = make_float4(T[12], T[13], T[14], T[15]);
        }
    }
    const unsigned long long m = __ballot(good);
    if (m == 0ull) return;
    int base = 0;
    const int lane = threadIdx.x & 63;
    if (lane == 0) base = atomicAdd(&S->n_ok, __popcll(m));
    base = __shfl(base, 0);
    if (good) {
        const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
        list[pos] = b;
        posmap[b] = pos;
    }
}
template <int NS>
__global__ void rs_hyp_kernel(const float4* __restrict__ P0, const float4* __restrict__ P1, int c,
                              unsigned long long seed, RState* __restrict__ S, float edge_thr, float* __restrict__ Ts, int* __restrict__ list,
                              int* __restrict__ posmap, int2* __restrict__ counts) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b - (int) threadIdx.x >= S->round_nb) return;   // (whole workgroup; S->stop leaves round_nb = 0)
    rs_hyp_item<NS>(b, P0, P1, c, seed, S, edge_thr, Ts, list, posmap, counts);
}

// candidates = survivors with enough inliers for the gate.  Up to 2^17 survivors ONE workgroup compacts them in list order (an ordered
// block scan per 1024 survivors): the metric kernel walks the inlier-mask column hpos[candidate] of every candidate, a 4-byte read per
// 128 KB row -- neighbouring workgroups then share the cache lines of neighbouring columns (with the candidates in arrival order of an
// atomic append the same launch took 1.9 instead of 0.5 ms at 9.5 k candidates).  Beyond that all workgroups append unordered.
__device__ __forceinline__ void rs_cand_body(const int tid, const int wg, const int n_wg, RState* S, const int2* counts, const int* list, int* list2, int* hpos) {
    __shared__ int scan[1024 / 64 + 2];
    const int n_ok = S->n_ok, mi = S->min_inliers;
    if (n_ok <= (1 << 17)) {
        if (wg != 0) return;
        int total = 0;
        for (int h0 = 0; h0 < n_ok; h0 += 1024) {
            const int h = h0 + tid;
            const bool cand = h < n_ok && counts[h].x >= mi;
            int tot;
            const int pos = total + block_excl_scan_1024(cand ? 1 : 0, scan, tid, &tot);
            if (cand) { list2[pos] = list[h]; hpos[pos] = h; }
            total += tot;
        }
        if (tid == 0) S->n_cand = total;
        return;
    }
    for (int h0 = wg * 1024; h0 < n_ok; h0 += n_wg * 1024) {
        const int h = h0 + tid;
        const bool cand = h < n_ok && counts[h].x >= mi;
        const unsigned long long m = __ballot(cand);
        if (m == 0ull) continue;
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&S->n_cand, __popcll(m));
        base = __shfl(base, 0);
        if (cand) {
            const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
            list2[pos] = list[h];
            hpos[pos] = h;
        }
    }
}

__global__ __launch_bounds__(1024) void rs_cand_kernel(RState* __restrict__ S, const int2* __restrict__ counts, const int* __restrict__ list,
                                                       int* __restrict__ list2, int* __restrict__ hpos) {
    rs_cand_body(threadIdx.x, blockIdx.x, gridDim.x, S, counts, list, list2, hpos);
}

__device__ __forceinline__ void rs_replay_body(const int tid, RState* S, const int* list, const int* list2, const float* metric, const int* ninl, const int2* counts,
                                               const int* posmap, const float* Ts, BatchStats* st_out) {
    __shared__ unsigned long long best_key[MAX_ROUND_BATCHES], rec_key[MAX_ROUND_BATCHES];
    __shared__ int n_ok_b[MAX_ROUND_BATCHES];
    if (S->stop || S->round_nb == 0) return;
    if (tid == 0) { S->tot_ok += S->n_ok; S->tot_cand += S->n_cand; }
    if (tid < MAX_ROUND_BATCHES) { best_key[tid] = 0ull; rec_key[tid] = 0ull; n_ok_b[tid] = 0; }
    __syncthreads();
    const int batch = S->batch, n_ok = S->n_ok, n_cand = S->n_cand;
    for (int j = tid; j < n_cand; j += blockDim.x) {
        const unsigned off = (unsigned) list2[j];
        const int bi = (int) (off / (unsigned) batch);
        atomicMax(&best_key[bi], ((unsigned long long) __float_as_uint(metric[j]) << 32) | (0xffffffffu - off));
        atomicMax(&rec_key[bi], ((unsigned long long) (unsigned) ninl[j] << 32) | (0xffffffffu - off));
    }
    for (int h = tid; h < n_ok; h += blockDim.x) atomicAdd(&n_ok_b[list[h] / batch], 1);
    __syncthreads();
    if (tid != 0) return;
    int done = S->done, bound = S->bound, largest = S->largest, num_rej = S->num_rejections, best_iter = S->best_iter;
    float final_metric = S->final_metric;
    const int round_first = done, max_it = S->max_iterations;
    int best_off = -1;
    for (int j = 0; j < S->round_batches && done < bound; ++j) {
        const int nbj = min(batch, max_it - done);
        num_rej += nbj - n_ok_b[j];
        st_out[j].n_ok = n_ok_b[j]; st_out[j].best_key = best_key[j]; st_out[j].rec_key = rec_key[j];
        if (best_key[j]) {
            const float m = __uint_as_float((unsigned) (best_key[j] >> 32));
            const int off = (int) (0xffffffffu - (unsigned) (best_key[j] & 0xffffffffull));
            if (final_metric < m) { final_metric = m; best_iter = round_first + off; best_off = off; }   // src/sac_prerejective_omp.cpp:232-235 / :251-254
        }
        if (rec_key[j]) {
            const int rec_inl = (int) (rec_key[j] >> 32);
            if (rec_inl > largest) {   // :224-228
                largest = rec_inl;
                const int off = (int) (0xffffffffu - (unsigned) (rec_key[j] & 0xffffffffull));
                bound = min(bound, est_from_support_dev(counts[posmap[off]].y, S->c, S->confidence, S->nr_samples));
            }
        }
        done += nbj;
        if (done >= max_it) break;
    }
    if (best_off >= 0)
        for (int i = 0; i < 16; ++i) S->best_T[i] = Ts[(size_t) best_off * 16 + i];
    S->done = done; S->bound = bound; S->largest = largest; S->num_rejections = num_rej; S->best_iter = best_iter; S->final_metric = final_metric;
    if (done >= bound || done >= max_it) S->stop = 1;
}
__global__ __launch_bounds__(1024) void rs_replay_kernel(RState* __restrict__ S, const int* __restrict__ list, const int* __restrict__ list2,
                                                          const float* __restrict__ metric, const int* __restrict__ ninl, const int2* __restrict__ counts,
                                                          const int* __restrict__ posmap, const float* __restrict__ Ts, BatchStats* __restrict__ st_out) {
    rs_replay_body(threadIdx.x, S, list, list2, metric, ninl, counts, posmap, Ts, st_out);
}

// ---------------------------------------------------------------------------------------------------- the resident loop
// north_star's "persistent-thread RANSAC": the phases of a round above as ONE kernel whose workgroups stay resident for the whole loop and hand
// over at grid barriers.  One workgroup of 1024 threads per CU (the metric phase's shape: 117 KB of LDS for the uniformity histogram); the
// hypothesis phase strides threads over the round's iterations, the counting phase strides WAVES over the (hypothesis block, chunk) items,
// the candidate / replay phases run in workgroup 0, the metric phase strides workgroups over the candidates.  The state (RState) and every
// list live in global memory exactly as in the launch chain, so the results are the chain's bit for bit.
//   * barrier: arrivals counted with a device-scope atomic, the last arrival bumps a generation word the others poll with device-scope loads;
//     a release fence before arriving and an acquire fence after leaving carry the phase's plain stores across the XCDs' L2s.
//   * every wave reaches the end of the kernel: the loop runs at most `max_rounds` rounds (the host's bound: every round consumes at least one
//     batch), and a workgroup that polls longer than RS_BARRIER_TICKS of the 100 MHz wall clock sets `abort`, on which every workgroup leaves
//     at its next poll (the host then reports LGR_ERR_HIP; it cannot happen unless a workgroup of the grid never becomes resident).
constexpr unsigned long long RS_BARRIER_TICKS = 200000000ull;   // 2 s
__device__ __forceinline__ bool rs_grid_barrier(RState* S, const unsigned n_wg) {
    __shared__ int s_go;   // the workgroup's one reading of `abort` (every wave takes the same way out)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();   // release: this workgroup's stores of the phase (the barrier above ordered the other waves' before this one)
        const unsigned gen = __hip_atomic_load(&S->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&S->bar_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n_wg - 1u) {
            __hip_atomic_store(&S->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&S->bar_gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(&S->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(8);
                if (__hip_atomic_load(&S->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (wall_clock64() - t0 > RS_BARRIER_TICKS) { __hip_atomic_store(&S->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __threadfence();   // acquire: the other workgroups' stores
        s_go = __hip_atomic_load(&S->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    }
    __syncthreads();
    return s_go != 0;
}

// what the loop WRITES (and reads back in a later phase) travels in this struct; what it only reads are kernel arguments of their own with
// const __restrict__: only those may go through scalar loads (count_item's wave-uniform pair records) -- a pointer out of a struct carries
// no such promise, and the counting phase then fetched every record once per LANE (measured: twice the time, 80 spilled registers)
struct ResidentArgs {
    int c; int n_samples;
    unsigned long long seed; float edge_thr;
    RState* S; BatchStats* st;
    float* Ts; int* list; int* posmap; int2* counts; int* list2; int* hpos; float* metric; int* ninl;
    unsigned* maskT; int mask_cap; float2* scratch;
    int metric_id, score_id, max_rounds;
};

__global__ __launch_bounds__(MB) void rs_resident_kernel(const ResidentArgs a, const float* __restrict__ in_src, const float* __restrict__ in_tgt,
                                                         const lgr_corr* __restrict__ in_corr, const CPair* __restrict__ in_PP,
                                                         const unsigned* __restrict__ in_pstats, const float4* __restrict__ in_P0,
                                                         const float4* __restrict__ in_P1, const float* __restrict__ in_sstar) {
    const int wg = blockIdx.x, n_wg = gridDim.x, tid = threadIdx.x;
    RState* const S = a.S;
    if (wg == 0) rs_begin_body(S, a.st, 1);
    unsigned long long t_prev = wall_clock64();
    auto stamp = [&](int phase) {   // (workgroup 0's view of where the loop's time goes: LGR_RANSAC_DEBUG prints it)
        if (wg == 0 && tid == 0) { const unsigned long long t = wall_clock64(); S->phase_ticks[phase] += t - t_prev; t_prev = t; }
    };
    unsigned long long busy[5] = {0, 0, 0, 0, 0}, t_in = t_prev;
    auto work_begin = [&]() { if (tid == 0) t_in = wall_clock64(); };
    auto work_end = [&](int phase) { if (tid == 0) busy[phase] += wall_clock64() - t_in; };
    for (int round = 0; round < a.max_rounds; ++round) {
        if (!rs_grid_barrier(S, n_wg)) return;
        stamp(0);
        if (S->stop) {
            if (tid == 0)
                for (int k = 1; k < 5; ++k) { atomicMax(&S->busy_max[k], busy[k]); atomicAdd(&S->busy_sum[k], busy[k]); }
            return;
        }
        work_begin();   // (written by workgroup 0 in front of the barrier: every workgroup reads the same value)
        // ---- sample -> prerejection -> transform
        {
            const int nb = S->round_nb;
            int tid_h = tid;
            asm volatile("" : "+v"(tid_h));   // (opaque per round and phase: per-thread values are recomputed in the phase that uses them instead of being carried -- spilled -- through the others)
            for (int b0 = wg * MB + (tid_h & ~63); b0 < nb; b0 += n_wg * MB) {
                const int b = b0 + (tid_h & 63);
                LGR_NS_DISPATCH(a.n_samples, (rs_hyp_item<NS>(b, in_P0, in_P1, a.c, a.seed, S, a.edge_thr, a.Ts, a.list, a.posmap, a.counts)));
            }
        }
        work_end(1);
        if (!rs_grid_barrier(S, n_wg)) return;
        stamp(1);
        work_begin();
        // ---- inlier counts of the survivors
        {
            const int nh = S->n_ok;
            if (nh > 0) {
                const int hb_n = (nh + CB - 1) / CB, cch = count_chunk(nh, a.c);
                const long long items = (long long) hb_n * ((a.c + cch - 1) / cch);
                unsigned* const mt = nh <= a.mask_cap ? a.maskT : nullptr;
                int tid_c = tid;
                asm volatile("" : "+v"(tid_c));
                const int wave = __builtin_amdgcn_readfirstlane(tid_c >> 6);   // (uniform for the compiler too: the item's correspondences come through scalar loads)
                for (long long it = (long long) wg * (MB / 64) + wave; it < items; it += (long long) n_wg * (MB / 64))
                    count_item((int) (it % hb_n), (int) (it / hb_n), a.Ts, a.list, nh, in_PP, in_pstats, a.c, a.counts, mt, cch, tid_c & 63);
            }
        }
        work_end(2);
        if (!rs_grid_barrier(S, n_wg)) return;
        stamp(2);
        work_begin();
        int tid_m = tid;
        asm volatile("" : "+v"(tid_m));
        rs_cand_body(tid_m, wg, n_wg, S, a.counts, a.list, a.list2, a.hpos);
        work_end(3);
        if (!rs_grid_barrier(S, n_wg)) return;
        stamp(3);
        work_begin();
        asm volatile("" : "+v"(tid_m));
        metric_body(tid_m, wg, n_wg, a.Ts, a.list2, 0, in_P0, in_P1, in_sstar, a.c, a.metric_id, a.score_id, a.metric, a.ninl, nullptr, nullptr, a.scratch, a.maskT, a.hpos,
                    a.mask_cap, nullptr, &S->n_cand, &S->n_ok);
        work_end(4);
        if (!rs_grid_barrier(S, n_wg)) return;
        stamp(4);
        if (wg == 0) {
            asm volatile("" : "+v"(tid_m));
            rs_replay_body(tid_m, S, a.list, a.list2, a.metric, a.ninl, a.counts, a.posmap, a.Ts, a.st);
            __syncthreads();
            rs_begin_body(S, a.st, 0);   // the next round's range (or `stop`)
        }
    }
}
// closest-plane / combination metrics inside the device-driven schedule (round 5): run_batch's plumbing kernels with their extents read from the
// RState (fixed grids striding over them)
__global__ void rs_plane_counts_kernel(const RState* __restrict__ S, const int* __restrict__ cnt, int2* __restrict__ counts) {
    const int n = S->n_ok;
    for (int h = blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) counts[h].x = cnt[h];
}
__global__ void rs_plane_pick_kernel(const RState* __restrict__ S, const int* __restrict__ hpos, const int* __restrict__ cnt, const float* __restrict__ cp,
                                     float* __restrict__ metric, int* __restrict__ ninl) {
    const int n = S->n_cand;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) { metric[j] = cp[hpos[j]]; ninl[j] = cnt[hpos[j]]; }
}
__global__ void rs_plane_mul_kernel(const RState* __restrict__ S, float* __restrict__ metric, const float* __restrict__ cp) {
    const int n = S->n_cand;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) metric[j] = metric[j] * cp[j];   // metric_cs * metric_cp (src/metric.cpp:248)
}
// the evaluation record of a single transform (metric, inliers, rmse as metric_kernel left them) into the RState
__global__ void rs_store_eval_kernel(const float* __restrict__ ev /* metric, n_inl bits, rmse */, float* __restrict__ dst3) {
    if (threadIdx.x < 3) dst3[threadIdx.x] = ev[threadIdx.x];
}
__global__ void rs_guess_kernel(RState* __restrict__ S, const float* __restrict__ ev) { S->final_metric = ev[0]; }

// include/utils.h:34-43 calculateCombinationOrMax<int>
int comb_or_max(int n, int k) {
    double result = 1.0;
    for (int i = 0; i < k; ++i) { result *= n - i; result /= i + 1; }
    int mx = INT_MAX;
    return result > mx ? mx : (int) result;
}

// src/metric.cpp:116-122 given the support count
int est_from_support(int count, int c, float confidence, int nr_samples) {
    float frac = (float) count / (float) c;
    frac /= 4.f;
    if (frac <= 0.0 || std::log(1.0 - std::pow(frac, nr_samples)) >= 0.0) return INT_MAX;
    double iterations = std::log(1.0 - confidence) / std::log(1.0 - std::pow(frac, nr_samples));
    return static_cast<int>(std::min((double) INT_MAX, iterations));
}

struct Packed { float4* P0; float4* P1; float* sstar; const CPair* PP; const unsigned* pstats; };

__global__ void corr_range_kernel(const lgr_corr* __restrict__ corr, int c, int ns, int nt, int* __restrict__ bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool b = false;
    if (i < c) {
        lgr_corr cr = corr[i];
        b = (unsigned) cr.index_query >= (unsigned) ns || (unsigned) cr.index_match >= (unsigned) nt;
    }
    if (__any(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}
}  // namespace

int lgr_check_corr(lgr_ctx* ctx, const lgr_corr* d_corr, int c, int ns, int nt) {
    if (ctx->corr_trusted || c <= 0) return LGR_OK;
    int* d_bad;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_STATS, 64, &d_bad));
    LGR_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    corr_range_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(d_corr, c, ns, nt, d_bad);
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) return lgr_fail(ctx, LGR_ERR_INVALID_ARG, "a correspondence index is outside its cloud (index_query in [0, ns), index_match in [0, nt))", __FILE__, __LINE__);
    return LGR_OK;
}

namespace {
int pack(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c, Packed* out) {
    LGR_TRY(lgr_check_corr(ctx, d_corr, c, ns, nt));
    const unsigned* bbk;
    LGR_TRY(lgr_bbox_launch(ctx, d_src, ns, &bbk));   // UniformityMetricEstimator::setSourceCloud (src/metric.cpp:167-170); stays on the device
    float4* P;
    const int cpad = (c + 63) & ~63;
    const size_t n4 = (size_t) c * 2 + (size_t) (c + 3) / 4 + 4;                     // P0, P1, sstar
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_PACK, n4 + (size_t) cpad * 2 + 8, &P));           // + PP (32 bytes per correspondence) + pstats
    out->P0 = P; out->P1 = P + c; out->sstar = (float*) (P + 2 * (size_t) c);
    float* PP = (float*) (P + n4);
    unsigned* pstats = (unsigned*) (P + n4 + (size_t) cpad * 2);
    out->PP = (const CPair*) PP; out->pstats = pstats;
    LGR_HIP(ctx, hipMemsetAsync(pstats, 0, 16, ctx->stream));
    if (c > 0)
        pack_kernel<<<std::min(cdiv(cpad, 256), 512), 256, 0, ctx->stream>>>(d_src, d_tgt, d_corr, c, bbk, out->P0, out->P1, out->sstar, PP, pstats, cpad);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

// refit over the inliers flagged in d_mask (NULL: all pairs): compaction in correspondence order, then refit_kernel
int refit_launch(lgr_ctx* ctx, const Packed& pk, int c, const uint8_t* d_mask, float* d_Tout) {
    if (!d_mask || c == 0) {
        refit_kernel<<<1, 256, 0, ctx->stream>>>(pk.P0, pk.P1, nullptr, c, d_Tout);
        LGR_HIP(ctx, hipGetLastError());
        return LGR_OK;
    }
    int* flags;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_HIST, (size_t) 2 * c + 16 + 8 * ((size_t) c + 4), &flags));
    int* pos = flags + c;
    float4* Q0 = (float4*) (flags + 2 * (size_t) c + 16 - ((2 * (size_t) c) & 3));
    Q0 = (float4*) (((uintptr_t) (flags + 2 * (size_t) c) + 15) & ~(uintptr_t) 15);
    float4* Q1 = Q0 + c;
    mask_flags_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(d_mask, c, flags);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, flags, pos, 0, (size_t) c, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, flags, pos, 0, (size_t) c, rocprim::plus<int>(), ctx->stream));
    compact_pairs_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(pk.P0, pk.P1, flags, pos, c, Q0, Q1);
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, pos + (c - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, flags + (c - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int n = h[0] + h[1];
    refit_kernel<<<1, 256, 0, ctx->stream>>>(Q0, Q1, nullptr, n, d_Tout);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

size_t metric_smem() { return (size_t) (30000 + 64) * 4; }

int metric_launch(lgr_ctx* ctx, const float* Ts, const int* list2, int nh2, const Packed& pk, int c, int metric_id, int score_id,
                  float* metric_out, int* ninl_out, float* rmse_out, uint8_t* mask,
                  const unsigned* maskT = nullptr, const int* hpos = nullptr, int mask_nh = 0) {
    // per device, so not cached in a process-wide flag (a process may hold contexts on several GPUs); the call is a host-side table update
    LGR_HIP(ctx, hipFuncSetAttribute((const void*) metric_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) metric_smem()));
    bool need_list = metric_id != LGR_METRIC_UNIFORMITY || rmse_out;
    // hypotheses are processed in waves of at most `wave` workgroups so that the ordered inlier lists stay bounded
    int wave = need_list ? std::max(1, std::min(nh2, 512)) : std::max(1, nh2);   // without lists: all candidates in one launch
    float2* scratch = nullptr;
    if (need_list) LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_LIST, (size_t) wave * std::max(c, 1), &scratch));
    for (int h0 = 0; h0 < nh2; h0 += wave) {
        int nh = std::min(wave, nh2 - h0);
        metric_kernel<<<nh, MB, metric_smem(), ctx->stream>>>(Ts, list2 ? list2 + h0 : nullptr, nh, pk.P0, pk.P1, pk.sstar, c, metric_id,
                                                              score_id, metric_out + h0, ninl_out + h0,
                                                              rmse_out ? rmse_out + h0 : nullptr, mask, scratch, maskT, hpos ? hpos + h0 : nullptr, mask_nh);
    }
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

struct EvalOut { int n_inl; float rmse; float metric; };
// single transform (device pointer d_T to 16 floats): mask + stats
int evaluate_one(lgr_ctx* ctx, const float* d_T, const Packed& pk, int c, int metric_id, int score_id, uint8_t* d_mask, EvalOut* out,
                 bool want_rmse = true /* false: no ordered inlier list when the metric itself does not need one (rmse = 0) */) {
    float* res;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, 64, &res));
    float* d_metric = res + 32; int* d_ninl = (int*) (res + 33); float* d_rmse = res + 34;
    if (!want_rmse) LGR_HIP(ctx, hipMemsetAsync(d_rmse, 0, 4, ctx->stream));
    if (metric_id == LGR_METRIC_UNIFORMITY && !want_rmse && c > 0) {
        // the counting half on the whole device, the entropy half in one workgroup
        int* ghist;
        LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_GHIST, (size_t) 30000 + 64, &ghist));
        LGR_HIP(ctx, hipMemsetAsync(ghist, 0, (30000 + 1) * 4, ctx->stream));
        inlier_hist_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(d_T, pk.P0, pk.P1, pk.sstar, c, d_mask, ghist);
        LGR_HIP(ctx, hipFuncSetAttribute((const void*) metric_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) metric_smem()));
        metric_kernel<<<1, MB, metric_smem(), ctx->stream>>>(d_T, nullptr, 1, pk.P0, pk.P1, pk.sstar, c, metric_id, score_id, d_metric, d_ninl, nullptr, nullptr,
                                                             nullptr, nullptr, nullptr, 0, ghist);
        LGR_HIP(ctx, hipGetLastError());
    } else {
        LGR_TRY(metric_launch(ctx, d_T, nullptr, 1, pk, c, metric_id, score_id, d_metric, d_ninl, want_rmse ? d_rmse : nullptr, d_mask));
    }
    float* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, d_metric, 12, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->metric = h[0]; memcpy(&out->n_inl, &h[1], 4); out->rmse = h[2];
    return LGR_OK;
}

}  // namespace

// single transform under a plane metric: counter = 0xFFFFFFFE / 0xFFFFFFFF for the two evaluations of the final block.
// closest_plane: inliers / rmse / metric from the plane test, pairs (sorted by source index) returned for the refit.
int evaluate_one_plane(lgr_ctx* ctx, const float* d_T, const Packed& pk, int c, int metric_id, int score_id, uint8_t* d_mask,
                       const lgr_plane_dev& plane, unsigned counter, EvalOut* out, std::vector<int2>* pairs) {
    int* pl;
    LGR_TRY(lgr_ws_t(ctx, WS_PLANE_OUT, (size_t) 8 + 2 * (size_t) std::max(plane.n_sp, 1) + 8, &pl));
    int* d_cnt = pl; float* d_cp = (float*) (pl + 1); float* d_rm = (float*) (pl + 2); int* d_np = pl + 3;
    int2* d_pairs = (int2*) (pl + 8);
    const bool want_pairs = pairs != nullptr;
    LGR_TRY(lgr_plane_eval(ctx, plane, d_T, nullptr, 1, counter, score_id, d_cnt, d_cp, d_rm, want_pairs ? d_pairs : nullptr, d_np));
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64 + (size_t) 8 * std::max(plane.n_sp, 1), (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, pl, 16, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int cnt = h[0];
    float cp, rm;
    memcpy(&cp, &h[1], 4); memcpy(&rm, &h[2], 4);
    if (want_pairs) {
        pairs->resize(cnt);
        if (cnt) {
            LGR_HIP(ctx, hipMemcpyAsync(h + 16, d_pairs, (size_t) cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
            LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            memcpy(pairs->data(), h + 16, (size_t) cnt * 8);
            std::sort(pairs->begin(), pairs->end(), [](const int2& a, const int2& b) { return a.x < b.x; });   // source indices are distinct
        }
    }
    if (metric_id == LGR_METRIC_CLOSEST_PLANE) {
        if (d_mask) LGR_HIP(ctx, hipMemsetAsync(d_mask, 0, (size_t) c, ctx->stream));
        out->n_inl = cnt; out->rmse = rm; out->metric = cp;
        return LGR_OK;
    }
    EvalOut e;
    LGR_TRY(evaluate_one(ctx, d_T, pk, c, LGR_METRIC_CORRESPONDENCES, LGR_SCORE_CONSTANT, d_mask, &e));
    out->n_inl = e.n_inl; out->rmse = e.rmse; out->metric = e.metric * cp;
    return LGR_OK;
}

extern "C" int lgr_evaluate_plane_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const float T16[16], int score_id,
                                      uint64_t seed, uint32_t counter, int* n_inliers, float* rmse, float* metric, float* threshold,
                                      int32_t* pairs, int* n_pairs) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && T16 && n_inliers && rmse && metric && ns > 0 && nt > 1 && score_id >= 0 && score_id <= 3, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    lgr_plane_dev plane;
    LGR_TRY(lgr_plane_setup(ctx, d_src, ns, d_tgt, nt, seed, &plane));
    float* dT;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, 64, &dT));
    LGR_HIP(ctx, hipMemcpyAsync(dT, T16, 64, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    Packed none{nullptr, nullptr, nullptr};
    EvalOut e;
    std::vector<int2> pr;
    LGR_TRY(evaluate_one_plane(ctx, dT, none, 0, LGR_METRIC_CLOSEST_PLANE, score_id, nullptr, plane, counter, &e, pairs ? &pr : nullptr));
    *n_inliers = e.n_inl; *rmse = e.rmse; *metric = e.metric;
    if (threshold) *threshold = plane.thr;
    if (pairs) {
        for (size_t i = 0; i < pr.size(); ++i) { pairs[2 * i] = pr[i].x; pairs[2 * i + 1] = pr[i].y; }
        if (n_pairs) *n_pairs = (int) pr.size();
    }
    return LGR_OK;
}

extern "C" int lgr_ransac_samples_n_dev(lgr_ctx* ctx, uint64_t seed, int first, int n, int n_corr, int n_samples, int32_t* d_tuples) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, n_samples >= LGR_MIN_SAMPLES && n_samples <= LGR_MAX_SAMPLES, LGR_ERR_UNSUPPORTED);
    LGR_CHECK(ctx, n >= 0 && n_corr >= n_samples && (d_tuples || n == 0) && first >= 0, LGR_ERR_INVALID_ARG);
    if (n == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    LGR_NS_DISPATCH(n_samples, (samples_kernel<NS><<<cdiv(n, 256), 256, 0, ctx->stream>>>(seed, first, n, n_corr, d_tuples)));
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}
extern "C" int lgr_ransac_samples_dev(lgr_ctx* ctx, uint64_t seed, int first, int n, int n_corr, int32_t* d_triples) {
    return lgr_ransac_samples_n_dev(ctx, seed, first, n, n_corr, 3, d_triples);
}

// lgr.h: one Philox4x32-10 block through the device's generator (known-answer tests)
__global__ void philox_kernel(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned* __restrict__ out) {
    unsigned w[4];
    lgr_philox4(seed, c0, c1, c2, c3, w);
    out[0] = w[0]; out[1] = w[1]; out[2] = w[2]; out[3] = w[3];
}
extern "C" int lgr_selfcheck_philox(lgr_ctx* ctx, uint64_t key, const uint32_t counter4[4], uint32_t out4[4]) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, counter4 && out4, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    unsigned* d;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, 64, &d));
    philox_kernel<<<1, 1, 0, ctx->stream>>>(key, counter4[0], counter4[1], counter4[2], counter4[3], d);
    LGR_HIP(ctx, hipGetLastError());
    LGR_HIP(ctx, hipMemcpyAsync(out4, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}

extern "C" int lgr_evaluate_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                                const float T16[16], int metric_id, int score_id,
                                uint8_t* d_mask, int* n_inliers, float* rmse, float* metric) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && (d_corr || c == 0) && T16 && c >= 0 && ns > 0 && nt > 0, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, metric_id == LGR_METRIC_UNIFORMITY || metric_id == LGR_METRIC_CORRESPONDENCES, LGR_ERR_UNSUPPORTED);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    Packed pk;
    LGR_TRY(pack(ctx, d_src, ns, d_tgt, nt, d_corr, c, &pk));
    float* dT;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, 64, &dT));
    LGR_HIP(ctx, hipMemcpyAsync(dT, T16, 64, hipMemcpyHostToDevice, ctx->stream));
    EvalOut e;
    LGR_TRY(evaluate_one(ctx, dT, pk, c, metric_id, score_id, d_mask, &e));
    if (n_inliers) *n_inliers = e.n_inl;
    if (rmse) *rmse = e.rmse;
    if (metric) *metric = e.metric;
    return LGR_OK;
}

// one batch: hypotheses -> ok-list -> counts -> candidate list -> metrics.  Returns device arrays + host counts.
struct BatchBuffers {
    float* Ts; int* ok; int* pos; int* list; int2* counts; int* flags2; int* pos2; int* list2; float* metric; int* ninl; int* hpos; BatchStats* st;
};
static int batch_buffers(lgr_ctx* ctx, int nb, BatchBuffers* b) {
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_T, (size_t) nb * 16, &b->Ts));
    int* s;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_STATS, (size_t) nb * 13 + 64 + MAX_ROUND_BATCHES * (sizeof(BatchStats) / 4), &s));
    b->ok = s; b->pos = s + nb; b->list = s + 2 * (size_t) nb; b->counts = (int2*) (s + 3 * (size_t) nb);
    b->flags2 = s + 5 * (size_t) nb; b->pos2 = s + 6 * (size_t) nb; b->list2 = s + 7 * (size_t) nb;
    b->metric = (float*) (s + 8 * (size_t) nb); b->ninl = s + 9 * (size_t) nb; b->hpos = s + 10 * (size_t) nb;
    b->st = (BatchStats*) (s + 11 * (size_t) nb + ((11 * (size_t) nb) & 1));
    return LGR_OK;
}

// runs one batch.  h_counts: [0] n_ok, [1] n_cand (hypotheses with >= MIN_NR_INLIERS inliers)
static int run_batch(lgr_ctx* ctx, const float* d_src, const float* d_tgt, const lgr_corr* d_corr, int c, const Packed& pk,
                     const lgr_params* p, uint64_t seed, int first, int nb, const int32_t* d_triples, BatchBuffers& b,
                     int* n_ok, int* n_cand, const lgr_plane_dev* plane = nullptr, int min_inliers = MIN_NR_INLIERS,
                     float best_prev = 0.f, int record_prev = 0) {
    LGR_NS_DISPATCH(p->n_samples, (hypotheses_kernel<NS><<<cdiv(nb, 128), 128, 0, ctx->stream>>>(d_src, d_tgt, d_corr, c, seed, first, nb, d_triples,
                                                                                                  p->edge_thr_coef, b.Ts, b.ok)));
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, b.ok, b.pos, 0, (size_t) nb, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, b.ok, b.pos, 0, (size_t) nb, rocprim::plus<int>(), ctx->stream));
    compact_kernel<<<cdiv(nb, 256), 256, 0, ctx->stream>>>(b.ok, b.pos, nb, nullptr, b.list);
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, b.pos + (nb - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, b.ok + (nb - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int nh = h[0] + h[1];
    *n_ok = nh; *n_cand = 0;
    if (nh == 0) return LGR_OK;
    LGR_HIP(ctx, hipMemsetAsync(b.counts, 0, (size_t) nh * 8, ctx->stream));
    // few hypotheses (the first round, the lr filter): shorter correspondence chunks, so that the launch still has a few thousand waves
    const int cch = count_chunk(nh, c);
    dim3 g(cdiv(nh, CB), cdiv(c, cch));
    // inlier bit masks for phase 2 (uniformity / correspondence-count metrics need the inlier set only); skipped when they
    // would not fit 2 GB (then phase 2 tests every correspondence again)
    unsigned* maskT = nullptr;
    const size_t mask_words = mask_pitch(c) * nh;
    if (!plane && p->metric_id == LGR_METRIC_UNIFORMITY && mask_words * 4 <= ((size_t) 2 << 30)) LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MASKT, mask_words, &maskT));
    count_kernel<<<g, CB, 0, ctx->stream>>>(b.Ts, b.list, nh, pk.PP, pk.pstats, c, b.counts, maskT, cch);
    int* pl_cnt = nullptr;
    float* pl_cp = nullptr;
    if (plane) {
        LGR_TRY(lgr_ws_t(ctx, WS_PLANE_OUT, (size_t) 2 * nb + 16, &pl_cnt));
        pl_cp = (float*) (pl_cnt + nb);
    }
    if (plane && p->metric_id == LGR_METRIC_CLOSEST_PLANE) {
        // every hypothesis that passed the prerejection is evaluated on its sparse subset; its plane inliers are "the inliers"
        // (gate: a hypothesis that can reach neither the best metric nor the record inlier count of the earlier batches is abandoned)
        LGR_TRY(lgr_plane_eval(ctx, *plane, b.Ts, b.list, nh, (unsigned) first, p->score_id, pl_cnt, pl_cp, nullptr, nullptr, nullptr,
                               best_prev, record_prev, nullptr));   // (record_prev == 0: no record yet, nothing is abandoned)
        plane_counts_kernel<<<cdiv(nh, 256), 256, 0, ctx->stream>>>(pl_cnt, nh, b.counts);
    }
    flag_ge_kernel<<<cdiv(nh, 256), 256, 0, ctx->stream>>>(b.counts, nh, min_inliers, b.flags2);
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, b.flags2, b.pos2, 0, (size_t) nh, rocprim::plus<int>(), ctx->stream));
    compact_kernel<<<cdiv(nh, 256), 256, 0, ctx->stream>>>(b.flags2, b.pos2, nh, b.list, b.list2);
    if (maskT) compact_kernel<<<cdiv(nh, 256), 256, 0, ctx->stream>>>(b.flags2, b.pos2, nh, nullptr, b.hpos);   // candidate -> ok-list position
    LGR_HIP(ctx, hipMemcpyAsync(h, b.pos2 + (nh - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, b.flags2 + (nh - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int nh2 = h[0] + h[1];
    *n_cand = nh2;
    if (nh2 == 0) return LGR_OK;
    if (plane && p->metric_id == LGR_METRIC_CLOSEST_PLANE) {
        plane_pick_kernel<<<cdiv(nh, 256), 256, 0, ctx->stream>>>(b.flags2, b.pos2, nh, pl_cnt, pl_cp, b.metric, b.ninl);
    } else if (plane) {   // combination: correspondences metric with the constant score (include/metric.h:191-192) x plane metric
        LGR_TRY(metric_launch(ctx, b.Ts, b.list2, nh2, pk, c, LGR_METRIC_CORRESPONDENCES, LGR_SCORE_CONSTANT, b.metric, b.ninl, nullptr, nullptr));
        LGR_TRY(lgr_plane_eval(ctx, *plane, b.Ts, b.list2, nh2, (unsigned) first, p->score_id, pl_cnt, pl_cp, nullptr, nullptr, nullptr,
                               best_prev, 0x7fffffff, b.metric));   // (records are correspondence counts here: only the metric gates)
        plane_mul_kernel<<<cdiv(nh2, 256), 256, 0, ctx->stream>>>(b.metric, pl_cp, nh2);
    } else {
        LGR_TRY(metric_launch(ctx, b.Ts, b.list2, nh2, pk, c, p->metric_id, p->score_id, b.metric, b.ninl, nullptr, nullptr, maskT, b.hpos, nh));
    }
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

// single-transform evaluation without a read-back: (metric, n_inl bits, rmse) -> d_out3 (device)
static int evaluate_one_dev(lgr_ctx* ctx, const float* d_T, const Packed& pk, int c, int metric_id, int score_id, uint8_t* d_mask, float* d_scratch3,
                            float* d_out3) {
    float* d_metric = d_scratch3; int* d_ninl = (int*) (d_scratch3 + 1); float* d_rmse = d_scratch3 + 2;
    LGR_HIP(ctx, hipMemsetAsync(d_rmse, 0, 4, ctx->stream));
    if (metric_id == LGR_METRIC_UNIFORMITY && c > 0) {
        int* ghist;
        LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_GHIST, (size_t) 30000 + 64, &ghist));
        LGR_HIP(ctx, hipMemsetAsync(ghist, 0, (30000 + 1) * 4, ctx->stream));
        inlier_hist_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(d_T, pk.P0, pk.P1, pk.sstar, c, d_mask, ghist);
        metric_kernel<<<1, MB, metric_smem(), ctx->stream>>>(d_T, nullptr, 1, pk.P0, pk.P1, pk.sstar, c, metric_id, score_id, d_metric, d_ninl, nullptr, nullptr,
                                                             nullptr, nullptr, nullptr, 0, ghist);
    } else {
        LGR_TRY(metric_launch(ctx, d_T, nullptr, 1, pk, c, metric_id, score_id, d_metric, d_ninl, nullptr, d_mask));
    }
    rs_store_eval_kernel<<<1, 64, 0, ctx->stream>>>(d_scratch3, d_out3);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}
// refit over the inliers flagged in d_mask, the number of inliers staying on the device
static int refit_launch_dev(lgr_ctx* ctx, const Packed& pk, int c, const uint8_t* d_mask, float* d_Tout) {
    int* flags;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_HIST, (size_t) 2 * c + 16 + 8 * ((size_t) c + 4), &flags));
    int* pos = flags + c;
    float4* Q0 = (float4*) (((uintptr_t) (flags + 2 * (size_t) c) + 15) & ~(uintptr_t) 15);
    float4* Q1 = Q0 + c;
    mask_flags_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(d_mask, c, flags);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, flags, pos, 0, (size_t) c, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, flags, pos, 0, (size_t) c, rocprim::plus<int>(), ctx->stream));
    compact_pairs_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(pk.P0, pk.P1, flags, pos, c, Q0, Q1);
    refit_kernel<<<1, 256, 0, ctx->stream>>>(Q0, Q1, nullptr, 0, d_Tout, pos + (c - 1), flags + (c - 1));
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

// The device-driven schedule (uniformity / correspondences metrics): see RState.  ONE host synchronisation per pair of rounds -- one per
// alignment whenever the loop ends within two rounds, i.e. for every max_iterations up to 17 batches and whenever a record inlier set
// brings the bound below that.
// plane != nullptr (closest_plane / combination, round 5): the same rounds with the plane evaluation in them; the loop only -- the final block
// (plane pairs sorted by source index for the refit) stays with the caller, which finds the loop's state in *state_out and the best transform
// on the device at *d_best_out.  initial_metric: the guess's metric (the caller evaluated it), or 0.
static int ransac_device_schedule(lgr_ctx* ctx, const float* d_src, const float* d_tgt, const lgr_corr* d_corr, int c, const Packed& pk, const lgr_params* p,
                                  uint64_t seed, int max_iterations, int batch, uint8_t* d_mask, lgr_result* res, const lgr_plane_dev* plane = nullptr,
                                  float initial_metric = 0.f, RState* state_out = nullptr, float** d_best_out = nullptr) {
    LGR_HIP(ctx, hipFuncSetAttribute((const void*) metric_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) metric_smem()));
    const int nb_max = (int) std::min<long long>((long long) batch * MAX_ROUND_BATCHES, std::max(max_iterations, 1));
    BatchBuffers b;
    LGR_TRY(batch_buffers(ctx, nb_max, &b));
    int* posmap = b.pos;   // [nb_max]: iteration offset in the round -> position in the survivors' list
    char* misc;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, sizeof(RState) + 256, &misc));
    RState* dS = (RState*) misc;
    float* d_ev = (float*) (misc + sizeof(RState));   // scratch of the single-transform evaluations
    RState* hS;
    LGR_TRY(lgr_pinned(ctx, sizeof(RState), (void**) &hS));
    memset(hS, 0, sizeof(RState));
    hS->bound = max_iterations; hS->max_iterations = max_iterations; hS->batch = batch; hS->round_cap = MAX_ROUND_BATCHES;
    hS->best_iter = -1; hS->metric_id = p->metric_id; hS->c = c; hS->nr_samples = p->n_samples; hS->confidence = p->confidence;
    hS->int_max = INT_MAX; hS->final_metric = initial_metric;
    for (int i = 0; i < 16; ++i) hS->best_T[i] = p->has_guess ? p->guess[i] : ((i % 5 == 0) ? 1.f : 0.f);
    LGR_HIP(ctx, hipMemcpyAsync(dS, hS, sizeof(RState), hipMemcpyHostToDevice, ctx->stream));
    if (p->has_guess && !plane) {
        // src/sac_prerejective_omp.cpp:134-147: the guess is the hypothesis to beat (final_tn / final_metric)
        uint8_t* d_gm;
        LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MASK, (size_t) c + 16, &d_gm));
        LGR_TRY(evaluate_one_dev(ctx, dS->best_T, pk, c, p->metric_id, p->score_id, d_gm, d_ev, d_ev + 4));
        rs_guess_kernel<<<1, 1, 0, ctx->stream>>>(dS, d_ev + 4);
    }
    // inlier bit masks for the uniformity metric: [survivors][mask_pitch(c)], as many rows as 2 GB hold (more survivors: the metric kernel
    // tests every correspondence again)
    unsigned* maskT = nullptr;
    int mask_cap = 0;
    if (p->metric_id == LGR_METRIC_UNIFORMITY) {
        const size_t words = mask_pitch(c);
        mask_cap = (int) std::min<size_t>((size_t) nb_max, ((size_t) 2 << 30) / (words * 4));
        if (mask_cap > 0) LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MASKT, words * (size_t) mask_cap, &maskT));
    }
    // plane metrics: the plane test's inlier count and metric per evaluated hypothesis (closest_plane: every survivor, indexed like the
    // survivors' list; combination: every candidate)
    int* pl_cnt = nullptr;
    float* pl_cp = nullptr;
    if (plane) {
        LGR_TRY(lgr_ws_t(ctx, WS_PLANE_OUT, (size_t) 2 * nb_max + 16, &pl_cnt));
        pl_cp = (float*) (pl_cnt + nb_max);
    }
    const bool closest = plane && p->metric_id == LGR_METRIC_CLOSEST_PLANE;
    const bool need_list = p->metric_id != LGR_METRIC_UNIFORMITY;
    const int g_metric = std::max(1, ctx->n_cu), g_count = 128 * std::max(1, ctx->n_cu);   // one 120 KB workgroup per CU; four times the resident single-wave workgroups (the tail evens out)
    float2* scratch = nullptr;
    if (need_list) LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_LIST, (size_t) g_metric * std::max(c, 1), &scratch));
    const bool ransac_debug = getenv("LGR_RANSAC_DEBUG") != nullptr;
    auto enqueue_round = [&](bool first) -> int {
        const int nb_up = first ? std::min(batch, nb_max) : nb_max;
        rs_begin_kernel<<<1, 64, 0, ctx->stream>>>(dS, b.st, first ? 1 : 0);
        LGR_NS_DISPATCH(p->n_samples, (rs_hyp_kernel<NS><<<cdiv(nb_up, 128), 128, 0, ctx->stream>>>(pk.P0, pk.P1, c, seed, dS, p->edge_thr_coef, b.Ts, b.list,
                                                                                                     posmap, b.counts)));
        count_list_kernel<<<g_count, CB, 0, ctx->stream>>>(b.Ts, b.list, &dS->n_ok, pk.PP, pk.pstats, c, b.counts, maskT, mask_cap);
        if (closest) {
            // every survivor on its sparse subset; its plane inliers are "the inliers" (gate: the loop's best metric and record so far)
            const lgr_plane_dyn dyn{&dS->n_ok, &dS->round_first, &dS->final_metric, &dS->largest};
            LGR_TRY(lgr_plane_eval(ctx, *plane, b.Ts, b.list, nb_up, 0u, p->score_id, pl_cnt, pl_cp, nullptr, nullptr, nullptr, 0.f, 0, nullptr, &dyn));
            rs_plane_counts_kernel<<<64, 256, 0, ctx->stream>>>(dS, pl_cnt, b.counts);
        }
        rs_cand_kernel<<<64, 1024, 0, ctx->stream>>>(dS, b.counts, b.list, b.list2, b.hpos);
        if (closest) {
            rs_plane_pick_kernel<<<64, 256, 0, ctx->stream>>>(dS, b.hpos, pl_cnt, pl_cp, b.metric, b.ninl);
        } else if (plane) {   // combination: correspondences metric with the constant score (include/metric.h:191-192) x plane metric of the candidates
            metric_kernel<<<g_metric, MB, metric_smem(), ctx->stream>>>(b.Ts, b.list2, 0, pk.P0, pk.P1, pk.sstar, c, LGR_METRIC_CORRESPONDENCES, LGR_SCORE_CONSTANT, b.metric, b.ninl,
                                                                        nullptr, nullptr, scratch, nullptr, b.hpos, 0, nullptr, &dS->n_cand, &dS->n_ok);
            const lgr_plane_dyn dyn{&dS->n_cand, &dS->round_first, &dS->final_metric, &dS->int_max};   // (records are correspondence counts here: only the metric gates)
            LGR_TRY(lgr_plane_eval(ctx, *plane, b.Ts, b.list2, nb_up, 0u, p->score_id, pl_cnt, pl_cp, nullptr, nullptr, nullptr, 0.f, 0x7fffffff, b.metric, &dyn));
            rs_plane_mul_kernel<<<64, 256, 0, ctx->stream>>>(dS, b.metric, pl_cp);
        } else
        metric_kernel<<<g_metric, MB, metric_smem(), ctx->stream>>>(b.Ts, b.list2, 0, pk.P0, pk.P1, pk.sstar, c, p->metric_id, p->score_id, b.metric, b.ninl,
                                                                    nullptr, nullptr, scratch, maskT, b.hpos, mask_cap, nullptr, &dS->n_cand, &dS->n_ok);
        rs_replay_kernel<<<1, 1024, 0, ctx->stream>>>(dS, b.list, b.list2, b.metric, b.ninl, b.counts, posmap, b.Ts, b.st);
        LGR_HIP(ctx, hipGetLastError());
        return LGR_OK;
    };
    // 0 (default) and 1: the launch chain; 2: the resident kernel (one launch for the whole loop; not for the plane metrics)
    const bool resident = ctx->opt.ransac_schedule == 2 && !plane;
    if (resident) {
        LGR_HIP(ctx, hipFuncSetAttribute((const void*) rs_resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) metric_smem()));
        int per_cu = 0;
        LGR_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*) rs_resident_kernel, MB, metric_smem()));
        LGR_CHECK(ctx, per_cu >= 1 && g_metric >= 1, LGR_ERR_HIP);   // (the grid must fit the device at once: one workgroup per CU)
        ResidentArgs ra{};
        ra.c = c; ra.n_samples = p->n_samples; ra.seed = seed; ra.edge_thr = p->edge_thr_coef;
        ra.S = dS; ra.st = b.st; ra.Ts = b.Ts; ra.list = b.list; ra.posmap = posmap; ra.counts = b.counts; ra.list2 = b.list2; ra.hpos = b.hpos;
        ra.metric = b.metric; ra.ninl = b.ninl;
        ra.maskT = maskT; ra.mask_cap = mask_cap; ra.scratch = scratch; ra.metric_id = p->metric_id; ra.score_id = p->score_id;
        ra.max_rounds = (int) std::min<long long>(((long long) max_iterations + batch - 1) / batch + 2, INT_MAX);
        rs_resident_kernel<<<g_metric, MB, metric_smem(), ctx->stream>>>(ra, d_src, d_tgt, d_corr, pk.PP, pk.pstats, pk.P0, pk.P1, pk.sstar);
        LGR_HIP(ctx, hipGetLastError());
        LGR_TRY(evaluate_one_dev(ctx, dS->best_T, pk, c, p->metric_id, p->score_id, d_mask, d_ev, &dS->e_metric));
        LGR_TRY(refit_launch_dev(ctx, pk, c, d_mask, dS->Tn));
        LGR_TRY(evaluate_one_dev(ctx, dS->Tn, pk, c, p->metric_id, p->score_id, d_mask, d_ev, &dS->e2_metric));
        LGR_HIP(ctx, hipMemcpyAsync(hS, dS, sizeof(RState), hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ransac_debug) fprintf(stderr, "[lgr] ransac (resident kernel, %d workgroups) after %d rounds: done %d bound %d largest %d best metric %.4f stop %d abort %d\n", g_metric,
                                  hS->rounds, hS->done, hS->bound, hS->largest, hS->final_metric, hS->stop, hS->abort);
        if (ransac_debug) fprintf(stderr, "[lgr]   %d survivors, %d candidates in all\n", hS->tot_ok, hS->tot_cand);
        if (ransac_debug) fprintf(stderr, "[lgr]   workgroup 0, us per phase incl. barrier: begin/replay %.1f hyp %.1f count %.1f cand %.1f metric %.1f\n", hS->phase_ticks[0] * 0.01,
                                  hS->phase_ticks[1] * 0.01, hS->phase_ticks[2] * 0.01, hS->phase_ticks[3] * 0.01, hS->phase_ticks[4] * 0.01);
        if (ransac_debug) fprintf(stderr, "[lgr]   own work per workgroup, us, max / mean: hyp %.1f / %.1f count %.1f / %.1f cand %.1f / %.1f metric %.1f / %.1f\n",
                                  hS->busy_max[1] * 0.01, hS->busy_sum[1] * 0.01 / g_metric, hS->busy_max[2] * 0.01, hS->busy_sum[2] * 0.01 / g_metric,
                                  hS->busy_max[3] * 0.01, hS->busy_sum[3] * 0.01 / g_metric, hS->busy_max[4] * 0.01, hS->busy_sum[4] * 0.01 / g_metric);
        if (hS->abort || !hS->stop) {
            ctx->err = "resident RANSAC kernel: a grid barrier timed out (a workgroup of the grid did not become resident) or the loop did not end";
            return LGR_ERR_HIP;
        }
    }
    bool first = true;
    while (!resident) {
        LGR_TRY(enqueue_round(first));
        first = false;
        LGR_TRY(enqueue_round(false));
        if (!plane) {
            // :265-296 final re-estimation (enqueued blind: redone when the loop turns out not to have ended)
            LGR_TRY(evaluate_one_dev(ctx, dS->best_T, pk, c, p->metric_id, p->score_id, d_mask, d_ev, &dS->e_metric));
            LGR_TRY(refit_launch_dev(ctx, pk, c, d_mask, dS->Tn));
            LGR_TRY(evaluate_one_dev(ctx, dS->Tn, pk, c, p->metric_id, p->score_id, d_mask, d_ev, &dS->e2_metric));
        }
        LGR_HIP(ctx, hipMemcpyAsync(hS, dS, sizeof(RState), hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ransac_debug) fprintf(stderr, "[lgr] ransac (device schedule) after %d rounds: done %d bound %d largest %d best metric %.4f stop %d (%d survivors, %d candidates in all)\n",
                                  hS->rounds, hS->done, hS->bound, hS->largest, hS->final_metric, hS->stop, hS->tot_ok, hS->tot_cand);
        if (hS->stop) break;
    }
    if (plane) {   // the caller's final block takes over
        *state_out = *hS;
        *d_best_out = dS->best_T;
        return LGR_OK;
    }
    int e_ninl, e2_ninl;
    memcpy(&e_ninl, &hS->e_ninl, 4); memcpy(&e2_ninl, &hS->e2_ninl, 4);
    const bool enough = e_ninl > MIN_NR_FINAL_INLIERS || (float) e_ninl > MIN_INLIER_RATE * (float) c;
    const float min_tol = p->metric_id == LGR_METRIC_UNIFORMITY ? 0.3f : 0.0f;   // include/metric.h:97-99 / 73-75
    memcpy(res->transformation, hS->Tn, 64);
    res->iterations = hS->done;
    res->converged = (enough && hS->e_metric > min_tol) ? 1 : 0;
    res->n_inliers = e2_ninl;
    res->metric = hS->e2_metric;
    res->best_metric_before_refit = hS->final_metric;
    res->best_iteration = hS->best_iter;
    res->num_rejections = hS->num_rejections;
    res->estimated_iters = hS->bound;
    return LGR_OK;
}

static int check_params(lgr_ctx* ctx, const lgr_params* p) {
    LGR_CHECK(ctx, p != nullptr, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, p->n_samples >= LGR_MIN_SAMPLES && p->n_samples <= LGR_MAX_SAMPLES, LGR_ERR_UNSUPPORTED);   // (fewer than 3 pairs leave Umeyama's rotation open)
    LGR_CHECK(ctx, p->metric_id == LGR_METRIC_UNIFORMITY || p->metric_id == LGR_METRIC_CORRESPONDENCES ||
                       p->metric_id == LGR_METRIC_CLOSEST_PLANE || p->metric_id == LGR_METRIC_COMBINATION,
              LGR_ERR_UNSUPPORTED);   // weighted_closest_plane (src/weights.cpp) is not built
    LGR_CHECK(ctx, p->score_id >= 0 && p->score_id <= 3, LGR_ERR_INVALID_ARG);
    return LGR_OK;
}

extern "C" int lgr_ransac_replay_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                                     const lgr_params* p, const int32_t* d_triples, int n,
                                     uint8_t* d_ok, float* d_T16, int32_t* d_n_inliers, float* d_metric) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_TRY(check_params(ctx, p));
    LGR_CHECK(ctx, p->metric_id == LGR_METRIC_UNIFORMITY || p->metric_id == LGR_METRIC_CORRESPONDENCES, LGR_ERR_UNSUPPORTED);   // plane metrics: lgr_evaluate_plane_dev
    LGR_CHECK(ctx, d_src && d_tgt && d_corr && d_triples && d_ok && d_T16 && d_n_inliers && d_metric && c >= p->n_samples && n >= 0 && ns > 0 && nt > 0, LGR_ERR_INVALID_ARG);
    if (n == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    Packed pk;
    LGR_TRY(pack(ctx, d_src, ns, d_tgt, nt, d_corr, c, &pk));
    BatchBuffers b;
    LGR_TRY(batch_buffers(ctx, n, &b));
    int n_ok = 0;
    // every prerejection survivor gets its metric here (no MIN_NR_INLIERS gate): replay reports per hypothesis
    LGR_NS_DISPATCH(p->n_samples, (hypotheses_kernel<NS><<<cdiv(n, 128), 128, 0, ctx->stream>>>(d_src, d_tgt, d_corr, c, 0, 0, n, d_triples, p->edge_thr_coef, b.Ts, b.ok)));
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, b.ok, b.pos, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, b.ok, b.pos, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    compact_kernel<<<cdiv(n, 256), 256, 0, ctx->stream>>>(b.ok, b.pos, n, nullptr, b.list);
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, b.pos + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, b.ok + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    n_ok = h[0] + h[1];
    LGR_HIP(ctx, hipMemsetAsync(d_n_inliers, 0, (size_t) n * 4, ctx->stream));
    LGR_HIP(ctx, hipMemsetAsync(d_metric, 0, (size_t) n * 4, ctx->stream));
    if (n_ok > 0) LGR_TRY(metric_launch(ctx, b.Ts, b.list, n_ok, pk, c, p->metric_id, p->score_id, b.metric, b.ninl, nullptr, nullptr));
    // scatter back to iteration order
    std::vector<int> hl(n_ok);
    std::vector<float> hm(n_ok);
    std::vector<int> hn(n_ok), hok(n);
    if (n_ok) {
        LGR_HIP(ctx, hipMemcpyAsync(hl.data(), b.list, (size_t) n_ok * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(hm.data(), b.metric, (size_t) n_ok * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(hn.data(), b.ninl, (size_t) n_ok * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    LGR_HIP(ctx, hipMemcpyAsync(hok.data(), b.ok, (size_t) n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<float> fm(n, 0.f);
    std::vector<int> fn(n, 0);
    std::vector<uint8_t> fo(n);
    for (int i = 0; i < n; ++i) fo[i] = (uint8_t) hok[i];
    for (int j = 0; j < n_ok; ++j) { fm[hl[j]] = hm[j]; fn[hl[j]] = hn[j]; }
    LGR_HIP(ctx, hipMemcpyAsync(d_metric, fm.data(), (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(d_n_inliers, fn.data(), (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(d_ok, fo.data(), (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(d_T16, b.Ts, (size_t) n * 64, hipMemcpyDeviceToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}

extern "C" int lgr_ransac_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                              const lgr_params* p, lgr_result* res, uint8_t* d_final_mask) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_TRY(check_params(ctx, p));
    LGR_CHECK(ctx, d_src && d_tgt && (d_corr || c == 0) && res && c >= 0 && ns > 0 && nt > 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    auto t_start = std::chrono::steady_clock::now();
    memset(res, 0, sizeof(*res));
    for (int i = 0; i < 16; ++i) res->transformation[i] = (i % 5 == 0) ? 1.f : 0.f;
    res->n_correspondences = c;
    if (c < p->n_samples) return LGR_OK;   // selectCorrespondences refuses (src/sac_prerejective_omp.cpp:36-42); identity, not converged
    uint64_t seed = p->fix_seed ? 566ull : p->seed;
    Packed pk;
    LGR_TRY(pack(ctx, d_src, ns, d_tgt, nt, d_corr, c, &pk));
    int max_iterations = std::min(comb_or_max(c, p->n_samples), p->max_iterations);
    int batch = std::max(1, p->ransac_batch);
    int bound = max_iterations, done = 0, largest = 0, num_rejections = 0, best_iter = -1;
    (void) largest;
    float final_metric = 0.f;
    float* d_best;   // device copy of the best transform so far
    static_assert(sizeof(RState) + 256 <= 256 * sizeof(float), "the device schedule's state fits the slot as requested here (no re-allocation under d_best)");
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, 256, &d_best));
    {
        float I[16];
        for (int i = 0; i < 16; ++i) I[i] = (i % 5 == 0) ? 1.f : 0.f;
        LGR_HIP(ctx, hipMemcpyAsync(d_best, I, 64, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const bool plane_metric = p->metric_id == LGR_METRIC_CLOSEST_PLANE || p->metric_id == LGR_METRIC_COMBINATION;
    if (!plane_metric) {
        // the loop, the final evaluation and the refit driven from the device: one host synchronisation (ransac_device_schedule)
        uint8_t* d_mask = d_final_mask;
        if (!d_mask) LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MASK, (size_t) c + 16, &d_mask));
        LGR_TRY(ransac_device_schedule(ctx, d_src, d_tgt, d_corr, c, pk, p, seed, max_iterations, batch, d_mask, res));
        res->time_te = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        return LGR_OK;
    }
    lgr_plane_dev plane;
    LGR_TRY(lgr_plane_setup(ctx, d_src, ns, d_tgt, nt, seed, &plane));
    if (p->has_guess) {
        // src/sac_prerejective_omp.cpp:134-147: the guess is the hypothesis to beat (final_tn / final_metric).  Its inliers only seed
        // the global largest_inlier_set, which the loop never reads (thread-local sets start empty, :177): the bound is unaffected.
        LGR_HIP(ctx, hipMemcpyAsync(d_best, p->guess, 64, hipMemcpyHostToDevice, ctx->stream));
        uint8_t* d_gm;
        LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MASK, (size_t) c + 16, &d_gm));
        EvalOut eg;
        if (plane_metric) LGR_TRY(evaluate_one_plane(ctx, d_best, pk, c, p->metric_id, p->score_id, d_gm, plane, 0xFFFFFFFDu, &eg, nullptr));
        else LGR_TRY(evaluate_one(ctx, d_best, pk, c, p->metric_id, p->score_id, d_gm, &eg, false));
        final_metric = eg.metric;
    }
    // The loop itself runs on the device-driven schedule (round 5; rounds 3-4 drove the plane metrics' batches from the host, one batch and three
    // synchronisations at a time): a round evaluates up to MAX_ROUND_BATCHES batches -- the sparse subset of a hypothesis is keyed by its ITERATION
    // (Philox counter = round_first + offset), so it does not matter which round or batch evaluates it; the gate uses the loop's state at the
    // start of the round (a looser gate than batch by batch: it still only abandons what can be neither the best nor a record).
    {
        RState hs_end;
        float* d_best_loop = nullptr;
        LGR_TRY(ransac_device_schedule(ctx, d_src, d_tgt, d_corr, c, pk, p, seed, max_iterations, batch, nullptr, res, &plane, final_metric, &hs_end, &d_best_loop));
        done = hs_end.done; bound = hs_end.bound; largest = hs_end.largest; num_rejections = hs_end.num_rejections; best_iter = hs_end.best_iter;
        final_metric = hs_end.final_metric;
        // (the schedule's RState shares WS_RANSAC_MISC with d_best: the best transform moves to the front, where the final block expects it)
        LGR_HIP(ctx, hipMemcpyAsync(d_best, d_best_loop, 64, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // :265-296 final re-estimation
    uint8_t* d_mask = d_final_mask;
    if (!d_mask) LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MASK, (size_t) c + 16, &d_mask));
    EvalOut e;
    std::vector<int2> plane_pairs;
    if (plane_metric)
        LGR_TRY(evaluate_one_plane(ctx, d_best, pk, c, p->metric_id, p->score_id, d_mask, plane, 0xFFFFFFFEu, &e,
                                   p->metric_id == LGR_METRIC_CLOSEST_PLANE ? &plane_pairs : nullptr));
    else LGR_TRY(evaluate_one(ctx, d_best, pk, c, p->metric_id, p->score_id, d_mask, &e, false));   // the final block uses inliers and metric only
    bool enough = e.n_inl > MIN_NR_FINAL_INLIERS || (float) e.n_inl > MIN_INLIER_RATE * (float) c;
    float min_tol = p->metric_id == LGR_METRIC_UNIFORMITY ? 0.3f : 0.0f;   // include/metric.h:97-99 / 73-75 / 124-126 / 198-200
    bool converged = enough && e.metric > min_tol;
    float* d_Tn = d_best + 16;
    if (p->metric_id == LGR_METRIC_CLOSEST_PLANE) {
        // estimateOptimalRigidTransformation over the plane pairs (source point, nearest target point), ascending source index
        const int np = (int) plane_pairs.size();
        Packed pp;
        float4* P;
        LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_LIST, (size_t) 3 * std::max(np, 1) + 4, &P));
        pp.P0 = P; pp.P1 = P + std::max(np, 1); pp.sstar = nullptr; pp.PP = nullptr; pp.pstats = nullptr;
        int2* d_pairs = (int2*) (P + 2 * (size_t) std::max(np, 1));
        if (np) {
            LGR_HIP(ctx, hipMemcpyAsync(d_pairs, plane_pairs.data(), (size_t) np * 8, hipMemcpyHostToDevice, ctx->stream));
            plane_pack_kernel<<<cdiv(np, 256), 256, 0, ctx->stream>>>(d_src, d_tgt, d_pairs, np, pp.P0, pp.P1);
            LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        LGR_TRY(refit_launch(ctx, pp, np, nullptr, d_Tn));
    } else {
        LGR_TRY(refit_launch(ctx, pk, c, d_mask, d_Tn));
    }
    EvalOut e2;
    if (plane_metric) LGR_TRY(evaluate_one_plane(ctx, d_Tn, pk, c, p->metric_id, p->score_id, d_mask, plane, 0xFFFFFFFFu, &e2, nullptr));
    else LGR_TRY(evaluate_one(ctx, d_Tn, pk, c, p->metric_id, p->score_id, d_mask, &e2, false));
    float* hT;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &hT));
    LGR_HIP(ctx, hipMemcpyAsync(hT, d_Tn, 64, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(res->transformation, hT, 64);
    res->iterations = done;
    res->converged = converged ? 1 : 0;
    res->n_inliers = e2.n_inl;
    res->metric = e2.metric;
    res->best_metric_before_refit = final_metric;
    res->best_iteration = best_iter;
    res->num_rejections = num_rejections;
    res->estimated_iters = bound;
    res->time_te = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return LGR_OK;
}

extern "C" int lgr_ransac(lgr_ctx* ctx, const float* src, int ns, const float* tgt, int nt, const lgr_corr* corr, int c,
                          const lgr_params* p, lgr_result* res, uint8_t* final_mask) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, src && tgt && (corr || c == 0) && res && ns > 0 && nt > 0 && c >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *ds, *dt;
    lgr_corr* dc;
    uint8_t* dm;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) ns * 12, &ds));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) nt * 12, &dt));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) c + 1, &dc));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_D, (size_t) c + 16, &dm));
    LGR_HIP(ctx, hipMemcpyAsync(ds, src, (size_t) ns * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(dt, tgt, (size_t) nt * 48, hipMemcpyHostToDevice, ctx->stream));
    if (c) LGR_HIP(ctx, hipMemcpyAsync(dc, corr, (size_t) c * 16, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_ransac_dev(ctx, ds, ns, dt, nt, dc, c, p, res, dm));
    if (final_mask && c >= p->n_samples) {
        LGR_HIP(ctx, hipMemcpyAsync(final_mask, dm, (size_t) c, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else if (final_mask && c > 0) memset(final_mask, 0, c);
    return LGR_OK;
}

extern "C" int lgr_refit_svd_dev(lgr_ctx* ctx, const float* d_src, const float* d_tgt, const lgr_corr* d_corr, int c,
                                 const uint8_t* d_mask, float T16[16]) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && (d_corr || c == 0) && T16 && c >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float4* P;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_PACK, (size_t) c * 2 + (size_t) (c + 3) / 4 + 4, &P));
    Packed pk{P, P + c, (float*) (P + 2 * (size_t) c), nullptr, nullptr};
    if (c > 0) pack_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(d_src, d_tgt, d_corr, c, nullptr, pk.P0, pk.P1, pk.sstar, nullptr, nullptr, c);
    float* dT;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, 64, &dT));
    LGR_TRY(refit_launch(ctx, pk, c, d_mask, dT));
    float* hT;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &hT));
    LGR_HIP(ctx, hipMemcpyAsync(hT, dT, 64, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(T16, hT, 64);
    return LGR_OK;
}

extern "C" int lgr_refit_svd(lgr_ctx* ctx, const float* src, const float* tgt, int ns, int nt, const lgr_corr* inliers, int n, float T16[16]) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, src && tgt && (inliers || n == 0) && T16 && ns > 0 && nt > 0 && n >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *ds, *dt;
    lgr_corr* dc;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) ns * 12, &ds));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) nt * 12, &dt));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) n + 1, &dc));
    LGR_HIP(ctx, hipMemcpyAsync(ds, src, (size_t) ns * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(dt, tgt, (size_t) nt * 48, hipMemcpyHostToDevice, ctx->stream));
    if (n) LGR_HIP(ctx, hipMemcpyAsync(dc, inliers, (size_t) n * 16, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_check_corr(ctx, dc, n, ns, nt));   // the _dev form mirrors estimateOptimalRigidTransformation(src, tgt, inliers, T) and has no sizes to check against
    return lgr_refit_svd_dev(ctx, ds, dt, dc, n, nullptr, T16);
}

// src/hypotheses.cpp:50-129 chooseBestHypothesis (compiled out in the reference like updateHypotheses): the decision --
// the hypothesis whose correspondence inliers are spread most uniformly (strict >, identity when none is positive).  The
// hypotheses.csv side output of the reference (inlier / overlap areas) is not produced.
extern "C" int lgr_choose_best_hypothesis_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                                              const float* tns16, int n, float T_out16[16], int* best_index, float* uniformities) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && (d_corr || c == 0) && (tns16 || n == 0) && T_out16 && n >= 0 && c >= 0 && ns > 0 && nt > 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < 16; ++i) T_out16[i] = (i % 5 == 0) ? 1.f : 0.f;
    int best_i = -1;
    float best = 0.f;
    if (n > 0) {
        Packed pk;
        LGR_TRY(pack(ctx, d_src, ns, d_tgt, nt, d_corr, c, &pk));
        float* dT;
        LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_T, (size_t) n * 16, &dT));
        LGR_HIP(ctx, hipMemcpyAsync(dT, tns16, (size_t) n * 64, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < n; ++i) {
            EvalOut e{0, 0.f, 0.f};
            if (c > 0) LGR_TRY(evaluate_one(ctx, dT + (size_t) i * 16, pk, c, LGR_METRIC_UNIFORMITY, LGR_SCORE_MSE, nullptr, &e));
            if (uniformities) uniformities[i] = e.metric;
            if (e.metric > best) { best = e.metric; best_i = i; memcpy(T_out16, tns16 + (size_t) i * 16, 64); }
        }
    }
    if (best_index) *best_index = best_i;
    return LGR_OK;
}

// src/hypotheses.cpp:14-48 updateHypotheses: pure host bookkeeping (the call sites are compiled out in the reference,
// SAVE_MULTIPLE_HYPOTHESES false, src/sac_prerejective_omp.cpp:11); tns16 = n column-major 4x4, capacity cap.
extern "C" int lgr_update_hypotheses(float* tns16, float* metrics, int n, int cap, const float* new_T16, float new_metric, float distance_thr) {
    if (!tns16 || !metrics || !new_T16 || n < 0 || cap < n) return LGR_ERR_INVALID_ARG;
    auto diff = [](const float* T1, const float* T2, float& angle, float& td) {
        // src/analysis.cpp:19-24: angle of R1^-1 R2, |t1 - t2|
        double R[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += (double) T1[4 * i + k] * (double) T2[4 * j + k];
                R[3 * i + j] = s;
            }
        double tr = R[0] + R[4] + R[8];
        double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
        angle = (float) std::atan2(0.5 * std::sqrt(vx * vx + vy * vy + vz * vz), 0.5 * (tr - 1.0));
        double dx = (double) T1[12] - T2[12], dy = (double) T1[13] - T2[13], dz = (double) T1[14] - T2[14];
        td = (float) std::sqrt(dx * dx + dy * dy + dz * dz);
    };
    std::vector<std::vector<float>> T(n, std::vector<float>(16));
    std::vector<float> M(metrics, metrics + n);
    for (int i = 0; i < n; ++i) memcpy(T[i].data(), tns16 + 16 * (size_t) i, 64);
    float best = n == 0 ? 0.f : *std::max_element(M.begin(), M.end());
    auto flush = [&]() {
        int m = (int) T.size();
        if (m > cap) return (int) LGR_ERR_INVALID_ARG;
        for (int i = 0; i < m; ++i) { memcpy(tns16 + 16 * (size_t) i, T[i].data(), 64); metrics[i] = M[i]; }
        return m;
    };
    if (new_metric < 0.1 * best) return flush();
    std::vector<int> similar;
    for (int i = (int) T.size() - 1; i >= 0; --i) {
        float r, t;
        diff(new_T16, T[i].data(), r, t);
        bool is_similar = r < (M_PI / 9) && t < 20 * distance_thr;
        if (is_similar) similar.push_back(i);
        if (is_similar && M[i] > new_metric) return flush();
    }
    for (int idx : similar) { T.erase(T.begin() + idx); M.erase(M.begin() + idx); }
    T.emplace_back(new_T16, new_T16 + 16);
    M.push_back(new_metric);
    if (new_metric > best)
        for (int i = (int) T.size() - 1; i >= 0; --i)
            if (M[i] < 0.1 * new_metric) { T.erase(T.begin() + i); M.erase(M.begin() + i); }
    return flush();
}
