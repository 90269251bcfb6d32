// lgr_features.hip -- voxel downsample, k-NN PCA normals, SPFH / FPFH (gfx950).
//
//   src/downsample.cpp:5-41 + include/downsample.h:6-30      -> lgr_downsample*
//   src/common.cpp:644-655, 593-628 (pcl::NormalEstimationOMP) -> lgr_normals_knn*
//   include/common.h:322-332 (pcl::FPFHEstimationOMP)          -> lgr_fpfh*
// Every parity-critical float sequence below restates the oracle op for op (oracle/src/orc_features.cpp,
// orc_math.h); the translation unit is compiled with -ffp-contract=off.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "lgr_grid.cuh"
#include "lgr_knn_wave.cuh"
#include "lgr_seqsum.h"
#include "lgr_math.cuh"
#include "lgr_libm.cuh"

namespace {

// ------------------------------------------------------------------------------------------------ downsample
__global__ void voxel_keys(const float* __restrict__ pts, int n, float bx, float by, float bz, float voxel,
                           unsigned long long* __restrict__ keys, int* __restrict__ vals, int* __restrict__ bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = pts[(size_t) i * 12], y = pts[(size_t) i * 12 + 1], z = pts[(size_t) i * 12 + 2];
    unsigned long long k = ~0ull;   // invalid points sort last and are dropped (isValid, src/downsample.cpp:25)
    if (lgr_finite3(x, y, z)) {
        // voxel_index = int(floor((p - voxel_min_bound) / voxel_size))   src/downsample.cpp:26-28
        int ix = (int) floorf((x - bx) / voxel), iy = (int) floorf((y - by) / voxel), iz = (int) floorf((z - bz) / voxel);
        if (ix < 0 || iy < 0 || iz < 0 || ix >= (1 << 21) || iy >= (1 << 21) || iz >= (1 << 21)) { atomicExch(bad, 1); ix = iy = iz = 0; }
        k = ((unsigned long long) iz << 42) | ((unsigned long long) iy << 21) | (unsigned long long) ix;
    }
    keys[i] = k; vals[i] = i;
}

__global__ void head_flags(const unsigned long long* __restrict__ keys, int n, int* __restrict__ flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = keys[i];
    flags[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// one thread per voxel: sequential intensity-weighted sums in input order (AccumulatedPoint::AddPoint /
// GetAveragePoint, include/downsample.h:8-27), written at the voxel's rank in (iz,iy,ix) order.
__global__ void voxel_accumulate(const float* __restrict__ pts, const unsigned long long* __restrict__ keys,
                                 const int* __restrict__ vals, const int* __restrict__ flags,
                                 const int* __restrict__ rank /* exclusive scan of flags */, int n,
                                 float* __restrict__ out, unsigned long long* __restrict__ out_keys, int* __restrict__ out_first) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    unsigned long long k = keys[i];
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f, snx = 0.f, sny = 0.f, snz = 0.f;
    for (int j = i; j < n && keys[j] == k; ++j) {
        const float4* p = reinterpret_cast<const float4*>(pts + (size_t) vals[j] * 12);
        float4 a = p[0], b = p[1], c = p[2];
        float w = c.x;
        sx += w * a.x; sy += w * a.y; sz += w * a.z;
        sw += w;
        snx += w * b.x; sny += w * b.y; snz += w * b.z;
    }
    float ax = snx / sw, ay = sny / sw, az = snz / sw;
    float norm = __builtin_sqrtf(ax * ax + ay * ay + az * az);
    norm = ((double) norm < 1e-5) ? 1.f : norm;
    int r = rank[i];
    float4* o = reinterpret_cast<float4*>(out + (size_t) r * 12);
    o[0] = make_float4(sx / sw, sy / sw, sz / sw, 1.f);
    o[1] = make_float4(ax / norm, ay / norm, az / norm, 0.f);
    o[2] = make_float4(sw, 0.f, 0.f, 0.f);
    if (out_keys) { out_keys[r] = k; out_first[r] = vals[i]; }
}

// ------------------------------------------------------------------------------------------------ normals
constexpr int NB = 128;

// covariance of the neighbours (in list order), its smallest eigenvector, orientation and curvature: the tail of the per-point work
template <class IndexOf>
__device__ __forceinline__ void normals_finish(float* __restrict__ p, float px, float py, float pz, int count, IndexOf&& index_of,
                                               const float* __restrict__ surf, float vpx, float vpy, float vpz) {
    const float nanv = __uint_as_float(0x7fc00000u);
    if (count < 3) { p[4] = nanv; p[5] = nanv; p[6] = nanv; p[9] = nanv; return; }
    const float* K = surf + (size_t) index_of(0) * 12;
    float Kx = K[0], Ky = K[1], Kz = K[2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f, a8 = 0.f;
    for (int j = 0; j < count; ++j) {
        const float* q = surf + (size_t) index_of(j) * 12;
        float x = q[0] - Kx, y = q[1] - Ky, z = q[2] - Kz;
        a0 += x * x; a1 += x * y; a2 += x * z; a3 += y * y; a4 += y * z; a5 += z * z;
        a6 += x; a7 += y; a8 += z;
    }
    float fn = (float) count;
    a0 /= fn; a1 /= fn; a2 /= fn; a3 /= fn; a4 /= fn; a5 /= fn; a6 /= fn; a7 /= fn; a8 /= fn;
    float C[9];
    C[0] = a0 - a6 * a6; C[1] = a1 - a6 * a7; C[2] = a2 - a6 * a8;
    C[4] = a3 - a7 * a7; C[5] = a4 - a7 * a8; C[8] = a5 - a8 * a8;
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    // pcl::solvePlaneParameters -> pcl::eigen33: smallest eigenvalue and its eigenvector, PCL's closed form (lgr_libm.cuh; rounds 1-4 used a
    // Jacobi solver here, a declared deviation of ~1e-6 that moved 0.3 % of the match indices: profiles/r5_pcl_order_by_piece_1M.json)
    float lambda_min, nx, ny, nz;
    lgr_pcl_eigen33(C, lambda_min, nx, ny, nz);
    float eig_sum = C[0] + C[4] + C[8];
    float curv = (eig_sum != 0.f) ? fabsf(lambda_min / eig_sum) : 0.f;
    float dx = vpx - px, dy = vpy - py, dz = vpz - pz;
    float cos_theta = (dx * nx + dy * ny + dz * nz);
    if (cos_theta < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    if (lgr_finite3(nx, ny, nz)) {
        float norm = __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
        nx /= norm; ny /= norm; nz /= norm;
    }
    p[4] = nx; p[5] = ny; p[6] = nz; p[9] = curv;
}

// pcl::NormalEstimationOMP (k-NN) + flipNormalTowardsViewpoint + postprocessNormals; see oracle orc_normals_knn.
// by_grid: the queries are the surface points themselves; thread t takes the point at sorted position t of the grid, so a
// wave's 64 queries sit in one or two cells and walk the same rings (coherent loops and loads); points that are not in the
// grid (non-finite) are written by the `else` branch of a second launch over the original order (by_grid = 2).
__global__ __launch_bounds__(NB) void normals_kernel(GridDev g, const float* __restrict__ surf, float* __restrict__ pts, int n,
                                                      int k, float vpx, float vpy, float vpz, int by_grid) {
    extern __shared__ float smem[];
    float* sd = smem;
    int* si = (int*) (smem + (size_t) k * NB);
    int i = blockIdx.x * NB + threadIdx.x;
    if (by_grid == 1) {
        i = lgr_xcd_tile(blockIdx.x, cdiv_dev(g.n, NB)) * NB + threadIdx.x;   // grid order, one contiguous range per XCD (lgr_grid.cuh)
        if (i >= g.n) return;
        i = __float_as_int(g.pxyz[i].w);
    } else if (i >= n) return;
    float* p = pts + (size_t) i * 12;
    float px = p[0], py = p[1], pz = p[2];
    if (by_grid == 2 && lgr_finite3(px, py, pz)) return;   // done by the grid-ordered launch
    KnnList<NB> L;
    L.init(sd, si, k, threadIdx.x);
    if (lgr_finite3(px, py, pz) && g.n > 0) lgr_knn_query(g, px, py, pz, L);
    normals_finish(p, px, py, pz, L.count, [&](int j) { return L.index(j); }, surf, vpx, vpy, vpz);
}

// The same with the wave-per-query search (lgr_knn_wave.cuh) for the grid-ordered launch: a wave finds the neighbours of its 64
// points one point at a time (64 lanes = 64 candidates), leaves the sorted index lists in LDS ([k][64] per wave), and then goes back
// to one point per lane for the covariance and its eigenvector.
constexpr int NW_WAVES = 4;
template <int KPL>
__global__ __launch_bounds__(64 * NW_WAVES) void normals_wave_kernel(GridDev g, const float* __restrict__ surf, float* __restrict__ pts, int k,
                                                                      float vpx, float vpy, float vpz, float r2_init) {
    extern __shared__ int slist[];   // [NW_WAVES][k][64]
    __shared__ unsigned long long sbuf[NW_WAVES][WaveKnn<KPL>::BUF];
    __shared__ int srow[NW_WAVES][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_tiles = cdiv_dev(g.n, 64 * NW_WAVES);
    const int tile = lgr_xcd_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int t = (tile * NW_WAVES + wv) * 64 + lane;
    int* list = slist + (size_t) wv * k * 64;
    int qi = -1;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (t < g.n) {
        const float4 q = g.pxyz[t];
        qi = __float_as_int(q.w); px = q.x; py = q.y; pz = q.z;
    }
    float guess = r2_init;
    int count = 0;
    lgr_wave_knn_tile<KPL>(g, qi >= 0, px, py, pz, k, guess, sbuf[wv], srow[wv], [&](int l, const WaveKnn<KPL>& W) {   // (grid points are finite)
        const int mk = min(W.m, k);
#pragma unroll
        for (int j = 0; j < KPL; ++j)
            if (W.rank[j] < mk) list[W.rank[j] * 64 + l] = (int) (unsigned) W.key[j];
        if (lane == l) count = mk;
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (qi < 0) return;
    normals_finish(pts + (size_t) qi * 12, px, py, pz, count, [&](int j) { return list[j * 64 + lane]; }, surf, vpx, vpy, vpz);
}

// ------------------------------------------------------------------------------------------------ FPFH
__device__ __forceinline__ float dot3e(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + az * bz) + ay * by;   // Eigen 4-float SSE reduction (a0+a2)+(a1+a3), a3 = 0
}

// pcl::computePairFeatures.  Returns false when the pair is skipped.
__device__ __forceinline__ bool pair_features(float p1x, float p1y, float p1z, float n1x, float n1y, float n1z,
                                              float p2x, float p2y, float p2z, float n2x, float n2y, float n2z,
                                              float& f1, float& f2, float& f3) {
    float dx = p2x - p1x, dy = p2y - p1y, dz = p2z - p1z;
    float f4 = __builtin_sqrtf(dot3e(dx, dy, dz, dx, dy, dz));
    if (f4 == 0.0f) return false;
    float angle1 = dot3e(n1x, n1y, n1z, dx, dy, dz) / f4;
    float angle2 = dot3e(n2x, n2y, n2z, dx, dy, dz) / f4;
    float ux = n1x, uy = n1y, uz = n1z, mx = n2x, my = n2y, mz = n2z;
    float a1 = fabsf(angle1), a2 = fabsf(angle2);
    if (lgr_glibc::acosf_(a1) > lgr_glibc::acosf_(a2)) {   // PCL: std::acos (std::fabs (angle1)) > std::acos (std::fabs (angle2)) on floats (lgr_libm.cuh)
        ux = n2x; uy = n2y; uz = n2z; mx = n1x; my = n1y; mz = n1z;
        dx = -dx; dy = -dy; dz = -dz;
        f3 = -angle2;
    } else {
        f3 = angle1;
    }
    float vx = dy * uz - dz * uy, vy = dz * ux - dx * uz, vz = dx * uy - dy * ux;
    float v_norm = __builtin_sqrtf(dot3e(vx, vy, vz, vx, vy, vz));
    if (v_norm == 0.0f) return false;
    vx /= v_norm; vy /= v_norm; vz /= v_norm;   // PCL: v /= v_norm; Eigen 3.3+ divides every component (no reciprocal)
    float wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx;
    f2 = dot3e(vx, vy, vz, mx, my, mz);
    float yy = dot3e(wx, wy, wz, mx, my, mz), xx = dot3e(ux, uy, uz, mx, my, mz);
    f1 = lgr_glibc::atan2f_(yy, xx);   // std::atan2 on floats (lgr_libm.cuh)
    return true;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- the three bin indices of a pair through a FILTER with a proven decision band (round 4) ----
// pcl::computePairFeatures is used here for three INTEGERS only (the bins of f1, f2, f3).  pair_bins_fast evaluates the same formulas
// with the hardware's approximate reciprocal / reciprocal square root (v_rcp_f32, v_rsq_f32: 1 ulp) instead of two IEEE square roots and
// seven IEEE divisions, and the bin arithmetic in float instead of double; every operation that is NOT a division or a square root is
// the very instruction of the canonical sequence (same operands as long as the swap decision is the same), so the two evaluations
// differ by bounded amounts:
//   u = 2^-24.  angle_k: canonical RN(dot / RN(sqrt)) carries 2u, the fast dot * rsq 3u + 2u  ->  |d angle| <= 5u |angle| < 4e-7      (DA = 1e-6)
//   v / |v|: canonical 2u, fast 3u per component (|v_i / |v|| <= 1)                           ->  |d v_i| <= 3e-7
//   w = u x v: two products of |.| <= 1 and a difference, three roundings per path            ->  |d w_i| <= 2 * 3e-7 + 6u < 1e-6
//   f2 = v . m (|m| ~ 1, sum |m_i| <= sqrt 3), five roundings per path                        ->  |d f2| <= 5.2e-7 + 10u < 1.2e-6           (DF2 = 3e-6)
//   yy = w . m                                                                                 ->  |d yy| <= 1.8e-6 + 10u < 2.4e-6           (DY = 5e-6)
//   f1 = atan2(yy, xx), xx identical in both: |d f1| <= DY / rho (rho^2 = xx^2 + yy^2, the sensitivity of atan2 to yy) + the two
//        evaluations' own distance to atan2 (glibc's atan2f: < 1 ulp of pi; fast: a Cephes polynomial on rcp quotients, <= 4 ulp of pi + 6u)   (DF1 = 5e-6 / rho + 3e-6)
// A bin is floor(t), t = 5.5 (f + 1) (f2, f3) or 11 (f1 + pi) / (2 pi_f) (f1) -- the canonical code evaluates t in double, i.e. exactly
// at this scale.  The fast t (one fused multiply-add, |t| <= 11: 11 u) is within 5.5 D + 1e-6 (f2, f3) / 1.751 D + 2e-6 (f1) of it, so
// when the fast t is farther than that from every integer its floor IS the canonical bin.  The swap decision (PCL: acosf(|angle1|) >
// acosf(|angle2|); acos decreases with slope <= -1 and glibc's acosf is within an ulp (1.2e-7) of it, so canonical magnitudes more than 1.2e-6
// apart order their acosf values strictly, and a magnitude above 1 gives NaN = "no swap") is certain when the two fast magnitudes are
// more than 2 DA apart and, for "swap", the larger is below 1 - DA.  Anything else -- a value
// inside a band, a degenerate pair (coincident points, d parallel to the normal), rho below 1e-2, a NaN anywhere (every test is written
// so that NaN fails it) -- is NOT decided here: the caller evaluates that pair with the canonical sequence.  On the 1M bench pair
// about one pair in 10^4 is; the rows are bit-identical to the oracle's (tests/test_gpu_parity_1m.py::test_fpfh_1m_full compares all
// 2 x 33 M values), and -DLGR_SPFH_CHECK builds a kernel that evaluates BOTH for every pair and counts disagreements (tools/exp_spfh_check.py).
__device__ __forceinline__ float lgr_atan2f_fast(float y, float x) {   // lgr_atan2f's polynomial on v_rcp_f32 quotients
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = (ax > ay) ? ax : ay;
    const float mn = (ax > ay) ? ay : ax;
    const float a = mn * __builtin_amdgcn_rcpf(mx);
    const bool hi = a > 0.41421356237f;
    const float z = hi ? (a - 1.0f) * __builtin_amdgcn_rcpf(a + 1.0f) : a;
    const float z2 = z * z;
    float p = 8.05374449538e-2f * z2 - 1.38776856032e-1f;
    p = p * z2 + 1.99777106478e-1f;
    p = p * z2 - 3.33329491539e-1f;
    p = p * z2 * z + z;
    float r = (hi ? 0.78539816339f : 0.0f) + p;
    if (ay > ax) r = 1.57079632679f - r;
    if (x < 0.0f) r = 3.14159265359f - r;
    if (y < 0.0f) r = -r;
    return r;
}
// floor of a fast t when it is farther than `margin` from every integer (clamped like bin11); returns false when it is not
__device__ __forceinline__ bool bin_decided(float t, float margin, int& b) {
    const float fl = floorf(t);
    const float fr = t - fl;                        // in [0, 1): exact (Sterbenz / small integers)
    b = (int) fminf(fmaxf(fl, 0.0f), 10.0f);
    return fr > margin && fr < 1.0f - margin;       // (NaN: false)
}
// One pair per lane (round 5; two per lane on packed operations before: 2 x 16 more operand registers for v_pk_* instructions that issue in the
// time of two scalar ones).  dec0: the pair's three bins were decided (b0 valid); otherwise the caller runs the canonical evaluation for it.
__device__ __forceinline__ void pair_bins_fast(float p1x, float p1y, float p1z, float n1x, float n1y, float n1z, float p2x, float p2y, float p2z, float n2x, float n2y, float n2z,
                                               int (&b0)[3], bool& dec0) {
    constexpr float DA = 1e-6f;
    float dx = p2x - p1x, dy = p2y - p1y, dz = p2z - p1z;
    const float f4s = dot3e(dx, dy, dz, dx, dy, dz);
    const float rs = __builtin_amdgcn_rsqf(f4s);
    const float angle1 = dot3e(n1x, n1y, n1z, dx, dy, dz) * rs;
    const float angle2 = dot3e(n2x, n2y, n2z, dx, dy, dz) * rs;
    const float a1 = fabsf(angle1), a2 = fabsf(angle2);
    // swap <=> acosf(a1) > acosf(a2) (canonical = PCL); certain "no" when a1 > a2 + 2 DA, certain "yes" when a1 < a2 - 2 DA and a2 < 1 - DA
    const bool s0 = a1 < a2;
    bool c0 = f4s > 1e-30f && (s0 ? (a2 - a1 > 2.0f * DA && a2 < 1.0f - DA) : (a1 - a2 > 2.0f * DA));
    const float ux = s0 ? n2x : n1x, uy = s0 ? n2y : n1y, uz = s0 ? n2z : n1z;
    const float mx = s0 ? n1x : n2x, my = s0 ? n1y : n2y, mz = s0 ? n1z : n2z;
    dx = s0 ? -dx : dx; dy = s0 ? -dy : dy; dz = s0 ? -dz : dz;
    const float f3 = s0 ? -angle2 : angle1;
    float vx = dy * uz - dz * uy, vy = dz * ux - dx * uz, vz = dx * uy - dy * ux;
    const float vn2 = dot3e(vx, vy, vz, vx, vy, vz);
    const float rv = __builtin_amdgcn_rsqf(vn2);
    c0 = c0 && vn2 > 1e-30f;
    vx = vx * rv; vy = vy * rv; vz = vz * rv;
    const float wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx;
    const float f2 = dot3e(vx, vy, vz, mx, my, mz);
    const float yy = dot3e(wx, wy, wz, mx, my, mz), xx = dot3e(ux, uy, uz, mx, my, mz);
    const float rho2 = xx * xx + yy * yy;
    const float f1 = lgr_atan2f_fast(yy, xx);
    const float irho = __builtin_amdgcn_rsqf(rho2);
    c0 = c0 && rho2 > 1e-4f;
    // t = 11 (f1 + pi) / (2 pi_f): scale and offset of the canonical double expression, rounded to float once
    constexpr float S1 = (float) (11.0 * (double) (1.0f / (2.0f * 3.14159274101257324f)));
    constexpr float O1 = (float) (11.0 * 3.14159265358979323846 * (double) (1.0f / (2.0f * 3.14159274101257324f)));
    const float t1 = __builtin_fmaf(f1, S1, O1), t2 = __builtin_fmaf(f2, 5.5f, 5.5f), t3 = __builtin_fmaf(f3, 5.5f, 5.5f);
    const float m1 = __builtin_fmaf(irho, 1.751f * 5e-6f, 1.751f * 3e-6f + 2e-6f);
    constexpr float M2 = 5.5f * 3e-6f + 1e-6f, M3 = 5.5f * DA + 1e-6f;
    const bool d01 = bin_decided(t1, m1, b0[0]), d02 = bin_decided(t2, M2, b0[1]), d03 = bin_decided(t3, M3, b0[2]);
    dec0 = d01 && d02 && d03 && c0;
}

__device__ __forceinline__ int bin11(double t) {
    double v = floor(11 * t);
    if (!(v == v)) return 0;
    if (v < 0) return 0;
    if (v >= 11) return 10;
    return (int) v;
}

// SPFH rows are kept in the SORTED order of the grid (row t = the point at sorted position t, so the points of a cell --
// the candidates the weighting step streams -- are consecutive rows) with a pitch of HP = 48 floats: bins 0..32 and 15 zeros
// = three 16-column MFMA operand tiles; the zeros make the third tile (bin 32 alone) a plain load.  Stored interleaved (spfh_slot).
constexpr int HP = 48;
// Slot of bin b in an SPFH row: the weighting kernel's lane i multiplies bins i, 16 + i and 32 + i (one column of each 16-bin MFMA tile), so those
// three are stored next to each other and arrive as ONE 12-byte load per lane and candidate (three 64-byte segments per candidate before).
__host__ __device__ constexpr int spfh_slot(int b) { return (b & 15) * 3 + (b >> 4); }
// SPFH rows for the surface points listed in the sorted order of the grid (thread t handles sorted position order[t]).
// Counters live in LDS as [bin][thread] (bank = thread % 32: conflict-free).
// SPFH rows of 16 surface points per wave with the expensive part (pcl::computePairFeatures: two square roots, three divisions,
// an atan2, three bin indices) run on COMPACTED work: a thread-per-point loop over the 27-cell candidates keeps only the ~35 % of
// its lanes busy that accept their current candidate.  Here lane = candidate: the points of the 27 cells around a run of tile
// points (same cell) are streamed 64 at a time, kept when within r of the run's bounding box and staged in LDS; for every tile
// point all lanes test their candidate (d2 < r2, the k count of the point = popcount of the ballot) and the accepted
// (point, candidate) pairs are appended to a queue in LDS; whenever 64 pairs are queued every lane takes one and does the
// pair-feature arithmetic -- all 64 lanes busy -- and three LDS atomic increments into the point's histogram.  Counts are
// order free (the canonical value of a bin is the sequential float sum of `count` copies of the increment), so any
// enumeration that meets every (point, neighbour) pair exactly once gives the oracle's rows bit for bit.
#ifdef LGR_SPFH_CHECK
__device__ unsigned long long g_spfh_check[4];   // pairs, pairs the filter left undecided, decided pairs whose bins differ from the canonical ones (must stay 0)
#endif
constexpr int ST = 16;          // surface points per wave
constexpr int SQ = 256;         // pair queue entries: (tile point << 8) | candidate slot
constexpr int SHP = 33;         // histogram pitch (lanes of one tile point hit bins = banks; the pad of rounds 1-4 bought nothing)

// Five waves per SIMD, as a MINIMUM, and one pair per lane (round 5): with two pairs per lane the allocator took 113 VGPRs and four waves; one pair
// needs 97, held to 96 it spills nothing, and the FPFH stage alone goes 2.89 -> 2.72 ms per cloud (40 % of the wave cycles were parked behind
// LDS / memory waits at four waves).  Two pairs per lane held to 96 VGPRs: 2.75 ms with 14 spilled registers; six waves need 80 VGPRs (60 B of
// scratch) and more LDS than a CU has for 24 of these workgroups.
#ifndef LGR_SPFH_WAVES
#define LGR_SPFH_WAVES 5
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LGR_SPFH_WAVES, LGR_SPFH_WAVES))) void spfh_tile_kernel(GridDev g, float r2, const int* __restrict__ order, float* __restrict__ spfh /* [n][HP], sorted positions */) {
    __shared__ float4 tp[ST], tn[ST];
    __shared__ float4 cp[64], cn[64];     // live candidates: tested when 64 are buffered (round 5: a chunk's overflow waits in registers, not in a second half of the buffer)
    __shared__ unsigned short queue[SQ];
    __shared__ int hist[2][ST][SHP];
    __shared__ int kcnt[ST];
    const int l = threadIdx.x;
    int pos = -1, cell = -1;   // sorted position / cell of tile point l (lanes 0..15)
    float x = 0.f, y = 0.f, z = 0.f;
    const int tile = lgr_xcd_tile(blockIdx.x, cdiv_dev(g.n, ST));   // one contiguous range of the sorted order per XCD (lgr_grid.cuh)
    if (tile * ST >= g.n) return;
    if (l < ST) {
        const int rnk = tile * ST + l;
        if (rnk < g.n) {
            pos = order ? order[rnk] : rnk;
            const float4 P = g.pxyz[pos];
            x = P.x; y = P.y; z = P.z;
            tp[l] = make_float4(P.x, P.y, P.z, __int_as_float(pos));
            tn[l] = g.pnrm[pos];
            const int cx = min(max(lgr_cellc(x, g.ox, g.h), 0), g.dx - 1), cy = min(max(lgr_cellc(y, g.oy, g.h), 0), g.dy - 1), cz = min(max(lgr_cellc(z, g.oz, g.h), 0), g.dz - 1);
            cell = (cz * g.dy + cy) * g.dx + cx;
        }
        kcnt[l] = 0;
    }
    for (int q = l; q < 2 * ST * SHP; q += 64) (&hist[0][0][0])[q] = 0;
    __syncthreads();
    const float r2box = r2 * 1.0001f + 1e-30f;
    const float d_pi = 1.0f / (2.0f * 3.14159274101257324f);   // 1.0f / (2.0f * static_cast<float>(M_PI))
    const double MPI = 3.14159265358979323846;
    int qh = 0, qt = 0;
    // (always_inline: left to its heuristics hipcc turns a lambda of this size into a CALL, with the captures in scratch memory)
    auto process = [&](int nb) __attribute__((always_inline)) {   // the first nb (<= 64) queued pairs, one per lane
        if (l < nb) {
            const unsigned e0 = queue[(qh + l) & (SQ - 1)];
            const int i0 = (int) (e0 >> 8);
            const float4 P0 = tp[i0], N0 = tn[i0], Q0 = cp[e0 & 255u], M0 = cn[e0 & 255u];
            // the bins through the filter (pair_bins_fast); the few pairs it does not decide go through the canonical sequence
            int ba[3];
            bool dec0, ok0 = true;
            pair_bins_fast(P0.x, P0.y, P0.z, N0.x, N0.y, N0.z, Q0.x, Q0.y, Q0.z, M0.x, M0.y, M0.z, ba, dec0);
#ifdef LGR_SPFH_CHECK
            const bool fdec0 = dec0;
            const int fa0 = ba[0], fa1 = ba[1], fa2 = ba[2];
            dec0 = false;   // evaluate the canonical sequence for EVERY pair and compare
#endif
            if (!dec0) {
                // the canonical sequence (two acosf, an atan2f: lgr_libm.cuh) for 2e-4 of the pairs; operands re-read from LDS, the fast path's registers are dead here
                const int cc = (int) (e0 & 255u);
                const float Px = tp[i0].x, Py = tp[i0].y, Pz = tp[i0].z, Nx = tn[i0].x, Ny = tn[i0].y, Nz = tn[i0].z;
                const float Qx = cp[cc].x, Qy = cp[cc].y, Qz = cp[cc].z, Mx = cn[cc].x, My = cn[cc].y, Mz = cn[cc].z;
                float f1, f2, f3;
                ok0 = pair_features(Px, Py, Pz, Nx, Ny, Nz, Qx, Qy, Qz, Mx, My, Mz, f1, f2, f3);
                ba[0] = bin11(((double) f1 + MPI) * (double) d_pi); ba[1] = bin11(((double) f2 + 1.0) * 0.5); ba[2] = bin11(((double) f3 + 1.0) * 0.5);
            }
#ifdef LGR_SPFH_CHECK
            {
                const bool bad0 = fdec0 && (!ok0 || fa0 != ba[0] || fa1 != ba[1] || fa2 != ba[2]);
                const unsigned long long n_pairs = __popcll(__ballot(true));
                const unsigned long long n_und = __popcll(__ballot(!fdec0));
                const unsigned long long n_bad = __popcll(__ballot(bad0));
                if (l == 0) { atomicAdd(&g_spfh_check[0], n_pairs); atomicAdd(&g_spfh_check[1], n_und); if (n_bad) atomicAdd(&g_spfh_check[2], n_bad); }
            }
#endif
            if (ok0) {
                int* h = &hist[l & 1][i0][0];
                atomicAdd(h + ba[0], 1); atomicAdd(h + 11 + ba[1], 1); atomicAdd(h + 22 + ba[2], 1);
            }
        }
        qh += nb;
    };
    for (int p = 0; p < ST;) {
        const int c = __builtin_amdgcn_readlane(cell, p);
        const bool active = l < ST && cell == c;
        const int n_run = __popcll(__ballot(active));
        const int p0 = p;
        p += n_run;
        if (c < 0) break;   // the rest of the tile lies beyond g.n
        float bx0 = active ? x : 3.4e38f, bx1 = active ? x : -3.4e38f, by0 = active ? y : 3.4e38f, by1 = active ? y : -3.4e38f;
        float bz0 = active ? z : 3.4e38f, bz1 = active ? z : -3.4e38f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            bx0 = fminf(bx0, __shfl_xor(bx0, o)); bx1 = fmaxf(bx1, __shfl_xor(bx1, o));
            by0 = fminf(by0, __shfl_xor(by0, o)); by1 = fmaxf(by1, __shfl_xor(by1, o));
            bz0 = fminf(bz0, __shfl_xor(bz0, o)); bz1 = fmaxf(bz1, __shfl_xor(bz1, o));
        }
        const int cz = c / (g.dx * g.dy), cy = (c / g.dx) % g.dy, cx = c % g.dx;
        int n_buf = 0;   // live candidates waiting in cp / cn [0, n_buf)
        // every tile point of the run against the first n_c buffered candidates (lane = candidate), accepted pairs queued and
        // processed 64 at a time; the queue is drained before the candidate slots are reused
        auto test_block = [&](int n_c) __attribute__((always_inline)) {
            float4 Q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l < n_c) Q = cp[l];
            for (int i = p0; i < p0 + n_run; ++i) {
                const float4 P_i = tp[i];
                const float d2 = lgr_dist2(P_i.x, P_i.y, P_i.z, Q.x, Q.y, Q.z);
                const bool acc = l < n_c && d2 < r2;
                const unsigned long long am = __ballot(acc);
                if (am == 0ull) continue;
                if (l == 0) kcnt[i] += __popcll(am);
                const bool enq = acc && __float_as_int(Q.w) != __float_as_int(P_i.w);   // if (s == t) return: the point itself
                const unsigned long long em = __ballot(enq);
                if (enq) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned) (em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) em, 0u));
                    queue[(qt + rank) & (SQ - 1)] = (unsigned short) ((i << 8) | l);
                }
                qt += __popcll(em);
                if (qt - qh >= 64) {
                    __syncthreads();
                    process(64);
                }
            }
            __syncthreads();
            while (qt - qh > 0) process(min(qt - qh, 64));
            __syncthreads();
        };
        for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.dz - 1); ++zz)
            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.dy - 1); ++yy) {
                const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
                const size_t c0 = ((size_t) zz * g.dy + yy) * g.dx;
                const int b = g.cell_start[c0 + x0], e = g.cell_start[c0 + x1 + 1];
                for (int t0 = b; t0 < e; t0 += 64) {
                    const int t = t0 + l;
                    bool live = false;
                    float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (t < e) {
                        P = g.pxyz[t];
                        const float ddx = fmaxf(fmaxf(bx0 - P.x, P.x - bx1), 0.f), ddy = fmaxf(fmaxf(by0 - P.y, P.y - by1), 0.f), ddz = fmaxf(fmaxf(bz0 - P.z, P.z - bz1), 0.f);
                        live = (ddx * ddx + ddy * ddy) + ddz * ddz <= r2box;
                    }
                    const unsigned long long lm = __ballot(live);
                    if (lm == 0ull) continue;
                    // the chunk's live candidates fill the 64-entry buffer; what does not fit stays in its lane's registers until the full buffer
                    // has been tested and goes to the front of the empty one (same candidates, same order as one 128-entry buffer: half the LDS)
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned) (lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) lm, 0u));
                    const int fit = min(__popcll(lm), 64 - n_buf);
                    float4 Nq = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (live) {
                        Nq = g.pnrm[t];
                        if (rank < fit) { cp[n_buf + rank] = make_float4(P.x, P.y, P.z, __int_as_float(t)); cn[n_buf + rank] = Nq; }
                    }
                    n_buf += fit;
                    __syncthreads();
                    if (n_buf == 64) {
                        test_block(64);
                        if (live && rank >= fit) { cp[rank - fit] = make_float4(P.x, P.y, P.z, __int_as_float(t)); cn[rank - fit] = Nq; }
                        n_buf = __popcll(lm) - fit;
                        __syncthreads();
                    }
                }
            }
        if (n_buf > 0) test_block(n_buf);
    }
    __syncthreads();
    // rows: bin value = sequential float sum of `count` copies of 100 / (k - 1) (lgr_seqsum: the loop's result without the loop); zero
    // padding up to the pitch
    for (int q = l; q < ST * 33; q += 64) {
        const int i = q / 33, b = q % 33;
        if (tile * ST + i >= g.n) continue;
        const int ps = __float_as_int(tp[i].w);
        const float incr = 100.0f / (float) (kcnt[i] - 1);
        spfh[(size_t) ps * HP + spfh_slot(b)] = lgr_seqsum(incr, hist[0][i][b] + hist[1][i][b]);
    }
    for (int q = l; q < ST * (HP - 33); q += 64) {
        const int i = q / (HP - 33), b = 33 + q % (HP - 33);
        if (tile * ST + i >= g.n) continue;
        spfh[(size_t) __float_as_int(tp[i].w) * HP + spfh_slot(b)] = 0.f;
    }
}

// FPFH rows of 16 key points per wave on the f32 matrix cores (include/common.h:322-332 -> pcl::FPFHEstimation::weightPointSPFHSignature).
//
//   out[kp][bin] = sum over the neighbours q of kp (d2 < r2, d2 != 0, canonical order)  (1 / d2) * SPFH[q][bin]
//
// is the product W (key points x candidates) . SPFH (candidates x 33) with W masked to the accepted pairs.  The canonical
// definition (oracle orc_fpfh) is one fmaf chain per bin over the accepted neighbours in grid order -- which is bit for bit what
// v_mfma_f32_16x16x4_f32 computes along k (a k-ordered fmaf chain, one rounding per product), and a masked pair (weight +0,
// SPFH finite) leaves the accumulator untouched: fma(+0, h, acc) == acc.  So ANY superset of a key point's neighbours, fed in
// ascending (cell z, y, x; sorted position) order, gives the canonical row.
//
// A wave takes 16 consecutive key points of the cell-sorted order (fine_keys: cell, then Morton code).  Per run of key points
// that share a cell: the points of the 27 surrounding cells are streamed 64 at a time, kept when they lie within r of the
// run's bounding box (a superset test), compacted in order into a ring in LDS, and consumed four at a time: lane (i, k) =
// (key point i, candidate k) computes d2 in the canonical op order, the mask, 1.0f / d2 (IEEE), and the wave issues three
// 16x16x4 MFMAs against the candidates' SPFH rows (three 16-bin column tiles; pitch-48 rows, third tile = bin 32 + zeros).
// Lanes of other runs in the same wave are masked (+0), so a tile may straddle cells.  Epilogue: accumulators -> LDS, one lane
// per (key point, 11-bin block): block sum in double (ascending bins), 100 / sum, scaled bins to the key point's output row.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
// 1.0f / x for the weights, without the IEEE division sequence (v_div_scale x 2, v_rcp, five fma, v_div_fmas, v_div_fixup: 11 instructions,
// a quarter of the weighting loop): v_rcp_f32 (1 ulp) and LGR_RCP_STEPS Newton steps y += y * (1 - x y), each two fma (packed: one
// v_pk_fma_f32 per two candidates).  The result is the correctly rounded quotient for every x in [LGR_RCP_LO, LGR_RCP_HI] -- not argued, CHECKED:
// lgr_selfcheck_rcp compares the sequence with the division for every float of that range on the device it runs on
// (tests/test_gpu_features.py; 2.0e9 values).  Outside the range (a squared distance below 1e-36) the callers divide.
#ifndef LGR_RCP_STEPS
#define LGR_RCP_STEPS 1
#endif
constexpr float LGR_RCP_LO = 1e-36f;   // (upper end 1e36: lgr_fpfh_dev checks the radius)
__device__ __forceinline__ v2f lgr_rcp2(v2f x) {
    v2f y{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    const v2f one{1.f, 1.f};
#pragma unroll
    for (int s = 0; s < LGR_RCP_STEPS; ++s) {
        const v2f e = __builtin_elementwise_fma(-x, y, one);
        y = __builtin_elementwise_fma(e, y, y);
    }
    return y;
}
__device__ __forceinline__ float lgr_rcp1(float x) {
    float y = __builtin_amdgcn_rcpf(x);
#pragma unroll
    for (int s = 0; s < LGR_RCP_STEPS; ++s) {
        const float e = fmaf(-x, y, 1.f);
        y = fmaf(e, y, y);
    }
    return y;
}
// out[0] += floats of [lo_bits, hi_bits] whose lgr_rcp1 / lgr_rcp2 differs from 1.0f / x, out[1] += floats tested
__global__ void rcp_check_kernel(unsigned lo_bits, unsigned hi_bits, unsigned long long* __restrict__ out) {
    unsigned long long bad = 0, cnt = 0;
    const unsigned long long span = (unsigned long long) hi_bits - lo_bits + 1ull;
    for (unsigned long long o = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; o < span; o += (unsigned long long) gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (unsigned) o);
        const float q = 1.0f / x;
        const v2f r2 = lgr_rcp2(v2f{x, x});
        bad += (__float_as_uint(lgr_rcp1(x)) != __float_as_uint(q)) || (__float_as_uint(r2.x) != __float_as_uint(q)) || (__float_as_uint(r2.y) != __float_as_uint(q));
        ++cnt;
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); cnt += __shfl_xor(cnt, o); }
    if ((threadIdx.x & 63) == 0) { if (bad) atomicAdd(&out[0], bad); atomicAdd(&out[1], cnt); }
}
constexpr int FT = 16;        // key points per wave
constexpr int FRING = 128;    // live-candidate ring entries (float4: x, y, z, bits(sorted position * HP))
constexpr int FP_PITCH = 36;  // LDS pitch of the epilogue rows (4 * 36 mod 32 == 16: the four row groups of a store hit different banks)

__global__ __launch_bounds__(64) void fpfh_mfma_kernel(GridDev g, const float* __restrict__ kps, const int* __restrict__ order, int m, float r2,
                                                       const float* __restrict__ Hs /* [g.n + 1][HP], last row zero */, float* __restrict__ out) {
    __shared__ float4 ring[FRING];
    __shared__ float fpl[FT * FP_PITCH];
    const int l = threadIdx.x, i = l & 15, k = l >> 4;
    const int tile = lgr_xcd_tile(blockIdx.x, cdiv_dev(m, FT));   // one contiguous range of the (cell, Morton) order per XCD (lgr_grid.cuh)
    if (tile * FT >= m) return;
    const int r = tile * FT + i;
    const float nanv = __uint_as_float(0x7fc00000u);
    int kp = -1;
    float x = nanv, y = nanv, z = nanv;
    if (r < m) {
        kp = order[r];
        x = kps[(size_t) kp * 12]; y = kps[(size_t) kp * 12 + 1]; z = kps[(size_t) kp * 12 + 2];
    }
    const bool fin = lgr_finite3(x, y, z);
    int cell = -1;   // clamped cell of the key point (the ordering key of fine_keys); -1: no neighbours at all
    if (fin && g.n > 0) {
        int cx = min(max(lgr_cellc(x, g.ox, g.h), 0), g.dx - 1), cy = min(max(lgr_cellc(y, g.oy, g.h), 0), g.dy - 1), cz = min(max(lgr_cellc(z, g.oz, g.h), 0), g.dz - 1);
        cell = (cz * g.dy + cy) * g.dx + cx;
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    bool found = false;
    const float r2box = r2 * 1.0001f + 1e-30f;
    const float* hload = Hs + 3 * i;   // lane (i, k) supplies column i of every 16-bin tile (spfh_slot: bins i, 16 + i, 32 + i are adjacent); row offsets in 32 bits (lgr_fpfh_dev checks the size)
    // runs of key points with the same cell (the tile is sorted by cell; -1 cells are skipped)
    for (int p = 0; p < FT;) {
        const int c = __builtin_amdgcn_readlane(cell, p);
        const bool active = cell == c;
        const unsigned long long am = __ballot(active) & 0xffffull;
        p += __popcll(am);
        if (c < 0) continue;
        // bounding box of the run's key points
        float bx0 = active ? x : 3.4e38f, bx1 = active ? x : -3.4e38f, by0 = active ? y : 3.4e38f, by1 = active ? y : -3.4e38f;
        float bz0 = active ? z : 3.4e38f, bz1 = active ? z : -3.4e38f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            bx0 = fminf(bx0, __shfl_xor(bx0, o)); bx1 = fmaxf(bx1, __shfl_xor(bx1, o));
            by0 = fminf(by0, __shfl_xor(by0, o)); by1 = fmaxf(by1, __shfl_xor(by1, o));
            bz0 = fminf(bz0, __shfl_xor(bz0, o)); bz1 = fmaxf(bz1, __shfl_xor(bz1, o));
        }
        // (the same in every lane after the reduction: kept in scalar registers, six vector registers fewer across the loops below)
        auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
        bx0 = uni(bx0); bx1 = uni(bx1); by0 = uni(by0); by1 = uni(by1); bz0 = uni(bz0); bz1 = uni(bz1);
        const int cz = c / (g.dx * g.dy), cy = (c / g.dx) % g.dy, cx = c % g.dx;
        int head = 0, tail = 0;   // ring positions (monotone; entry e lives at e & (FRING - 1))
        // A round of G groups (4 G candidates): the ring entries are read together, the SPFH loads of the round (one 12-byte load per lane and
        // group) are in flight together while the squared distances (lgr_dist2's operations, two candidates per packed instruction) and the
        // reciprocals (lgr_rcp2) are computed, and the groups' MFMAs follow each other in candidate order -- the same products in the same
        // order as one group at a time.  Every group is loaded and multiplied, used or not: a candidate no key point of the run accepts has
        // weight +0 in all its lanes and leaves the accumulators as they are (the skip cost a ballot and a branch per group to save ~10 % of
        // the loads).  (History: one group per round left the MFMAs waiting for a dependent L2 round trip each time; the IEEE division and a
        // 64-bit multiply per address were 40 % of the loop's vector instructions.)
        auto round = [&](auto GC) {
            constexpr int G = decltype(GC)::value;
            float4 e[G];
            f32x3 hv[G];
#pragma unroll
            for (int u = 0; u < G; ++u) e[u] = ring[(head + 4 * u + k) & (FRING - 1)];
#pragma unroll
            for (int u = 0; u < G; ++u) hv[u] = *reinterpret_cast<const f32x3*>(hload + (unsigned) __float_as_int(e[u].w));   // (the ring holds the row offsets)
            const v2f X{x, x}, Y{y, y}, Z{z, z};
            float d2[G], w[G];
#pragma unroll
            for (int h2 = 0; h2 < G / 2; ++h2) {
                const v2f dx = X - v2f{e[2 * h2].x, e[2 * h2 + 1].x}, dy = Y - v2f{e[2 * h2].y, e[2 * h2 + 1].y}, dz = Z - v2f{e[2 * h2].z, e[2 * h2 + 1].z};
                const v2f dd = (dx * dx + dy * dy) + dz * dz;   // lgr_dist2
                const v2f ww = lgr_rcp2(dd);
                d2[2 * h2] = dd.x; d2[2 * h2 + 1] = dd.y; w[2 * h2] = ww.x; w[2 * h2 + 1] = ww.y;
            }
            bool odd = false;
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const bool in = active & (d2[u] < r2);
                found = found | in;
                const bool use = in & (d2[u] != 0.f);
                odd = odd | (use & !(d2[u] >= LGR_RCP_LO));
                w[u] = use ? w[u] : 0.f;
            }
            if (__ballot(odd) != 0ull) {   // (never on real clouds: a squared distance below 1e-36 -- the division, as the canonical definition has it)
#pragma unroll
                for (int u = 0; u < G; ++u) w[u] = (active & (d2[u] < r2) & (d2[u] != 0.f)) ? 1.0f / d2[u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u], hv[u].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u], hv[u].y, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u], hv[u].z, acc2, 0, 0, 0);
            }
            head += 4 * G;
        };
        auto consume = [&](int n_groups) {
            int gq = 0;
            for (; gq + 4 <= n_groups; gq += 4) round(std::integral_constant<int, 4>{});
            for (; gq < n_groups; ++gq, head += 4) {
                const float4 e = ring[(head + k) & (FRING - 1)];
                const float d2 = lgr_dist2(x, y, z, e.x, e.y, e.z);
                const bool in = active & (d2 < r2);
                found = found | in;
                const bool use = in & (d2 != 0.f);
                if (__ballot(use) == 0ull) continue;
                float w = lgr_rcp1(d2);
                if (__ballot(use & !(d2 >= LGR_RCP_LO)) != 0ull) w = 1.0f / d2;
                w = use ? w : 0.f;
                const f32x3 hv = *reinterpret_cast<const f32x3*>(hload + (unsigned) __float_as_int(e.w));
                const float b0 = hv.x, b1 = hv.y, b2 = hv.z;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b2, acc2, 0, 0, 0);
            }
        };
        for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.dz - 1); ++zz)
            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.dy - 1); ++yy) {
                const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
                const size_t c0 = ((size_t) zz * g.dy + yy) * g.dx;
                const int b = g.cell_start[c0 + x0], e = g.cell_start[c0 + x1 + 1];   // three x-cells are contiguous
                for (int t0 = b; t0 < e; t0 += 64) {
                    const int t = t0 + l;
                    bool live = false;
                    float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (t < e) {
                        P = g.pxyz[t];
                        const float ddx = fmaxf(fmaxf(bx0 - P.x, P.x - bx1), 0.f), ddy = fmaxf(fmaxf(by0 - P.y, P.y - by1), 0.f), ddz = fmaxf(fmaxf(bz0 - P.z, P.z - bz1), 0.f);
                        live = (ddx * ddx + ddy * ddy) + ddz * ddz <= r2box;
                    }
                    const unsigned long long lm = __ballot(live);
                    if (lm == 0ull) continue;
                    if (live) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned) (lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) lm, 0u));
                        ring[(tail + rank) & (FRING - 1)] = make_float4(P.x, P.y, P.z, __uint_as_float((unsigned) t * (unsigned) HP));   // .w: element offset of the SPFH row
                    }
                    tail += __popcll(lm);
                    __syncthreads();   // one wave per workgroup: orders the LDS stores before the reads below
                    consume((tail - head) >> 2);
                    __syncthreads();   // ... and those reads before the next chunk's stores
                }
            }
        // the last 1..3 candidates of the run: pad the group with the all-zero SPFH row (position g.n) far away
        if (tail > head) {
            if (l < 4 - (tail - head)) ring[(tail + l) & (FRING - 1)] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __uint_as_float((unsigned) g.n * (unsigned) HP));
            __syncthreads();
            consume(1);
            __syncthreads();
        }
    }
    // The last MFMAs were issued in another basic block (end of the run loop): hipcc's hazard recognizer does not carry their
    // "XDL write -> VALU read" wait states across the branches in between, and the v_accvgpr_read below then returns the
    // accumulators WITHOUT the last group's products (seen on gfx950 / ROCm 7.2: every row off by a few ulp after normalisation;
    // any build that happened to put more instructions in between was exact).  18 wait states cover an 8-pass MFMA.
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // key point i found a neighbour in any of its four candidate slots?
    const unsigned long long fm = __ballot(found);
    // accumulators -> LDS: lane l holds rows 4 * (l >> 4) + q, column l & 15 of every tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 4 * k + q;
        fpl[row * FP_PITCH + i] = acc0[q];
        fpl[row * FP_PITCH + 16 + i] = acc1[q];
        if (i == 0) fpl[row * FP_PITCH + 32] = acc2[q];
    }
    __syncthreads();
    if (l < 3 * FT) {
        const int kq = l / 3, blk = l % 3;
        const int rr = tile * FT + kq;
        if (rr < m) {
            const bool any = ((fm >> kq) | (fm >> (kq + 16)) | (fm >> (kq + 32)) | (fm >> (kq + 48))) & 1ull;
            float* o = out + (size_t) order[rr] * 33 + 11 * blk;
            const float* f = fpl + kq * FP_PITCH + 11 * blk;
            if (!any) {
#pragma unroll
                for (int q = 0; q < 11; ++q) o[q] = nanv;
            } else {
                double sum = 0.0;
#pragma unroll
                for (int q = 0; q < 11; ++q) sum += (double) f[q];
                if (sum != 0) sum = 100.0 / sum;
#pragma unroll
                for (int q = 0; q < 11; ++q) o[q] = (float) ((double) f[q] * sum);
            }
        }
    }
}

// ---- FPFH weighting exactly as PCL writes it (lgr_ctx_options.arithmetic = LGR_ARITH_PCL) ----
// pcl::FPFHEstimation::weightPointSPFHSignature [3P, PCL 1.12.1 features/impl/fpfh.hpp; call site include/common.h:326-331]: the neighbours
// of a key point in the order radiusSearch returns them -- ascending squared distance (ties: index, the oracle's rule; FLANN leaves them
// open) --, per neighbour and bin  val = hist * (1.0f / d2)  ROUNDED to float,  fpfh[bin] += val  in float,  sum_block += val  in double, and
// at the end fpfh[bin] * (100.0 / sum_block) in double, stored as float.  The default mode's fused chain in grid order (fpfh_mfma_kernel)
// differs from this in rounding only (all three pieces measured alone: profiles/r5_pcl_order_by_piece_1M.json); this kernel restates
// oracle orc_fpfh under ORC_ARITH_PCL bit for bit.  One wave per key point:
//   gather  every candidate of the 27 cells with d2 < r2 as (bits(d2) << 32 | original index, sorted position) into LDS, up to PW_CAP;
//           a key point with more neighbours than that is processed in SHELLS of ascending key ranges [lo, hi) found by bisection on the
//           64-bit keys (any neighbour count works, the order of processing is the same);
//   sort    bitonic, ascending keys;
//   weigh   lane b < 33 owns bin b: val, float add and a double sum of its own vals, neighbour by neighbour in sorted order.
// The block sums: PCL adds the 11 vals of a neighbour, neighbour after neighbour, into one double.  All vals are non-negative floats, i.e.
// multiples of U = the smallest ulp among them; when the total T satisfies T < 2^53 U every partial sum of ANY order is a multiple of U
// below 2^53 U and therefore exact -- all orders give the same double, so the per-lane sums added across the block's lanes ARE PCL's
// sum.  The kernel tracks U and T and checks that condition at the end; when it fails (a neighbour ~1e-6 of the radius away, NaN / inf
// values) the key point is done again in STRICT mode: the vals of a neighbour go through LDS and one lane per block adds them in PCL's
// order.  tests/test_gpu_pcl_arith.py drives both paths and the shells (PW_CAP is a template parameter).
template <int PW_CAP>
__global__ __launch_bounds__(64) void fpfh_pcl_kernel(GridDev g, const float* __restrict__ kps, const int* __restrict__ order, int m, float r2,
                                                      const float* __restrict__ Hs /* [g.n][HP] */, float* __restrict__ out) {
    __shared__ unsigned long long skey[PW_CAP];
    __shared__ unsigned spos[PW_CAP];
    __shared__ float sval[64];
    const int l = threadIdx.x;
    const int tile = lgr_xcd_tile(blockIdx.x, m);
    if (tile >= m) return;
    const int kp = order[tile];
    const float x = kps[(size_t) kp * 12], y = kps[(size_t) kp * 12 + 1], z = kps[(size_t) kp * 12 + 2];
    const float nanv = __uint_as_float(0x7fc00000u);
    float* o = out + (size_t) kp * 33;
    if (!lgr_finite3(x, y, z) || g.n == 0) {
        if (l < 33) o[l] = nanv;
        return;
    }
    const int cx = min(max(lgr_cellc(x, g.ox, g.h), 0), g.dx - 1), cy = min(max(lgr_cellc(y, g.oy, g.h), 0), g.dy - 1), cz = min(max(lgr_cellc(z, g.oz, g.h), 0), g.dz - 1);
    const unsigned long long key_end = (unsigned long long) __float_as_uint(r2) << 32;   // keys of neighbours are below it (d2 < r2, both non-negative)
    // one pass over the candidates: count the neighbours with lo <= key < hi and (store) keep the first PW_CAP of them
    auto scan = [&](unsigned long long lo, unsigned long long hi, bool store) -> int {
        int cnt = 0;
        for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.dz - 1); ++zz)
            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.dy - 1); ++yy) {
                const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
                const size_t c0 = ((size_t) zz * g.dy + yy) * g.dx;
                const int b = g.cell_start[c0 + x0], e = g.cell_start[c0 + x1 + 1];
                for (int t0 = b; t0 < e; t0 += 64) {
                    const int t = t0 + l;
                    bool in = false;
                    unsigned long long key = 0ull;
                    if (t < e) {
                        const float4 P = g.pxyz[t];
                        const float d2 = lgr_dist2(x, y, z, P.x, P.y, P.z);
                        key = ((unsigned long long) __float_as_uint(d2) << 32) | (unsigned) __float_as_int(P.w);
                        in = d2 < r2 && key >= lo && key < hi;
                    }
                    const unsigned long long bm = __ballot(in);
                    if (bm == 0ull) continue;
                    if (store && in) {
                        const int slot = cnt + __builtin_amdgcn_mbcnt_hi((unsigned) (bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) bm, 0u));
                        if (slot < PW_CAP) { skey[slot] = key; spos[slot] = (unsigned) t; }
                    }
                    cnt += __popcll(bm);
                }
            }
        return cnt;
    };
    const int bin = min(l, 32);
    const float* hcol = Hs + spfh_slot(bin);
    for (int strict = 0; strict < 2; ++strict) {
        float fp = 0.f;
        double dsum = 0.0;          // strict: lanes 0..2 hold the block sums; otherwise every bin lane holds the double sum of its own vals
        int emin = 255;             // smallest biased exponent among the nonzero vals of this lane (0 counted as 1: denormals share the ulp of the first binade)
        bool bad = false;           // a val that is NaN, inf or negative: the exactness argument does not apply
        int total = 0;
        unsigned long long lo = 0ull;
        while (lo < key_end) {
            unsigned long long hi = key_end;
            __syncthreads();        // the previous shell's reads of skey / spos are done
            int cnt = scan(lo, hi, true);
            if (cnt > PW_CAP) {
                // more neighbours than the buffer holds: a shell [lo, hi) with PW_CAP / 4 <= count <= PW_CAP (keys are unique, so a one-key range holds <= 1)
                unsigned long long a = lo, bnd = hi;   // count(lo, a) <= PW_CAP, count(lo, bnd) > PW_CAP
                for (;;) {
                    const unsigned long long mid = a + ((bnd - a) >> 1);
                    const int c = scan(lo, mid, false);
                    if (c > PW_CAP) bnd = mid;
                    else { a = mid; if (c >= PW_CAP / 4 || bnd - a <= 1ull) break; }
                }
                hi = a;
                __syncthreads();
                cnt = scan(lo, hi, true);
            }
            total += cnt;
            // bitonic sort of skey[0, n_pad) with spos as payload
            int n_pad = 64;
            while (n_pad < cnt) n_pad <<= 1;
            for (int q = cnt + l; q < n_pad; q += 64) { skey[q] = ~0ull; spos[q] = 0u; }
            __syncthreads();
            for (int k2 = 2; k2 <= n_pad; k2 <<= 1)
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    for (int q = l; q < (n_pad >> 1); q += 64) {
                        const int i0 = ((q & ~(j - 1)) << 1) | (q & (j - 1)), i1 = i0 | j;   // the q-th pair of this step
                        const unsigned long long ka = skey[i0], kb = skey[i1];
                        const bool up = (i0 & k2) == 0;
                        if ((ka > kb) == up) {
                            skey[i0] = kb; skey[i1] = ka;
                            const unsigned pa = spos[i0]; spos[i0] = spos[i1]; spos[i1] = pa;
                        }
                    }
                    __syncthreads();
                }
            // entries -> (bits(1.0f / d2) << 32 | sorted position * HP); weight 0 marks the key point itself (d2 == 0: skipped)
            for (int q = l; q < cnt; q += 64) {
                const float d2 = __uint_as_float((unsigned) (skey[q] >> 32));
                const float w = d2 == 0.f ? 0.f : 1.0f / d2;
                skey[q] = ((unsigned long long) __float_as_uint(w) << 32) | (spos[q] * (unsigned) HP);
            }
            __syncthreads();
            if (!strict) {
                int q = 0;
                for (; q + 4 <= cnt; q += 4) {   // four SPFH loads in flight; the adds in neighbour order
                    const unsigned long long e0 = skey[q], e1 = skey[q + 1], e2 = skey[q + 2], e3 = skey[q + 3];
                    const float h0 = hcol[(unsigned) e0], h1 = hcol[(unsigned) e1], h2 = hcol[(unsigned) e2], h3 = hcol[(unsigned) e3];
                    const float w0 = __uint_as_float((unsigned) (e0 >> 32)), w1 = __uint_as_float((unsigned) (e1 >> 32)), w2 = __uint_as_float((unsigned) (e2 >> 32)), w3 = __uint_as_float((unsigned) (e3 >> 32));
                    const float v[4] = {h0 * w0, h1 * w1, h2 * w2, h3 * w3};
                    const float ws[4] = {w0, w1, w2, w3};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ws[u] == 0.f) continue;   // (wave-uniform)
                        fp += v[u];
                        dsum += (double) v[u];
                        const int eb = (int) ((__float_as_uint(v[u]) >> 23) & 0xffu);
                        bad = bad || !(v[u] >= 0.f) || eb == 255;
                        if (v[u] != 0.f) emin = min(emin, max(eb, 1));
                    }
                }
                for (; q < cnt; ++q) {
                    const unsigned long long e0 = skey[q];
                    const float w0 = __uint_as_float((unsigned) (e0 >> 32));
                    if (w0 == 0.f) continue;
                    const float v0 = hcol[(unsigned) e0] * w0;
                    fp += v0;
                    dsum += (double) v0;
                    const int eb = (int) ((__float_as_uint(v0) >> 23) & 0xffu);
                    bad = bad || !(v0 >= 0.f) || eb == 255;
                    if (v0 != 0.f) emin = min(emin, max(eb, 1));
                }
            } else {
                for (int q = 0; q < cnt; ++q) {
                    const unsigned long long e0 = skey[q];
                    const float w0 = __uint_as_float((unsigned) (e0 >> 32));
                    if (w0 == 0.f) continue;
                    const float v0 = hcol[(unsigned) e0] * w0;
                    fp += v0;
                    sval[l] = v0;
                    __syncthreads();
                    if (l < 3) {
#pragma unroll
                        for (int u = 0; u < 11; ++u) dsum += (double) sval[11 * l + u];   // sum_f += val_f, bin after bin
                    }
                    __syncthreads();
                }
            }
            lo = hi;
        }
        if (total == 0) {           // no neighbour at all (the key point itself counts as one): a NaN row
            if (l < 33) o[l] = nanv;
            return;
        }
        // the block sums in the lanes of their bins
        double bsum;
        if (!strict) {
            // exactness of every partial sum of every order: T < 2^53 U, U = 2^(emin - 150), T <= the sum of all lane sums (+ slack for their own rounding)
            double t_all = l < 33 ? dsum : 0.0;
            int e_all = l < 33 ? emin : 255;
            bool b_all = l < 33 && bad;
#pragma unroll
            for (int s2 = 1; s2 < 64; s2 <<= 1) {
                t_all += __shfl_xor(t_all, s2);
                e_all = min(e_all, __shfl_xor(e_all, s2));
            }
            b_all = __ballot(b_all) != 0ull;
            const bool exact_here = !b_all && (e_all == 255 || t_all * 1.000001 < ldexp(1.0, 53 + e_all - 150));
            if (__ballot(!exact_here) != 0ull) continue;   // (one decision per wave) again, in PCL's literal order
            // sum of the block's 11 lane sums (exact, any order): lanes 11 blk .. 11 blk + 10
            const int blk = bin / 11;
            bsum = 0.0;
#pragma unroll
            for (int u = 0; u < 11; ++u) bsum += __shfl(dsum, 11 * blk + u);
        } else {
            bsum = __shfl(dsum, bin / 11);
        }
        if (bsum != 0) bsum = 100.0 / bsum;
        if (l < 33) o[l] = (float) ((double) fp * bsum);
        return;
    }
}

// Processing order of key points / surface points: grid cell, then a Morton code of the position inside the cell
// (2^sb steps per axis).  A wave then holds 64 spatially close points: its lanes walk the same 27 cells in lockstep
// (wave-uniform loads) and mostly agree on which candidates lie within the radius, so fewer lanes idle through the
// accept branch and candidates nobody accepts are skipped by the whole wave.  Scheduling only: every point still visits
// its neighbours in the canonical order, and outputs go to the original index.
__global__ void fine_keys(GridDev g, const float* __restrict__ pts, int stride, int m, int sb, unsigned* __restrict__ keys, int* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float x = pts[(size_t) i * stride], y = pts[(size_t) i * stride + 1], z = pts[(size_t) i * stride + 2];
    unsigned k = 0xffffffffu;
    if (lgr_finite3(x, y, z)) {
        int cx = min(max(lgr_cellc(x, g.ox, g.h), 0), g.dx - 1), cy = min(max(lgr_cellc(y, g.oy, g.h), 0), g.dy - 1), cz = min(max(lgr_cellc(z, g.oz, g.h), 0), g.dz - 1);
        k = (unsigned) ((cz * g.dy + cy) * g.dx + cx);
        if (sb > 0) {
            const float q = (float) (1 << sb);
            int sx = min(max((int) (((x - g.ox) / g.h - (float) cx) * q), 0), (1 << sb) - 1);
            int sy = min(max((int) (((y - g.oy) / g.h - (float) cy) * q), 0), (1 << sb) - 1);
            int sz = min(max((int) (((z - g.oz) / g.h - (float) cz) * q), 0), (1 << sb) - 1);
            unsigned mort = 0;
            for (int b = 0; b < sb; ++b) mort |= (((unsigned) sx >> b) & 1u) << (3 * b) | (((unsigned) sy >> b) & 1u) << (3 * b + 1) | (((unsigned) sz >> b) & 1u) << (3 * b + 2);
            k = (k << (3 * sb)) | mort;
        }
    }
    keys[i] = k; vals[i] = i;
}

struct Key3 { int x, y, z; bool operator==(const Key3& o) const { return x == o.x && y == o.y && z == o.z; } };
// include/common.h:212-223 HashEigen<Eigen::Vector3i> (std::hash<int> is the identity)
struct HashKey3 {
    std::size_t operator()(const Key3& k) const {
        std::size_t seed = 0;
        const int e[3] = {k.x, k.y, k.z};
        for (int i = 0; i < 3; ++i) seed ^= std::hash<int>()(e[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};

}  // namespace

// device pipeline; n_out on host.  keys_first (optional, host vectors) receive per-voxel key and first input index.
static int downsample_core(lgr_ctx* ctx, const float* d_pts, int n, float voxel, float* d_out, int* n_out,
                           std::vector<unsigned long long>* h_keys, std::vector<int>* h_first) {
    LGR_CHECK(ctx, voxel > 0.f, LGR_ERR_INVALID_ARG);   // reference only prints an error; a non-positive voxel is meaningless
    *n_out = 0;
    if (n == 0) return LGR_OK;
    float bb[12];
    LGR_TRY(lgr_bbox_host(ctx, d_pts, n, bb));
    // voxel_min_bound = min_point_AABB - voxel_size * 0.5  (quirk bbox, src/downsample.cpp:13-14)
    float bx = bb[6] - voxel * 0.5f, by = bb[7] - voxel * 0.5f, bz = bb[8] - voxel * 0.5f;
    unsigned long long *keys, *keys2;
    int *vals, *vals2, *flags, *rank, *bad;
    LGR_TRY(lgr_ws_t(ctx, WS_DS_KEYS, (size_t) n, &keys));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_KEYS2, (size_t) n, &keys2));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_VALS, (size_t) n, &vals));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_VALS2, (size_t) n, &vals2));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_FLAGS, (size_t) 2 * n + 16, &flags));
    rank = flags + n;
    bad = rank + n;
    LGR_HIP(ctx, hipMemsetAsync(bad, 0, 4, ctx->stream));
    voxel_keys<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_pts, n, bx, by, bz, voxel, keys, vals, bad);
    size_t tb2 = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb2, flags, rank, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb2, &tmp));
    {
        // only the bits a voxel index of this cloud can set are sorted: index <= floor((max - bound) / voxel), the kernel's own
        // expression on the largest coordinate (float subtraction, division and floor are monotone).  The z field is sorted one
        // bit wider: that bit is 0 in every valid key and 1 in the all-ones key of an invalid point, which therefore sorts last.
        const float bnd[3] = {bx, by, bz};
        int shifts[3] = {0, 21, 42}, widths[3];
        for (int a = 0; a < 3; ++a) {
            const float q = std::floor((bb[9 + a] - bnd[a]) / voxel);
            long long imax = (q >= 0.f && q < 2097152.f) ? (long long) q : 2097151;
            int b = 1;
            while ((1ll << b) <= imax) ++b;
            widths[a] = std::min(21, b);
        }
        widths[2] = std::min(22, widths[2] + 1);
        LGR_TRY(lgr_sort_pairs_u64(ctx, keys, keys2, vals, vals2, (size_t) n, shifts, widths, 3));
    }
    head_flags<<<cdiv(n, 256), 256, 0, ctx->stream>>>(keys2, n, flags);
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb2, flags, rank, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, rank + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, flags + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 2, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int nv = h[0] + h[1];
    if (h[2]) return lgr_fail(ctx, LGR_ERR_VOXEL_TOO_SMALL, "voxel index exceeds 21 bits per axis", __FILE__, __LINE__);
    unsigned long long* okeys = nullptr;
    int* ofirst = nullptr;
    if (h_keys) {
        LGR_TRY(lgr_ws_t(ctx, WS_DS_MISC, (size_t) nv * 3 + 4, (int**) &ofirst));
        okeys = (unsigned long long*) (ofirst + ((nv + 1) & ~1));
    }
    // out may alias the input: accumulate into a staging buffer first when it does
    float* stage = d_out;
    bool alias = d_out == d_pts;
    if (alias) LGR_TRY(lgr_ws_t(ctx, WS_HOST_F, (size_t) nv * 12 + 4, &stage));
    voxel_accumulate<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_pts, keys2, vals2, flags, rank, n, stage, okeys, ofirst);
    if (alias) LGR_HIP(ctx, hipMemcpyAsync(d_out, stage, (size_t) nv * 48, hipMemcpyDeviceToDevice, ctx->stream));
    if (h_keys) {
        h_keys->resize(nv); h_first->resize(nv);
        LGR_HIP(ctx, hipMemcpyAsync(h_keys->data(), okeys, (size_t) nv * 8, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(h_first->data(), ofirst, (size_t) nv * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    LGR_HIP(ctx, hipGetLastError());
    *n_out = nv;
    return LGR_OK;
}

extern "C" int lgr_downsample_dev(lgr_ctx* ctx, const float* d_pts, int n, float voxel, float* d_out, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_pts || n == 0) && (d_out || n == 0) && n_out && n >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    return downsample_core(ctx, d_pts, n, voxel, d_out, n_out, nullptr, nullptr);
}

extern "C" int lgr_downsample(lgr_ctx* ctx, const float* pts, int n, float voxel, int order, float* out, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (pts || n == 0) && out && n_out && n >= 0, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, order == LGR_ORDER_REFERENCE || order == LGR_ORDER_CANONICAL, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    if (n == 0) return LGR_OK;
    float *dp, *dout;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) n * 12, &dp));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) n * 12, &dout));
    LGR_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t) n * 48, hipMemcpyHostToDevice, ctx->stream));
    std::vector<unsigned long long> keys;
    std::vector<int> first;
    int nv = 0;
    LGR_TRY(downsample_core(ctx, dp, n, voxel, dout, &nv, order == LGR_ORDER_REFERENCE ? &keys : nullptr, &first));
    if (order == LGR_ORDER_CANONICAL) {
        LGR_HIP(ctx, hipMemcpyAsync(out, dout, (size_t) nv * 48, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        // reference order = iteration order of std::unordered_map<Vector3i, AccumulatedPoint, HashEigen>
        // (src/downsample.cpp:20,32-34).  Replay the insertion sequence (voxels by first input index) through
        // the very same container and walk it; the values were computed on the device.
        std::vector<float> tmp((size_t) nv * 12);
        LGR_HIP(ctx, hipMemcpyAsync(tmp.data(), dout, (size_t) nv * 48, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int> by_first(nv);
        for (int i = 0; i < nv; ++i) by_first[i] = i;
        std::sort(by_first.begin(), by_first.end(), [&](int a, int b) { return first[a] < first[b]; });
        std::unordered_map<Key3, int, HashKey3> m;
        for (int v : by_first) {
            unsigned long long k = keys[v];
            Key3 k3{(int) (k & 0x1fffff), (int) ((k >> 21) & 0x1fffff), (int) ((k >> 42) & 0x1fffff)};
            m[k3] = v;
        }
        size_t o = 0;
        for (const auto& kv : m) { memcpy(out + 12 * o, tmp.data() + 12 * (size_t) kv.second, 48); ++o; }
    }
    *n_out = nv;
    return LGR_OK;
}

// lgr.h: every float of [lo_bits, hi_bits] through the weighting kernel's reciprocal (lgr_rcp1 / lgr_rcp2) and through the division
extern "C" int lgr_selfcheck_rcp(lgr_ctx* ctx, unsigned lo_bits, unsigned hi_bits, unsigned long long* out2) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, out2 && lo_bits <= hi_bits, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long* d;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) 2, &d));
    LGR_HIP(ctx, hipMemsetAsync(d, 0, 16, ctx->stream));
    rcp_check_kernel<<<8 * ctx->n_cu, 256, 0, ctx->stream>>>(lo_bits, hi_bits, d);
    LGR_HIP(ctx, hipGetLastError());
    LGR_HIP(ctx, hipMemcpyAsync(out2, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}

// lgr.h: lgr_libm.cuh element-wise on the device (fn 0 acosf(a), 1 atanf(a), 2 atan2f(a, b), 3 sinf(a), 4 cosf(a)); host arrays
__global__ void libm_eval_kernel(int fn, const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ out) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const float x = a[i];
        float r;
        switch (fn) {
            case 0: r = lgr_glibc::acosf_(x); break;
            case 1: r = lgr_glibc::atanf_(x); break;
            case 2: r = lgr_glibc::atan2f_(x, b[i]); break;
            case 3: r = lgr_glibc::sinf_(x); break;
            default: r = lgr_glibc::cosf_(x); break;
        }
        out[i] = r;
    }
}
extern "C" int lgr_selfcheck_libm(lgr_ctx* ctx, int fn, const float* a, const float* b, long long n, float* out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, fn >= 0 && fn <= 4 && a && out && n >= 0 && (fn != 2 || b), LGR_ERR_INVALID_ARG);
    if (n == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *da, *db, *dout;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) n, &da));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) n, &db));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) n, &dout));
    LGR_HIP(ctx, hipMemcpyAsync(da, a, (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (fn == 2) LGR_HIP(ctx, hipMemcpyAsync(db, b, (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    libm_eval_kernel<<<8 * ctx->n_cu, 256, 0, ctx->stream>>>(fn, da, db, n, dout);
    LGR_HIP(ctx, hipGetLastError());
    LGR_HIP(ctx, hipMemcpyAsync(out, dout, (size_t) n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}

#ifdef LGR_SPFH_CHECK
// diagnostics of the -DLGR_SPFH_CHECK build (tools/exp_spfh_check.py): {pairs, pairs the filter left undecided, decided pairs whose bins differ}
extern "C" int lgr_debug_spfh_check(unsigned long long* out3, int reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out3, HIP_SYMBOL(g_spfh_check), 24) != hipSuccess) return LGR_ERR_HIP;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_spfh_check), z, 32) != hipSuccess) return LGR_ERR_HIP;
    return LGR_OK;
}
#endif

extern "C" int lgr_normals_knn_dev(lgr_ctx* ctx, float* d_pts, int n, const float* d_surf, int ns, int k, const float* vp3, int normals_available) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_pts || n == 0) && n >= 0 && k >= 1 && k <= 64, LGR_ERR_INVALID_ARG);
    (void) normals_available;   // no observable effect in the reference (src/common.cpp:597-598 compares a point with itself)
    if (n == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    const float* S = d_surf;
    if (!S) {
        // surface = the cloud itself: normals are written in place, so query a snapshot
        float* snap;
        LGR_TRY(lgr_ws_t(ctx, WS_HOST_F, (size_t) n * 12, &snap));
        LGR_HIP(ctx, hipMemcpyAsync(snap, d_pts, (size_t) n * 48, hipMemcpyDeviceToDevice, ctx->stream));
        S = snap; ns = n;
    }
    GridDev g;
    // cell size: about 6 points per cell for the per-thread heaps (measured at 1M, k = 30: 2 / 3 / 4 / 6 / 8 / 12 / 16 points per cell -> 6.2 / 5.3 /
    // 4.8 / 4.1 / 4.0 / 4.1 / 4.0 ms for the two clouds' normals, the later stages slowing down slightly from 8 on), about 10 for the wave
    // search (6 / 8 / 10 / 12 / 14 -> normals stage 1.87 / 1.61 / 1.56 / 1.63 / 1.64 ms; more queries share a cell's candidate set, at more
    // candidates per scan).  The neighbours do not depend on it.
    const bool wave = !d_surf && k >= 16;
    const float ppc = wave ? 10.f : 6.f;
    LGR_TRY(lgr_grid_build(ctx, WS_GRID_A, S, ns, 0.f, ppc, &g));
    size_t sm = (size_t) k * NB * 8;
    const float vx = vp3 ? vp3[0] : 0.f, vy = vp3 ? vp3[1] : 0.f, vz = vp3 ? vp3[2] : 0.f;
    if (wave) {
        // (below k = 16 the per-thread heaps win: a wave per query leaves most of its lanes without a candidate)
        const float r2i = g.h * g.h * 1.25f * (float) k / (3.14159265f * ppc);
        const int grid = lgr_xcd_grid(cdiv(g.n, 64 * NW_WAVES));
        const size_t sml = (size_t) NW_WAVES * k * 64 * sizeof(int);
        if (g.n > 0) {
            if (k <= 40) normals_wave_kernel<1><<<grid, 64 * NW_WAVES, sml, ctx->stream>>>(g, S, d_pts, k, vx, vy, vz, r2i);
            else {
                // k >= 58: the index lists (k KB) + the static buffers pass 64 KB per workgroup, which a launch only gets when the kernel says so
                if (sml + 8192 > 65536) LGR_HIP(ctx, hipFuncSetAttribute((const void*) normals_wave_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sml));
                normals_wave_kernel<2><<<grid, 64 * NW_WAVES, sml, ctx->stream>>>(g, S, d_pts, k, vx, vy, vz, r2i);
            }
        }
        if (g.n < n) normals_kernel<<<cdiv(n, NB), NB, sm, ctx->stream>>>(g, S, d_pts, n, k, vx, vy, vz, 2);   // non-finite points: NaN normals
    } else if (!d_surf) {
        if (g.n > 0) normals_kernel<<<lgr_xcd_grid(cdiv(g.n, NB)), NB, sm, ctx->stream>>>(g, S, d_pts, n, k, vx, vy, vz, 1);
        if (g.n < n) normals_kernel<<<cdiv(n, NB), NB, sm, ctx->stream>>>(g, S, d_pts, n, k, vx, vy, vz, 2);   // non-finite points: NaN normals
    } else {
        normals_kernel<<<cdiv(n, NB), NB, sm, ctx->stream>>>(g, S, d_pts, n, k, vx, vy, vz, 0);
    }
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

extern "C" int lgr_normals_knn(lgr_ctx* ctx, float* pts, int n, const float* surf, int ns, int k, const float* vp3, int normals_available) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (pts || n == 0) && n >= 0, LGR_ERR_INVALID_ARG);
    if (n == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *dp, *ds = nullptr;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) n * 12, &dp));
    LGR_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t) n * 48, hipMemcpyHostToDevice, ctx->stream));
    if (surf) {
        LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) ns * 12 + 4, &ds));
        LGR_HIP(ctx, hipMemcpyAsync(ds, surf, (size_t) ns * 48, hipMemcpyHostToDevice, ctx->stream));
    }
    LGR_TRY(lgr_normals_knn_dev(ctx, dp, n, ds, ns, k, vp3, normals_available));
    LGR_HIP(ctx, hipMemcpyAsync(pts, dp, (size_t) n * 48, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}

extern "C" int lgr_fpfh_dev(lgr_ctx* ctx, const float* d_kps, int m, const float* d_surf, int n, float radius, float* d_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_kps || m == 0) && (d_surf || n == 0) && (d_out || m == 0) && m >= 0 && n >= 0 && radius > 0.f, LGR_ERR_INVALID_ARG);
    // the weighting kernel's reciprocal is checked for squared distances up to 1e36 (lgr_rcp2) and addresses SPFH rows with 32-bit element offsets
    LGR_CHECK(ctx, radius <= 1e18f && (size_t) n + 1 <= ((size_t) 1 << 32) / HP - 1, LGR_ERR_UNSUPPORTED);
    if (m == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    GridDev g;
    LGR_TRY(lgr_grid_build(ctx, WS_GRID_B, d_surf, n, radius * 1.001f, 0.f, &g));
    float r2 = radius * radius;
    float* spfh;   // [g.n + 1][HP] in the grid's sorted order; row g.n is all zero (padding candidates of the MFMA weighting)
    LGR_TRY(lgr_ws_t(ctx, WS_SPFH, ((size_t) n + 1) * HP + 64, &spfh));
    LGR_HIP(ctx, hipMemsetAsync(spfh + (size_t) g.n * HP, 0, HP * sizeof(float), ctx->stream));
    // PCL computes SPFH only for surface points within r of some keypoint (spfh_indices); rows outside that set are
    // never read by the weighting step, so computing all rows gives the same FPFH output and saves the marking pass.
    // processing orders (fine_keys): cell bits + 3 * sb Morton bits must fit the 32-bit sort key
    const long long n_cells = (long long) g.dx * g.dy * g.dz;
    int cell_bits = 1;
    while ((1ll << cell_bits) < n_cells) ++cell_bits;
    const int sb = std::max(0, std::min(2, (32 - cell_bits) / 3));
    const int key_bits = std::min(32, cell_bits + 3 * sb);
    const int mx = std::max(m, g.n);
    // Both processing orders are sorted BEFORE the SPFH kernel: the weighting kernel then follows it directly in the stream.  (With the key
    // points' sort between the two, its short launches starved behind the OTHER cloud's SPFH kernel -- 2 ms for a 13 us scatter -- and the
    // two clouds' big kernels ran strictly one after the other; now the latency-bound weighting of one cloud shares the device with the
    // VALU-bound SPFH of the other.)
    unsigned *keys, *keys2, *kkeys, *kkeys2;
    int *vals, *vals2, *kvals, *kvals2;
    LGR_TRY(lgr_ws_t(ctx, WS_KP_ORDER, (size_t) mx * 8 + 16, &keys));
    keys2 = keys + mx; vals = (int*) (keys2 + mx); vals2 = vals + mx;
    kkeys = (unsigned*) (vals2 + mx); kkeys2 = kkeys + mx; kvals = (int*) (kkeys2 + mx); kvals2 = kvals + mx;
    fine_keys<<<cdiv(m, 256), 256, 0, ctx->stream>>>(g, d_kps, 12, m, sb, kkeys, kvals);
    LGR_TRY(lgr_sort_pairs_u32(ctx, kkeys, kkeys2, kvals, kvals2, (size_t) m, 0, key_bits));
    if (g.n > 0) {
        fine_keys<<<cdiv(g.n, 256), 256, 0, ctx->stream>>>(g, reinterpret_cast<const float*>(g.pxyz), 4, g.n, sb, keys, vals);
        LGR_TRY(lgr_sort_pairs_u32(ctx, keys, keys2, vals, vals2, (size_t) g.n, 0, key_bits));
        spfh_tile_kernel<<<lgr_xcd_grid(cdiv(g.n, ST)), 64, 0, ctx->stream>>>(g, r2, vals2, spfh);
    }
    if (ctx->opt.arithmetic == LGR_ARITH_PCL) {
        // PCL's own weighting order and rounding steps (fpfh_pcl_kernel); pcl_neighbour_cap: neighbours sorted at once (tests shrink it to drive the shells)
        if (ctx->opt.pcl_neighbour_cap == 64) fpfh_pcl_kernel<64><<<lgr_xcd_grid(m), 64, 0, ctx->stream>>>(g, d_kps, kvals2, m, r2, spfh, d_out);
        else if (ctx->opt.pcl_neighbour_cap == 1024) fpfh_pcl_kernel<1024><<<lgr_xcd_grid(m), 64, 0, ctx->stream>>>(g, d_kps, kvals2, m, r2, spfh, d_out);
        else fpfh_pcl_kernel<512><<<lgr_xcd_grid(m), 64, 0, ctx->stream>>>(g, d_kps, kvals2, m, r2, spfh, d_out);   // (default: 6 KB of LDS per wave; ~260 neighbours on average, more go in shells)
    } else {
        fpfh_mfma_kernel<<<lgr_xcd_grid(cdiv(m, FT)), 64, 0, ctx->stream>>>(g, d_kps, kvals2, m, r2, spfh, d_out);
    }
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

extern "C" int lgr_fpfh(lgr_ctx* ctx, const float* kps, int m, const float* surf, int n, float radius, float* out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (kps || m == 0) && (surf || n == 0) && (out || m == 0) && m >= 0 && n >= 0, LGR_ERR_INVALID_ARG);
    if (m == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *dk, *ds, *dout;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) m * 12, &dk));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) n * 12 + 4, &ds));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) m * 33, &dout));
    LGR_HIP(ctx, hipMemcpyAsync(dk, kps, (size_t) m * 48, hipMemcpyHostToDevice, ctx->stream));
    if (n) LGR_HIP(ctx, hipMemcpyAsync(ds, surf, (size_t) n * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_fpfh_dev(ctx, dk, m, ds, n, radius, dout));
    LGR_HIP(ctx, hipMemcpyAsync(out, dout, (size_t) m * 132, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}
