// lgr_gror.hip -- GROR initial alignment (BASELINE config 5) for gfx950.
//
// Replaces alignGror (reference src/alignment.cpp:21-35), i.e. pcl::registration::GRORInitialAlignment::computeTransformation
// (reference include/gror/ia_gror.hpp:367-415) with resolution = distance_thr and K_optimal = 800.
//
// Split of the work:
//   device  node reliability (ia_gror.hpp:126-170): all C^2 edge-length comparisons, one lane per node, the other
//           nodes streamed through LDS; 64-bit radix sort (degree desc, index asc) for the K most reliable nodes
//   host    edge stage over the K <= 800 selected nodes (ia_gror.hpp:82-124,199-259,472-501,620-747): a few 1e5
//           scalar operations with a sequential dependency through best_count_ and double-precision libm calls
//   device  refinement (ia_gror.hpp:261-316): inlier test of every correspondence under the global transform,
//           ordered compaction, Umeyama sums accumulated sequentially in correspondence order by one wave
//
// Canonical tie orders (std::sort leaves them open in the reference): node degree ties by correspondence index,
// graph rows of equal size by node order, interval ends of equal location by insertion order.
#include <algorithm>
#include <chrono>
#include <cmath>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "lgr_internal.h"
#include "lgr_math.cuh"

namespace {

// ------------------------------------------------------------------------------------------------ device: nodes
__global__ void gror_pack_kernel(const float* __restrict__ src, const float* __restrict__ tgt, const lgr_corr* __restrict__ corr, int c,
                                 float4* __restrict__ S, float4* __restrict__ T) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    lgr_corr cr = corr[i];
    const float* s = src + 12 * (size_t) cr.index_query;
    const float* t = tgt + 12 * (size_t) cr.index_match;
    S[i] = make_float4(s[0], s[1], s[2], 0.f);
    T[i] = make_float4(t[0], t[1], t[2], 0.f);
}

__device__ __forceinline__ float edge_len(float4 a, float4 b) {   // pcl::geometry::distance
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
}

constexpr int DEG_THREADS = 256;
constexpr int DEG_TILE = 512;
// degree[i] += #{ j in this block's slice, j != i : | |s_i s_j| - |t_i t_j| | < 2 * resolution }
__global__ __launch_bounds__(DEG_THREADS) void gror_degree_kernel(const float4* __restrict__ S, const float4* __restrict__ T, int c, int slice,
                                                                  float two_res, int* __restrict__ degree) {
    __shared__ float4 Ss[DEG_TILE], Ts[DEG_TILE];
    int i = blockIdx.x * DEG_THREADS + threadIdx.x;
    bool live = i < c;
    float4 si = live ? S[i] : make_float4(0, 0, 0, 0), ti = live ? T[i] : make_float4(0, 0, 0, 0);
    int j0 = blockIdx.y * slice, j1 = min(c, j0 + slice);
    int cnt = 0;
    for (int base = j0; base < j1; base += DEG_TILE) {
        int m = min(DEG_TILE, j1 - base);
        __syncthreads();
        for (int k = threadIdx.x; k < m; k += DEG_THREADS) { Ss[k] = S[base + k]; Ts[k] = T[base + k]; }
        __syncthreads();
        // The decision |  |s_i s_j| - |t_i t_j|  | < 2 r with the hardware's one-instruction square root (1 ulp) wherever that decides it -- the two
        // lengths are within 1.5 ulp of the correctly rounded ones, so the difference is within 4 ulp of the larger length of the reference's: a pair
        // goes through edge_len's correctly rounded square roots (a dozen instructions each) only when it lies within 1e-6 of the larger length of the
        // threshold, which some lane of a wave does for ~1e-4 of the steps (round 5: 3.3 -> ~2 ms for 50 000 correspondences; dead lanes take part:
        // their points are the origin).
        for (int k = 0; k < m; ++k) {
            const float4 sj = Ss[k], tj = Ts[k];
            const float ax = si.x - sj.x, ay = si.y - sj.y, az = si.z - sj.z, bx = ti.x - tj.x, by = ti.y - tj.y, bz = ti.z - tj.z;
            const float la = __builtin_amdgcn_sqrtf((ax * ax + ay * ay) + az * az), lb = __builtin_amdgcn_sqrtf((bx * bx + by * by) + bz * bz);
            const float d = fabsf(la - lb);
            bool in = d < two_res;
            if (__any(!(fabsf(d - two_res) > 1e-6f * fmaxf(la, lb) + 1e-30f)))   // (NaN: taken)
                in = fabsf(edge_len(si, sj) - edge_len(ti, tj)) < two_res;
            cnt += (in && live && base + k != i) ? 1 : 0;
        }
    }
    if (live && cnt) atomicAdd(&degree[i], cnt);
}

__global__ void gror_keys_kernel(const int* __restrict__ degree, int c, unsigned long long* __restrict__ keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) keys[i] = ((unsigned long long) (0xffffffffu - (unsigned) degree[i]) << 32) | (unsigned) i;
}

__global__ void gror_gather_kernel(const unsigned long long* __restrict__ keys, int K, int use_keys, const float4* __restrict__ S,
                                   const float4* __restrict__ T, float4* __restrict__ out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    int i = use_keys ? (int) (keys[k] & 0xffffffffu) : k;
    float4 s = S[i], t = T[i];
    s.w = __int_as_float(i);
    out[2 * k] = s;
    out[2 * k + 1] = t;
}

// ------------------------------------------------------------------------------------------------ device: refine
// pcl::transformPointCloudWithNormals: x*c0 + (y*c1 + (z*c2 + c3)) per output lane; G column-major
__global__ void gror_inlier_kernel(const float4* __restrict__ S, const float4* __restrict__ T, int c, const float* __restrict__ G, float two_res,
                                   int* __restrict__ flags, uint8_t* __restrict__ mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    float4 s = S[i], t = T[i];
    float4 m;
    m.x = G[0] * s.x + (G[4] * s.y + (G[8] * s.z + G[12]));
    m.y = G[1] * s.x + (G[5] * s.y + (G[9] * s.z + G[13]));
    m.z = G[2] * s.x + (G[6] * s.y + (G[10] * s.z + G[14]));
    int in = edge_len(t, m) < two_res ? 1 : 0;
    flags[i] = in;
    if (mask) mask[i] = (uint8_t) in;
}

__global__ void gror_compact_kernel(const float4* __restrict__ S, const float4* __restrict__ T, const int* __restrict__ flags,
                                    const int* __restrict__ pos, int c, float4* __restrict__ Sc, float4* __restrict__ Tc, int* __restrict__ n_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    if (flags[i]) { Sc[pos[i]] = S[i]; Tc[pos[i]] = T[i]; }
    if (i == c - 1) *n_out = pos[i] + flags[i];
}

// pcl::umeyama(src, tgt, false) over the compacted inliers.  Lanes 0..5 carry the six mean sums, lanes 0..8 the nine
// covariance sums, each strictly in correspondence order; lane 0 finishes with the 3x3 SVD.
// (Round 5: the terms come through LDS, component-major, staged by the whole workgroup -- in the second pass as the nine finished products, the same
// subtractions and multiplication by another thread -- so that a summing lane is left with consecutive LDS reads running ahead of its one dependent
// addition per term.  Straight from global memory every term waited for its own load: 1.4 ms for 4 900 inliers.)
constexpr int GU_CH = 1024, GU_THREADS = 256;
__global__ __launch_bounds__(GU_THREADS) void gror_umeyama_kernel(const float4* __restrict__ Sc, const float4* __restrict__ Tc, const int* __restrict__ n_ptr, float* __restrict__ Tout) {
    __shared__ float mean[6], sig[9];
    __shared__ float sp[9 * GU_CH];   // pass 1: [6 components][GU_CH]; pass 2: [9 products][GU_CH]
    const int l = threadIdx.x, n = *n_ptr;
    const float inv_n = 1.0f / (float) n;
    {
        float acc = 0.f;
        for (int i0 = 0; i0 < n; i0 += GU_CH) {
            __syncthreads();
            for (int i = l; i < GU_CH && i0 + i < n; i += GU_THREADS) {
                const float4 s4 = Sc[i0 + i], t4 = Tc[i0 + i];
                sp[0 * GU_CH + i] = s4.x; sp[1 * GU_CH + i] = s4.y; sp[2 * GU_CH + i] = s4.z;
                sp[3 * GU_CH + i] = t4.x; sp[4 * GU_CH + i] = t4.y; sp[5 * GU_CH + i] = t4.z;
            }
            __syncthreads();
            if (l < 6) {
                const int m = min(GU_CH, n - i0);
                const float* col = sp + l * GU_CH;
#pragma unroll 16
                for (int i = 0; i < m; ++i) acc += col[i];
            }
        }
        if (l < 6) mean[l] = acc * inv_n;
    }
    {
        float acc = 0.f;
        for (int i0 = 0; i0 < n; i0 += GU_CH) {
            __syncthreads();   // (the first one also publishes mean[])
            float mm[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) mm[k] = mean[k];
            for (int i = l; i < GU_CH && i0 + i < n; i += GU_THREADS) {
                const float4 s4 = Sc[i0 + i], t4 = Tc[i0 + i];
                const float ds[3] = {s4.x - mm[0], s4.y - mm[1], s4.z - mm[2]}, dt[3] = {t4.x - mm[3], t4.y - mm[4], t4.z - mm[5]};
#pragma unroll
                for (int a = 0; a < 3; ++a)      // sigma(a, b) = sum (tgt_a - mean_t_a) * (src_b - mean_s_b)
#pragma unroll
                    for (int b = 0; b < 3; ++b) sp[(3 * a + b) * GU_CH + i] = dt[a] * ds[b];
            }
            __syncthreads();
            if (l < 9) {
                const int m = min(GU_CH, n - i0);
                const float* col = sp + l * GU_CH;
#pragma unroll 16
                for (int i = 0; i < m; ++i) acc += col[i];
            }
        }
        if (l < 9) sig[l] = acc * inv_n;
    }
    __syncthreads();
    if (l == 0) {
        float A[9], U[9], Sg[3], V[9];
        _Pragma("unroll") for (int k = 0; k < 9; ++k) A[k] = sig[k];
        lgr_svd3(A, U, Sg, V);
        float flip = (lgr_det3(U) * lgr_det3(V) < 0.f) ? -1.f : 1.f;
        float R[9];
        _Pragma("unroll") for (int r = 0; r < 3; ++r)
            _Pragma("unroll") for (int q = 0; q < 3; ++q) R[3 * r + q] = (U[3 * r] * V[3 * q] + U[3 * r + 1] * V[3 * q + 1]) + (U[3 * r + 2] * flip) * V[3 * q + 2];
        _Pragma("unroll") for (int k = 0; k < 16; ++k) Tout[k] = 0.f;
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {
            _Pragma("unroll") for (int q = 0; q < 3; ++q) Tout[4 * q + r] = R[3 * r + q];
            Tout[12 + r] = mean[3 + r] - ((R[3 * r] * mean[0] + R[3 * r + 1] * mean[1]) + R[3 * r + 2] * mean[2]);
        }
        Tout[15] = 1.f;
    }
}

// ------------------------------------------------------------------------------------------------ host: edge stage
struct Vec { float x, y, z; };
static inline Vec operator-(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float dot3(Vec a, Vec b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float len3(Vec a) { return std::sqrt(dot3(a, a)); }
static inline Vec unit(Vec v) { float z = dot3(v, v); if (!(z > 0.f)) return v; float n = std::sqrt(z); return {v.x / n, v.y / n, v.z / n}; }

struct Mat3 {
    float a[9];   // row-major
    float operator()(int r, int c) const { return a[3 * r + c]; }
    float& operator()(int r, int c) { return a[3 * r + c]; }
};
static Mat3 eye3() { return Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
static Mat3 matmul(const Mat3& A, const Mat3& B) {
    Mat3 C;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = (A(r, 0) * B(0, c) + A(r, 1) * B(1, c)) + A(r, 2) * B(2, c);
    return C;
}
static Vec matvec(const Mat3& A, Vec v) {
    return {(A(0, 0) * v.x + A(0, 1) * v.y) + A(0, 2) * v.z, (A(1, 0) * v.x + A(1, 1) * v.y) + A(1, 2) * v.z, (A(2, 0) * v.x + A(2, 1) * v.y) + A(2, 2) * v.z};
}
static Vec matTvec(const Mat3& A, Vec v) {
    return {(A(0, 0) * v.x + A(1, 0) * v.y) + A(2, 0) * v.z, (A(0, 1) * v.x + A(1, 1) * v.y) + A(2, 1) * v.z, (A(0, 2) * v.x + A(1, 2) * v.y) + A(2, 2) * v.z};
}
// rotation taking unit vector a onto unit vector b: I + [v]x + [v]x^2 / (1 + a.b)   (ia_gror.hpp:450-478)
static Mat3 rot_between(Vec a, Vec b) {
    Vec v{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    float k = 1.0f / (1.0f + dot3(a, b));
    Mat3 X{{0.f, -1.0f * v.z, v.y, v.z, 0.f, -1.0f * v.x, -1.0f * v.y, v.x, 0.f}};
    Mat3 X2 = matmul(X, X), R = eye3();
    for (int e = 0; e < 9; ++e) R.a[e] = (R.a[e] + X.a[e]) + X2.a[e] * k;
    return R;
}
struct Pose { Mat3 R; Vec t; };
static Pose chain(const Pose& A, const Pose& B) {   // 4x4 product A * B of two affine matrices
    Vec rt = matvec(A.R, B.t);
    return Pose{matmul(A.R, B.R), Vec{rt.x + A.t.x, rt.y + A.t.y, rt.z + A.t.z}};
}
static Vec move(const Pose& P, Vec p) {   // pcl::detail::Transformer::se3 order
    return {P.R(0, 0) * p.x + (P.R(0, 1) * p.y + (P.R(0, 2) * p.z + P.t.x)), P.R(1, 0) * p.x + (P.R(1, 1) * p.y + (P.R(1, 2) * p.z + P.t.y)),
            P.R(2, 0) * p.x + (P.R(2, 1) * p.y + (P.R(2, 2) * p.z + P.t.z))};
}
static float fast_atan2(float y, float x) {   // ia_gror.h:291-311
    const float c3 = 0.1821F, c1 = 0.9675F;
    float ay = std::fabs(y), r, ang;
    if (x >= 0) { r = (x - ay) / (x + ay); ang = (float) (3.1415926f / 4); }
    else { r = (x + ay) / (ay - x); ang = (float) (3 * 3.1415926f / 4); }
    ang += (c3 * r * r - c1) * r;
    return y < 0 ? -ang : ang;
}

struct Stab { double at; int open; };   // interval end: location, +1 start / 0 end

struct EdgeStage {
    int K;
    std::vector<Vec> S, T;
    float two_res_f;     // 2 * resolution_ in float (RCFS, refine)
    double two_res_d;    // 2.0 * resolution_ in double (graph, TCFS)
    // result
    int best_count = 3;
    Pose best{eye3(), Vec{0, 0, 0}};
    Vec axis{0, 0, 1}, origin{0, 0, 0};
    float angle = 0.f;
    int tcfs_rows = 0;

    // ia_gror.hpp:620-747 + :555-617 (one_to_one branch): best stabbing count of the rotation-angle intervals
    void tcfs(const Pose& two_pt, Vec ax, Vec org, float* out_angle, int* out_count, std::vector<Stab>& ends) const {
        Pose to_axis{rot_between(ax, Vec{0, 0, 1}), Vec{0, 0, 0}};
        to_axis.t = matvec(to_axis.R, Vec{-org.x, -org.y, -org.z});
        Pose src_to_axis = chain(to_axis, two_pt);
        ends.clear();
        const float twopi_f = (float) (2.0 * M_PI);
        const double TWOPI = twopi_f;
        auto add = [&](double b, double e) { ends.push_back({b, 1}); ends.push_back({e, 0}); };
        for (int k = 0; k < K; ++k) {
            Vec m = move(src_to_axis, S[k]), b = move(to_axis, T[k]);
            float m_len = std::sqrt(m.x * m.x + m.y * m.y), b_len = std::sqrt(b.x * b.x + b.y * b.y);
            float m_azi = fast_atan2(m.y, m.x), b_azi = fast_atan2(b.y, b.x);
            double dz = b.z - m.z, d = b_len - m_len;
            double room = two_res_d * two_res_d - dz * dz;
            if (!(d * d <= room)) continue;
            double rth = std::sqrt(room);
            if (m_len <= 1e-12) { add(0, TWOPI); continue; }
            double dev;   // circleIntersection(R = m_len, d = b_len, r = rth), ia_gror.hpp:517-552
            if ((double) b_len <= 1e-12) dev = M_PI;
            else {
                double Rr = m_len, dd = b_len;
                double ratio = ((dd * dd - rth * rth + Rr * Rr) / (2 * dd)) / Rr;
                dev = ratio <= -1.0 ? M_PI : std::acos(ratio);
            }
            if (std::fabs(dev - M_PI) <= 1e-12) { add(0, TWOPI); continue; }
            double lo = std::fmod(b_azi - dev - m_azi, TWOPI), hi = std::fmod(b_azi + dev - m_azi, TWOPI);
            if (lo < 0) lo += TWOPI;
            if (hi < 0) hi += TWOPI;
            if (hi >= lo) add(lo, hi);
            else { add(lo, TWOPI); add(0, hi); }
        }
        std::stable_sort(ends.begin(), ends.end(), [](const Stab& p, const Stab& q) { return p.at < q.at; });
        int level = 0, closed = 0, top = 0;
        double top_at = 0, cursor = 0;
        for (const Stab& e : ends) {
            if (e.open) { if (++level > top) { top = level; top_at = e.at; } }
            else ++closed;
            if (e.at > cursor) { level -= closed; closed = 0; cursor = e.at; }
        }
        *out_angle = (float) top_at;
        *out_count = top;
    }

    void run() {
        // edge graph (ia_gror.hpp:82-124): only the row sizes and each row's first edge are ever used
        std::vector<int> row_size(K, 0), row_first(K, -1), order(K);
        for (int i = 0; i < K; ++i)
            for (int j = i + 1; j < K; ++j) {
                float dl = std::fabs(len3(S[i] - S[j]) - len3(T[i] - T[j]));
                if (dl < two_res_d) { if (row_size[i]++ == 0) row_first[i] = j; }
            }
        for (int i = 0; i < K; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return row_size[p] > row_size[q]; });
        std::vector<Stab> ends;
        ends.reserve(4 * (size_t) K);
        for (int r = 0; r < K; ++r) {
            int i = order[r];
            if (row_size[i] < 10) break;   // rows are sorted: every later row is skipped as well
            int j = row_first[i];
            // twoPairPointsAlign (ia_gror.hpp:417-448)
            Vec us = unit(S[i] - S[j]), ut = unit(T[i] - T[j]);
            Pose two_pt;
            two_pt.R = rot_between(us, ut);
            Vec d1 = T[i] - matvec(two_pt.R, S[i]), d2 = T[j] - matvec(two_pt.R, S[j]);
            two_pt.t = Vec{0.5f * (d1.x + d2.x), 0.5f * (d1.y + d2.y), 0.5f * (d1.z + d2.z)};
            // relaxed constraint (ia_gror.hpp:472-501): same distance to the first node and same height along the axis
            Vec axis_s = matTvec(two_pt.R, ut);
            int relaxed = 0;
            for (int k = 0; k < K; ++k) {
                Vec dt = T[k] - T[i], ds = S[k] - S[i];
                if (std::fabs(len3(dt) - len3(ds)) < two_res_f && std::fabs(dot3(dt, ut) - dot3(ds, axis_s)) < two_res_f) ++relaxed;
            }
            if (relaxed <= best_count) continue;
            ++tcfs_rows;
            float ang;
            int tight;
            tcfs(two_pt, ut, T[i], &ang, &tight, ends);
            if (tight > best_count) { best_count = tight; best = two_pt; axis = ut; origin = T[i]; angle = ang; }
        }
    }

    // IdM_3 * IdM_2 * IdM_1 * two_point_tran_mat (ia_gror.hpp:404-412), column-major 4x4
    void global_transform(float G[16]) const {
        float sn = std::sin(angle), cs = std::cos(angle);   // Eigen::AngleAxisf::toRotationMatrix
        Vec sa{sn * axis.x, sn * axis.y, sn * axis.z}, ca{(1 - cs) * axis.x, (1 - cs) * axis.y, (1 - cs) * axis.z};
        Mat3 R;
        float w;
        w = ca.x * axis.y; R(0, 1) = w - sa.z; R(1, 0) = w + sa.z;
        w = ca.x * axis.z; R(0, 2) = w + sa.y; R(2, 0) = w - sa.y;
        w = ca.y * axis.z; R(1, 2) = w - sa.x; R(2, 1) = w + sa.x;
        R(0, 0) = ca.x * axis.x + cs; R(1, 1) = ca.y * axis.y + cs; R(2, 2) = ca.z * axis.z + cs;
        Pose back{eye3(), origin}, spin{R, Vec{0, 0, 0}}, fwd{eye3(), Vec{-origin.x, -origin.y, -origin.z}};
        Pose g = chain(chain(chain(back, spin), fwd), best);
        for (int k = 0; k < 16; ++k) G[k] = 0.f;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) G[4 * c + r] = g.R(r, c);
        G[12] = g.t.x; G[13] = g.t.y; G[14] = g.t.z; G[15] = 1.f;
    }
};

struct GrorBuffers { float4 *S, *T; int* degree; };

int gror_pack(lgr_ctx* ctx, const float* d_src, const float* d_tgt, const lgr_corr* d_corr, int c, GrorBuffers* b) {
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_PACK, (size_t) 2 * c + 4, &b->S));
    b->T = b->S + c;
    gror_pack_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(d_src, d_tgt, d_corr, c, b->S, b->T);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

int gror_degrees(lgr_ctx* ctx, const GrorBuffers& b, int c, float resolution, int* d_degree) {
    LGR_HIP(ctx, hipMemsetAsync(d_degree, 0, (size_t) c * 4, ctx->stream));
    int iblocks = cdiv(c, DEG_THREADS);
    // enough workgroups for 256 CUs: split the j range when there are few i blocks
    int want = std::max(1, (8 * ctx->n_cu) / iblocks);
    int nsplit = std::min(want, cdiv(c, DEG_TILE));
    int slice = cdiv(cdiv(c, nsplit), DEG_TILE) * DEG_TILE;
    nsplit = cdiv(c, slice);
    gror_degree_kernel<<<dim3(iblocks, nsplit), DEG_THREADS, 0, ctx->stream>>>(b.S, b.T, c, slice, 2.0f * resolution, d_degree);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

}  // namespace

extern "C" int lgr_gror_node_degree_dev(lgr_ctx* ctx, const float* d_src, const float* d_tgt, const lgr_corr* d_corr, int c, float resolution,
                                        int32_t* d_degree) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && (d_corr || c == 0) && (d_degree || c == 0) && c >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    if (c == 0) return LGR_OK;
    GrorBuffers b;
    LGR_TRY(gror_pack(ctx, d_src, d_tgt, d_corr, c, &b));
    return gror_degrees(ctx, b, c, resolution, d_degree);
}

extern "C" int lgr_gror_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c, float resolution,
                            int k_optimal, lgr_result* res, uint8_t* d_inlier_mask) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && res && ns > 0 && nt > 0 && (d_corr || c == 0) && c >= 0, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, resolution > 0.f && k_optimal > 0 && k_optimal <= 4096, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof *res);
    auto t0 = std::chrono::steady_clock::now();
    // c == 0: the reference reads uninitialised matrices; report "not converged" with the identity
    if (c == 0) {
        for (int k = 0; k < 4; ++k) res->transformation[5 * k] = 1.f;
        res->iterations = 1;
        return LGR_OK;
    }
    LGR_TRY(lgr_check_corr(ctx, d_corr, c, ns, nt));
    GrorBuffers b;
    LGR_TRY(gror_pack(ctx, d_src, d_tgt, d_corr, c, &b));
    int K = c >= k_optimal ? k_optimal : c;
    // one scratch block: sort keys first, later flags | pos | n | compacted pairs
    int* hist;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_HIST, (size_t) 2 * c + 8 + 8 * ((size_t) c + 4), &hist));
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    if (c >= k_optimal) {   // optimalSelectionBasedOnNodeReliability; below K_optimal the input order is kept (:183-185)
        LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_LIST, (size_t) c, &b.degree));
        LGR_TRY(gror_degrees(ctx, b, c, resolution, b.degree));
        keys = (unsigned long long*) hist;
        keys2 = keys + c;
        gror_keys_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(b.degree, c, keys);
        size_t tb = 0;
        LGR_HIP(ctx, rocprim::radix_sort_keys(nullptr, tb, keys, keys2, (size_t) c, 0, 64, ctx->stream));
        void* tmp;
        LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
        LGR_HIP(ctx, rocprim::radix_sort_keys(tmp, tb, keys, keys2, (size_t) c, 0, 64, ctx->stream));
    }
    float4* d_sel;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_MISC, (size_t) 2 * K + 16, &d_sel));
    gror_gather_kernel<<<cdiv(K, 256), 256, 0, ctx->stream>>>(keys2, K, keys2 != nullptr, b.S, b.T, d_sel);
    float4* h_sel;
    LGR_TRY(lgr_pinned(ctx, (size_t) 2 * K * 16 + 256, (void**) &h_sel));
    LGR_HIP(ctx, hipMemcpyAsync(h_sel, d_sel, (size_t) 2 * K * 16, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));

    EdgeStage es;
    es.K = K;
    es.S.resize(K);
    es.T.resize(K);
    for (int k = 0; k < K; ++k) { es.S[k] = Vec{h_sel[2 * k].x, h_sel[2 * k].y, h_sel[2 * k].z}; es.T[k] = Vec{h_sel[2 * k + 1].x, h_sel[2 * k + 1].y, h_sel[2 * k + 1].z}; }
    es.two_res_f = 2 * resolution;
    es.two_res_d = 2.0 * resolution;
    es.run();
    float* h_G = (float*) ((char*) h_sel + (size_t) 2 * K * 16);
    es.global_transform(h_G);

    // refineTransformationMatrix
    float* d_G;
    LGR_TRY(lgr_ws_t(ctx, WS_RANSAC_T, (size_t) 64, &d_G));
    LGR_HIP(ctx, hipMemcpyAsync(d_G, h_G, 64, hipMemcpyHostToDevice, ctx->stream));
    int* flags = hist;   // the keys are dead: the gather above has completed
    int* pos = flags + c;
    float4* Sc = (float4*) (((uintptr_t) (flags + 2 * (size_t) c + 4) + 15) & ~(uintptr_t) 15);
    float4* Tc = Sc + c;
    int* d_n = pos + c;
    gror_inlier_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(b.S, b.T, c, d_G, 2 * resolution, flags, d_inlier_mask);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, flags, pos, 0, (size_t) c, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, flags, pos, 0, (size_t) c, rocprim::plus<int>(), ctx->stream));
    gror_compact_kernel<<<cdiv(c, 256), 256, 0, ctx->stream>>>(b.S, b.T, flags, pos, c, Sc, Tc, d_n);
    gror_umeyama_kernel<<<1, GU_THREADS, 0, ctx->stream>>>(Sc, Tc, d_n, d_G + 16);
    LGR_HIP(ctx, hipGetLastError());
    float* h_out = h_G + 16;
    LGR_HIP(ctx, hipMemcpyAsync(h_out, d_G + 16, 64, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h_out + 16, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(res->transformation, h_out, 64);
    res->iterations = 1;                       // AlignmentResult{..., 1, true, t} (src/alignment.cpp:34)
    res->converged = 1;
    res->n_inliers = *(int*) (h_out + 16);
    res->metric = (float) es.best_count;       // size of the maximum consistent set found by the edge stage
    res->best_metric_before_refit = (float) es.best_count;
    res->best_iteration = es.tcfs_rows;        // rows that reached the tight-constraint stage
    res->estimated_iters = K;
    res->n_correspondences = c;
    res->time_te = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return LGR_OK;
}

extern "C" int lgr_gror(lgr_ctx* ctx, const float* src, int ns, const float* tgt, int nt, const lgr_corr* corr, int c, float resolution,
                        int k_optimal, lgr_result* res, uint8_t* inlier_mask) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, src && tgt && res && ns > 0 && nt > 0 && (corr || c == 0) && c >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *ds, *dt;
    lgr_corr* dc;
    uint8_t* dm = nullptr;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) ns * 12, &ds));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) nt * 12, &dt));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) c + 1, &dc));
    if (inlier_mask) LGR_TRY(lgr_ws_t(ctx, WS_HOST_D, (size_t) c + 1, &dm));
    LGR_HIP(ctx, hipMemcpyAsync(ds, src, (size_t) ns * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(dt, tgt, (size_t) nt * 48, hipMemcpyHostToDevice, ctx->stream));
    if (c) LGR_HIP(ctx, hipMemcpyAsync(dc, corr, (size_t) c * 16, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_gror_dev(ctx, ds, ns, dt, nt, dc, c, resolution, k_optimal, res, dm));
    if (inlier_mask && c) {
        LGR_HIP(ctx, hipMemcpyAsync(inlier_mask, dm, (size_t) c, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return LGR_OK;
}
