// lgr_match.hip -- brute-force FPFH matching (both directions from one MFMA pass) for gfx950.
//
// Replaces include/matching.h:594-634 matchBF<FPFH> (cv::BFMatcher(NORM_L2)::knnMatch, k = 1) and the cross-block
// merge src/common.cpp:517-529.  Result contract (bit-exact with the oracle): for every valid query row, the train
// row minimising the CANONICAL distance d = sqrtf(normL2Sqr) -- OpenCV 4.5.1 SSE lane order, see exact_l2() -- with
// ties broken "highest bf block, then lowest index inside the block"; NaN rows never match.
//
// Structure (DESIGN.md "matcher"):
//   1. cluster      : 16 k-means centres of the descriptors (Lloyd on a sample, on the device); every row is assigned
//                     to its nearest centre and both sets are sorted by (cluster, distance to centre).
//   2. pack         : MFMA operands, K = 34: A' = [-2(a - c_p), 1] for a in cluster p, and one column set per cluster,
//                     B'(p) = [b - c_p, |b - c_p|^2].  Distances are translation invariant, so for a row of cluster p
//                     S = A'.B'(p) = |b - c_p|^2 - 2 (a - c_p).(b - c_p) = d2 - |a - c_p|^2 -- and its rounding error
//                     scales with (|a - c_p| + |b - c_p|)^2, i.e. it is tiny exactly for the near pairs that matter
//                     (FPFH data is full of near-duplicate "flat surface" rows far from the global mean).
//   3. match_mfma   : the brute-force contraction on v_mfma_f32_32x32x2_f32 with a fused epilogue that keeps only
//                     min_b d2~ per (row, column group) and min_a d2~ per (column, row group), d2~ = S + |a'|^2.
//                     FILTER only.
//   4. rerank_*     : per query, a group is a candidate when its lower bound (value - proven error) does not exceed
//                     the smallest upper bound; candidate groups are rescanned with the exact canonical distance and
//                     a packed 64-bit atomicMin applies the reference's tie rules (order independent).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include "lgr_internal.h"

namespace {

constexpr int KK = 17;              // K = 34 -> 17 MFMA steps of k = 2
constexpr int TILE = 32;
constexpr int RW = 1;               // row tiles per wave
constexpr int WAVES = 8;
constexpr int NTHR = WAVES * 64;    // threads per workgroup of the MFMA kernel
constexpr int BLOCK_ROWS = TILE * RW * WAVES;   // 256
constexpr int STAGE_TILES = 4;
constexpr int STAGE_COLS = STAGE_TILES * TILE;  // 128
constexpr int CHUNK_COLS = 4096;
constexpr int PAD = 256;
constexpr int KCL = 16;             // k-means centres (operand centring)
constexpr int SUBMAX = 64;          // second-level centres per cluster ("leaves": sort order + skip bounds)
constexpr int MAXLEAF = KCL * SUBMAX;
#ifndef LGR_KM_SAMPLE
#define LGR_KM_SAMPLE 16384
#endif
constexpr int KM_SAMPLE = LGR_KM_SAMPLE;    // sample rows per side
#ifndef LGR_KM_ITERS
#define LGR_KM_ITERS 16   // Lloyd iterations, first / second level (6 / 4 -> 10 / 8 -> 16 / 8: 81.7 -> 80.2 -> 79.5 ms per 1M pair; tighter leaves)
#endif
constexpr int KM_ITERS = LGR_KM_ITERS;
#ifndef LGR_KM2_ITERS
#define LGR_KM2_ITERS 8
#endif
constexpr int KM2_ITERS = LGR_KM2_ITERS;
#ifndef LGR_MM_OCC
#define LGR_MM_OCC 4          // waves per SIMD of match_mfma (2: 256 VGPRs, one workgroup per CU; 4: 128 VGPRs, two)
#endif
constexpr int NEAR_T = 48;          // pass 0 visits the NEAR_T nearest leaves of a row block / row blocks of a leaf (measured optimum at 1M with the box bounds: 32 / 48 / 64 / 96 -> 84.2 / 83.3 / 84.4 / 87.0 ms per pair)
#ifndef LGR_PRUNE_BETAS
#define LGR_PRUNE_BETAS 1.0f   // intermediate thresholds (e.g. 0.5f, 1.0f) were measured: no gain over one final pass
#endif
#ifndef LGR_GROUP_COLS
#define LGR_GROUP_COLS 1024
#endif
constexpr int GROUP_COLS = LGR_GROUP_COLS;     // largest column group of the row-minimum table (leaves are cut into such pieces)
constexpr int STAGES_PER_CHUNK = CHUNK_COLS / STAGE_COLS;   // 32 -> one 32-bit stage mask per (row block, chunk)
constexpr float FLT_BIG = 3.4028234663852886e38f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MFMA operand formats.  F32: v_mfma_f32_32x32x2_f32, K = 34 (33 dims + norm slot) -> 17 steps, fragment = 1 float.
// F16: v_mfma_f32_32x32x16_f16 (16x the f32 rate) on two-term f16 splits of the scaled f32 operands,
//      x * 2^s = h1 + h2 (+ residual <= 2^-22 |x|):  a.b ~ a1.b1 + a1.b2 + a2.b1  -> concatenated K = 3 * 33 + 6 norm slots
//      = 105, padded to 112 = 7 steps, fragment = 8 halves (lane l: row l & 31, k = 16 * step + 8 * (l >> 5) + j).
// F16R: the same on 30 coordinates.  Every 11-bin block of an FPFH row sums to 100, so differences of rows have no component
//      along the block's all-ones direction; in a Helmert basis of the block that direction is one coordinate, the other
//      10 carry the whole distance.  K = 3 * 30 + 6 = 96 = 6 steps (-1/7 of the MFMA work, LDS reads and operand bytes).
//      Used only when the dropped coordinates are (numerically) constant over both sets; their largest measured energy
//      enters the error bound, so any input stays exact (match_impl, "rot").
enum { FMT_F32 = 0, FMT_F16 = 1, FMT_F16R = 2 };
template <int FMT> struct OpFmt;
template <> struct OpFmt<FMT_F32> { typedef float frag; static constexpr int KS = 17; };
template <> struct OpFmt<FMT_F16> { typedef f16x8 frag; static constexpr int KS = 7; };
template <> struct OpFmt<FMT_F16R> { typedef f16x8 frag; static constexpr int KS = 6; };
struct F16Scale { float s_mul; float inv_s2; float a_norm[3]; };   // 2^s, 2^-2s, the three a-side norm-slot constants

__device__ __forceinline__ unsigned f2key(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}
__device__ __forceinline__ bool row_finite(const float* __restrict__ r, float* v) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 33; ++k) { v[k] = r[k]; ok = ok && (fabsf(v[k]) <= FLT_BIG); }
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
// 1. clustering (any centres are valid -- they only shape the error bound -- so float atomics are fine here)
__global__ void km_sample(const float* __restrict__ A, int ma, const float* __restrict__ B, int mb, int per_side,
                          float* __restrict__ smp, int* __restrict__ smp_ok) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 2 * per_side) return;
    const float* X = s < per_side ? A : B;
    int m = s < per_side ? ma : mb;
    int t = s < per_side ? s : s - per_side;
    float v[33];
    bool ok = false;
    if (m > 0) {
        long long i = (long long) t * m / per_side;
        ok = row_finite(X + (size_t) i * 33, v);
    }
    for (int k = 0; k < 33; ++k) smp[(size_t) s * 33 + k] = ok ? v[k] : 0.f;
    smp_ok[s] = ok ? 1 : 0;
}
__global__ void km_init(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns, float* __restrict__ cen) {
    int c = threadIdx.x;
    if (c >= KCL) return;
    int s = (int) ((long long) c * ns / KCL);
    int tries = 0;
    while (!smp_ok[s] && tries < ns) { s = (s + 1) % ns; ++tries; }
    for (int k = 0; k < 33; ++k) cen[c * 33 + k] = smp_ok[s] ? smp[(size_t) s * 33 + k] : 0.f;
}
__device__ __forceinline__ int nearest_centre(const float* v, const float* __restrict__ cen, float& best) {
    int bi = 0;
    best = __uint_as_float(0x7f800000u);
#pragma unroll 1
    for (int c = 0; c < KCL; ++c) {
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) { float t = v[k] - cen[c * 33 + k]; d = d + t * t; }
        if (d < best) { best = d; bi = c; }
    }
    return bi;
}
// second level: `sub` centres inside every cluster, seeded with evenly spaced sample members of the cluster
__global__ void km_label(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns, const float* __restrict__ cen,
                         int* __restrict__ label) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    int c = -1;
    if (smp_ok[s]) {
        float v[33], d;
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = smp[(size_t) s * 33 + k];
        c = nearest_centre(v, cen, d);
    }
    label[s] = c;
}
__global__ void km2_init(const float* __restrict__ smp, const int* __restrict__ label, int ns, const float* __restrict__ cen, int sub,
                         float* __restrict__ cen2) {
    const int p = blockIdx.x, lane = threadIdx.x;   // one wave per cluster
    for (int e = lane; e < sub * 33; e += 64) cen2[(size_t) p * sub * 33 + e] = cen[p * 33 + e % 33];
    __threadfence_block();
    __syncthreads();
    int cnt = 0;
    for (int base = 0; base < ns; base += 64) {
        int s = base + lane;
        bool m = s < ns && label[s] == p;
        cnt += __popcll(__ballot(m));
    }
    if (cnt == 0) return;
    int rank0 = 0;
    for (int base = 0; base < ns; base += 64) {
        int s = base + lane;
        bool m = s < ns && label[s] == p;
        unsigned long long bal = __ballot(m);
        if (m) {
            int r = rank0 + __popcll(bal & ((1ull << lane) - 1ull));
            int j = (int) ((long long) r * sub / cnt);
            bool first = r == 0 || (int) ((long long) (r - 1) * sub / cnt) != j;
            if (first)
                for (int k = 0; k < 33; ++k) cen2[((size_t) p * sub + j) * 33 + k] = smp[(size_t) s * 33 + k];
        }
        rank0 += __popcll(bal);
    }
}
__device__ __forceinline__ int nearest_sub(const float* v, const float* __restrict__ c2 /* [sub][33] of the row's cluster */, int sub, float& best) {
    int bj = 0;
    best = __uint_as_float(0x7f800000u);
#pragma unroll 1
    for (int j = 0; j < sub; ++j) {
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) { float t = v[k] - c2[j * 33 + k]; d = d + t * t; }
        if (d < best) { best = d; bj = j; }
    }
    return bj;
}
constexpr int KM2_THREADS = 512;
// Lloyd step of the second level in two deterministic kernels (no float atomics: the same centres, hence the same tile
// schedule and timing, on every run).  km2_label: leaf of every sample (sub-centres of all clusters in LDS, odd pitch per
// cluster as in assign_kernel).  km2_centres: one wave per leaf sums its samples in sample order and writes the new centre.
__global__ __launch_bounds__(KM2_THREADS) void km2_label(const float* __restrict__ smp, const int* __restrict__ label, int ns, const float* __restrict__ cen2, int sub,
                                                         int* __restrict__ leaf_of /* [ns], -1: no cluster */) {
    extern __shared__ float c2s[];
    const int pitch = sub * 33 + 1;
    for (int e = threadIdx.x; e < KCL * sub * 33; e += blockDim.x) c2s[(e / (sub * 33)) * pitch + e % (sub * 33)] = cen2[e];
    __syncthreads();
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    int p = label[s];
    int leaf = -1;
    if (p >= 0) {
        float v[33], d;
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = smp[(size_t) s * 33 + k];
        leaf = p * sub + nearest_sub(v, c2s + p * pitch, sub, d);
    }
    leaf_of[s] = leaf;
}
// level 1: 16 centres over the whole sample -> 1024 threads per centre, the 16 wave sums combined in a fixed order
constexpr int KMC_THREADS = 1024;
__global__ __launch_bounds__(KMC_THREADS) void km_centres(const float* __restrict__ smp, const int* __restrict__ label, int ns, float* __restrict__ cen) {
    __shared__ float part[KMC_THREADS / 64][34];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) acc[k] = 0.f;
    int n = 0;
    for (int s0 = threadIdx.x; s0 < ns; s0 += KMC_THREADS * 8) {
        int lb[8];   // eight label loads in flight (a label per iteration is one exposed load latency per iteration)
#pragma unroll
        for (int u = 0; u < 8; ++u) lb[u] = s0 + KMC_THREADS * u < ns ? label[s0 + KMC_THREADS * u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (lb[u] != c) continue;
            const int s = s0 + KMC_THREADS * u;
            ++n;
#pragma unroll
            for (int k = 0; k < 33; ++k) acc[k] += smp[(size_t) s * 33 + k];
        }
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
#pragma unroll
    for (int k = 0; k < 33; ++k) {
        float a = acc[k];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) part[wave][k] = a;
    }
    if (lane == 0) part[wave][33] = (float) n;
    __syncthreads();
    if (threadIdx.x < 33) {
        float cnt = 0.f, sum = 0.f;
        for (int w = 0; w < KMC_THREADS / 64; ++w) { cnt += part[w][33]; sum += part[w][threadIdx.x]; }
        if (cnt > 0.f) cen[c * 33 + threadIdx.x] = sum / cnt;
    }
}
__global__ __launch_bounds__(64) void km2_centres(const float* __restrict__ smp, const int* __restrict__ leaf_of, int ns, float* __restrict__ cen2) {
    const int leaf = blockIdx.x, lane = threadIdx.x;
    float acc[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) acc[k] = 0.f;
    int n = 0;
    for (int s0 = lane; s0 < ns; s0 += 64 * 16) {
        int lb[16];   // sixteen label loads in flight
#pragma unroll
        for (int u = 0; u < 16; ++u) lb[u] = s0 + 64 * u < ns ? leaf_of[s0 + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (lb[u] != leaf) continue;
            const int s = s0 + 64 * u;
            ++n;
#pragma unroll
            for (int k = 0; k < 33; ++k) acc[k] += smp[(size_t) s * 33 + k];
        }
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (n == 0) return;   // empty leaf: the centre stays
#pragma unroll
    for (int k = 0; k < 33; ++k) {
        float a = acc[k];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) cen2[(size_t) leaf * 33 + k] = a / (float) n;
    }
}

// key = (leaf << 22) | (bits(r2) >> 9), leaf = cluster * sub + sub-centre: sort by cluster, leaf, then distance to the
// cluster centre.  Invalid rows: 0xffffffff.  counts[leaf] / counts[MAXLEAF] (invalid) and the squared leaf radii
// rmax[leaf] = max |x - c_leaf|^2 (float bits) are accumulated through LDS.
constexpr int ASSIGN_THREADS = 512;
__global__ __launch_bounds__(ASSIGN_THREADS) void assign_kernel(const float* __restrict__ X, int m, const float* __restrict__ cen, const float* __restrict__ cen2, int sub,
                                                                unsigned* __restrict__ keys, int* __restrict__ vals, uint8_t* __restrict__ valid,
                                                                int* __restrict__ counts /* [MAXLEAF+1] */, unsigned* __restrict__ rmax /* [MAXLEAF] */) {
    // All sub-centres live in LDS (dynamic; up to 16 x 64 x 33 floats = 135 KB): every lane walks the sub-centres of ITS
    // cluster, which from global memory is a per-lane gather of 33 x sub words.  The odd pitch per cluster keeps lanes of
    // different clusters on different banks; lanes of one cluster read the same word (broadcast).
    extern __shared__ float c2s[];
    const int pitch = sub * 33 + 1;
    int* lc = (int*) (c2s + KCL * pitch);
    unsigned* lr = (unsigned*) (lc + MAXLEAF + 1);
    for (int e = threadIdx.x; e < KCL * sub * 33; e += blockDim.x) c2s[(e / (sub * 33)) * pitch + e % (sub * 33)] = cen2[e];
    for (int i = threadIdx.x; i <= MAXLEAF; i += blockDim.x) { lc[i] = 0; if (i < MAXLEAF) lr[i] = 0u; }
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        float v[33], r2;
        bool ok = row_finite(X + (size_t) i * 33, v);
        unsigned key = 0xffffffffu;
        if (ok) {
            // a finite row whose squared distance to every centre overflows float stays a valid row (its exact distance to
            // a duplicate of itself is 0 in the reference); it lands in leaf 0 of cluster 0 with an infinite radius, and
            // the overflow sends the whole call down the exact dense path (match_impl, force_dense)
            int c = nearest_centre(v, cen, r2);
            float rl2;
            int j = nearest_sub(v, c2s + c * pitch, sub, rl2);
            if (!(r2 < FLT_BIG)) r2 = __uint_as_float(0x7f800000u);
            if (!(rl2 < FLT_BIG)) rl2 = __uint_as_float(0x7f800000u);
            int leaf = c * sub + j;
            key = ((unsigned) leaf << 22) | (__float_as_uint(r2) >> 9);
            atomicAdd(&lc[leaf], 1);
            atomicMax(&lr[leaf], __float_as_uint(rl2));
        }
        if (!ok) atomicAdd(&lc[MAXLEAF], 1);
        keys[i] = key; vals[i] = i; valid[i] = ok ? 1 : 0;
    }
    __syncthreads();
    for (int l = threadIdx.x; l <= MAXLEAF; l += blockDim.x) {
        if (lc[l]) atomicAdd(&counts[l], lc[l]);
        if (l < MAXLEAF && lr[l]) atomicMax(&rmax[l], lr[l]);
    }
}

// sorted position s -> padded position (leaves / clusters start at multiples of their pad units)
__global__ void place_kernel(const unsigned* __restrict__ keys_sorted, const int* __restrict__ vals_sorted, int n_valid,
                             const int* __restrict__ sorted_start /* [leaf] */, const int* __restrict__ pad_start /* [leaf] */,
                             int* __restrict__ perm) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_valid) return;
    int l = (int) (keys_sorted[s] >> 22);
    perm[pad_start[l] + (s - sorted_start[l])] = vals_sorted[s];
}

// 2. pack.  P layout: [tile][kk][half][i] floats (tile = 32 rows): MFMA lane l of step kk reads P[(tile*KK+kk)*64 + l].
// role 0 (rows): centre = the row's own cluster (blkcl[pos / 256]), operand [-2 x', 1], nrm = |x'|^2
// role 1 (cols): blockIdx.y = cluster set p, centre c_p for every column, operand [x', |x'|^2]
// padding positions (perm < 0): rows [0.., 1] / cols [0.., +inf], nrm = +inf
__global__ void pack_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, int role,
                            const float* __restrict__ cen, const int* __restrict__ blkcl,
                            float* __restrict__ P, float* __restrict__ nrm, unsigned* __restrict__ ovf) {
    int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n_pad) return;
    int set = role == 1 ? blockIdx.y : 0;
    int o = perm[pos];
    int c = role == 1 ? set : blkcl[pos / BLOCK_ROWS];
    float v[33];
    float n2 = 0.f;
    if (o >= 0) {
#pragma unroll
        for (int k = 0; k < 33; ++k) { v[k] = X[(size_t) o * 33 + k] - cen[c * 33 + k]; n2 = n2 + v[k] * v[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = 0.f;
        n2 = __uint_as_float(0x7f800000u);
    }
    nrm[(size_t) set * n_pad + pos] = n2;
    if (o >= 0 && !(n2 < FLT_BIG)) *ovf = 1u;   // |x'|^2 overflows float: the filter cannot represent this row
    int tile = pos >> 5, r = pos & 31;
    float* base = P + ((size_t) set * (n_pad / TILE) + tile) * KK * 64 + r;
#pragma unroll
    for (int k = 0; k < 34; ++k) {
        float val;
        if (k < 33) val = role == 0 ? -2.0f * v[k] : v[k];
        else val = role == 0 ? 1.0f : n2;
        base[(k >> 1) * 64 + (k & 1) * 32] = val;
    }
}

// f16-split operands (OpFmt<true>).  Same roles / sets / nrm output as pack_kernel; P holds f16x8 fragments:
// fragment (tile, step, lane) at ((set * tiles + tile) * 7 + step) * 64 + lane, lane = row | (khalf << 5).
// Concatenated K index c: [0,33) a1.b1, [33,66) a1.b2, [66,99) a2.b1, 99..104 norm slots, rest 0.
// rows: h = split(-2 x' 2^s);  cols: h = split(x' 2^s);  a norm enters as the three-term f16 expansion of
// N = |x'|^2 2^2s against the constants A1..A3 on the other side (N = A1 B1 + A2 B2 + A3 B3 up to 2^-33 N or the f16
// flush limit).
// Helmert coordinates of one 11-bin block: y_k = (x_0 + .. + x_{k-1} - k x_k) / sqrt(k (k + 1)), k = 1..10 (orthonormal, all
// orthogonal to (1,..,1)); *u = (x_0 + .. + x_10) / sqrt(11) is the dropped coordinate.
__device__ __forceinline__ void helmert11(const float* __restrict__ x, float* __restrict__ y, float* u) {
    const float rs[10] = {0.70710678118654752f, 0.40824829046386302f, 0.28867513459481288f, 0.22360679774997897f, 0.18257418583505537f,
                          0.15430334996209191f, 0.13363062095621219f, 0.11785113019775793f, 0.10540925533894598f, 0.09534625892455924f};
    float pre = x[0];
#pragma unroll
    for (int k = 1; k <= 10; ++k) {
        y[k - 1] = (pre - (float) k * x[k]) * rs[k - 1];
        pre = pre + x[k];
    }
    *u = pre * 0.30151134457776363f;
}

template <bool ROT, bool NORMS_ONLY>
__global__ __launch_bounds__(256) void pack16_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, int role,
                              const float* __restrict__ cen, const int* __restrict__ blkcl, F16Scale sc,
                              _Float16* __restrict__ P, float* __restrict__ nrm, unsigned* __restrict__ drop_max) {
    int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = pos < n_pad;
    if (!in_range) pos = n_pad - 1;           // keep whole waves alive for the reduction below; nothing is stored
    // the row is read once (a 132-byte gather) and packed for every set: 16 column sets, one per centre, or the row set
    const int n_sets = role == 1 ? KCL : 1;
    const int o = perm[pos];
    float x0[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) x0[k] = o >= 0 ? X[(size_t) o * 33 + k] : 0.f;
    float drop = 0.f;
#pragma unroll 1
    for (int set = 0; set < n_sets; ++set) {
    const int c = role == 1 ? set : blkcl[pos / BLOCK_ROWS];
    float v[33];
    float n2 = 0.f;
    if (o >= 0) {
#pragma unroll
        for (int k = 0; k < 33; ++k) { v[k] = x0[k] - cen[c * 33 + k]; n2 = n2 + v[k] * v[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = 0.f;
        n2 = __uint_as_float(0x7f800000u);
    }
    if (in_range) nrm[(size_t) set * n_pad + pos] = n2;   // |x'|^2 in all 33 coordinates: the magnitude the error bounds are stated in
    if (NORMS_ONLY) {
        // first pass: norms (the scale is chosen from the largest one) and the largest energy of the three coordinates
        // the rotated format drops, u_k = (sum of block k of x') / sqrt(11): summed in double (exact for any realistic
        // exponent spread), rounded up; one atomic per wave
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 11; ++k) { s0 += (double) v[k]; s1 += (double) v[11 + k]; s2 += (double) v[22 + k]; }
        float d2 = (o >= 0 && in_range) ? (float) (((s0 * s0 + s1 * s1) + s2 * s2) * (1.0001 / 11.0)) * 1.000001f + 1e-20f * n2 : 0.f;
        drop = fmaxf(drop, d2);
        if (o >= 0 && in_range && !(n2 < FLT_BIG)) drop_max[1] = 1u;   // |x'|^2 overflows float: the filter cannot represent this row
        continue;
    }
    float y[30], u0, u1, u2;
    if (ROT) { helmert11(v, y, &u0); helmert11(v + 11, y + 10, &u1); helmert11(v + 22, y + 20, &u2); }
    if (!in_range) continue;
    constexpr int nd = ROT ? 30 : 33, ks = ROT ? OpFmt<FMT_F16R>::KS : OpFmt<FMT_F16>::KS;
    int tile = pos >> 5, r = pos & 31;
    // the row's K = 16 ks halves are assembled in registers (all indices are compile-time constants) and leave as
    // 2 ks 16-byte pieces: piece (step, khalf) of row r sits at fragment (step * 64 + khalf * 32 + r)
    _Float16 hv[ks * 16];
    auto put = [&](int cidx, _Float16 h) { hv[cidx] = h; };
    const float mul = role == 0 ? -2.0f * sc.s_mul : sc.s_mul;
    float n2m = n2;                           // the norm the MFMA chain must see: of the operand coordinates
    if (ROT && o >= 0) {
        n2m = 0.f;
#pragma unroll
        for (int k = 0; k < 30; ++k) n2m = n2m + y[k] * y[k];
    }
#pragma unroll
    for (int k = 0; k < nd; ++k) {
        float x = (ROT ? y[k] : v[k]) * mul;  // exact (power of two)
        _Float16 h1 = (_Float16) x;           // round to nearest
        _Float16 h2 = (_Float16) (x - (float) h1);
        if (role == 0) { put(k, h1); put(nd + k, h1); put(2 * nd + k, h2); }
        else { put(k, h1); put(nd + k, h2); put(2 * nd + k, h1); }
    }
    // norm slots: 3 nd .. 3 nd + 2 carry |b'|^2 (expansion on the column side, constants on the row side), the next three
    // |a'|^2 the other way round, so d2~ 2^2s = |b'|^2 - 2 a'.b' + |a'|^2 comes out of the MFMA chain with C = 0
    const bool rows = role == 0;
    const _Float16 c0 = (_Float16) sc.a_norm[0], c1 = (_Float16) sc.a_norm[1], c2 = (_Float16) sc.a_norm[2];
    _Float16 b1, b2, b3;
    if (o >= 0) {
        float N = n2m * (sc.s_mul * sc.s_mul);
        b1 = (_Float16) (N / sc.a_norm[0]);
        float r1 = __builtin_fmaf(-sc.a_norm[0], (float) b1, N);
        b2 = (_Float16) (r1 / sc.a_norm[1]);
        float r2 = __builtin_fmaf(-sc.a_norm[1], (float) b2, r1);
        b3 = (_Float16) (r2 / sc.a_norm[2]);
    } else {
        b1 = (_Float16) __uint_as_float(0x7f800000u); b2 = (_Float16) 0.f; b3 = (_Float16) 0.f;   // padding: +inf
    }
    // columns: [expansion | constants], rows: [constants | expansion]
    put(3 * nd + 0, rows ? c0 : b1); put(3 * nd + 1, rows ? c1 : b2); put(3 * nd + 2, rows ? c2 : b3);
    put(3 * nd + 3, rows ? b1 : c0); put(3 * nd + 4, rows ? b2 : c1); put(3 * nd + 5, rows ? b3 : c2);
#pragma unroll
    for (int cidx = 3 * nd + 6; cidx < ks * 16; ++cidx) put(cidx, (_Float16) 0.f);
    f16x8* base = reinterpret_cast<f16x8*>(P) + ((size_t) set * (n_pad / TILE) + tile) * ks * 64;
#pragma unroll
    for (int piece = 0; piece < 2 * ks; ++piece) {
        f16x8 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = hv[piece * 8 + j];
        base[(piece >> 1) * 64 + ((piece & 1) << 5) + r] = w;
    }
    }   // sets
    if (NORMS_ONLY) {
        for (int sh = 32; sh > 0; sh >>= 1) drop = fmaxf(drop, __shfl_xor(drop, sh));
        // (a plain look first: same-address atomics from every wave would serialise in L2; a stale value only costs an atomic)
        if ((threadIdx.x & 63) == 0 && drop > 0.f && __float_as_uint(drop) > *(volatile unsigned*) drop_max) atomicMax(drop_max, __float_as_uint(drop));
    }
}

// largest finite |x - c|^2 over all rows and sets (float bits through atomicMax; values are >= 0): one atomic per block
__global__ __launch_bounds__(256) void norm_max_kernel(const float* __restrict__ nrm, size_t n, unsigned* __restrict__ out) {
    float v = 0.f;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        float t = nrm[i];
        if (t < FLT_BIG) v = fmaxf(v, t);
    }
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        if (v > 0.f) atomicMax(out, __float_as_uint(v));
    }
}

// original rows in padded (cluster-sorted) order, contiguous, for the exact rerank (padding rows are never read)
__global__ void gather_rows_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, float* __restrict__ out) {
    size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) n_pad * 33) return;
    int pos = (int) (e / 33), k = (int) (e % 33);
    int o = perm[pos];
    out[e] = o >= 0 ? X[(size_t) o * 33 + k] : 0.f;
}

// per-group maxima of sqrt(norm) (finite entries only): out[set][g]; groups are [starts[g], starts[g+1]) or, with
// starts == nullptr, fixed windows of `group` positions
__global__ void group_max_kernel(const float* __restrict__ nrm, int n_pad, int group, const int* __restrict__ starts, float* __restrict__ out) {
    int g = blockIdx.x, set = blockIdx.y, n_groups = gridDim.x;
    float m = 0.f;
    int p0 = starts ? starts[g] : g * group, p1 = starts ? starts[g + 1] : min(n_pad, (g + 1) * group);
    for (int pos = p0 + threadIdx.x; pos < p1; pos += blockDim.x) {
        float v = nrm[(size_t) set * n_pad + pos];
        if (v < FLT_BIG) m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int) (blockDim.x >> 6); ++w) m = fmaxf(m, sh[w]);
        out[(size_t) set * n_groups + g] = sqrtf(m) * 1.0000002f;   // rounded up
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3. MFMA filter kernel.  One work item = one row group (item_rb row blocks of 256) x one 4096-column chunk (32 stages of 128).
//    Wave w of row block rb owns row tile rb*8 + w; all waves share the column stage staged in LDS.
//    The column operand set is chosen per row block: Bp + blkcl[rb] * bset_stride.
//    stage_mask[rb][chunk] (optional) selects the stages to compute: bound-based skipping, section 3b.
//    Row minima are flushed per column group (tile_group[tile], a leaf of the train side) with an integer atomicMin
//    on the float bits; column minima per row group with a read-modify-write (one owner per entry).  Both tables must
//    be initialised to +inf bits, so several masked passes accumulate into the same tables.
#ifdef EXP_PROF
__device__ unsigned long long g_prof[16];
#define PROF_T(var) unsigned long long var = wall_clock64()
#define PROF_ADD(slot, a, b) do { if (tid == 0) atomicAdd(&g_prof[slot], (b) - (a)); } while (0)
#define PROF_CNT(slot) do { if (tid == 0) atomicAdd(&g_prof[slot], 1ull); } while (0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, a, b)
#define PROF_CNT(slot)
#endif
__device__ __forceinline__ f32x16 mfma_step(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_step(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <bool COLDIR, int FMT>
__global__ __launch_bounds__(NTHR, LGR_MM_OCC) void match_mfma(const typename OpFmt<FMT>::frag* __restrict__ Ap, const typename OpFmt<FMT>::frag* __restrict__ Bp,
                                                     size_t bset_stride /* fragments */, float c_scale /* 2^2s, F16 only */, float out_scale /* 2^-2s */,
                                                     const int* __restrict__ blkcl, const float* __restrict__ nA, int ma_pad, int mb_pad,
                                                     int rg_rows, const int* __restrict__ tile_group, const unsigned* __restrict__ stage_mask,
                                                     int* __restrict__ rowmin /* [n_groups][ma_pad] */,
                                                     int* __restrict__ colmin /* [ma_pad/rg_rows][mb_pad] */,
                                                     int n_cc, int item_rb, const int2* __restrict__ items, const int* __restrict__ xcd_start,
                                                     int* __restrict__ xcd_ctr) {
    // column stage double buffered in LDS: the next stage is prefetched into registers while the current one is
    // consumed and written to the other buffer afterwards -> one barrier per stage, global latency hidden
    constexpr bool F16 = FMT != FMT_F32;
    typedef typename OpFmt<FMT>::frag frag;
    constexpr int KS = OpFmt<FMT>::KS;
    constexpr int STAGE_FRAGS = STAGE_TILES * KS * 64;
    constexpr int STAGE_VEC4 = STAGE_FRAGS * (int) sizeof(frag) / 16;   // 16-byte pieces per stage
    __shared__ __attribute__((aligned(16))) frag Bs[2][STAGE_FRAGS];
    __shared__ int cmin_s[CHUNK_COLS];
    __shared__ int tg_s[CHUNK_COLS / TILE];
    __shared__ int item_s;

    // Persistent workgroups over a compacted work list.  An item is (column chunk, item_rb row blocks) with at least
    // one stage to compute.  Hardware places workgroup i on XCD i % 8; the list is partitioned per XCD (XCD x owns the
    // chunks x, x + 8, ...; items ordered by chunk, then rows), and the workgroups of an XCD pull items in order from
    // a shared counter: a chunk's B operand stays in one L2 while its items run, chunks of different cost interleave
    // across the XCDs, and nobody idles behind a static partition.  (Speed only: any item order gives the same tables.)
    const int xcd = blockIdx.x % 8;
    const int item0 = xcd_start[xcd], n_items = xcd_start[xcd + 1] - item0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int rg_blocks = rg_rows / BLOCK_ROWS;
    const int n_rb_total = ma_pad / BLOCK_ROWS;
    constexpr int IINF = 0x7f800000;   // +inf as bits
    static_assert(STAGE_VEC4 % 64 == 0, "a stage is a whole number of 1 KB DMA pieces");
    if (COLDIR) {
        for (int i = tid; i < CHUNK_COLS; i += NTHR) cmin_s[i] = IINF;   // every flush leaves the array at +inf again
    }
    int cur_cc = -1, col_tile0 = 0, n_coltiles = 0;
    unsigned full = 0u;
    unsigned tend[STAGE_TILES] = {0u, 0u, 0u, 0u};
    PROF_T(t_wg0);
    PROF_CNT(8);

  for (;;) {
    __syncthreads();   // all waves are done with the previous item (tg_s, item_s, cmin_s)
    if (tid == 0) item_s = atomicAdd(&xcd_ctr[xcd], 1);
    __syncthreads();
    const int it = item_s;
    if (it >= n_items) break;
    const int2 item = items[item0 + it];
    const int cc = item.x, rb0 = item.y;
    const int n_rb = min(item_rb, n_rb_total - rb0);
    if (cc != cur_cc) {
        cur_cc = cc;
        col_tile0 = cc * (CHUNK_COLS / TILE);
        n_coltiles = min(CHUNK_COLS / TILE, mb_pad / TILE - col_tile0);
        const int n_stages = n_coltiles / STAGE_TILES;
        full = n_stages >= 32 ? 0xffffffffu : ((1u << n_stages) - 1u);
        // column group (train leaf) of every 32-column tile of this chunk; tend[ct] bit st = tile ct of stage st is
        // the last tile of its group (uniform registers: nothing is loaded between the MFMA chains)
        if (tid < CHUNK_COLS / TILE) tg_s[tid] = tid < n_coltiles ? tile_group[col_tile0 + tid] : -1;
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < STAGE_TILES; ++ct) {
            int t = (lane & 31) * STAGE_TILES + ct;
            bool e = t < n_coltiles && (t == n_coltiles - 1 || tg_s[t + 1] != tg_s[t]);
            tend[ct] = __builtin_amdgcn_readfirstlane((unsigned) (__ballot(e) & 0xffffffffull));
        }
    }
    // the stage masks of the item's row blocks, fetched once (lane rbi holds the mask of row block rb0 + rbi)
    unsigned my_mask = full;
    if (stage_mask) my_mask = lane < n_rb ? (stage_mask[(size_t) (rb0 + lane) * n_cc + cc] & full) : 0u;
    bool col_dirty = false;

    for (int rbi = 0; rbi < n_rb; ++rbi) {
        const int rb = rb0 + rbi;
        unsigned mask = stage_mask ? __builtin_amdgcn_readlane(my_mask, rbi) : full;   // uniform over the workgroup
        if (mask) {
            PROF_T(t_v0);
            PROF_CNT(9);
            col_dirty = true;
            const int row_tile = rb * (BLOCK_ROWS / TILE) + wave * RW;
            const frag* Bset = Bp + (size_t) blkcl[rb] * bset_stride + (size_t) col_tile0 * KS * 64;
            // A fragments (coalesced 256-B loads) and the |a'|^2 of the 16 rows each lane's accumulators cover
            frag a[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a[kk] = Ap[((size_t) row_tile * KS + kk) * 64 + lane];
            f32x16 nav = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (!F16) {   // f32 operands: |a'|^2 through the accumulator input (the f16 format carries it in K slots)
#pragma unroll
                for (int g = 0; g < 16; ++g) nav[g] = nA[row_tile * TILE + (g & 3) + 8 * (g >> 2) + 4 * half];
            }
            int rmin[16];   // float bit patterns, see the epilogue note
#pragma unroll
            for (int g = 0; g < 16; ++g) rmin[g] = IINF;

            // Column stages go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers; each wave
            // instruction copies one contiguous 1 KB piece, the stage image has the same order in memory and in LDS).
            auto stage_dma = [&](int stage, int to_buf) {
                const char* src = reinterpret_cast<const char*>(Bset + (size_t) stage * STAGE_FRAGS);
                char* dst = reinterpret_cast<char*>(Bs[to_buf]);
#pragma unroll
                for (int piece = wave; piece < STAGE_VEC4 / 64; piece += WAVES)
                    __builtin_amdgcn_global_load_lds((const void*) (src + piece * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*) (dst + piece * 1024), 16, 0, 0);
            };
            // first active stage of this row block (barrier first: every wave is past the previous row block's LDS reads)
            int st = __builtin_ctz(mask);
            __syncthreads();
            stage_dma(st, 0);
            __syncthreads();   // waits for the DMA (vmcnt(0)) and makes the stage visible
            PROF_T(t_v1);
            PROF_ADD(0, t_v0, t_v1);
            // Stage loop: the DMA of the next active stage into the other buffer is issued before the current stage is
            // consumed; the barrier at the end of the stage waits for it.
            // On gfx950 the f32 MFMA runs on the FP32 lanes the VALU uses (equal peak rate; no co-execution was
            // measured: removing the epilogue saved exactly its VALU time), so the epilogue is kept minimal:
            //  * |a'|^2 enters through the accumulator input of the first MFMA step (f32) or through spare K slots
            //    (f16): d2~ = S + |a'|^2 costs nothing;
            //  * minima are taken on the bit patterns with v_min_i32 / v_min3_i32 (one instruction per slot, no
            //    canonicalising v_max pair as a float min of raw MFMA output needs).  Signed-int order equals float
            //    order except among negative values, where it keeps the one closest to zero; d2~ < 0 only within
            //    the proven error eps of a true distance >= 0, so the filtered minimum stays within eps;
            //  * a VALU lane swap instead of an LDS shuffle folds the two lane halves of the column chain.
            // The B fragment of the next tile is fetched from LDS before the epilogue runs.
            // Epilogue of one finished 32x32 tile: tile ct of stage st; nxt = the stage computed after st.
            auto row_min1 = [&](const f32x16& acc) {
#pragma unroll
                for (int g = 0; g < 16; ++g) rmin[g] = min(rmin[g], __float_as_int(acc[g]));
            };
            auto col_min = [&](const f32x16& acc, int st, int ct) {
                if (COLDIR) {
                    int cm = min(__float_as_int(acc[0]), __float_as_int(acc[1]));
#pragma unroll
                    for (int g = 2; g < 16; g += 2) cm = min(min(cm, __float_as_int(acc[g])), __float_as_int(acc[g + 1]));
                    // fold the two lane halves (rows 4*half + ...) with the VALU lane swap of gfx950
                    auto sw = __builtin_amdgcn_permlane32_swap((unsigned) cm, (unsigned) cm, false, false);
                    int other = (int) (half ? sw[0] : sw[1]);
                    cm = min(cm, other);
                    // both halves hold the folded minimum: all 64 lanes issue the LDS atomic (no exec-mask branch in
                    // the MFMA block; the two lanes of a column hit the same word with the same value)
                    atomicMin(&cmin_s[(st * STAGE_TILES + ct) * TILE + (lane & 31)], cm);
                }
            };
            auto tile_ends_group = [&](int st, int ct) {
                const unsigned te = ct == 0 ? tend[0] : ct == 1 ? tend[1] : ct == 2 ? tend[2] : tend[3];
                return ((te >> st) & 1u) != 0u;
            };
            auto maybe_flush = [&](int st, int ct, int nxt) {
                // flush the row minima when the column group (train leaf) ends, or before skipped stages
                if (tile_ends_group(st, ct) || (ct == STAGE_TILES - 1 && nxt != st + 1)) {
                    PROF_CNT(10);
                    const int grp = tg_s[st * STAGE_TILES + ct];
                    // Halving butterfly over the 32 lanes of each half wave: at every step a lane keeps half of
                    // its registers and receives the partner's copy of them, so 16 registers x 32 lanes reduce to
                    // one value per lane with 16 + 8 + 4 + 2 + 1 exchanges instead of 16 x 5; lane bits 4..1 then
                    // select the register (= row) the lane ends up holding, and one atomic instruction with 16
                    // active lanes per half wave writes all rows.
                    int w8[8], w4[4], w2[2], w1;
                    {
                        const bool up = (lane & 16) != 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            int keep = up ? rmin[8 + j] : rmin[j], send = up ? rmin[j] : rmin[8 + j];
                            w8[j] = min(keep, __shfl_xor(send, 16));
                        }
                    }
                    {
                        const bool up = (lane & 8) != 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            int keep = up ? w8[4 + j] : w8[j], send = up ? w8[j] : w8[4 + j];
                            w4[j] = min(keep, __shfl_xor(send, 8));
                        }
                    }
                    {
                        const bool up = (lane & 4) != 0;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            int keep = up ? w4[2 + j] : w4[j], send = up ? w4[j] : w4[2 + j];
                            w2[j] = min(keep, __shfl_xor(send, 4));
                        }
                    }
                    {
                        const bool up = (lane & 2) != 0;
                        int keep = up ? w2[1] : w2[0], send = up ? w2[0] : w2[1];
                        w1 = min(keep, __shfl_xor(send, 2));
                    }
                    w1 = min(w1, __shfl_xor(w1, 1));
                    // register index held by this lane: bit 3 <- lane bit 4, bit 2 <- bit 3, bit 1 <- bit 2, bit 0 <- bit 1
                    const int g = (lane >> 1) & 15;
                    if (F16) w1 = __float_as_int(__int_as_float(w1) * out_scale);   // back to d2~ (monotonic)
                    if ((lane & 1) == 0 && w1 != IINF)
                        atomicMin(&rowmin[(size_t) grp * ma_pad + row_tile * TILE + (g & 3) + 8 * (g >> 2) + 4 * half], w1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) rmin[r] = IINF;
                }
            };
            // Epilogue of one finished 32x32 tile: tile ct of stage st; nxt = the stage computed after st.
            auto epilogue = [&](const f32x16& acc, int st, int ct, int nxt) {
                row_min1(acc);
                col_min(acc, st, ct);
                maybe_flush(st, ct, nxt);
            };
            // Schedules tried and measured on a dense 400k x 400k probe (35.9 ms as is; MFMA chains alone 24.9 ms: the
            // chip holds ~1.4 GHz under this f16 MFMA load): deferring a tile's epilogue behind the next tile's MFMA
            // chain (software pipeline, with and without register double buffering of the B fragments) -1..-2 % at
            // 4 waves/SIMD with spills, +8 % at 2 waves/SIMD; no stage DMA -14 %; no barrier 0 %; no column minima -5 %;
            // two column tiles per epilogue (one v_min3 per accumulator pair for the row minima, 8 fewer vector
            // instructions per tile): +13 % at 4 waves/SIMD (spills), -1.4 % at 2 waves/SIMD.
            auto compute = [&](int st, int buf, int nxt) {
                frag b[KS];
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = Bs[buf][kk * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < STAGE_TILES; ++ct) {
                    f32x16 acc = mfma_step(a[0], b[0], nav);
#pragma unroll
                    for (int kk = 1; kk < KS; ++kk) acc = mfma_step(a[kk], b[kk], acc);
                    if (ct + 1 < STAGE_TILES) {
#pragma unroll
                        for (int kk = 0; kk < KS; ++kk) b[kk] = Bs[buf][((ct + 1) * KS + kk) * 64 + lane];
                    }
                    epilogue(acc, st, ct, nxt);
                }
            };
            mask &= mask - 1u;   // st is taken
            int buf = 0;
            while (true) {
                int nxt = -1;
                if (mask) { nxt = __builtin_ctz(mask); mask &= mask - 1u; }
                if (nxt >= 0) stage_dma(nxt, buf ^ 1);
                compute(st, buf, nxt);
                if (nxt < 0) break;
                PROF_T(t_s0);
                __syncthreads();   // DMA landed (vmcnt(0)) and visible; all waves done with the buffer refilled next
                PROF_T(t_s1);
                PROF_ADD(2, t_s0, t_s1);
                buf ^= 1;
                st = nxt;
            }
            PROF_T(t_v2);
            PROF_ADD(1, t_v1, t_v2);
        }
    }
    // column minima of this item -> table (the item covers exactly one row group: single owner, plain read-modify-write)
    if (COLDIR && col_dirty) {
        PROF_T(t_c0);
        __syncthreads();
        int rg = rb0 / rg_blocks;
        int ncols = n_coltiles * TILE;
        int* dst = colmin + (size_t) rg * mb_pad + col_tile0 * TILE;
        constexpr int NCM = CHUNK_COLS / NTHR;   // 8 columns per thread: all loads in flight before the merge
        int cur[NCM], old[NCM];
#pragma unroll
        for (int j = 0; j < NCM; ++j) {
            int i = tid + NTHR * j;
            cur[j] = i < ncols ? cmin_s[i] : IINF;
            old[j] = cur[j] != IINF ? dst[i] : IINF;
        }
#pragma unroll
        for (int j = 0; j < NCM; ++j) {
            int i = tid + NTHR * j;
            if (cur[j] != IINF) {
                int v = F16 ? __float_as_int(__int_as_float(cur[j]) * out_scale) : cur[j];
                if (v < old[j]) dst[i] = v;
                cmin_s[i] = IINF;
            }
        }
        PROF_T(t_c1);
        PROF_ADD(3, t_c0, t_c1);
    }
  }
    PROF_T(t_wg1);
    PROF_ADD(4, t_wg0, t_wg1);
}

// work list of match_mfma: flag every (XCD-major chunk, item row) that has something to compute, scan, emit
__global__ void items_flag_kernel(const unsigned* __restrict__ mask, int n_rb, int n_cc, int item_rb, int n_ir, int ccx, int* __restrict__ flags) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 8 * ccx * n_ir) return;
    int xcd = j / (ccx * n_ir), rem = j % (ccx * n_ir);
    int cc = (rem / n_ir) * 8 + xcd, ir = rem % n_ir;
    int f = 0;
    if (cc < n_cc) {
        if (!mask) f = 1;
        else
            for (int r = ir * item_rb; r < min(n_rb, (ir + 1) * item_rb); ++r) f |= mask[(size_t) r * n_cc + cc] != 0u ? 1 : 0;
    }
    flags[j] = f;
}
__global__ void items_emit_kernel(const int* __restrict__ flags, const int* __restrict__ pos, int item_rb, int n_ir, int ccx,
                                  int2* __restrict__ items, int* __restrict__ xcd_start /* [9] */) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_xcd = ccx * n_ir;
    if (j >= 8 * per_xcd) return;
    int xcd = j / per_xcd, rem = j % per_xcd;
    if (flags[j]) items[pos[j]] = make_int2((rem / n_ir) * 8 + xcd, (rem % n_ir) * item_rb);
    if (rem == 0) xcd_start[xcd] = pos[j];
    if (j == 8 * per_xcd - 1) xcd_start[8] = pos[j] + flags[j];
}

// ---------------------------------------------------------------------------------------------------------------
// 3b. bound-based stage skipping (exact).  Train rows are sorted by leaf (two-level k-means), so a column group g is a
// ball (centre c_g, radius r_g).  For a 256-row block rb,  LB(rb, g) = max(0, min_i |a_i - c_g| - r_g)  is a lower bound
// of every distance in the tile rb x g.  Pass 1 computes the NEAR_T nearest groups of each row block and the NEAR_T
// nearest row blocks of each group; from its minima every row / column gets an upper bound U of its nearest-neighbour
// distance.  Pass 2 computes the remaining tiles with LB <= max U of the block's rows or of the group's columns.  A
// skipped tile holds only pairs with d >= LB > U >= (nearest distance), so it can contain neither the nearest
// neighbour nor a tie of any row or column.  All comparisons carry relative slack far above float rounding.
constexpr float LB_SHRINK = 0.99999f, LB_GROW = 1.00001f;

// LBsq[rb][g]; +inf when the row block has no valid row or the leaf is empty
__global__ __launch_bounds__(256) void lb_kernel(const float* __restrict__ Asorted, const int* __restrict__ permA, const float* __restrict__ cen2,
                                                 const unsigned* __restrict__ r2max, const int* __restrict__ leaf_count, int n_leaves,
                                                 float* __restrict__ LBsq) {
    constexpr int ROW_LD = 34;   // even row pitch: the packed loads below stay 8-byte aligned
    __shared__ __attribute__((aligned(16))) float rows[BLOCK_ROWS * ROW_LD];
    __shared__ int okr[BLOCK_ROWS];
    const int rb = blockIdx.x;
    for (int e = threadIdx.x; e < BLOCK_ROWS * 33; e += 256) rows[(e / 33) * ROW_LD + e % 33] = Asorted[(size_t) rb * BLOCK_ROWS * 33 + e];
    okr[threadIdx.x] = permA[rb * BLOCK_ROWS + threadIdx.x] >= 0;
    __syncthreads();
    typedef float v2f __attribute__((ext_vector_type(2)));
    // |a - c|^2 on packed fp32 math (v_pk_add_f32 / v_pk_fma_f32: two coordinates per instruction, even and odd coordinates
    // in separate accumulators); any summation order is fine here, the bound carries 1e-5 of slack.  Two leaves per thread
    // and row pass: every row read from LDS feeds two centres.
    for (int g0 = threadIdx.x; g0 < n_leaves; g0 += 2 * 256) {
        const int g1 = g0 + 256;
        const bool h1 = g1 < n_leaves;
        v2f c0[16], c1[16];
        float c032, c132;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            c0[k].x = cen2[(size_t) g0 * 33 + 2 * k]; c0[k].y = cen2[(size_t) g0 * 33 + 2 * k + 1];
            c1[k].x = h1 ? cen2[(size_t) g1 * 33 + 2 * k] : 0.f; c1[k].y = h1 ? cen2[(size_t) g1 * 33 + 2 * k + 1] : 0.f;
        }
        c032 = cen2[(size_t) g0 * 33 + 32]; c132 = h1 ? cen2[(size_t) g1 * 33 + 32] : 0.f;
        float dmin0 = __uint_as_float(0x7f800000u), dmin1 = dmin0;
        for (int i = 0; i < BLOCK_ROWS; ++i) {
            if (!okr[i]) continue;
            const float* __restrict__ r = rows + i * ROW_LD;
            v2f d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma clang fp contract(fast)
                v2f a = *reinterpret_cast<const v2f*>(r + 2 * k);
                v2f t0 = a - c0[k], t1 = a - c1[k];
                d0 = t0 * t0 + d0; d1 = t1 * t1 + d1;
            }
            const float r32 = r[32];
            float t0 = r32 - c032, t1 = r32 - c132;
            dmin0 = fminf(dmin0, __builtin_fmaf(t0, t0, d0.x + d0.y));
            dmin1 = fminf(dmin1, __builtin_fmaf(t1, t1, d1.x + d1.y));
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int g = w ? g1 : g0;
            if (g >= n_leaves) break;
            const float dmin = w ? dmin1 : dmin0;
            float out = __uint_as_float(0x7f800000u);
            if (leaf_count[g] > 0 && dmin < FLT_BIG) {
                float lb = sqrtf(dmin) * LB_SHRINK - sqrtf(__uint_as_float(r2max[g])) * LB_GROW;
                lb = lb > 0.f ? lb : 0.f;
                out = lb * lb * LB_SHRINK;
            }
            LBsq[(size_t) rb * n_leaves + g] = out;
        }
    }
}

// ---- bounding-box bounds (section 3b).  A leaf's ball is a poor container in 33 dimensions; its axis-aligned box in a
// fixed orthonormal basis (the principal axes of a sample of both sets) excludes a fifth more tiles at 1M x 1M.  For a row
// block with box [amin, amax] and a leaf with box [blo, bhi] every pair is at least sqrt(sum_k gap_k^2) apart,
// gap_k = max(0, amin_k - bhi_k, blo_k - amax_k).  Any orthonormal V gives a valid bound; float rounding of the rotation
// is taken off every gap (4.1e-6 * largest |x - mu|), the rest is covered like the ball bound's roundings (LB_SHRINK).
constexpr int COV_ROWS = 384, COV_THREADS = 640;   // rows per block; 33 sums + 561 products (a <= b) + the row count = 595 workers
__global__ __launch_bounds__(COV_THREADS) void cov_kernel(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns,
                                                          float* __restrict__ part /* [blocks][34 * 33 + 1]: sums, products (a <= b), row count */) {
    __shared__ float rows[COV_ROWS * 33];
    __shared__ int okr[COV_ROWS];
    const int r0 = blockIdx.x * COV_ROWS, nr = min(COV_ROWS, ns - r0);
    for (int i = threadIdx.x; i < nr * 33; i += COV_THREADS) rows[i] = smp[(size_t) r0 * 33 + i];
    for (int i = threadIdx.x; i < nr; i += COV_THREADS) okr[i] = smp_ok[r0 + i];
    __syncthreads();
    const int w = threadIdx.x;
    if (w > 594) return;
    float* out = part + (size_t) blockIdx.x * (34 * 33 + 1);
    float acc = 0.f;
    if (w < 33) {
        for (int r = 0; r < nr; ++r) if (okr[r]) acc += rows[r * 33 + w];
        out[w] = acc;
    } else if (w < 594) {
        int p = w - 33, a = 0;
        while (p >= 33 - a) { p -= 33 - a; ++a; }   // pair (a, b = a + p)
        const int b2 = a + p;
        for (int r = 0; r < nr; ++r) if (okr[r]) acc = __builtin_fmaf(rows[r * 33 + a], rows[r * 33 + b2], acc);
        out[33 + a * 33 + b2] = acc;
    } else {
        for (int r = 0; r < nr; ++r) acc += okr[r] ? 1.f : 0.f;
        out[34 * 33] = acc;
    }
}
// block partials -> totals, summed in block order (deterministic basis, hence a deterministic tile schedule)
__global__ void cov_reduce(const float* __restrict__ part, int n_blocks, float* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > 34 * 33) return;
    float acc = 0.f;
    for (int b = 0; b < n_blocks; ++b) acc += part[(size_t) b * (34 * 33 + 1) + e];
    out[e] = acc;
}
// boxes of row segments in the basis (rows of V, y = V (x - mu)); segments: fixed 256-row blocks (starts == nullptr) or
// [starts[s], starts[s + 1]).  box[s][0..32] = min, [33..65] = max (transposed: box[c][s]); rmax2: largest |x - mu|^2 seen.
__global__ __launch_bounds__(256) void box_kernel(const float* __restrict__ Xs, const int* __restrict__ perm, const int* __restrict__ starts, int n_seg,
                                                  const float* __restrict__ V /* [33][33] */, const float* __restrict__ mu, int transposed,
                                                  float* __restrict__ box, unsigned* __restrict__ rmax2) {
    __shared__ float Vs[33 * 33 + 33];
    __shared__ float red[4][66];
    for (int i = threadIdx.x; i < 33 * 33; i += 256) Vs[i] = V[i];
    if (threadIdx.x < 33) Vs[33 * 33 + threadIdx.x] = mu[threadIdx.x];
    __syncthreads();
    const int seg = blockIdx.x;
    const int b = starts ? starts[seg] : seg * BLOCK_ROWS, e = starts ? starts[seg + 1] : (seg + 1) * BLOCK_ROWS;
    const float inf = __uint_as_float(0x7f800000u);
    float mn[33], mx[33], r2 = 0.f;
#pragma unroll
    for (int k = 0; k < 33; ++k) { mn[k] = inf; mx[k] = -inf; }
    for (int r = b + threadIdx.x; r < e; r += 256) {
        if (perm[r] < 0) continue;
        float x[33];
        float n2 = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) { x[k] = Xs[(size_t) r * 33 + k] - Vs[33 * 33 + k]; n2 = __builtin_fmaf(x[k], x[k], n2); }
        r2 = fmaxf(r2, n2);
#pragma unroll 3
        for (int k = 0; k < 33; ++k) {
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < 33; ++j) y = __builtin_fmaf(Vs[k * 33 + j], x[j], y);
            mn[k] = fminf(mn[k], y); mx[k] = fmaxf(mx[k], y);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 33; ++k) {
        float a = mn[k], c = mx[k];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); c = fmaxf(c, __shfl_xor(c, o)); }
        if (lane == 0) { red[wave][k] = a; red[wave][33 + k] = c; }
    }
    for (int o = 32; o > 0; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o));
    if (lane == 0 && r2 > 0.f && __float_as_uint(r2) > *(volatile unsigned*) rmax2) atomicMax(rmax2, __float_as_uint(r2));
    __syncthreads();
    if (threadIdx.x < 66) {
        const int c = threadIdx.x;
        float v = red[0][c];
        for (int w = 1; w < 4; ++w) v = c < 33 ? fminf(v, red[w][c]) : fmaxf(v, red[w][c]);
        box[transposed ? (size_t) c * n_seg + seg : (size_t) seg * 66 + c] = v;
    }
}
// LBsq[rb][leaf] = max(ball bound, box bound)
__global__ __launch_bounds__(256) void box_lb_kernel(const float* __restrict__ boxA /* [n_rb][66] */, const float* __restrict__ boxBt /* [66][n_leaves] */,
                                                     int n_leaves, const unsigned* __restrict__ rmax2, float* __restrict__ LBsq) {
    __shared__ float a[66];
    const int rb = blockIdx.x;
    if (threadIdx.x < 66) a[threadIdx.x] = boxA[(size_t) rb * 66 + threadIdx.x];
    __syncthreads();
    const float delta = 4.1e-6f * sqrtf(__uint_as_float(*rmax2)) * 1.01f;
    for (int g = threadIdx.x; g < n_leaves; g += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) {
            float gap = fmaxf(a[k] - boxBt[(size_t) (33 + k) * n_leaves + g], boxBt[(size_t) k * n_leaves + g] - a[33 + k]) - delta;
            if (gap > 0.f) s = __builtin_fmaf(gap, gap, s);
        }
        const size_t idx = (size_t) rb * n_leaves + g;
        const float old = LBsq[idx];
        const float lb = s * (LB_SHRINK * LB_SHRINK * LB_SHRINK);
        if (old < FLT_BIG && lb > old && lb < FLT_BIG) LBsq[idx] = lb;   // (the box alone: 19.4 % of the tiles, the ball alone 22.2 %, both 17.4 %)
    }
}

// the near_t smallest finite entries of a strided vector -> need1 = 1; ties go to the lowest index.  One 256-thread block
// per vector: the vector is read once into LDS (dynamic: len words), a bitwise radix select finds the near_t-th smallest
// key (entries are >= 0, so the float bits order like the values), then everything below it and the first ties are marked.
constexpr int NEAR_THREADS = 256;
constexpr int NEAR_LDS_MAX = 36 * 1024;   // entries that fit the dynamic LDS slab (144 KB); longer vectors are re-read from global
template <bool IN_LDS>
__global__ __launch_bounds__(NEAR_THREADS) void near_kernel(int near_t, const float* __restrict__ LBsq, int n_vec, int len, size_t vec_stride, size_t elem_stride,
                                                            uint8_t* __restrict__ need1, size_t need_vec_stride, size_t need_elem_stride) {
    extern __shared__ unsigned keys[];
    __shared__ int cnt_s, base_s;
    __shared__ int wave_cnt[NEAR_THREADS / 64];
    const int vec = blockIdx.x, tid = threadIdx.x;
    if (vec >= n_vec) return;
    constexpr unsigned INF = 0x7f800000u;
    auto load = [&](int e) {
        unsigned k = __float_as_uint(LBsq[vec * vec_stride + e * elem_stride]);
        return k > INF ? INF : k;                   // negative values / NaN cannot occur; anything odd counts as "not finite"
    };
    auto key_of = [&](int e) { return IN_LDS ? keys[e] : load(e); };
    int n_fin = 0;
    for (int e = tid; e < len; e += NEAR_THREADS) {
        unsigned k = load(e);
        if (IN_LDS) keys[e] = k;
        n_fin += k < INF ? 1 : 0;
    }
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) n_fin += __shfl_xor(n_fin, o);
    if ((tid & 63) == 0) atomicAdd(&cnt_s, n_fin);
    __syncthreads();
    int k = min(near_t, cnt_s);                     // how many to mark
    __syncthreads();
    if (k == 0) return;
    // k-th smallest key (1-based) by radix select from the top bit; `k` becomes its rank among the equal keys
    unsigned prefix = 0u;
    for (int bit = 30; bit >= 0; --bit) {           // bit 31 is 0 everywhere
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        const unsigned hi_mask = ~((2u << bit) - 1u);   // bits above `bit`
        int c0 = 0;
        for (int e = tid; e < len; e += NEAR_THREADS) {
            unsigned key = key_of(e);
            c0 += ((key & hi_mask) == prefix && !((key >> bit) & 1u)) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) c0 += __shfl_xor(c0, o);
        if ((tid & 63) == 0 && c0) atomicAdd(&cnt_s, c0);
        __syncthreads();
        const int zeros = cnt_s;
        __syncthreads();
        if (k > zeros) { k -= zeros; prefix |= 1u << bit; }
    }
    // mark the keys below the k-th ...
    for (int e = tid; e < len; e += NEAR_THREADS)
        if (key_of(e) < prefix) need1[vec * need_vec_stride + e * need_elem_stride] = 1;
    // ... and the first k entries equal to it, in index order (rows of NEAR_THREADS consecutive entries)
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int e0 = 0; e0 < len; e0 += NEAR_THREADS) {
        const int e = e0 + tid;
        const bool tie = e < len && key_of(e) == prefix;
        const unsigned long long bal = __ballot(tie);
        if ((tid & 63) == 0) wave_cnt[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = base_s;
        for (int w = 0; w < (tid >> 6); ++w) before += wave_cnt[w];
        before += __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if (tie && before < k) need1[vec * need_vec_stride + e * need_elem_stride] = 1;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < NEAR_THREADS / 64; ++w) t += wave_cnt[w]; base_s += t; }
        __syncthreads();
        if (base_s >= k) break;   // uniform
    }
}

// ---------------------------------------------------------------------------------------------------------------
// exact canonical distance: cv::hal::normL2Sqr_ (OpenCV 4.5.1, SSE baseline: 4 lanes x 4 accumulators over blocks
// of 16 floats, mul then add, reduce ((acc0+acc1)+acc2)+acc3 then (s0+s2)+(s1+s3), scalar tail) followed by sqrt.
// Must stay op-for-op identical to oracle/src/orc_matching.cpp:l2sqr33 (compiled with -ffp-contract=off).
__device__ __forceinline__ float exact_l2(const float* __restrict__ a, const float* __restrict__ b) {
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[i][l] = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                int j = 16 * blk + 4 * i + l;
                float t = a[j] - b[j];
                acc[i][l] = t * t + acc[i][l];
            }
    float s[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) s[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
    float d = (s[0] + s[2]) + (s[1] + s[3]);
    float t = a[32] - b[32];
    d = d + t * t;
    return __builtin_sqrtf(d);   // IEEE-correct sequence (NOT __fsqrt_rn, which is the 1-ulp v_sqrt_f32 on gfx950)
}

// tie rank of train index j: highest bf block first, lowest index inside a block first (smaller rank wins)
__device__ __forceinline__ unsigned tie_rank(int j, int block, int nblocks) {
    int blk = j / block;
    return (unsigned) ((nblocks - 1 - blk) * (long long) block + (j - blk * block));
}

// ---------------------------------------------------------------------------------------------------------------
// 4a. candidate groups per query.  table[g][q_pad] holds, for padded query position i and train group g, the
// filtered minimum v of d2~ = S + |a'|^2.  Proven bound of |filtered - true| for
// every pair of (query i, group g)  (DESIGN.md "matcher margin"): centring (2 roundings) + fma chain of 34 products
// + norm rounding + the column-direction add:  eps = 4 g40 (x + y)^2, g40 = 40u/(1-40u), u = 2^-24, where x, y are
// |q - c| and the group's max |t - c| for the centre c the pair was computed with.
//   upper = v + eps, lower = v - eps;  UB = min_g upper;  group g is a candidate iff lower_g <= UB + slack, with
//   slack = 1e-5 * d2(UB) (two rows whose true d2 differ by less may tie or swap in the canonical float distance)
//         + float rounding of the comparison.
struct RerankCounters { unsigned n_items; unsigned n_dense; unsigned pad0; unsigned pad1; };

// extra terms of the f16-split operand path (0 on the f32 path): eps += lin * (x + y) + abs
struct EpsExtra { float lin, abs, quad; };   // quad: multiplier of the 4 g40 (x + y)^2 term (1 on the f32 path)

template <bool ROWDIR>
__device__ __forceinline__ float group_eps(int i, int g, float xq, const float* __restrict__ nT_sets, const float* __restrict__ gmax,
                                           int n_groups, int p_of_query, const int* __restrict__ cl_of_group, int t_pad, EpsExtra ex) {
    // ROWDIR: query = row i of cluster p (xq = |a'|), train group g of columns: y = gmaxB[p][g]
    // COLDIR: query = column i, train group g = row group of cluster p(g): x = gmaxA[g], y = |b - c_p(g)| (per set)
    // evaluated in float, inflated by 1e-5 (the five roundings below are worth 3e-7): an upper bound of the proven eps
    float x, y;
    if (ROWDIR) { x = xq; y = gmax[(size_t) p_of_query * n_groups + g]; }
    else {
        int p = cl_of_group[g];
        x = gmax[g];
        y = sqrtf(nT_sets[(size_t) p * t_pad + i]) * 1.0000002f;
    }
    const float c_quad = 9.5367477e-6f * ex.quad;   // 4 g40 = 4 * 40 u / (1 - 40 u) = 9.53677e-6, rounded up
    const float s = x + y;
    return ((c_quad * s) * s + ex.lin * s + ex.abs) * 1.00001f + 1e-30f;
}

// Table scan of one query: calls f(group, value) for every computed, finite entry table[g][i].  Four loads are in flight
// before the first value is used (the loop body is short; one dependent global load per iteration was the whole cost).
// `own` (columns): the computed-flags of the query's own leaf.  A block of 256 columns can span two leaves, so the
// block's list is a superset; entries that were never computed are never initialised (init_tables_kernel) and must not
// be read.  nullptr: the list is exact (rows: one row block per workgroup) or everything was computed.
template <class F>
__device__ __forceinline__ void scan_groups(const float* __restrict__ table, size_t q_pad, int i, int n_list, int n_groups,
                                            const int* __restrict__ list_s, const uint8_t* __restrict__ own, F&& f) {
    const int n_it = n_list < 0 ? n_groups : n_list;
    const float inf = __uint_as_float(0x7f800000u);
    int k = 0;
    for (; k + 4 <= n_it; k += 4) {
        int g[4];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[j] = n_list < 0 ? k + j : list_s[k + j];
            v[j] = (!own || own[g[j]]) ? table[(size_t) g[j] * q_pad + i] : inf;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (v[j] < FLT_BIG) f(g[j], v[j]);
    }
    for (; k < n_it; ++k) {
        const int g = n_list < 0 ? k : list_s[k];
        const float v = (!own || own[g]) ? table[(size_t) g * q_pad + i] : inf;
        if (v < FLT_BIG) f(g, v);
    }
}

// Which table entries were computed at all (skipping leaves most of them at +inf): byte matrices derived from the
// done | scheduled tiles, so the table scans below read only the entries that can be finite.
//   rows:  comp_r[row block][column group]      cols:  comp_c[leaf][row group]
struct CompView { const uint8_t* m; int stride; const int* row_of_tile; };   // m == nullptr: everything was computed
__device__ __forceinline__ const uint8_t* comp_row(const CompView& c, int i, int block_row) {
    if (!c.m) return nullptr;
    int r = c.row_of_tile ? c.row_of_tile[i / TILE] : block_row;
    return c.m + (size_t) r * c.stride;
}
// compact list (dynamic LDS) of the groups computed for any query of this block; returns its length, or -1 when
// nothing was skipped (iterate all groups).  Every thread of the block must call it.
__device__ __forceinline__ int comp_list(const CompView& c, int i0, int n_i, int n_groups, int* list_s, int span = 0, bool* several = nullptr) {
    __shared__ int cnt_s;
    if (several) *several = false;
    if (!c.m) return -1;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    const int t0 = i0 / TILE, t1 = (min(i0 + (span ? span : (int) blockDim.x), n_i) - 1) / TILE;
    if (several && c.row_of_tile) *several = c.row_of_tile[t0] != c.row_of_tile[t1];   // tiles are sorted by leaf
    for (int g = threadIdx.x; g < n_groups; g += blockDim.x) {
        uint8_t f = 0;
        if (c.row_of_tile) {
            int prev = -1;
            for (int t = t0; t <= t1; ++t) { int r = c.row_of_tile[t]; if (r != prev) { f |= c.m[(size_t) r * c.stride + g]; prev = r; } }
        } else f = c.m[(size_t) (i0 / BLOCK_ROWS) * c.stride + g];
        if (f) list_s[atomicAdd(&cnt_s, 1)] = g;
    }
    __syncthreads();
    return cnt_s;
}
__global__ void comp_rows_kernel(const uint8_t* __restrict__ done, const uint8_t* __restrict__ sched, const int* __restrict__ group_leaf,
                                 int n_rb, int n_leaves, int n_groups, uint8_t* __restrict__ comp_r) {
    size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t) n_rb * n_groups) return;
    int rb = (int) (idx / n_groups), g = (int) (idx % n_groups);
    size_t t = (size_t) rb * n_leaves + group_leaf[g];
    comp_r[idx] = done[t] | sched[t];
}
__global__ void comp_cols_kernel(const uint8_t* __restrict__ done, const uint8_t* __restrict__ sched, int n_rb, int n_leaves, int n_rg,
                                 int rg_blocks, uint8_t* __restrict__ comp_c) {
    size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t) n_leaves * n_rg) return;
    int l = (int) (idx / n_rg), rg = (int) (idx % n_rg);
    uint8_t v = 0;
    for (int rb = rg * rg_blocks; rb < min(n_rb, (rg + 1) * rg_blocks); ++rb) v |= done[(size_t) rb * n_leaves + l] | sched[(size_t) rb * n_leaves + l];
    comp_c[idx] = v;
}

// upper bounds after a masked pass (section 3b): largest over the row block / the leaf of  min_g (filtered + eps)
__global__ __launch_bounds__(BLOCK_ROWS) void row_u_kernel(const float* __restrict__ table, int n_groups, int q_pad, const int* __restrict__ permQ,
                                                            const float* __restrict__ nQ, const int* __restrict__ blkclQ,
                                                            const float* __restrict__ gmax, EpsExtra ex, CompView comp, float* __restrict__ u_rb) {
    extern __shared__ int list_s[];
    const int i = blockIdx.x * BLOCK_ROWS + threadIdx.x;
    const int n_list = comp_list(comp, blockIdx.x * BLOCK_ROWS, q_pad, n_groups, list_s);
    float ub = -1.f;   // padding rows need nothing
    if (i < q_pad && permQ[i] >= 0) {
        int p = blkclQ[blockIdx.x];
        float xq = sqrtf(nQ[i]) * 1.0000002f;
        ub = __uint_as_float(0x7f800000u);
        scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, nullptr, [&](int g, float v) {
            float e = group_eps<true>(i, g, xq, nullptr, gmax, n_groups, p, nullptr, q_pad, ex);
            ub = fminf(ub, v + e);
        });
    }
    for (int o = 32; o > 0; o >>= 1) ub = fmaxf(ub, __shfl_xor(ub, o));
    __shared__ float sh[BLOCK_ROWS / 64];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ub;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLOCK_ROWS / 64; ++w) ub = fmaxf(ub, sh[w]);
        u_rb[blockIdx.x] = ub;
    }
}
__global__ void col_u_kernel(const float* __restrict__ table, int n_rg, int t_pad, const int* __restrict__ permT,
                             const float* __restrict__ nT_sets, const float* __restrict__ gmaxA, const int* __restrict__ cl_of_rg,
                             const int* __restrict__ tile_group, EpsExtra ex, CompView comp, unsigned* __restrict__ u_leaf /* float bits, >= 0 */) {
    extern __shared__ int list_s[];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    bool several;
    const int n_list = comp_list(comp, blockIdx.x * blockDim.x, t_pad, n_rg, list_s, 0, &several);
    float ub = 0.f;   // padding columns need nothing
    if (j < t_pad && permT[j] >= 0) {
        ub = __uint_as_float(0x7f800000u);
        scan_groups(table, (size_t) t_pad, j, n_list, n_rg, list_s, several ? comp_row(comp, j, 0) : nullptr, [&](int g, float v) {
            float e = group_eps<false>(j, g, 0.f, nT_sets, gmaxA, n_rg, 0, cl_of_rg, t_pad, ex);
            ub = fminf(ub, v + e);
        });
        ub = ub > 0.f ? ub : 0.f;
    }
    // the 32 columns of a tile share a leaf: one atomic per tile (t_pad is a multiple of the block size, so whole waves get here)
    for (int o = 16; o > 0; o >>= 1) ub = fmaxf(ub, __shfl_xor(ub, o));
    if ((threadIdx.x & 31) == 0 && j < t_pad && ub > 0.f) atomicMax(&u_leaf[tile_group[j / TILE]], __float_as_uint(ub));
}
// +inf for the table entries a masked pass is about to compute for the first time (the tables hold 10 GB at 1M x 1M and
// only a fifth of them is ever computed or read: no blanket fill).  Row table: (group of a newly scheduled leaf, the 256
// rows of the block).  Column table: the columns of the leaf in the block's row group, written by the lowest newly
// scheduled block of the group unless an earlier pass already computed that (row group, leaf).  Entries that only a
// boundary stage touches (a stage is computed when any leaf it overlaps is scheduled) may hold anything: nothing reads
// them until their own (block, leaf) is scheduled, and that initialises them here.
__global__ __launch_bounds__(BLOCK_ROWS) void init_tables_kernel(const uint8_t* __restrict__ sched, const uint8_t* __restrict__ done, int n_rb, int n_leaves,
                                                                  const int* __restrict__ leaf_g0 /* [n_leaves + 1] */, const int* __restrict__ group_start,
                                                                  int rg_blocks, int* __restrict__ rowmin, size_t ma_pad, int* __restrict__ colmin, size_t mb_pad) {
    __shared__ uint8_t s_s[MAXLEAF];   // 0: nothing, 1: rows only, 3: rows and columns
    const int rb = blockIdx.x, tid = threadIdx.x;
    const int rg = rb / rg_blocks, rb_lo = rg * rg_blocks, rb_hi = min(n_rb, rb_lo + rg_blocks);
    for (int l = tid; l < n_leaves; l += BLOCK_ROWS) {
        uint8_t f = sched[(size_t) rb * n_leaves + l] ? 1 : 0;
        if (f && colmin) {
            bool first = true;
            for (int r = rb_lo; r < rb_hi; ++r) {
                if (done[(size_t) r * n_leaves + l]) first = false;
                if (r < rb && sched[(size_t) r * n_leaves + l]) first = false;
            }
            if (first) f = 3;
        }
        s_s[l] = f;
    }
    __syncthreads();
    constexpr int IINF = 0x7f800000;
    for (int l = 0; l < n_leaves; ++l) {
        const uint8_t f = s_s[l];
        if (!f) continue;
        const int g0 = leaf_g0[l], g1 = leaf_g0[l + 1];
        for (int g = g0; g < g1; ++g) rowmin[(size_t) g * ma_pad + (size_t) rb * BLOCK_ROWS + tid] = IINF;
        if ((f & 2) && g0 < g1)
            for (int col = group_start[g0] + tid; col < group_start[g1]; col += BLOCK_ROWS) colmin[(size_t) rg * mb_pad + col] = IINF;
    }
}

// tile scheduling of one pass (section 3b).  sched_kernel: tiles of the previous pass become done; a tile not yet
// done is scheduled when  LBsq <= beta_sq * U  of its row block or (both directions) of its leaf.  mask_kernel turns the
// scheduled (row block, leaf) tiles into stage masks: a stage is computed when any leaf it overlaps is scheduled.
struct MaskStats { unsigned long long stages[8]; };
__global__ void sched_kernel(int both, float beta_sq, const float* __restrict__ LBsq, const float* __restrict__ u_rb,
                             const unsigned* __restrict__ u_leaf, int n_rb, int n_leaves, uint8_t* __restrict__ done, uint8_t* __restrict__ sched) {
    const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t) n_rb * n_leaves) return;
    const int rb = (int) (idx / n_leaves), g = (int) (idx % n_leaves);
    uint8_t d = done[idx] | sched[idx];
    done[idx] = d;
    uint8_t s = 0;
    if (!d) {
        float lb = LBsq[idx], urb = u_rb[rb];
        bool need = urb >= 0.f && lb <= beta_sq * (urb * LB_GROW + 1e-12f);
        if (both) { float ug = __uint_as_float(u_leaf[g]); need = need || lb <= beta_sq * (ug * LB_GROW + 1e-12f); }
        s = need ? 1 : 0;
    }
    sched[idx] = s;
}
__global__ void mask_kernel(int pass, const uint8_t* __restrict__ sched, const int* __restrict__ tile_group,
                            int n_rb, int n_cc, int n_leaves, int n_stage_total, unsigned* __restrict__ mask, MaskStats* __restrict__ stats) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned m = 0u;
    if (idx < n_rb * n_cc) {
        const int rb = idx / n_cc, cc = idx % n_cc;
        for (int s = 0; s < STAGES_PER_CHUNK; ++s) {
            int gst = cc * STAGES_PER_CHUNK + s;
            if (gst >= n_stage_total) break;
            bool on = false;
            int gprev = -1;
            for (int ct = 0; ct < STAGE_TILES; ++ct) {
                int g = tile_group[gst * STAGE_TILES + ct];
                if (g == gprev) continue;
                gprev = g;
                on = on || sched[(size_t) rb * n_leaves + g] != 0;
            }
            if (on) m |= 1u << s;
        }
        mask[idx] = m;
    }
    unsigned c = __popc(m);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&stats->stages[pass], (unsigned long long) c);
}

// Self-check of the filter bound (LGR_MATCH_CHECK=1, test sizes only): for sampled queries and every computed group,
// |filtered minimum - exact minimum of the squared distance (double)| / eps, maximised through atomicMax on the float
// bits.  eps is a proven bound, so the ratio must stay <= 1; tests assert it on both operand formats.
template <bool ROWDIR>
__global__ void check_kernel(const float* __restrict__ table, int n_groups, int q_pad, int group_size, const int* __restrict__ starts,
                             const float* __restrict__ Qsorted, const int* __restrict__ permQ, const float* __restrict__ Tsorted,
                             const int* __restrict__ permT, int t_pad, const float* __restrict__ nQ, const int* __restrict__ blkclQ,
                             const float* __restrict__ nQ_sets, const float* __restrict__ gmax, const int* __restrict__ cl_of_group,
                             EpsExtra ex, CompView comp, int stride, const uint8_t* __restrict__ done, const uint8_t* __restrict__ sched,
                             int n_leaves, unsigned* __restrict__ worst) {
    extern __shared__ int list_s[];
    const int i = blockIdx.x * stride;   // sampled padded query position
    // the computed groups of the row block (rows) / of the leaf (columns) this query lives in
    const int n_list = comp_list(comp, ROWDIR ? (i / BLOCK_ROWS) * BLOCK_ROWS : (i / TILE) * TILE, q_pad, n_groups, list_s, ROWDIR ? BLOCK_ROWS : TILE);
    // columns: a row group is computed row block by row block; the table holds the minimum over the computed ones only
    const int my_leaf = (!ROWDIR && done && i < q_pad) ? comp.row_of_tile[i / TILE] : -1;
    if (i >= q_pad || permQ[i] < 0) return;
    const int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    const float xq = ROWDIR ? sqrtf(nQ[i]) * 1.0000002f : 0.f;
    float q[33];
    for (int k = 0; k < 33; ++k) q[k] = Qsorted[(size_t) i * 33 + k];
    for (int kk = 0; kk < (n_list < 0 ? n_groups : n_list); ++kk) {
        const int g = n_list < 0 ? kk : list_s[kk];
        const float v = table[(size_t) g * q_pad + i];
        const int j0 = starts ? starts[g] : g * group_size, j1 = starts ? starts[g + 1] : min(t_pad, j0 + group_size);
        double best = 1e300;
        for (int j = j0 + (int) threadIdx.x; j < j1; j += blockDim.x) {
            if (permT[j] < 0) continue;
            if (my_leaf >= 0) {
                size_t t = (size_t) (j / BLOCK_ROWS) * n_leaves + my_leaf;
                if (!(done[t] | sched[t])) continue;
            }
            double d = 0;
            for (int k = 0; k < 33; ++k) { double t = (double) q[k] - (double) Tsorted[(size_t) j * 33 + k]; d += t * t; }
            best = d < best ? d : best;
        }
        for (int o = 32; o > 0; o >>= 1) { double other = __shfl_xor(best, o); best = other < best ? other : best; }
        __shared__ double sh[4];
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int) (blockDim.x >> 6); ++w) best = sh[w] < best ? sh[w] : best;
            if (best < 1e299) {   // the group has valid rows: the table entry must be finite and within eps
                float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
                float ratio = (v < FLT_BIG) ? (float) (fabs((double) v - best) / (double) e) : 1e30f;
                atomicMax(worst, __float_as_uint(ratio));
            }
        }
    }
}

constexpr int CAND_KEEP = 4;   // smallest lower bounds kept per query by rerank_count (up to CAND_KEEP - 1 candidates without a rescan)
template <bool ROWDIR>
__global__ void rerank_count(const float* __restrict__ table, int n_groups, int q_pad, const int* __restrict__ permQ,
                             const float* __restrict__ nQ /* ROWDIR: |a'|^2 per padded row */, const int* __restrict__ blkclQ,
                             const float* __restrict__ nQ_sets /* COLDIR: |b - c_p|^2 [KCL][q_pad] */, const float* __restrict__ gmax,
                             const int* __restrict__ cl_of_group, int dense_limit /* < 0: every query takes the dense path */, EpsExtra ex, CompView comp,
                             float* __restrict__ thr_out, int* __restrict__ counts, int* __restrict__ cand /* [q_pad][CAND_KEEP] */,
                             unsigned* __restrict__ dense, RerankCounters* __restrict__ cnt) {
    extern __shared__ int list_s[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool several;
    const int n_list = comp_list(comp, blockIdx.x * blockDim.x, q_pad, n_groups, list_s, 0, &several);
    if (i >= q_pad) return;
    counts[i] = 0;
    int o = permQ[i];
    if (o < 0) return;
    if (dense_limit < 0) {   // the filter is not usable for this call (a centred norm overflows float): exact brute force
        unsigned pos = atomicAdd(&cnt->n_dense, 1u);
        dense[pos] = (unsigned) o;
        return;
    }
    int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    float nq = ROWDIR ? nQ[i] : 0.f;
    float xq = ROWDIR ? sqrtf(nq) * 1.0000002f : 0.f;
    // one scan: the smallest upper bound, and the CAND_KEEP smallest lower bounds with their groups.  Candidates are the
    // groups whose lower bound does not exceed thr (derived from the smallest upper bound); there is about one per query,
    // so they are almost always among the kept ones and neither a second scan here nor one in rerank_emit is needed.
    float ub = __uint_as_float(0x7f800000u);
    float lo[CAND_KEEP];
    int lg[CAND_KEEP];
#pragma unroll
    for (int j = 0; j < CAND_KEEP; ++j) { lo[j] = __uint_as_float(0x7f800000u); lg[j] = -1; }
    const uint8_t* own = (ROWDIR || !several) ? nullptr : comp_row(comp, i, 0);   // the list is exact unless the block spans two leaves
    scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, own, [&](int g, float v) {
        float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
        ub = fminf(ub, v + e);
        float l = v - e;
        int gi = g;
        if (l < lo[CAND_KEEP - 1]) {
#pragma unroll
            for (int j = 0; j < CAND_KEEP; ++j)
                if (l < lo[j]) { float tl = lo[j]; int tg = lg[j]; lo[j] = l; lg[j] = gi; l = tl; gi = tg; }
        }
    });
    if (!(ub < FLT_BIG)) return;     // no valid train row at all
    double d2 = fmax((double) ub, 0.0);   // both tables hold d2~ = S + |a'|^2
    float thr = (float) ((double) ub + 1e-5 * d2 + 8.0 * 5.9604644775390625e-8 * fabs((double) ub) + 1e-30);
    if (thr < ub) thr = ub;
    int nc = 0;
    if (lo[CAND_KEEP - 1] <= thr) {
        // the kept list may be incomplete: count by a second scan, rerank_emit rescans too (cand[0] = -1)
        scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, own, [&](int g, float v) {
            float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
            nc += (v - e <= thr) ? 1 : 0;
        });
        cand[(size_t) i * CAND_KEEP] = -1;
    } else {
#pragma unroll
        for (int j = 0; j < CAND_KEEP - 1; ++j)
            if (lo[j] <= thr) { cand[(size_t) i * CAND_KEEP + nc] = lg[j]; ++nc; }
    }
    if (nc > dense_limit) {
        unsigned pos = atomicAdd(&cnt->n_dense, 1u);
        dense[pos] = (unsigned) o;
        return;
    }
    thr_out[i] = thr;
    counts[i] = nc;
}

template <bool ROWDIR>
__global__ void rerank_emit(const float* __restrict__ table, int n_groups, int q_pad, const float* __restrict__ nQ,
                            const int* __restrict__ blkclQ, const float* __restrict__ nQ_sets, const float* __restrict__ gmax,
                            const int* __restrict__ cl_of_group, EpsExtra ex, CompView comp, const float* __restrict__ thr_in,
                            const int* __restrict__ counts, const int* __restrict__ cand, const int* __restrict__ offs,
                            unsigned* __restrict__ item_q, unsigned* __restrict__ item_g) {
    extern __shared__ int list_s[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool several;
    const int n_list = comp_list(comp, blockIdx.x * blockDim.x, q_pad, n_groups, list_s, 0, &several);
    if (i >= q_pad || counts[i] == 0) return;
    int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    float xq = ROWDIR ? sqrtf(nQ[i]) * 1.0000002f : 0.f;
    float thr = thr_in[i];
    int pos = offs[i];
    if (cand[(size_t) i * CAND_KEEP] >= 0) {   // the candidates rerank_count kept
        for (int j = 0; j < counts[i]; ++j) { item_q[pos] = (unsigned) i; item_g[pos] = (unsigned) cand[(size_t) i * CAND_KEEP + j]; ++pos; }
        return;
    }
    scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, (ROWDIR || !several) ? nullptr : comp_row(comp, i, 0), [&](int g, float v) {
        float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
        if (v - e <= thr) { item_q[pos] = (unsigned) i; item_g[pos] = (unsigned) g; ++pos; }
    });
}

// 4b. exact distances of the (query position, train group) items, sorted by group: a workgroup takes 256 consecutive
// items (almost always one group) and every thread scans the group's train rows for its own query with the canonical
// distance.  The train row address is wave uniform (made explicit with readfirstlane), so the rows arrive through the
// scalar cache as SGPR operands of the VALU ops: no vector loads, no LDS in the inner loop.
constexpr int RQ_THREADS = 256;
__global__ __launch_bounds__(RQ_THREADS) void rerank_grouped(const float* __restrict__ Q, const int* __restrict__ permQ,
                                                             const float* __restrict__ Tsorted, const int* __restrict__ permT, int t_pad,
                                                             int group_size, const int* __restrict__ starts /* variable groups, or nullptr */,
                                                             int block, int nblocks, const unsigned* __restrict__ item_g,
                                                             const unsigned* __restrict__ item_q, unsigned n_items,
                                                             unsigned long long* __restrict__ best) {
    __shared__ unsigned next_g;
    const int tid = threadIdx.x;
    const unsigned idx = blockIdx.x * RQ_THREADS + tid;
    const bool act = idx < n_items;
    const unsigned g = act ? item_g[idx] : 0xffffffffu;
    const int qo = act ? permQ[item_q[idx]] : -1;
    float q[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) q[k] = act ? Q[(size_t) qo * 33 + k] : 0.f;
    unsigned long long bk = ~0ull;
    unsigned cur = item_g[blockIdx.x * RQ_THREADS];   // items are sorted: the first one has the smallest group
    while (cur != 0xffffffffu) {
        const int j0 = __builtin_amdgcn_readfirstlane(starts ? starts[cur] : (int) cur * group_size);
        const int j1 = __builtin_amdgcn_readfirstlane(starts ? starts[cur + 1] : min(t_pad, j0 + group_size));
        // waves without an item of this group skip it (wave-uniform branch)
        if (__ballot(act && g == cur) != 0ull) {
            for (int j = j0; j < j1; ++j) {
                const int to = __builtin_amdgcn_readfirstlane(permT[j]);
                if (to < 0) continue;                // padding
                const float* __restrict__ tp = Tsorted + (size_t) j * 33;   // wave-uniform address -> scalar loads
                float t[33];
#pragma unroll
                for (int k = 0; k < 33; ++k) t[k] = tp[k];
                float d = exact_l2(q, t);
                if (act && g == cur && d < FLT_BIG) {   // batchDistance keeps only d < FLT_MAX
                    unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(to, block, nblocks);
                    bk = key < bk ? key : bk;
                }
            }
        }
        __syncthreads();
        if (tid == 0) next_g = 0xffffffffu;
        __syncthreads();
        if (act && g > cur) atomicMin(&next_g, g);
        __syncthreads();
        cur = next_g;
    }
    if (act && bk != ~0ull) atomicMin(&best[qo], bk);
}

// 4c. dense fallback (degenerate data: more than half of all groups qualify, e.g. huge sets of identical rows):
// plain exact brute force over the original train rows, parallel over (256 dense queries) x (column chunk).
constexpr int DENSE_CHUNK = 8192;
__global__ __launch_bounds__(256) void rerank_dense(const float* __restrict__ Q, const float* __restrict__ T,
                                                    const uint8_t* __restrict__ validT, int nt, int block, int nblocks,
                                                    const unsigned* __restrict__ dense, unsigned n_dense,
                                                    unsigned long long* __restrict__ best) {
    __shared__ float Ts[64 * 33];
    __shared__ uint8_t vTs[64];
    int c0 = blockIdx.y * DENSE_CHUNK, c1 = min(nt, c0 + DENSE_CHUNK);
    for (unsigned base = blockIdx.x * 256; base < n_dense; base += gridDim.x * 256) {
        unsigned di = base + threadIdx.x;
        bool act = di < n_dense;
        unsigned qi = act ? dense[di] : 0;
        float q[33];
#pragma unroll
        for (int k = 0; k < 33; ++k) q[k] = act ? Q[(size_t) qi * 33 + k] : 0.f;
        unsigned long long bk = ~0ull;
        for (int j0 = c0; j0 < c1; j0 += 64) {
            __syncthreads();
            int nj = min(64, c1 - j0);
            for (int i = threadIdx.x; i < nj * 33; i += 256) Ts[i] = T[(size_t) j0 * 33 + i];
            if (threadIdx.x < nj) vTs[threadIdx.x] = validT[j0 + threadIdx.x];
            __syncthreads();
            if (act) {
                for (int jj = 0; jj < nj; ++jj) {
                    if (!vTs[jj]) continue;
                    float d = exact_l2(q, Ts + jj * 33);
                    if (!(d < FLT_BIG)) continue;
                    unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(j0 + jj, block, nblocks);
                    bk = key < bk ? key : bk;
                }
            }
        }
        if (act && bk != ~0ull) atomicMin(&best[qi], bk);
        __syncthreads();
    }
}

__global__ void fill_u64(unsigned long long* __restrict__ p, int n, unsigned long long v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void rerank_finalize(const unsigned long long* __restrict__ best, int nq, int block, int nblocks,
                                int32_t* __restrict__ idx, float* __restrict__ dist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    unsigned long long k = best[i];
    if (k == ~0ull) { idx[i] = -1; dist[i] = 0.f; return; }
    unsigned rank = (unsigned) (k & 0xffffffffu);
    int qb = rank / block, r = rank % block;
    int blk = nblocks - 1 - qb;
    idx[i] = blk * block + r;
    dist[i] = __uint_as_float((unsigned) (k >> 32));
}

int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// one side (A or B) after clustering
struct Side {
    int m = 0, n_valid = 0, n_pad = 0;
    int* perm = nullptr;          // [n_pad] padded position -> original row or -1
    uint8_t* valid = nullptr;     // [m]
    int* blkcl = nullptr;         // [n_pad / 256] cluster of each 256-row block (device)
    int* leaf_start = nullptr;    // [n_leaves + 1] padded start of every leaf (device); leaf l covers [start[l], start[l+1])
    int* leaf_count = nullptr;    // [MAXLEAF + 1] valid rows per leaf (device; [MAXLEAF] = invalid rows)
    unsigned* r2max = nullptr;    // [MAXLEAF] squared leaf radius bits (device)
    std::vector<int> h_blkcl;     // host copies
    std::vector<int> h_leaf_start;
};

// assign + sort + place one side.  Leaves start at multiples of leaf_unit, clusters at multiples of cluster_unit
// (a multiple of 256 and of leaf_unit); padding positions carry perm = -1.
int build_side(lgr_ctx* ctx, const float* d_x, int m, const float* cen, const float* cen2, int sub, int leaf_unit, int cluster_unit,
               int ws_keys, int ws_perm, Side* s) {
    s->m = m;
    const int n_leaves = KCL * sub;
    unsigned *keys, *keys2;
    int *vals, *vals2;
    char* kbuf;
    size_t body = (((size_t) m * 17 + 255) & ~(size_t) 255);
    LGR_TRY(lgr_ws_t(ctx, ws_keys, body + 16384, &kbuf));
    keys = (unsigned*) kbuf; keys2 = keys + m; vals = (int*) (keys2 + m); vals2 = vals + m;
    s->valid = (uint8_t*) (vals2 + m);
    int* counts = (int*) (kbuf + body);               // [MAXLEAF + 1]
    unsigned* rmax = (unsigned*) (kbuf + body + 8192);   // [MAXLEAF]
    s->leaf_count = counts; s->r2max = rmax;
    LGR_HIP(ctx, hipMemsetAsync(counts, 0, 16384, ctx->stream));
    const size_t assign_lds = ((size_t) KCL * (sub * 33 + 1) + 2 * MAXLEAF + 1) * 4;
    if (assign_lds > 64 * 1024) LGR_HIP(ctx, hipFuncSetAttribute((const void*) assign_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) assign_lds));
    assign_kernel<<<cdiv(m, ASSIGN_THREADS), ASSIGN_THREADS, assign_lds, ctx->stream>>>(d_x, m, cen, cen2, sub, keys, vals, s->valid, counts, rmax);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, keys, keys2, vals, vals2, (size_t) m, 0, 32, ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::radix_sort_pairs(tmp, tb, keys, keys2, vals, vals2, (size_t) m, 0, 32, ctx->stream));
    int* h;
    LGR_TRY(lgr_pinned(ctx, 8192, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, counts, (MAXLEAF + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> starts(2 * (size_t) MAXLEAF + 2, 0);   // [0..MAXLEAF): sorted start, [MAXLEAF..2*MAXLEAF]: padded start
    int acc = 0, pacc = 0;
    s->h_blkcl.clear();
    s->h_leaf_start.assign(n_leaves + 1, 0);
    for (int c = 0; c < KCL; ++c) {
        int cluster_begin = pacc;
        for (int j = 0; j < sub; ++j) {
            int l = c * sub + j;
            starts[l] = acc; starts[MAXLEAF + l] = pacc;
            s->h_leaf_start[l] = pacc;
            acc += h[l];
            pacc += pad_to(h[l], leaf_unit);
        }
        pacc = pad_to(pacc, cluster_unit);
        for (int b = 0; b < (pacc - cluster_begin) / BLOCK_ROWS; ++b) s->h_blkcl.push_back(c);
    }
    s->h_leaf_start[n_leaves] = pacc;
    starts[MAXLEAF + n_leaves] = pacc;
    s->n_valid = acc; s->n_pad = pacc;
    if (s->n_pad == 0) return LGR_OK;
    int* pbuf;
    LGR_TRY(lgr_ws_t(ctx, ws_perm, (size_t) s->n_pad + s->h_blkcl.size() + starts.size() + 64, &pbuf));
    s->perm = pbuf; s->blkcl = pbuf + s->n_pad;
    int* d_starts = s->blkcl + s->h_blkcl.size();
    s->leaf_start = d_starts + MAXLEAF;
    LGR_HIP(ctx, hipMemsetAsync(s->perm, 0xff, (size_t) s->n_pad * 4, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(s->blkcl, s->h_blkcl.data(), s->h_blkcl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host staging buffers go out of scope
    if (s->n_valid) place_kernel<<<cdiv(s->n_valid, 256), 256, 0, ctx->stream>>>(keys2, vals2, s->n_valid, d_starts, d_starts + MAXLEAF, s->perm);
    return LGR_OK;
}

template <bool ROWDIR>
int run_rerank(lgr_ctx* ctx, EpsExtra ex, CompView comp, const float* table, int n_groups, int group_size, const int* starts, const float* Q, const Side& qs,
               const float* nQ, const float* nQ_sets, const float* gmax, const int* cl_of_group,
               const float* T, const float* Tsorted, const Side& ts, int block, unsigned long long* best, int32_t* d_idx, float* d_dist,
               unsigned* stat_items, unsigned* stat_dense, bool force_dense) {
    const int q_pad = qs.n_pad;
    unsigned* dense;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_DENSE, (size_t) q_pad * (4 + CAND_KEEP) + 64, &dense));
    float* thr = (float*) (dense + q_pad);
    int* counts = (int*) (dense + 2 * (size_t) q_pad);
    int* offs = (int*) (dense + 3 * (size_t) q_pad);
    int* cand = (int*) (dense + 4 * (size_t) q_pad);
    char* misc;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, 4096, &misc));
    RerankCounters* cnt = (RerankCounters*) (misc + 64);
    LGR_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(RerankCounters), ctx->stream));
    int nblocks = (ts.m + block - 1) / block;
    int dense_limit = force_dense ? -1 : std::max(64, n_groups / 2);
    rerank_count<ROWDIR><<<cdiv(q_pad, 256), 256, (size_t) (n_groups + 8) * 4, ctx->stream>>>(table, n_groups, q_pad, qs.perm, nQ, qs.blkcl, nQ_sets, gmax,
                                                                   cl_of_group, dense_limit, ex, comp, thr, counts, cand, dense, cnt);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, counts, offs, 0, (size_t) q_pad, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, counts, offs, 0, (size_t) q_pad, rocprim::plus<int>(), ctx->stream));
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, offs + (q_pad - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, counts + (q_pad - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 2, cnt, sizeof(RerankCounters), hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned n_items = (unsigned) (h[0] + h[1]);
    unsigned n_dense = ((RerankCounters*) (h + 2))->n_dense;
    *stat_items = n_items; *stat_dense = n_dense;
    if (n_items) {
        unsigned* ib;
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ITEMS, (size_t) 4 * n_items + 64, &ib));
        unsigned *item_q = ib, *item_g = ib + n_items, *item_q2 = ib + 2 * (size_t) n_items, *item_g2 = ib + 3 * (size_t) n_items;
        rerank_emit<ROWDIR><<<cdiv(q_pad, 256), 256, (size_t) (n_groups + 8) * 4, ctx->stream>>>(table, n_groups, q_pad, nQ, qs.blkcl, nQ_sets, gmax, cl_of_group,
                                                                      ex, comp, thr, counts, cand, offs, item_q, item_g);
        int bits = 1;
        while ((1 << bits) < n_groups) ++bits;
        size_t sb = 0;
        LGR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, sb, item_g, item_g2, item_q, item_q2, (size_t) n_items, 0, bits, ctx->stream));
        void* stmp;
        LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, sb, &stmp));
        LGR_HIP(ctx, rocprim::radix_sort_pairs(stmp, sb, item_g, item_g2, item_q, item_q2, (size_t) n_items, 0, bits, ctx->stream));
        rerank_grouped<<<cdiv(n_items, RQ_THREADS), RQ_THREADS, 0, ctx->stream>>>(Q, qs.perm, Tsorted, ts.perm, ts.n_pad, group_size, starts, block, nblocks,
                                                                                 item_g2, item_q2, n_items, best);
    }
    if (n_dense) {
        dim3 g(std::min(cdiv(n_dense, 256), 64), cdiv(ts.m, DENSE_CHUNK));
        rerank_dense<<<g, 256, 0, ctx->stream>>>(Q, T, ts.valid, ts.m, block, nblocks, dense, n_dense, best);
    }
    rerank_finalize<<<cdiv(qs.m, 256), 256, 0, ctx->stream>>>(best, qs.m, block, nblocks, d_idx, d_dist);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

}  // namespace

// statistics of the last match call (bench/diagnostics): candidate (query, group) items and dense-fallback queries
// per direction, group counts, and the column stages the MFMA passes executed out of all (row block, stage) pairs
struct lgr_match_stats { unsigned items_ab, dense_ab, items_ba, dense_ba; int sub_cols, rg_rows; double stages_done, stages_all; int f16; };
// diagnostics of the calling thread's last match call (one context per host thread)
static thread_local lgr_match_stats g_last_stats;
static thread_local double g_last_check[2] = {-1, -1};
extern "C" int lgr_match_last_stats(unsigned* out6) {
    out6[0] = g_last_stats.items_ab; out6[1] = g_last_stats.dense_ab; out6[2] = g_last_stats.items_ba;
    out6[3] = g_last_stats.dense_ba; out6[4] = (unsigned) g_last_stats.sub_cols; out6[5] = (unsigned) g_last_stats.rg_rows;
    return LGR_OK;
}
// fraction of the (row block x column stage) tiles of the last match call that the MFMA passes computed (1 = dense)
extern "C" int lgr_match_last_work(double* executed_fraction) {
    if (!executed_fraction) return LGR_ERR_INVALID_ARG;
    *executed_fraction = g_last_stats.stages_all > 0 ? g_last_stats.stages_done / g_last_stats.stages_all : 1.0;
    return LGR_OK;
}

// MFMA operand format of the last match call: 1 = f16-split operands on v_mfma_f32_32x32x16_f16 (224 FLOP per pair),
// 0 = f32 operands on v_mfma_f32_32x32x2_f32 (68 FLOP per pair)
extern "C" int lgr_match_last_format(int* f16) {
    if (!f16) return LGR_ERR_INVALID_ARG;
    *f16 = g_last_stats.f16;
    return LGR_OK;
}

// LGR_MATCH_CHECK=1 (tests): worst |filtered - exact| / eps over the sampled table entries of the last call, per direction
// (rows, columns); -1 when the check did not run.  A proven bound: must be <= 1.
extern "C" int lgr_match_last_check(double* out2) {
    if (!out2) return LGR_ERR_INVALID_ARG;
    out2[0] = g_last_check[0]; out2[1] = g_last_check[1];
    return LGR_OK;
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Orthonormal basis for the box bounds: principal axes of the k-means sample (both sets).  Covariance on the device,
// cyclic Jacobi on the host (33 x 33), rows of V = eigenvectors, mu = sample mean.  LGR_MATCH_BOX=2: raw coordinates.
static int box_basis(lgr_ctx* ctx, const float* smp, const int* smp_ok, int ns, float* d_basis /* [34][33] + 1: V rows, then mu */) {
    std::vector<float> h(34 * 33 + 1, 0.f);
    const bool raw = env_int("LGR_MATCH_BOX", 1) == 2;
    if (!raw) {
        const int nb = cdiv(ns, COV_ROWS);
        float* part;
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_DENSE, (size_t) nb * (34 * 33 + 1) + 64, &part));   // scratch: the rerank buffers are not live yet
        LGR_HIP(ctx, hipMemsetAsync(part, 0, (size_t) nb * (34 * 33 + 1) * 4, ctx->stream));   // the a > b product slots are never written
        cov_kernel<<<nb, COV_THREADS, 0, ctx->stream>>>(smp, smp_ok, ns, part);
        cov_reduce<<<cdiv(34 * 33 + 1, 256), 256, 0, ctx->stream>>>(part, nb, d_basis);
        LGR_HIP(ctx, hipMemcpyAsync(h.data(), d_basis, (34 * 33 + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const double n = h[34 * 33];
    std::vector<double> C(33 * 33, 0.0), mu(33, 0.0), Vd(33 * 33, 0.0);
    for (int i = 0; i < 33; ++i) Vd[i * 33 + i] = 1.0;
    if (!raw && n >= 2) {
        for (int k = 0; k < 33; ++k) mu[k] = h[k] / n;
        for (int a = 0; a < 33; ++a)
            for (int b = a; b < 33; ++b) {
                double c = h[33 + a * 33 + b] / n - mu[a] * mu[b];
                C[a * 33 + b] = c; C[b * 33 + a] = c;
            }
        // cyclic Jacobi; rows of Vd become the eigenvectors.  Whatever it converges to, Vd stays a product of plane
        // rotations, i.e. orthonormal -- which is all the bound needs.
        for (int sweep = 0; sweep < 12; ++sweep)
            for (int p = 0; p < 32; ++p)
                for (int q = p + 1; q < 33; ++q) {
                    double apq = C[p * 33 + q];
                    if (std::fabs(apq) < 1e-300) continue;
                    double theta = (C[q * 33 + q] - C[p * 33 + p]) / (2.0 * apq);
                    double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                    for (int k = 0; k < 33; ++k) {
                        double ckp = C[k * 33 + p], ckq = C[k * 33 + q];
                        C[k * 33 + p] = c * ckp - sn * ckq; C[k * 33 + q] = sn * ckp + c * ckq;
                    }
                    for (int k = 0; k < 33; ++k) {
                        double cpk = C[p * 33 + k], cqk = C[q * 33 + k];
                        C[p * 33 + k] = c * cpk - sn * cqk; C[q * 33 + k] = sn * cpk + c * cqk;
                    }
                    for (int k = 0; k < 33; ++k) {
                        double vpk = Vd[p * 33 + k], vqk = Vd[q * 33 + k];
                        Vd[p * 33 + k] = c * vpk - sn * vqk; Vd[q * 33 + k] = sn * vpk + c * vqk;
                    }
                }
    }
    for (int i = 0; i < 33 * 33; ++i) h[i] = (float) Vd[i];
    for (int k = 0; k < 33; ++k) h[33 * 33 + k] = (float) mu[k];
    LGR_HIP(ctx, hipMemcpyAsync(d_basis, h.data(), 34 * 33 * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // h goes out of scope
    return LGR_OK;
}

static int match_impl(lgr_ctx* ctx, const float* d_a, int ma, const float* d_b, int mb, int block,
                      int32_t* d_ab_idx, float* d_ab_dist, int32_t* d_ba_idx, float* d_ba_dist) {
    LGR_CHECK(ctx, ctx && (d_a || ma == 0) && (d_b || mb == 0) && (d_ab_idx || ma == 0) && (d_ab_dist || ma == 0), LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, ma >= 0 && mb >= 0 && block > 0, LGR_ERR_INVALID_ARG);
    bool both = d_ba_idx != nullptr && mb > 0;
    if (d_ba_idx) LGR_CHECK(ctx, d_ba_dist != nullptr, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    memset(&g_last_stats, 0, sizeof g_last_stats);
    g_last_check[0] = g_last_check[1] = -1;
    ctx->mfma_timed = 0;
    // default result: unmatched
    if (ma) { LGR_HIP(ctx, hipMemsetAsync(d_ab_idx, 0xff, (size_t) ma * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ab_dist, 0, (size_t) ma * 4, ctx->stream)); }
    if (mb && d_ba_idx) { LGR_HIP(ctx, hipMemsetAsync(d_ba_idx, 0xff, (size_t) mb * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ba_dist, 0, (size_t) mb * 4, ctx->stream)); }
    if (ma == 0 || mb == 0) return LGR_OK;

    // leaves per cluster: about 1024 rows per leaf on the larger side.  LGR_MATCH_SUB / LGR_MATCH_PRUNE override the
    // automatic choices, LGR_MATCH_NEAR the pass-0 width (tests force the skipping path on small inputs); results never
    // depend on them.
    int sub = 1;
    while (sub < SUBMAX && (long long) KCL * sub * 1024 < std::max(ma, mb)) sub *= 2;
    sub = std::min(SUBMAX, std::max(1, env_int("LGR_MATCH_SUB", sub)));
    const int n_leaves = KCL * sub;
    const int prune_mode = env_int("LGR_MATCH_PRUNE", -1);   // -1 auto, 0 off, 1 on
    const int near_t = std::max(1, env_int("LGR_MATCH_NEAR", NEAR_T));
    const bool prune = prune_mode == 1 || (prune_mode != 0 && (double) ma * mb >= 65536.0 * 65536.0);

    // ---- 1. k-means centres on a sample: KCL clusters, then `sub` leaves inside every cluster
    const int ns = 2 * KM_SAMPLE;
    char* misc;
    size_t off = 8192;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t) 255; return o; };
    const size_t o_cen2 = carve((size_t) MAXLEAF * 33 * 4), o_sums2 = carve((size_t) MAXLEAF * 34 * 4);
    const size_t o_smp = carve((size_t) ns * 33 * 4), o_ok = carve((size_t) ns * 4), o_label = carve((size_t) ns * 4);
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, off, &misc));
    float* cen = (float*) (misc + 256);                 // [KCL][33]
    float* cen2 = (float*) (misc + o_cen2);             // [n_leaves][33]
    float* sums2 = (float*) (misc + o_sums2);
    float* smp = (float*) (misc + o_smp);
    int* smp_ok = (int*) (misc + o_ok);
    int* label = (int*) (misc + o_label);
    km_sample<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(d_a, ma, d_b, mb, KM_SAMPLE, smp, smp_ok);
    km_init<<<1, 64, 0, ctx->stream>>>(smp, smp_ok, ns, cen);
    for (int it = 0; it < KM_ITERS; ++it) {   // Lloyd, deterministic (no float atomics)
        km_label<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(smp, smp_ok, ns, cen, label);
        km_centres<<<KCL, KMC_THREADS, 0, ctx->stream>>>(smp, label, ns, cen);
    }
    km_label<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(smp, smp_ok, ns, cen, label);
    km2_init<<<KCL, 64, 0, ctx->stream>>>(smp, label, ns, cen, sub, cen2);
    if (sub > 1) {
        int* leaf_of = (int*) sums2;   // [ns] (the slab of the former atomic sums: MAXLEAF * 34 floats >= 2 * KM_SAMPLE)
        static_assert((size_t) MAXLEAF * 34 >= 2 * (size_t) KM_SAMPLE, "leaf_of fits the sums2 slab");
        const size_t km2_lds = (size_t) KCL * (sub * 33 + 1) * 4;
        if (km2_lds > 64 * 1024) LGR_HIP(ctx, hipFuncSetAttribute((const void*) km2_label, hipFuncAttributeMaxDynamicSharedMemorySize, (int) km2_lds));
        for (int it = 0; it < KM2_ITERS; ++it) {
            km2_label<<<cdiv(ns, KM2_THREADS), KM2_THREADS, km2_lds, ctx->stream>>>(smp, label, ns, cen2, sub, leaf_of);
            km2_centres<<<n_leaves, 64, 0, ctx->stream>>>(smp, leaf_of, ns, cen2);
        }
    }

    // ---- 2. assign / sort / place
    auto pick_group = [](size_t q_count, size_t t_count) {   // table [t/g][q] floats kept under ~6 GB
        int g = 1024;
        while (g < 4096 && (t_count / g + 1) * q_count * 4 > ((size_t) 6 << 30)) g *= 2;
        return g;
    };
    int rg_rows = both ? pick_group((size_t) mb, (size_t) ma) : BLOCK_ROWS;   // row groups (column direction table)
    if (ma <= 65536) rg_rows = BLOCK_ROWS;                                     // small inputs: keep the cluster padding small
    Side A, B;
    LGR_TRY(build_side(ctx, d_a, ma, cen, cen2, sub, 1, rg_rows, WS_MATCH_NA, WS_MATCH_AP, &A));
    LGR_TRY(build_side(ctx, d_b, mb, cen, cen2, sub, TILE, PAD, WS_MATCH_NB, WS_MATCH_BP, &B));
    if (A.n_valid == 0 || B.n_valid == 0) return LGR_OK;
    const int ma_pad = A.n_pad, mb_pad = B.n_pad;
    g_last_stats.rg_rows = rg_rows;
    const int ta = ma_pad / TILE, tb = mb_pad / TILE;
    const int n_rb = ma_pad / BLOCK_ROWS, n_stage_total = mb_pad / STAGE_COLS;

    // ---- 3. pack operands, group maxima, stage -> leaf map
    const bool f16 = env_int("LGR_MATCH_F16", 1) != 0;
    g_last_stats.f16 = f16 ? 1 : 0;
    EpsExtra ex{0.f, 0.f, 1.f};
    float c_scale = 1.f, out_scale = 1.f;
    F16Scale sc{1.f, 1.f, {1.f, 1.f, 1.f}};
    bool rot = false;
    float *nAp, *nBp;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_NORMS, (size_t) ma_pad + (size_t) KCL * mb_pad + 64, &nAp));
    nBp = nAp + ma_pad;
    unsigned* d_max = (unsigned*) (misc + 128);   // [0] largest finite norm, [1] dropped energy, [2] norm overflow flag
    LGR_HIP(ctx, hipMemsetAsync(d_max, 0, 12, ctx->stream));
    bool force_dense = false;
    if (f16) {
        // norms first: the power-of-two scale 2^s puts the largest operand (2 |a'| 2^s, |b'| 2^s) just under 2^15; the same
        // pass measures the largest energy of the three coordinates the rotated 30-D format would drop
        pack16_kernel<false, true><<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, sc, nullptr, nAp, d_max + 1);
        pack16_kernel<false, true><<<cdiv(mb_pad, 256), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, sc, nullptr, nBp, d_max + 1);
        norm_max_kernel<<<std::min(cdiv(ma_pad, 256), 2048), 256, 0, ctx->stream>>>(nAp, (size_t) ma_pad, d_max);
        norm_max_kernel<<<std::min(cdiv((long long) KCL * mb_pad, 256), 2048), 256, 0, ctx->stream>>>(nBp, (size_t) KCL * mb_pad, d_max);
        unsigned* h_max;
        LGR_TRY(lgr_pinned(ctx, 64, (void**) &h_max));
        LGR_HIP(ctx, hipMemcpyAsync(h_max, d_max, 12, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float r2, drop2;
        memcpy(&r2, h_max, 4);
        memcpy(&drop2, h_max + 1, 4);
        force_dense = h_max[2] != 0u;
        // Rotated format (FMT_F16R) when what it drops is negligible: with u the dropped coordinates of a row relative to a
        // centre, d2 = d2_30 + |u_a - u_b|^2 and 0 <= |u_a - u_b|^2 <= 4 max |u|^2 -- that bound joins the absolute error
        // term, so the choice below only trades speed.  FPFH rows: every block sums to 100 -> max |u|^2 ~ 1e-7.
        const int rot_env = env_int("LGR_MATCH_ROT", -1);
        rot = rot_env >= 0 ? rot_env != 0 : (4.0 * (double) drop2 <= 1e-8 * (double) r2);
        double R = std::sqrt((double) r2);
        int sexp = R > 0 ? (int) std::floor(std::log2(16384.0 / R)) : 14;
        sexp = std::max(-40, std::min(14, sexp));
        double N_max = (double) r2 * std::ldexp(1.0, 2 * sexp);
        int e1 = N_max > 32768.0 ? (int) std::ceil(std::log2(N_max / 32768.0)) : 0;
        e1 = std::min(15, std::max(0, e1));
        sc.s_mul = (float) std::ldexp(1.0, sexp);
        sc.inv_s2 = (float) std::ldexp(1.0, -2 * sexp);
        sc.a_norm[0] = (float) std::ldexp(1.0, e1);
        sc.a_norm[1] = (float) std::ldexp(1.0, std::max(0, e1 - 11));
        sc.a_norm[2] = (float) std::ldexp(1.0, std::max(0, e1 - 22));
        c_scale = (float) std::ldexp(1.0, 2 * sexp);
        out_scale = sc.inv_s2;
        // error terms of the split: elements whose second half would be an f16 subnormal may be flushed
        // (tau per element, linear term); three-term norm expansion (absolute term); the 112-product f32 accumulation
        // chain and the 2^-22 split residual are covered by doubling the quadratic term
        const double tau = std::ldexp(1.0, -14 - sexp);
        ex.lin = (float) (12.0 * tau);
        ex.abs = (float) (2.0 * std::ldexp(1.0, -14) * (sc.a_norm[2] + sc.a_norm[1] / 2048.0 + sc.a_norm[0] / 4194304.0) * (double) sc.inv_s2 * 1.01);   // both norms
        ex.quad = 2.f;
        if (rot) {
            // Helmert coordinates are computed in f32: |dy| <= 10.4 u |x'| per vector (prefix sums of <= 11 terms, one
            // rounded constant) -> 20.8 u (x + y)^2 on d2, 0.13 of the unit 4 g40 (x + y)^2; the dropped energy is absolute
            ex.quad = 2.2f;
            ex.abs = (float) ((double) ex.abs + 4.0 * (double) drop2 * 1.0001);
        }
    }
    g_last_stats.f16 = f16 ? (rot ? 2 : 1) : 0;
    const int KS = !f16 ? OpFmt<FMT_F32>::KS : rot ? OpFmt<FMT_F16R>::KS : OpFmt<FMT_F16>::KS;
    const size_t frag_bytes = f16 ? sizeof(f16x8) : sizeof(float);
    const size_t a_op_bytes = (size_t) ta * KS * 64 * frag_bytes;
    const size_t bset_stride = (size_t) tb * KS * 64;   // fragments per column set
    char *Aop, *Bop;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ROWMIN, a_op_bytes + 256, &Aop));
    const size_t b_op_bytes = KCL * bset_stride * frag_bytes;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_COLMIN, b_op_bytes + 256, &Bop));
    if (!f16) {
        pack_kernel<<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, (float*) Aop, nAp, d_max + 2);
        pack_kernel<<<dim3(cdiv(mb_pad, 256), KCL), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, (float*) Bop, nBp, d_max + 2);
        unsigned* h_ovf;
        LGR_TRY(lgr_pinned(ctx, 64, (void**) &h_ovf));
        LGR_HIP(ctx, hipMemcpyAsync(h_ovf, d_max + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        force_dense = h_ovf[0] != 0u;
    } else {
        if (rot) {
            pack16_kernel<true, false><<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, sc, (_Float16*) Aop, nAp, nullptr);
            pack16_kernel<true, false><<<cdiv(mb_pad, 256), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, sc, (_Float16*) Bop, nBp, nullptr);
        } else {
            pack16_kernel<false, false><<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, sc, (_Float16*) Aop, nAp, nullptr);
            pack16_kernel<false, false><<<cdiv(mb_pad, 256), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, sc, (_Float16*) Bop, nBp, nullptr);
        }
    }
    const int n_rg = cdiv(ma_pad, rg_rows);
    // column groups of the row-minimum table: a leaf, cut into pieces of at most GROUP_COLS columns (k-means leaves of
    // near-duplicate descriptors can hold tens of thousands of rows; the exact rerank scans a whole group per item)
    std::vector<int> h_group_start, h_tiles(2 * (size_t) tb);   // [tile] -> group, [tb + tile] -> leaf
    int group_cols = GROUP_COLS;   // larger pieces for very large inputs: keep the table [groups][ma_pad] under ~24 GB
    while (group_cols < 65536 && ((size_t) mb_pad / group_cols + n_leaves) * (size_t) ma_pad * 4 > ((size_t) 24 << 30)) group_cols *= 2;
    for (int l = 0; l < n_leaves; ++l)
        for (int s0 = B.h_leaf_start[l]; s0 < B.h_leaf_start[l + 1]; s0 += group_cols) {
            int s1 = std::min(B.h_leaf_start[l + 1], s0 + group_cols), g = (int) h_group_start.size();
            h_group_start.push_back(s0);
            for (int t = s0 / TILE; t < s1 / TILE; ++t) { h_tiles[t] = g; h_tiles[tb + t] = l; }
        }
    const int n_groups = (int) h_group_start.size();
    g_last_stats.sub_cols = n_groups;
    h_group_start.push_back(mb_pad);
    float *gmaxB, *gmaxA;
    int *cl_of_rg, *tile_group, *tile_leaf, *group_start;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BEST_B, (size_t) KCL * n_groups + 2 * (size_t) n_rg + 2 * (size_t) tb + n_groups + 65, &gmaxB));
    gmaxA = gmaxB + (size_t) KCL * n_groups;
    cl_of_rg = (int*) (gmaxA + n_rg);
    tile_group = cl_of_rg + n_rg;
    tile_leaf = tile_group + tb;
    group_start = tile_leaf + tb;
    {
        std::vector<int> h(n_rg);
        for (int g = 0; g < n_rg; ++g) h[g] = A.h_blkcl[(size_t) g * (rg_rows / BLOCK_ROWS)];
        LGR_HIP(ctx, hipMemcpyAsync(cl_of_rg, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(tile_group, h_tiles.data(), h_tiles.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(group_start, h_group_start.data(), h_group_start.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    group_max_kernel<<<dim3(n_groups, KCL), 256, 0, ctx->stream>>>(nBp, mb_pad, 0, group_start, gmaxB);
    group_max_kernel<<<dim3(n_rg, 1), 256, 0, ctx->stream>>>(nAp, ma_pad, rg_rows, nullptr, gmaxA);
    float *sortedA = nullptr, *sortedB;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_SORTED_B, (size_t) mb_pad * 33, &sortedB));
    gather_rows_kernel<<<cdiv((long long) mb_pad * 33, 256), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, sortedB);
    if (both || prune) {
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_SORTED_A, (size_t) ma_pad * 33, &sortedA));
        gather_rows_kernel<<<cdiv((long long) ma_pad * 33, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, sortedA);
    }

    // ---- 4. MFMA passes into the two minimum tables (+inf initialised)
    int *rowmin, *colmin = nullptr;
    const size_t tab_floats = (size_t) n_groups * ma_pad + (both ? (size_t) n_rg * mb_pad : 0);
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BEST_A, tab_floats + 2 * ((size_t) ma + mb) + 64, &rowmin));
    if (both) colmin = rowmin + (size_t) n_groups * ma_pad;
    unsigned long long* bestA = (unsigned long long*) (rowmin + tab_floats + (tab_floats & 1));
    unsigned long long* bestB = bestA + ma;
    fill_u64<<<cdiv(ma + mb, 256), 256, 0, ctx->stream>>>(bestA, ma + mb, ~0ull);
    // dense mode: +inf everywhere; skipping mode: init_tables_kernel covers what each pass computes (LGR_MATCH_POISON=1, tests: the
    // rest is filled with 0 -- the most harmful value a stale entry could have -- to show that nothing reads it)
    if (!prune) LGR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t) rowmin, 0x7f800000, tab_floats, ctx->stream));
    else if (env_int("LGR_MATCH_POISON", 0)) LGR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t) rowmin, 0, tab_floats, ctx->stream));
    const int n_cc = cdiv(mb_pad, CHUNK_COLS);
    // work items of the persistent MFMA kernel: one row group (the owner of its column minima) x one column chunk
    const int item_rb = both ? std::min(rg_rows / BLOCK_ROWS, 16) : 4;
    const int n_ir = cdiv(n_rb, item_rb), ccx = cdiv(n_cc, 8), n_flags = 8 * ccx * n_ir;
    int* ibuf;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ITEMS2, (size_t) 4 * n_flags + 64, &ibuf));
    int *iflags = ibuf, *ipos = ibuf + n_flags;
    int2* ilist = (int2*) (ibuf + 2 * (size_t) n_flags);
    int* xcd_start = ibuf + 4 * (size_t) n_flags;   // [9]
    int* xcd_ctr = xcd_start + 16;                   // [8]
    const int mfma_grid = 8 * (LGR_MM_OCC / 2) * std::max(1, ctx->n_cu / 8);   // resident workgroups: LGR_MM_OCC / 2 per CU
    auto launch_mfma = [&](const unsigned* mask) -> int {
        items_flag_kernel<<<cdiv(n_flags, 256), 256, 0, ctx->stream>>>(mask, n_rb, n_cc, item_rb, n_ir, ccx, iflags);
        size_t sb = 0;
        LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, sb, iflags, ipos, 0, (size_t) n_flags, rocprim::plus<int>(), ctx->stream));
        void* stmp;
        LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, sb, &stmp));
        LGR_HIP(ctx, rocprim::exclusive_scan(stmp, sb, iflags, ipos, 0, (size_t) n_flags, rocprim::plus<int>(), ctx->stream));
        items_emit_kernel<<<cdiv(n_flags, 256), 256, 0, ctx->stream>>>(iflags, ipos, item_rb, n_ir, ccx, ilist, xcd_start);
        LGR_HIP(ctx, hipMemsetAsync(xcd_ctr, 0, 32, ctx->stream));
        LGR_CHECK(ctx, ctx->mfma_timed < 8, LGR_ERR_INVALID_ARG);
        (void) hipEventRecord(ctx->ev[9 + 2 * ctx->mfma_timed], ctx->stream);
#define LGR_MFMA_ARGS bset_stride, c_scale, out_scale, A.blkcl, nAp, ma_pad, mb_pad, rg_rows, tile_group, mask, rowmin, colmin, n_cc, item_rb, ilist, xcd_start, xcd_ctr
        if (f16 && rot) {
            if (both) match_mfma<true, FMT_F16R><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
            else match_mfma<false, FMT_F16R><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
        } else if (f16) {
            if (both) match_mfma<true, FMT_F16><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
            else match_mfma<false, FMT_F16><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
        } else {
            if (both) match_mfma<true, FMT_F32><<<mfma_grid, NTHR, 0, ctx->stream>>>((const float*) Aop, (const float*) Bop, LGR_MFMA_ARGS);
            else match_mfma<false, FMT_F32><<<mfma_grid, NTHR, 0, ctx->stream>>>((const float*) Aop, (const float*) Bop, LGR_MFMA_ARGS);
        }
#undef LGR_MFMA_ARGS
        (void) hipEventRecord(ctx->ev[10 + 2 * ctx->mfma_timed], ctx->stream);
        ctx->mfma_timed += 1;
        LGR_HIP(ctx, hipGetLastError());
#ifdef EXP_PROF
        {
            LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            unsigned long long hp[16];
            (void) hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_prof), sizeof hp);
            int hx[9];
            (void) hipMemcpy(hx, xcd_start, sizeof hx, hipMemcpyDeviceToHost);
            fprintf(stderr, "[lgr] prof launch %d (10 ns ticks): prologue %llu stages %llu (barrier+dma wait %llu, - %llu) colflush %llu wg_total %llu | wgs %llu visits %llu rowflush %llu | items %d (per xcd %d %d %d %d %d %d %d %d)\n",
                    ctx->mfma_timed - 1, hp[0], hp[1], hp[2], hp[5], hp[3], hp[4], hp[8], hp[9], hp[10], hx[8], hx[1] - hx[0], hx[2] - hx[1], hx[3] - hx[2],
                    hx[4] - hx[3], hx[5] - hx[4], hx[6] - hx[5], hx[7] - hx[6], hx[8] - hx[7]);
            unsigned long long z[16] = {0};
            (void) hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z);
        }
#endif
        return LGR_OK;
    };
    g_last_stats.stages_all = (double) n_rb * n_stage_total;
    CompView comp_rows{nullptr, 0, nullptr}, comp_cols{nullptr, 0, nullptr};
    const uint8_t *chk_done = nullptr, *chk_sched = nullptr;
    if (!prune) {
        LGR_TRY(launch_mfma(nullptr));
        g_last_stats.stages_done = g_last_stats.stages_all;
    } else {
        // section 3b: lower bounds, pass 1 (nearest tiles), upper bounds, pass 2 (everything the bounds cannot exclude)
        char* pb;
        size_t poff = 0;
        auto pcarve = [&](size_t bytes) { size_t o = poff; poff += (bytes + 255) & ~(size_t) 255; return o; };
        const size_t o_lb = pcarve((size_t) n_rb * n_leaves * 4), o_done = pcarve((size_t) n_rb * n_leaves), o_sched = pcarve((size_t) n_rb * n_leaves);
        const size_t o_mask = pcarve((size_t) n_rb * n_cc * 4), o_urb = pcarve((size_t) n_rb * 4), o_ul = pcarve((size_t) MAXLEAF * 4);
        const size_t o_stats = pcarve(sizeof(MaskStats));
        const size_t o_cr = pcarve((size_t) n_rb * n_groups), o_cc = pcarve((size_t) n_leaves * n_rg), o_gl = pcarve((size_t) n_groups * 4);
        const size_t o_lg = pcarve((size_t) (n_leaves + 1) * 4);
        const size_t o_boxa = pcarve((size_t) n_rb * 66 * 4), o_boxb = pcarve((size_t) n_leaves * 66 * 4), o_basis = pcarve((size_t) (34 * 33 + 64) * 4);   // V, mu, count / rmax2
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_PRUNE, poff, &pb));
        float* LBsq = (float*) (pb + o_lb);
        uint8_t* done = (uint8_t*) (pb + o_done);
        uint8_t* sched = (uint8_t*) (pb + o_sched);
        unsigned* mask = (unsigned*) (pb + o_mask);
        float* u_rb = (float*) (pb + o_urb);
        unsigned* u_leaf = (unsigned*) (pb + o_ul);
        MaskStats* mstats = (MaskStats*) (pb + o_stats);
        uint8_t* comp_r = (uint8_t*) (pb + o_cr);
        uint8_t* comp_c = (uint8_t*) (pb + o_cc);
        int* group_leaf = (int*) (pb + o_gl);
        int* leaf_g0 = (int*) (pb + o_lg);
        LGR_HIP(ctx, hipMemsetAsync(pb + o_done, 0, o_cr - o_done, ctx->stream));   // done, sched, masks, bounds, stats
        {
            std::vector<int> h(n_groups), hl(n_leaves + 1, n_groups);
            for (int g = 0; g < n_groups; ++g) h[g] = h_tiles[tb + h_group_start[g] / TILE];
            for (int g = n_groups - 1; g >= 0; --g) hl[h[g]] = g;                     // first group of every leaf that has one
            for (int l = n_leaves - 1; l >= 0; --l) hl[l] = std::min(hl[l], hl[l + 1]);   // empty leaves: g0 == g1
            LGR_HIP(ctx, hipMemcpyAsync(group_leaf, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            LGR_HIP(ctx, hipMemcpyAsync(leaf_g0, hl.data(), hl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        auto build_comp = [&]() {
            comp_rows_kernel<<<cdiv((long long) n_rb * n_groups, 256), 256, 0, ctx->stream>>>(done, sched, group_leaf, n_rb, n_leaves, n_groups, comp_r);
            if (both) comp_cols_kernel<<<cdiv((long long) n_leaves * n_rg, 256), 256, 0, ctx->stream>>>(done, sched, n_rb, n_leaves, n_rg, rg_rows / BLOCK_ROWS, comp_c);
        };
        chk_done = done; chk_sched = sched;
        comp_rows = CompView{comp_r, n_groups, nullptr};
        comp_cols = CompView{comp_c, n_rg, tile_leaf};
        lb_kernel<<<n_rb, 256, 0, ctx->stream>>>(sortedA, A.perm, cen2, B.r2max, B.leaf_count, n_leaves, LBsq);
        if (env_int("LGR_MATCH_BOX", 1)) {
            float* boxA = (float*) (pb + o_boxa);
            float* boxBt = (float*) (pb + o_boxb);
            float* basis = (float*) (pb + o_basis);              // V [33][33], mu [33]
            unsigned* rmax2 = (unsigned*) (basis + 34 * 33 + 8);
            LGR_TRY(box_basis(ctx, smp, smp_ok, ns, basis));
            LGR_HIP(ctx, hipMemsetAsync(rmax2, 0, 4, ctx->stream));
            box_kernel<<<n_rb, 256, 0, ctx->stream>>>(sortedA, A.perm, nullptr, n_rb, basis, basis + 33 * 33, 0, boxA, rmax2);
            box_kernel<<<n_leaves, 256, 0, ctx->stream>>>(sortedB, B.perm, B.leaf_start, n_leaves, basis, basis + 33 * 33, 1, boxBt, rmax2);
            box_lb_kernel<<<n_rb, 256, 0, ctx->stream>>>(boxA, boxBt, n_leaves, rmax2, LBsq);
        }
        // pass 0: the NEAR_T nearest leaves of every row block and the NEAR_T nearest row blocks of every leaf
        auto launch_near = [&](int n_vec, int len, size_t vs, size_t es) -> int {
            if (len <= NEAR_LDS_MAX) {
                if ((size_t) len * 4 > 64 * 1024)
                    LGR_HIP(ctx, hipFuncSetAttribute((const void*) near_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, len * 4));
                near_kernel<true><<<n_vec, NEAR_THREADS, (size_t) len * 4, ctx->stream>>>(near_t, LBsq, n_vec, len, vs, es, sched, vs, es);
            } else {
                near_kernel<false><<<n_vec, NEAR_THREADS, 0, ctx->stream>>>(near_t, LBsq, n_vec, len, vs, es, sched, vs, es);
            }
            return LGR_OK;
        };
        LGR_TRY(launch_near(n_rb, n_leaves, (size_t) n_leaves, 1));
        LGR_TRY(launch_near(n_leaves, n_rb, 1, (size_t) n_leaves));
        // passes 1..: tiles within beta * U of the bounds known so far; the last pass (beta = 1) takes everything the
        // bounds cannot exclude
        static const float betas[] = {LGR_PRUNE_BETAS};
        const int n_beta = (int) (sizeof betas / sizeof betas[0]);
        for (int pass = 0; pass <= n_beta; ++pass) {
            if (pass > 0) {
                build_comp();
                row_u_kernel<<<n_rb, BLOCK_ROWS, (size_t) (n_groups + 8) * 4, ctx->stream>>>((const float*) rowmin, n_groups, ma_pad, A.perm, nAp, A.blkcl, gmaxB, ex, comp_rows, u_rb);
                if (both) {
                    LGR_HIP(ctx, hipMemsetAsync(u_leaf, 0, (size_t) MAXLEAF * 4, ctx->stream));
                    col_u_kernel<<<cdiv(mb_pad, 256), 256, (size_t) (n_rg + 8) * 4, ctx->stream>>>((const float*) colmin, n_rg, mb_pad, B.perm, nBp, gmaxA, cl_of_rg, tile_leaf, ex, comp_cols, u_leaf);
                }
                float bsq = betas[pass - 1] * betas[pass - 1];
                sched_kernel<<<cdiv((long long) n_rb * n_leaves, 256), 256, 0, ctx->stream>>>(both ? 1 : 0, bsq, LBsq, u_rb, u_leaf, n_rb, n_leaves, done, sched);
            }
            mask_kernel<<<cdiv((long long) n_rb * n_cc, 256), 256, 0, ctx->stream>>>(pass, sched, tile_leaf, n_rb, n_cc, n_leaves, n_stage_total, mask, mstats);
            init_tables_kernel<<<n_rb, BLOCK_ROWS, 0, ctx->stream>>>(sched, done, n_rb, n_leaves, leaf_g0, group_start, rg_rows / BLOCK_ROWS, rowmin, (size_t) ma_pad,
                                                                     colmin, (size_t) mb_pad);
            LGR_TRY(launch_mfma(mask));
        }
        build_comp();   // final state for the rerank scans
        MaskStats* hs;
        LGR_TRY(lgr_pinned(ctx, 256, (void**) &hs));
        LGR_HIP(ctx, hipMemcpyAsync(hs, mstats, sizeof(MaskStats), hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        g_last_stats.stages_done = 0;
        for (int k = 0; k <= n_beta; ++k) g_last_stats.stages_done += (double) hs->stages[k];
        if (env_int("LGR_MATCH_DEBUG", 0)) {
            fprintf(stderr, "[lgr] stages per pass:");
            for (int k = 0; k <= n_beta; ++k) fprintf(stderr, " %llu", hs->stages[k]);
            fprintf(stderr, " of %.0f (n_rb %d n_cc %d item_rb %d leaves %d)\n", g_last_stats.stages_all, n_rb, n_cc, item_rb, n_leaves);
            // what the schedule asks for at leaf granularity (the stages computed above also cover the neighbours' boundary tiles)
            std::vector<uint8_t> hd((size_t) n_rb * n_leaves), hsch((size_t) n_rb * n_leaves);
            LGR_HIP(ctx, hipMemcpy(hd.data(), done, hd.size(), hipMemcpyDeviceToHost));
            LGR_HIP(ctx, hipMemcpy(hsch.data(), sched, hsch.size(), hipMemcpyDeviceToHost));
            double need_cols = 0;
            for (int rb = 0; rb < n_rb; ++rb)
                for (int l = 0; l < n_leaves; ++l)
                    if (hd[(size_t) rb * n_leaves + l] | hsch[(size_t) rb * n_leaves + l]) need_cols += B.h_leaf_start[l + 1] - B.h_leaf_start[l];
            fprintf(stderr, "[lgr] scheduled (row block, leaf) pairs cover %.4f of the tiles; computed stages %.4f\n",
                    need_cols / ((double) n_rb * mb_pad), g_last_stats.stages_done / g_last_stats.stages_all);
        }
    }
    LGR_HIP(ctx, hipGetLastError());

    if (env_int("LGR_MATCH_CHECK", 0) && sortedA) {
        unsigned* d_worst = (unsigned*) (misc + 192);
        LGR_HIP(ctx, hipMemsetAsync(d_worst, 0, 8, ctx->stream));
        const int stride = 37;
        check_kernel<true><<<cdiv(ma_pad, stride), 256, (size_t) (n_groups + 8) * 4, ctx->stream>>>(
            (const float*) rowmin, n_groups, ma_pad, 0, group_start, sortedA, A.perm, sortedB, B.perm, mb_pad, nAp, A.blkcl, nullptr, gmaxB, nullptr,
            ex, comp_rows, stride, nullptr, nullptr, 0, d_worst);
        if (both)
            check_kernel<false><<<cdiv(mb_pad, stride), 256, (size_t) (n_rg + 8) * 4, ctx->stream>>>(
                (const float*) colmin, n_rg, mb_pad, rg_rows, nullptr, sortedB, B.perm, sortedA, A.perm, ma_pad, nullptr, nullptr, nBp, gmaxA, cl_of_rg,
                ex, comp_cols, stride, chk_done, chk_sched, n_leaves, d_worst + 1);
        unsigned* hw;
        LGR_TRY(lgr_pinned(ctx, 64, (void**) &hw));
        LGR_HIP(ctx, hipMemcpyAsync(hw, d_worst, 8, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float r0, r1;
        memcpy(&r0, hw, 4); memcpy(&r1, hw + 1, 4);
        g_last_check[0] = r0; g_last_check[1] = r1;
    }

    // ---- 5. exact rerank
    LGR_TRY((run_rerank<true>(ctx, ex, comp_rows, (const float*) rowmin, n_groups, 0, group_start, d_a, A, nAp, nullptr, gmaxB, nullptr, d_b, sortedB, B, block, bestA,
                              d_ab_idx, d_ab_dist, &g_last_stats.items_ab, &g_last_stats.dense_ab, force_dense)));
    if (both)
        LGR_TRY((run_rerank<false>(ctx, ex, comp_cols, (const float*) colmin, n_rg, rg_rows, nullptr, d_b, B, nullptr, nBp, gmaxA, cl_of_rg, d_a, sortedA, A, block, bestB,
                                   d_ba_idx, d_ba_dist, &g_last_stats.items_ba, &g_last_stats.dense_ba, force_dense)));
    return LGR_OK;
}

// duration of the match_mfma launch(es) of the last match call in ms (hipEvents on the ctx stream); -1 if none
extern "C" int lgr_match_last_kernel_ms(lgr_ctx* ctx, float* ms) {
    if (!ctx || !ms) return LGR_ERR_INVALID_ARG;
    *ms = -1.f;
    if (!ctx->mfma_timed) return LGR_OK;
    *ms = 0.f;
    for (int k = 0; k < ctx->mfma_timed; ++k) {   // one event pair per masked pass
        float t = 0.f;
        LGR_HIP(ctx, hipEventSynchronize(ctx->ev[10 + 2 * k]));
        LGR_HIP(ctx, hipEventElapsedTime(&t, ctx->ev[9 + 2 * k], ctx->ev[10 + 2 * k]));
        if (env_int("LGR_MATCH_DEBUG", 0)) fprintf(stderr, "[lgr] match_mfma pass %d: %.2f ms\n", k, t);
        *ms += t;
    }
    return LGR_OK;
}

extern "C" int lgr_match_bf_dev(lgr_ctx* ctx, const float* d_q33, int mq, const float* d_t33, int mt, int block,
                                int32_t* d_idx, float* d_dist) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    return match_impl(ctx, d_q33, mq, d_t33, mt, block, d_idx, d_dist, nullptr, nullptr);
}

extern "C" int lgr_match_bf2_dev(lgr_ctx* ctx, const float* d_a33, int ma, const float* d_b33, int mb, int block,
                                 int32_t* d_ab_idx, float* d_ab_dist, int32_t* d_ba_idx, float* d_ba_dist) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_ba_idx && d_ba_dist) || mb == 0, LGR_ERR_INVALID_ARG);
    return match_impl(ctx, d_a33, ma, d_b33, mb, block, d_ab_idx, d_ab_dist, d_ba_idx, d_ba_dist);
}

extern "C" int lgr_match_bf(lgr_ctx* ctx, const float* q33, int mq, const float* t33, int mt, int block,
                            int32_t* idx, float* dist) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (q33 || mq == 0) && (t33 || mt == 0) && (idx || mq == 0) && (dist || mq == 0) && mq >= 0 && mt >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *dq, *dt, *dd;
    int32_t* di;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) mq * 33 + 1, &dq));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) mt * 33 + 1, &dt));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) mq + 1, &di));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_D, (size_t) mq + 1, &dd));
    if (mq) LGR_HIP(ctx, hipMemcpyAsync(dq, q33, (size_t) mq * 132, hipMemcpyHostToDevice, ctx->stream));
    if (mt) LGR_HIP(ctx, hipMemcpyAsync(dt, t33, (size_t) mt * 132, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_match_bf_dev(ctx, dq, mq, dt, mt, block, di, dd));
    if (mq) {
        LGR_HIP(ctx, hipMemcpyAsync(idx, di, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(dist, dd, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}
