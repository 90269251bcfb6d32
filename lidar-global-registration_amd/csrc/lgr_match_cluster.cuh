// lgr_match_cluster.cuh -- 1. two-level k-means on a sample, assignment and placement of the rows.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
#pragma once
#include "lgr_match_common.cuh"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// 1. clustering (any centres are valid -- they only shape the error bounds and the tile schedule; the Lloyd steps are
//    deterministic all the same, so that schedule and timing repeat from run to run)
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


}  // namespace
