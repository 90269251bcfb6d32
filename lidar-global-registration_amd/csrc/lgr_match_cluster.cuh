// lgr_match_cluster.cuh -- 1. two-level k-means on a sample, assignment and placement of the rows.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
#pragma once
#include "lgr_match_common.cuh"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// 1. clustering (any centres are valid -- they only shape the error bounds and the tile schedule).  Lloyd steps with
//    ORDER-FREE sums: every sample coordinate is turned into a 64-bit integer on a per-call power-of-two grid
//    (2^40 steps up to the largest sample magnitude) and added with integer atomics, so the centres -- hence the tile
//    schedule and the timing -- repeat from run to run whatever order the atomics land in, and a Lloyd step is one
//    kernel (label + accumulate) instead of a labelling kernel plus a deterministic tree reduction per centre.
struct KmAcc { long long sum[33]; long long cnt; };
__device__ __forceinline__ double km_scale(unsigned kmax_bits) {   // 2^(40 - e), 2^e <= largest |sample value| < 2^(e+1)
    return kmax_bits ? ldexp(1.0, 40 - ((int) (kmax_bits >> 23) - 127)) : 1.0;
}
__global__ void km_sample(const float* __restrict__ A, int ma, const float* __restrict__ B, int mb, int per_side,
                          float* __restrict__ smp, int* __restrict__ smp_ok, unsigned* __restrict__ kmax /* zeroed: max |v| bits */) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 2 * per_side) return;
    // B == nullptr: all 2 * per_side samples from A (evenly spaced), otherwise per_side from each set
    const float* X = (!B || s < per_side) ? A : B;
    int m = (!B || s < per_side) ? ma : mb;
    int t = (!B || s < per_side) ? s : s - per_side;
    float v[33];
    bool ok = false;
    if (m > 0) {
        long long i = (long long) t * m / (B ? per_side : 2 * per_side);
        ok = row_finite(X + (size_t) i * 33, v);
    }
    float mx = 0.f;
    for (int k = 0; k < 33; ++k) { smp[(size_t) s * 33 + k] = ok ? v[k] : 0.f; if (ok) mx = fmaxf(mx, fabsf(v[k])); }
    smp_ok[s] = ok ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(kmax, __float_as_uint(mx));
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
// One Lloyd step of the first level: the centres of this step come from the previous step's sums (an empty cluster keeps
// its centre), every sample is labelled and, when acc_out is given, added to its centre's sums.  The last launch
// (acc_out = nullptr) only labels and leaves the final centres in cen_out.
constexpr int KM1_THREADS = 256;
__global__ __launch_bounds__(KM1_THREADS) void km1_step(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns, const unsigned* __restrict__ kmax,
                                                        const float* __restrict__ cen_prev, const KmAcc* __restrict__ acc_prev, KmAcc* __restrict__ acc_out,
                                                        float* __restrict__ cen_out, int* __restrict__ label) {
    __shared__ float cen_s[KCL * 33];
    __shared__ long long acc_s[KCL * 34];
    const double scale = km_scale(*kmax);
    for (int e = threadIdx.x; e < KCL * 33; e += KM1_THREADS) {
        const int c = e / 33, k = e % 33;
        float v = cen_prev[e];
        if (acc_prev && acc_prev[c].cnt > 0) v = (float) (((double) acc_prev[c].sum[k] / scale) / (double) acc_prev[c].cnt);
        cen_s[e] = v;
        if (blockIdx.x == 0) cen_out[e] = v;
    }
    for (int e = threadIdx.x; e < KCL * 34; e += KM1_THREADS) acc_s[e] = 0;
    __syncthreads();
    const int s = blockIdx.x * KM1_THREADS + threadIdx.x;
    if (s < ns) {
        int c = -1;
        if (smp_ok[s]) {
            float v[33], d;
#pragma unroll
            for (int k = 0; k < 33; ++k) v[k] = smp[(size_t) s * 33 + k];
            c = nearest_centre(v, cen_s, d);
            if (acc_out) {
#pragma unroll
                for (int k = 0; k < 33; ++k) atomicAdd((unsigned long long*) &acc_s[c * 34 + k], (unsigned long long) (long long) rint((double) v[k] * scale));
                atomicAdd((unsigned long long*) &acc_s[c * 34 + 33], 1ull);
            }
        }
        label[s] = c;
    }
    if (!acc_out) return;
    __syncthreads();
    long long* out = (long long*) acc_out;
    for (int e = threadIdx.x; e < KCL * 34; e += KM1_THREADS)
        if (acc_s[e] != 0) atomicAdd((unsigned long long*) &out[e], (unsigned long long) acc_s[e]);
}
// second level: `sub` centres inside every cluster, seeded with evenly spaced sample members of the cluster (in sample
// order).  One workgroup per cluster; thread t owns the samples [t * per, (t + 1) * per), ranks by a workgroup scan.
constexpr int KM2I_THREADS = 1024;
__global__ __launch_bounds__(KM2I_THREADS) void km2_init(const float* __restrict__ smp, const int* __restrict__ label, int ns, const float* __restrict__ cen, int sub,
                                                         float* __restrict__ cen2) {
    __shared__ int wsum[KM2I_THREADS / 64];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < sub * 33; e += KM2I_THREADS) cen2[(size_t) p * sub * 33 + e] = cen[p * 33 + e % 33];
    const int per = (ns + KM2I_THREADS - 1) / KM2I_THREADS;
    const int s0 = min(ns, tid * per), s1 = min(ns, s0 + per);
    int mine = 0;
    for (int s = s0; s < s1; ++s) mine += label[s] == p;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();   // also: the default centres above are written before any seed below
    int before = incl - mine, cnt = 0;
    for (int w = 0; w < KM2I_THREADS / 64; ++w) { if (w < wave) before += wsum[w]; cnt += wsum[w]; }
    if (cnt == 0) return;
    int r = before;
    for (int s = s0; s < s1; ++s) {
        if (label[s] != p) continue;
        const int j = (int) ((long long) r * sub / cnt);
        const bool first = r == 0 || (int) ((long long) (r - 1) * sub / cnt) != j;
        if (first)
            for (int k = 0; k < 33; ++k) cen2[((size_t) p * sub + j) * 33 + k] = smp[(size_t) s * 33 + k];
        ++r;
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
// Lloyd step of the second level: km2_step labels every sample with the nearest sub-centre of its cluster (sub-centres of
// all clusters in LDS, odd pitch per cluster as in assign_kernel) and adds it to the leaf's integer sums; km2_finalize turns
// the sums into the new centres (an empty leaf keeps its centre) and clears them for the next step.
__global__ __launch_bounds__(KM2_THREADS) void km2_step(const float* __restrict__ smp, const int* __restrict__ label, int ns, const unsigned* __restrict__ kmax,
                                                        const float* __restrict__ cen2, int sub, KmAcc* __restrict__ acc2) {
    extern __shared__ float c2s[];
    const int pitch = sub * 33 + 1;
    for (int e = threadIdx.x; e < KCL * sub * 33; e += blockDim.x) c2s[(e / (sub * 33)) * pitch + e % (sub * 33)] = cen2[e];
    __syncthreads();
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    int p = label[s];
    if (p < 0) return;
    const double scale = km_scale(*kmax);
    float v[33], d;
#pragma unroll
    for (int k = 0; k < 33; ++k) v[k] = smp[(size_t) s * 33 + k];
    const int leaf = p * sub + nearest_sub(v, c2s + p * pitch, sub, d);
    long long* out = (long long*) &acc2[leaf];
#pragma unroll
    for (int k = 0; k < 33; ++k) atomicAdd((unsigned long long*) &out[k], (unsigned long long) (long long) rint((double) v[k] * scale));
    atomicAdd((unsigned long long*) &out[33], 1ull);
}
__global__ __launch_bounds__(64) void km2_finalize(KmAcc* __restrict__ acc2, const unsigned* __restrict__ kmax, float* __restrict__ cen2) {
    const int leaf = blockIdx.x, k = threadIdx.x;
    const long long cnt = acc2[leaf].cnt;
    __syncthreads();
    if (k < 33) {
        if (cnt > 0) cen2[(size_t) leaf * 33 + k] = (float) (((double) acc2[leaf].sum[k] / km_scale(*kmax)) / (double) cnt);
        acc2[leaf].sum[k] = 0;
    } else if (k == 33) acc2[leaf].cnt = 0;
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
