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
constexpr int RB_PER_SUPER = 16;
constexpr int SUPER_ROWS = BLOCK_ROWS * RB_PER_SUPER;   // 4096
constexpr int STAGE_TILES = 4;
constexpr int STAGE_COLS = STAGE_TILES * TILE;  // 128
constexpr int CHUNK_COLS = 4096;
constexpr int STAGE_FLOATS = STAGE_TILES * KK * 64;   // 4352
constexpr int PAD = 256;
constexpr int KCL = 16;             // k-means centres
constexpr int KM_SAMPLE = 8192;     // sample rows per side
constexpr int KM_ITERS = 6;
constexpr float FLT_BIG = 3.4028234663852886e38f;

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
__global__ void km_accum(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns, const float* __restrict__ cen,
                         float* __restrict__ sums /* [KCL][34] */) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns || !smp_ok[s]) return;
    float v[33], d;
#pragma unroll
    for (int k = 0; k < 33; ++k) v[k] = smp[(size_t) s * 33 + k];
    int c = nearest_centre(v, cen, d);
    for (int k = 0; k < 33; ++k) atomicAdd(&sums[c * 34 + k], v[k]);
    atomicAdd(&sums[c * 34 + 33], 1.0f);
}
__global__ void km_update(float* __restrict__ cen, float* __restrict__ sums) {
    int c = threadIdx.x;
    if (c >= KCL) return;
    float n = sums[c * 34 + 33];
    for (int k = 0; k < 33; ++k) { if (n > 0.f) cen[c * 33 + k] = sums[c * 34 + k] / n; sums[c * 34 + k] = 0.f; }
    sums[c * 34 + 33] = 0.f;
}

// key = (cluster << 27) | (bits(r2) >> 5): sort by cluster, then by distance to the centre.  Invalid rows: 0xffffffff.
__global__ void assign_kernel(const float* __restrict__ X, int m, const float* __restrict__ cen, unsigned* __restrict__ keys,
                              int* __restrict__ vals, uint8_t* __restrict__ valid, int* __restrict__ counts /* [KCL+1] */) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float v[33], r2;
    bool ok = row_finite(X + (size_t) i * 33, v);
    unsigned key = 0xffffffffu;
    if (ok) {
        int c = nearest_centre(v, cen, r2);
        if (r2 < FLT_BIG) { key = ((unsigned) c << 27) | (__float_as_uint(r2) >> 5); atomicAdd(&counts[c], 1); }
        else ok = false;
    }
    if (!ok) atomicAdd(&counts[KCL], 1);
    keys[i] = key; vals[i] = i; valid[i] = ok ? 1 : 0;
}

// sorted position s -> padded position (every cluster starts at a multiple of the pad unit)
__global__ void place_kernel(const unsigned* __restrict__ keys_sorted, const int* __restrict__ vals_sorted, int n_valid,
                             const int* __restrict__ sorted_start /* [KCL] */, const int* __restrict__ pad_start /* [KCL] */,
                             int* __restrict__ perm) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_valid) return;
    int c = (int) (keys_sorted[s] >> 27);
    perm[pad_start[c] + (s - sorted_start[c])] = vals_sorted[s];
}

// 2. pack.  P layout: [tile][kk][half][i] floats (tile = 32 rows): MFMA lane l of step kk reads P[(tile*KK+kk)*64 + l].
// role 0 (rows): centre = the row's own cluster (blkcl[pos / 256]), operand [-2 x', 1], nrm = |x'|^2
// role 1 (cols): blockIdx.y = cluster set p, centre c_p for every column, operand [x', |x'|^2]
// padding positions (perm < 0): rows [0.., 1] / cols [0.., +inf], nrm = +inf
__global__ void pack_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, int role,
                            const float* __restrict__ cen, const int* __restrict__ blkcl,
                            float* __restrict__ P, float* __restrict__ nrm) {
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

// original rows in padded (cluster-sorted) order, contiguous, for the exact rerank (padding rows are never read)
__global__ void gather_rows_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, float* __restrict__ out) {
    size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) n_pad * 33) return;
    int pos = (int) (e / 33), k = (int) (e % 33);
    int o = perm[pos];
    out[e] = o >= 0 ? X[(size_t) o * 33 + k] : 0.f;
}

// per-group maxima of sqrt(norm) (finite entries only): out[set][g]
__global__ void group_max_kernel(const float* __restrict__ nrm, int n_pad, int group, float* __restrict__ out) {
    int g = blockIdx.x, set = blockIdx.y, n_groups = gridDim.x;
    float m = 0.f;
    for (int i = threadIdx.x; i < group; i += blockDim.x) {
        int pos = g * group + i;
        if (pos < n_pad) { float v = nrm[(size_t) set * n_pad + pos]; if (v < FLT_BIG) m = fmaxf(m, v); }
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
// 3. MFMA filter kernel.  One workgroup = 4096 rows (16 row blocks of 256) x 4096 columns.
//    wave w of row block rb owns row tiles (rb*8 + 2w, +1); all waves share the column stage staged in LDS.
//    The column operand set is chosen per row block: Bp + blkcl[rb] * bset_stride.
template <bool COLDIR>
__global__ __launch_bounds__(NTHR, 4) void match_mfma(const float* __restrict__ Ap, const float* __restrict__ Bp, size_t bset_stride,
                                                     const int* __restrict__ blkcl, const float* __restrict__ nA, int ma_pad, int mb_pad,
                                                     int sub_cols, int rg_rows,
                                                     float* __restrict__ rowmin /* [mb_pad/sub_cols][ma_pad] */,
                                                     float* __restrict__ colmin /* [ma_pad/rg_rows][mb_pad] */,
                                                     int n_cc, int n_sr) {
    // column stage double buffered in LDS: the next stage is prefetched into registers while the current one is
    // consumed and written to the other buffer afterwards -> one barrier per stage, global latency hidden
    __shared__ float Bs[2][STAGE_FLOATS];
    __shared__ int cmin_s[CHUNK_COLS];

    // XCD-aware remap: workgroups that share a column chunk (the B operand) are placed on one XCD (speed only).
    int nwg = n_cc * n_sr;
    int orig = blockIdx.x;
    int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    int cc = wgid / n_sr, sr = wgid % n_sr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int col_tile0 = cc * (CHUNK_COLS / TILE);
    const int n_coltiles = min(CHUNK_COLS / TILE, mb_pad / TILE - col_tile0);
    const int n_stages = n_coltiles / STAGE_TILES;
    const int sub_stages = sub_cols / STAGE_COLS;
    const int rb0 = sr * RB_PER_SUPER;
    const int n_rb = min(RB_PER_SUPER, ma_pad / BLOCK_ROWS - rb0);
    const int rg_blocks = rg_rows / BLOCK_ROWS;
    constexpr int IINF = 0x7f800000;   // +inf as bits
    constexpr int NPRE = (STAGE_FLOATS / 4 + NTHR - 1) / NTHR;   // float4 per thread per stage (5, the last one partial)

    if (COLDIR) {
        for (int i = tid; i < CHUNK_COLS; i += NTHR) cmin_s[i] = IINF;
    }

    for (int rbi = 0; rbi < n_rb; ++rbi) {
        const int rb = rb0 + rbi;
        const int row_tile = rb * (BLOCK_ROWS / TILE) + wave * RW;
        const float* Bset = Bp + (size_t) blkcl[rb] * bset_stride + (size_t) col_tile0 * KK * 64;
        // A fragments (coalesced 256-B loads) and the |a'|^2 of the 16 rows each lane's accumulators cover
        float a[RW][KK];
        float na[RW][16];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) a[r][kk] = Ap[((size_t) (row_tile + r) * KK + kk) * 64 + lane];
#pragma unroll
            for (int g = 0; g < 16; ++g)
                na[r][g] = nA[(row_tile + r) * TILE + (g & 3) + 8 * (g >> 2) + 4 * half];
        }
        f32x16 nav;
#pragma unroll
        for (int g = 0; g < 16; ++g) nav[g] = na[0][g];
        int rmin[RW][16];   // float bit patterns, see the epilogue note
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) rmin[r][g] = IINF;

        // stage 0 of this row block (barrier first: every wave is past the previous row block's LDS reads)
        __syncthreads();
        {
            const float4* src = reinterpret_cast<const float4*>(Bset);
            float4* dst = reinterpret_cast<float4*>(Bs[0]);
            for (int i = tid; i < STAGE_FLOATS / 4; i += NTHR) dst[i] = src[i];
        }
        __syncthreads();

        for (int st = 0; st < n_stages; ++st) {
            const int buf = st & 1;
            const bool more = st + 1 < n_stages;
            float4 pre[NPRE];
            if (more) {
                const float4* src = reinterpret_cast<const float4*>(Bset + (size_t) (st + 1) * STAGE_FLOATS);
#pragma unroll
                for (int j = 0; j < NPRE; ++j) {
                    int i = tid + NTHR * j;
                    if (i < STAGE_FLOATS / 4) pre[j] = src[i];
                }
            }
            // On gfx950 the f32 MFMA runs on the FP32 lanes the VALU uses (equal peak rate; no co-execution was
            // measured: removing the epilogue saved exactly its VALU time), so the epilogue is kept minimal:
            //  * |a'|^2 enters through the accumulator input of the first MFMA step, so d2~ = S + |a'|^2 costs nothing;
            //  * minima are taken on the bit patterns with v_min_i32 / v_min3_i32 (one instruction per slot, no
            //    canonicalising v_max pair as a float min of raw MFMA output needs).  Signed-int order equals float
            //    order except among negative values, where it keeps the one closest to zero; d2~ < 0 only within the
            //    proven error eps of a true distance >= 0, so the filtered minimum stays within eps;
            //  * a VALU lane swap instead of an LDS shuffle folds the two lane halves of the column chain.
            // The B fragment of the next tile is fetched from LDS before the epilogue runs.
            float b[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) b[kk] = Bs[buf][kk * 64 + lane];
#pragma unroll 1
            for (int ct = 0; ct < STAGE_TILES; ++ct) {
                f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][0], b[0], nav, 0, 0, 0);
#pragma unroll
                for (int kk = 1; kk < KK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], b[kk], acc, 0, 0, 0);
                if (ct + 1 < STAGE_TILES) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) b[kk] = Bs[buf][((ct + 1) * KK + kk) * 64 + lane];
                }
                int v[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) { v[g] = __float_as_int(acc[g]); rmin[0][g] = min(rmin[0][g], v[g]); }
                if (COLDIR) {
                    int cm = min(v[0], v[1]);
#pragma unroll
                    for (int g = 2; g < 16; g += 2) cm = min(min(cm, v[g]), v[g + 1]);
                    // fold the two lane halves (rows 4*half + ...) with the VALU lane swap of gfx950
                    auto sw = __builtin_amdgcn_permlane32_swap((unsigned) cm, (unsigned) cm, false, false);
                    int other = (int) (half ? sw[0] : sw[1]);
                    cm = min(cm, other);
                    if (lane < 32) atomicMin(&cmin_s[(st * STAGE_TILES + ct) * TILE + lane], cm);
                }
            }
            // flush the row minima of this column group
            if (((st + 1) % sub_stages) == 0 || !more) {
                int sub = (col_tile0 * TILE + st * STAGE_COLS) / sub_cols;
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        int v = rmin[r][g];
                        v = min(v, __shfl_xor(v, 1));
                        v = min(v, __shfl_xor(v, 2));
                        v = min(v, __shfl_xor(v, 4));
                        v = min(v, __shfl_xor(v, 8));
                        v = min(v, __shfl_xor(v, 16));
                        if ((lane & 31) == 0)
                            rowmin[(size_t) sub * ma_pad + (row_tile + r) * TILE + (g & 3) + 8 * (g >> 2) + 4 * half] = __int_as_float(v);
                        rmin[r][g] = IINF;
                    }
            }
            if (more) {
                float4* dst = reinterpret_cast<float4*>(Bs[buf ^ 1]);
#pragma unroll
                for (int j = 0; j < NPRE; ++j) {
                    int i = tid + NTHR * j;
                    if (i < STAGE_FLOATS / 4) dst[i] = pre[j];
                }
                __syncthreads();   // next buffer visible; all waves done with the buffer that is refilled next
            }
        }
        if (COLDIR && (((rbi + 1) % rg_blocks) == 0 || rbi + 1 == n_rb)) {
            __syncthreads();
            int rg = (rb * BLOCK_ROWS) / rg_rows;
            int ncols = n_coltiles * TILE;
            for (int i = tid; i < ncols; i += NTHR) {
                colmin[(size_t) rg * mb_pad + col_tile0 * TILE + i] = __int_as_float(cmin_s[i]);
                cmin_s[i] = IINF;
            }
            // the next row block's first __syncthreads() orders these resets before any new atomicMin
        }
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

template <bool ROWDIR>
__device__ __forceinline__ float group_eps(int i, int g, float xq, const float* __restrict__ nT_sets, const float* __restrict__ gmax,
                                           int n_groups, int p_of_query, const int* __restrict__ cl_of_group, int t_pad) {
    // ROWDIR: query = row i of cluster p (xq = |a'|), train group g of columns: y = gmaxB[p][g]
    // COLDIR: query = column i, train group g = row group of cluster p(g): x = gmaxA[g], y = |b - c_p(g)| (per set)
    double x, y;
    if (ROWDIR) { x = (double) xq; y = (double) gmax[(size_t) p_of_query * n_groups + g]; }
    else {
        int p = cl_of_group[g];
        x = (double) gmax[g];
        float nb = nT_sets[(size_t) p * t_pad + i];
        y = sqrt((double) nb) * 1.0000002;
    }
    const double u = 5.9604644775390625e-8;
    const double g40 = 40 * u / (1 - 40 * u);
    double s = x + y;
    return (float) (4.0 * g40 * s * s * 1.000001 + 1e-30);
}

template <bool ROWDIR>
__global__ void rerank_count(const float* __restrict__ table, int n_groups, int q_pad, const int* __restrict__ permQ,
                             const float* __restrict__ nQ /* ROWDIR: |a'|^2 per padded row */, const int* __restrict__ blkclQ,
                             const float* __restrict__ nQ_sets /* COLDIR: |b - c_p|^2 [KCL][q_pad] */, const float* __restrict__ gmax,
                             const int* __restrict__ cl_of_group, int dense_limit,
                             float* __restrict__ thr_out, int* __restrict__ counts, unsigned* __restrict__ dense,
                             RerankCounters* __restrict__ cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q_pad) return;
    counts[i] = 0;
    int o = permQ[i];
    if (o < 0) return;
    int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    float nq = ROWDIR ? nQ[i] : 0.f;
    float xq = ROWDIR ? sqrtf(nq) * 1.0000002f : 0.f;
    float ub = __uint_as_float(0x7f800000u);
    for (int g = 0; g < n_groups; ++g) {
        float v = table[(size_t) g * q_pad + i];
        if (!(v < FLT_BIG)) continue;
        float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad);
        ub = fminf(ub, v + e);
    }
    if (!(ub < FLT_BIG)) return;     // no valid train row at all
    double d2 = fmax((double) ub, 0.0);   // both tables hold d2~ = S + |a'|^2
    float thr = (float) ((double) ub + 1e-5 * d2 + 8.0 * 5.9604644775390625e-8 * fabs((double) ub) + 1e-30);
    if (thr < ub) thr = ub;
    int nc = 0;
    for (int g = 0; g < n_groups; ++g) {
        float v = table[(size_t) g * q_pad + i];
        if (!(v < FLT_BIG)) continue;
        float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad);
        nc += (v - e <= thr) ? 1 : 0;
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
                            const int* __restrict__ cl_of_group, const float* __restrict__ thr_in,
                            const int* __restrict__ counts, const int* __restrict__ offs, uint2* __restrict__ items) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q_pad || counts[i] == 0) return;
    int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    float xq = ROWDIR ? sqrtf(nQ[i]) * 1.0000002f : 0.f;
    float thr = thr_in[i];
    int pos = offs[i];
    for (int g = 0; g < n_groups; ++g) {
        float v = table[(size_t) g * q_pad + i];
        if (!(v < FLT_BIG)) continue;
        float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad);
        if (v - e <= thr) items[pos++] = make_uint2((unsigned) i, (unsigned) g);
    }
}

// 4b. one wave per (query position, train group) item: exact distances to the group's train rows (original arrays).
__global__ __launch_bounds__(256) void rerank_items(const float* __restrict__ Q, const int* __restrict__ permQ,
                                                    const float* __restrict__ Tsorted, const int* __restrict__ permT, int t_pad,
                                                    int group_size, int block, int nblocks, const uint2* __restrict__ items,
                                                    unsigned n_items, unsigned long long* __restrict__ best) {
    int lane = threadIdx.x & 63;
    for (unsigned it = blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += gridDim.x * 4) {
        uint2 w = items[it];
        int qo = permQ[w.x];
        float q[33];
        const float* qp = Q + (size_t) qo * 33;
#pragma unroll
        for (int k = 0; k < 33; ++k) q[k] = qp[k];
        int j0 = (int) w.y * group_size, j1 = min(t_pad, j0 + group_size);
        unsigned long long bk = ~0ull;
        for (int j = j0 + lane; j < j1; j += 64) {
            int to = permT[j];
            if (to < 0) continue;                // padding
            float t[33];
            const float* tp = Tsorted + (size_t) j * 33;   // same values as T[to], contiguous in padded order
#pragma unroll
            for (int k = 0; k < 33; ++k) t[k] = tp[k];
            float d = exact_l2(q, t);
            if (!(d < FLT_BIG)) continue;        // batchDistance keeps only d < FLT_MAX
            unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(to, block, nblocks);
            bk = key < bk ? key : bk;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long ok = __shfl_xor(bk, o);
            bk = ok < bk ? ok : bk;
        }
        if (lane == 0 && bk != ~0ull) atomicMin(&best[qo], bk);
    }
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
    int m = 0, n_valid = 0, n_pad = 0, unit = PAD;
    int* perm = nullptr;          // [n_pad] padded position -> original row or -1
    uint8_t* valid = nullptr;     // [m]
    int* blkcl = nullptr;         // [n_pad / 256] cluster of each 256-row block (device)
    std::vector<int> h_blkcl;     // host copy
};

// assign + sort + place one side
int build_side(lgr_ctx* ctx, const float* d_x, int m, const float* cen, int unit, int ws_keys, int ws_perm, Side* s) {
    s->m = m; s->unit = unit;
    unsigned *keys, *keys2;
    int *vals, *vals2;
    char* kbuf;
    size_t body = (((size_t) m * 17 + 255) & ~(size_t) 255);
    LGR_TRY(lgr_ws_t(ctx, ws_keys, body + 1024, &kbuf));
    keys = (unsigned*) kbuf; keys2 = keys + m; vals = (int*) (keys2 + m); vals2 = vals + m;
    s->valid = (uint8_t*) (vals2 + m);
    int* counts = (int*) (kbuf + body);
    LGR_HIP(ctx, hipMemsetAsync(counts, 0, 256, ctx->stream));
    assign_kernel<<<cdiv(m, 128), 128, 0, ctx->stream>>>(d_x, m, cen, keys, vals, s->valid, counts);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, keys, keys2, vals, vals2, (size_t) m, 0, 32, ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::radix_sort_pairs(tmp, tb, keys, keys2, vals, vals2, (size_t) m, 0, 32, ctx->stream));
    int* h;
    LGR_TRY(lgr_pinned(ctx, 256, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, counts, (KCL + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int starts[2 * KCL];
    int acc = 0, pacc = 0;
    s->h_blkcl.clear();
    for (int c = 0; c < KCL; ++c) {
        starts[c] = acc; starts[KCL + c] = pacc;
        acc += h[c];
        int padded = pad_to(h[c], unit);
        for (int b = 0; b < padded / BLOCK_ROWS; ++b) s->h_blkcl.push_back(c);
        pacc += padded;
    }
    s->n_valid = acc; s->n_pad = pacc;
    if (s->n_pad == 0) return LGR_OK;
    int* pbuf;
    LGR_TRY(lgr_ws_t(ctx, ws_perm, (size_t) s->n_pad + s->h_blkcl.size() + 2 * KCL + 64, &pbuf));
    s->perm = pbuf; s->blkcl = pbuf + s->n_pad;
    int* d_starts = s->blkcl + s->h_blkcl.size();
    LGR_HIP(ctx, hipMemsetAsync(s->perm, 0xff, (size_t) s->n_pad * 4, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(s->blkcl, s->h_blkcl.data(), s->h_blkcl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(d_starts, starts, sizeof starts, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `starts` is a stack buffer
    if (s->n_valid) place_kernel<<<cdiv(s->n_valid, 256), 256, 0, ctx->stream>>>(keys2, vals2, s->n_valid, d_starts, d_starts + KCL, s->perm);
    return LGR_OK;
}

template <bool ROWDIR>
int run_rerank(lgr_ctx* ctx, const float* table, int n_groups, int group_size, const float* Q, const Side& qs,
               const float* nQ, const float* nQ_sets, const float* gmax, const int* cl_of_group,
               const float* T, const float* Tsorted, const Side& ts, int block, unsigned long long* best, int32_t* d_idx, float* d_dist,
               unsigned* stat_items, unsigned* stat_dense) {
    const int q_pad = qs.n_pad;
    unsigned* dense;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_DENSE, (size_t) q_pad * 4 + 64, &dense));
    float* thr = (float*) (dense + q_pad);
    int* counts = (int*) (dense + 2 * (size_t) q_pad);
    int* offs = (int*) (dense + 3 * (size_t) q_pad);
    char* misc;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, 4096, &misc));
    RerankCounters* cnt = (RerankCounters*) (misc + 64);
    LGR_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(RerankCounters), ctx->stream));
    int nblocks = (ts.m + block - 1) / block;
    int dense_limit = std::max(64, n_groups / 2);
    rerank_count<ROWDIR><<<cdiv(q_pad, 256), 256, 0, ctx->stream>>>(table, n_groups, q_pad, qs.perm, nQ, qs.blkcl, nQ_sets, gmax,
                                                                   cl_of_group, dense_limit, thr, counts, dense, cnt);
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
        uint2* items;
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ITEMS, (size_t) n_items, &items));
        rerank_emit<ROWDIR><<<cdiv(q_pad, 256), 256, 0, ctx->stream>>>(table, n_groups, q_pad, nQ, qs.blkcl, nQ_sets, gmax, cl_of_group,
                                                                      thr, counts, offs, items);
        int grid = (int) std::min<unsigned>((n_items + 3) / 4, (unsigned) ctx->n_cu * 16);
        rerank_items<<<grid, 256, 0, ctx->stream>>>(Q, qs.perm, Tsorted, ts.perm, ts.n_pad, group_size, block, nblocks, items, n_items, best);
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
// per direction, group sizes
struct lgr_match_stats { unsigned items_ab, dense_ab, items_ba, dense_ba; int sub_cols, rg_rows; };
static lgr_match_stats g_last_stats;
extern "C" int lgr_match_last_stats(unsigned* out6) {
    out6[0] = g_last_stats.items_ab; out6[1] = g_last_stats.dense_ab; out6[2] = g_last_stats.items_ba;
    out6[3] = g_last_stats.dense_ba; out6[4] = (unsigned) g_last_stats.sub_cols; out6[5] = (unsigned) g_last_stats.rg_rows;
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
    // default result: unmatched
    if (ma) { LGR_HIP(ctx, hipMemsetAsync(d_ab_idx, 0xff, (size_t) ma * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ab_dist, 0, (size_t) ma * 4, ctx->stream)); }
    if (mb && d_ba_idx) { LGR_HIP(ctx, hipMemsetAsync(d_ba_idx, 0xff, (size_t) mb * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ba_dist, 0, (size_t) mb * 4, ctx->stream)); }
    if (ma == 0 || mb == 0) return LGR_OK;

    // ---- 1. k-means centres on a sample
    char* misc;
    const size_t misc_bytes = 8192 + (size_t) 2 * KM_SAMPLE * 34 * 4;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, misc_bytes, &misc));
    float* cen = (float*) (misc + 256);                 // [KCL][33]
    float* sums = (float*) (misc + 4096);               // [KCL][34]
    float* smp = (float*) (misc + 8192);
    int* smp_ok = (int*) (smp + (size_t) 2 * KM_SAMPLE * 33);
    const int ns = 2 * KM_SAMPLE;
    km_sample<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(d_a, ma, d_b, mb, KM_SAMPLE, smp, smp_ok);
    km_init<<<1, 64, 0, ctx->stream>>>(smp, smp_ok, ns, cen);
    LGR_HIP(ctx, hipMemsetAsync(sums, 0, KCL * 34 * 4, ctx->stream));
    for (int it = 0; it < KM_ITERS; ++it) {
        km_accum<<<cdiv(ns, 128), 128, 0, ctx->stream>>>(smp, smp_ok, ns, cen, sums);
        km_update<<<1, 64, 0, ctx->stream>>>(cen, sums);
    }

    // ---- 2. assign / sort / place, group sizes
    auto pick_group = [](size_t q_count, size_t t_count) {   // table [t/g][q] floats kept under ~6 GB
        int g = 1024;
        while (g < 4096 && (t_count / g + 1) * q_count * 4 > ((size_t) 6 << 30)) g *= 2;
        return g;
    };
    int rg_rows = both ? pick_group((size_t) mb, (size_t) ma) : BLOCK_ROWS;   // row groups (column direction table)
    if (ma <= 65536) rg_rows = BLOCK_ROWS;                                     // small inputs: keep the cluster padding small
    Side A, B;
    LGR_TRY(build_side(ctx, d_a, ma, cen, rg_rows, WS_MATCH_NA, WS_MATCH_AP, &A));
    LGR_TRY(build_side(ctx, d_b, mb, cen, PAD, WS_MATCH_NB, WS_MATCH_BP, &B));
    if (A.n_valid == 0 || B.n_valid == 0) return LGR_OK;
    const int ma_pad = A.n_pad, mb_pad = B.n_pad;
    const int sub_cols = pick_group((size_t) ma_pad, (size_t) mb_pad);
    g_last_stats.sub_cols = sub_cols; g_last_stats.rg_rows = rg_rows;
    const int ta = ma_pad / TILE, tb = mb_pad / TILE;

    // ---- 3. pack operands
    float *Ap, *Bp, *nAp, *nBp;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ROWMIN, (size_t) ta * KK * 64 + ma_pad, &Ap));
    nAp = Ap + (size_t) ta * KK * 64;
    const size_t bset_stride = (size_t) tb * KK * 64;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_COLMIN, KCL * bset_stride + (size_t) KCL * mb_pad, &Bp));
    nBp = Bp + KCL * bset_stride;
    pack_kernel<<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, Ap, nAp);
    pack_kernel<<<dim3(cdiv(mb_pad, 256), KCL), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, Bp, nBp);
    const int n_sub = cdiv(mb_pad, sub_cols), n_rg = cdiv(ma_pad, rg_rows);
    float *gmaxB, *gmaxA;
    int* cl_of_rg;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BEST_B, (size_t) KCL * n_sub + 2 * (size_t) n_rg + 64, &gmaxB));
    gmaxA = gmaxB + (size_t) KCL * n_sub;
    cl_of_rg = (int*) (gmaxA + n_rg);
    group_max_kernel<<<dim3(n_sub, KCL), 256, 0, ctx->stream>>>(nBp, mb_pad, sub_cols, gmaxB);
    group_max_kernel<<<dim3(n_rg, 1), 256, 0, ctx->stream>>>(nAp, ma_pad, rg_rows, gmaxA);
    {
        std::vector<int> h(n_rg);
        for (int g = 0; g < n_rg; ++g) h[g] = A.h_blkcl[(size_t) g * (rg_rows / BLOCK_ROWS)];
        LGR_HIP(ctx, hipMemcpyAsync(cl_of_rg, h.data(), (size_t) n_rg * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }

    // ---- 4. MFMA pass
    float *rowmin, *colmin = nullptr;
    const size_t tab_floats = (size_t) n_sub * ma_pad + (both ? (size_t) n_rg * mb_pad : 0);
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BEST_A, tab_floats + 2 * ((size_t) ma + mb) + 64, &rowmin));
    if (both) colmin = rowmin + (size_t) n_sub * ma_pad;
    unsigned long long* bestA = (unsigned long long*) (rowmin + tab_floats + (tab_floats & 1));
    unsigned long long* bestB = bestA + ma;
    fill_u64<<<cdiv(ma + mb, 256), 256, 0, ctx->stream>>>(bestA, ma + mb, ~0ull);
    int n_cc = cdiv(mb_pad, CHUNK_COLS), n_sr = cdiv(ma_pad, SUPER_ROWS);
    (void) hipEventRecord(ctx->ev[9], ctx->stream);
    if (both)
        match_mfma<true><<<n_cc * n_sr, NTHR, 0, ctx->stream>>>(Ap, Bp, bset_stride, A.blkcl, nAp, ma_pad, mb_pad, sub_cols, rg_rows, rowmin, colmin, n_cc, n_sr);
    else
        match_mfma<false><<<n_cc * n_sr, NTHR, 0, ctx->stream>>>(Ap, Bp, bset_stride, A.blkcl, nAp, ma_pad, mb_pad, sub_cols, rg_rows, rowmin, colmin, n_cc, n_sr);
    (void) hipEventRecord(ctx->ev[10], ctx->stream);
    ctx->mfma_timed = 1;
    LGR_HIP(ctx, hipGetLastError());

    // ---- 5. exact rerank
    float *sortedA = nullptr, *sortedB;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_SORTED_B, (size_t) mb_pad * 33, &sortedB));
    gather_rows_kernel<<<cdiv((long long) mb_pad * 33, 256), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, sortedB);
    if (both) {
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_SORTED_A, (size_t) ma_pad * 33, &sortedA));
        gather_rows_kernel<<<cdiv((long long) ma_pad * 33, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, sortedA);
    }
    LGR_TRY((run_rerank<true>(ctx, rowmin, n_sub, sub_cols, d_a, A, nAp, nullptr, gmaxB, nullptr, d_b, sortedB, B, block, bestA,
                              d_ab_idx, d_ab_dist, &g_last_stats.items_ab, &g_last_stats.dense_ab)));
    if (both)
        LGR_TRY((run_rerank<false>(ctx, colmin, n_rg, rg_rows, d_b, B, nullptr, nBp, gmaxA, cl_of_rg, d_a, sortedA, A, block, bestB,
                                   d_ba_idx, d_ba_dist, &g_last_stats.items_ba, &g_last_stats.dense_ba)));
    return LGR_OK;
}

// duration of the last match_mfma launch in ms (hipEvents on the ctx stream); -1 if none
extern "C" int lgr_match_last_kernel_ms(lgr_ctx* ctx, float* ms) {
    if (!ctx || !ms) return LGR_ERR_INVALID_ARG;
    *ms = -1.f;
    if (!ctx->mfma_timed) return LGR_OK;
    LGR_HIP(ctx, hipEventSynchronize(ctx->ev[10]));
    LGR_HIP(ctx, hipEventElapsedTime(ms, ctx->ev[9], ctx->ev[10]));
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
