// lgr_match.hip -- brute-force FPFH matching (both directions from one MFMA pass) for gfx950.
//
// Replaces include/matching.h:594-634 matchBF<FPFH> (cv::BFMatcher(NORM_L2)::knnMatch, k = 1) and the cross-block
// merge src/common.cpp:517-529.  Result contract (bit-exact with the oracle): for every valid query row, the train
// row minimising the CANONICAL distance d = sqrtf(normL2Sqr) -- OpenCV 4.5.1 SSE lane order, see exact_l2() -- with
// ties broken "highest bf block, then lowest index inside the block"; NaN rows never match.
//
// Structure (DESIGN.md "matcher"):
//   1. match_prep      : pack rows into MFMA operand order, K = 34: A' = [-2a, 1], B' = [b, |b|^2]; norms; validity.
//   2. match_mfma      : S = A'.B' = |b|^2 - 2a.b on v_mfma_f32_32x32x2_f32; fused epilogue keeps only
//                        min_b S per (row, column group) and min_a (S + |a|^2) per (column, row group).
//                        This is a FILTER: its rounding error is bounded by `margin`.
//   3. rerank_*        : every group whose filtered minimum is within the proven error margin of the row's best is
//                        rescanned with the exact canonical distance; ties resolved with the reference's rules.
#include "lgr_internal.h"

namespace {

constexpr int KK = 17;              // K = 34 -> 17 MFMA steps of k = 2
constexpr int TILE = 32;
constexpr int RW = 2;               // row tiles per wave
constexpr int WAVES = 4;
constexpr int BLOCK_ROWS = TILE * RW * WAVES;   // 256
constexpr int RB_PER_SUPER = 16;
constexpr int SUPER_ROWS = BLOCK_ROWS * RB_PER_SUPER;   // 4096
constexpr int STAGE_TILES = 4;
constexpr int STAGE_COLS = STAGE_TILES * TILE;  // 128
constexpr int CHUNK_COLS = 4096;
constexpr int STAGE_FLOATS = STAGE_TILES * KK * 64;   // 4352
constexpr int PAD = 256;            // both sides padded to a multiple of this

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned f2key(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// ---------------------------------------------------------------------------------------------------------------
// 1. prep.  role 0: row operand [-2x, 1];  role 1: column operand [x, |x|^2].
// P layout: [tile][kk][half][i] floats (tile = 32 rows) so that MFMA lane l of step kk reads P[(tile*KK+kk)*64 + l].
// Invalid (non-finite) or padding rows: row operand -> [0.., 1], norm = +inf; column operand -> [0.., +inf].
__global__ void match_prep(const float* __restrict__ X, int m, int m_pad, int role, float* __restrict__ P,
                           float* __restrict__ nrm, unsigned* __restrict__ maxnorm_bits) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_pad) return;
    float v[33];
    bool valid = i < m;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 33; ++k) {
            v[k] = X[(size_t) i * 33 + k];
            valid = valid && (fabsf(v[k]) <= 3.4028234663852886e38f);   // finite test (NaN compares false)
        }
    }
    float n2 = 0.f;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 33; ++k) n2 = n2 + v[k] * v[k];
        atomicMax(maxnorm_bits, __float_as_uint(n2));
    } else {
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = 0.f;
        n2 = __uint_as_float(0x7f800000u);
    }
    nrm[i] = n2;
    int tile = i >> 5, r = i & 31;
    float* base = P + (size_t) tile * KK * 64 + r;
#pragma unroll
    for (int k = 0; k < 34; ++k) {
        float val;
        if (k < 33) val = role == 0 ? -2.0f * v[k] : v[k];
        else val = role == 0 ? 1.0f : n2;
        base[(k >> 1) * 64 + (k & 1) * 32] = val;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 2. MFMA filter kernel.  One workgroup = 4096 rows (16 row blocks of 256) x 4096 columns.
//    wave w of row block rb owns row tiles (rb*8 + 2w, +1); all waves share the column stage staged in LDS.
template <bool COLDIR>
__global__ __launch_bounds__(256) void match_mfma(const float* __restrict__ Ap, const float* __restrict__ Bp,
                                                  const float* __restrict__ nA, int ma_pad, int mb_pad,
                                                  int sub_cols, int rg_rows,
                                                  float* __restrict__ rowmin /* [mb_pad/sub_cols][ma_pad] */,
                                                  float* __restrict__ colmin /* [ma_pad/rg_rows][mb_pad] */,
                                                  int n_cc, int n_sr) {
    __shared__ float Bs[STAGE_FLOATS];
    __shared__ unsigned cmin_s[CHUNK_COLS];

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
    const float INF = __uint_as_float(0x7f800000u);

    if (COLDIR) {
        for (int i = tid; i < CHUNK_COLS; i += 256) cmin_s[i] = 0xffffffffu;
    }

    for (int rbi = 0; rbi < n_rb; ++rbi) {
        const int rb = rb0 + rbi;
        const int row_tile = rb * (BLOCK_ROWS / TILE) + wave * RW;
        // A fragments (coalesced 256-B loads) and the |a|^2 of the 16 rows each lane's accumulators cover
        float a[RW][KK];
        float na[RW][16];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) a[r][kk] = Ap[((size_t) (row_tile + r) * KK + kk) * 64 + lane];
            if (COLDIR) {
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    na[r][g] = nA[(row_tile + r) * TILE + (g & 3) + 8 * (g >> 2) + 4 * half];
            }
        }
        float rmin[RW][16];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) rmin[r][g] = INF;

        for (int st = 0; st < n_stages; ++st) {
            __syncthreads();
            {
                const float4* src = reinterpret_cast<const float4*>(Bp + ((size_t) (col_tile0 + st * STAGE_TILES)) * KK * 64);
                float4* dst = reinterpret_cast<float4*>(Bs);
                for (int i = tid; i < STAGE_FLOATS / 4; i += 256) dst[i] = src[i];
            }
            __syncthreads();
#pragma unroll 1
            for (int ct = 0; ct < STAGE_TILES; ++ct) {
                float b[KK];
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) b[kk] = Bs[(ct * KK + kk) * 64 + lane];
                f32x16 acc[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[r][g] = 0.f;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int r = 0; r < RW; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][kk], b[kk], acc[r], 0, 0, 0);
                float cm = INF;
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        float s = acc[r][g];
                        rmin[r][g] = __builtin_fminf(rmin[r][g], s);
                        if (COLDIR) cm = __builtin_fminf(cm, s + na[r][g]);
                    }
                if (COLDIR) {
                    cm = __builtin_fminf(cm, __shfl_xor(cm, 32));
                    if (lane < 32) atomicMin(&cmin_s[(st * STAGE_TILES + ct) * TILE + lane], f2key(cm));
                }
            }
            // flush the row minima of this column group
            if (((st + 1) % sub_stages) == 0 || st + 1 == n_stages) {
                int sub = (col_tile0 * TILE + st * STAGE_COLS) / sub_cols;
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        float v = rmin[r][g];
                        v = __builtin_fminf(v, __shfl_xor(v, 1));
                        v = __builtin_fminf(v, __shfl_xor(v, 2));
                        v = __builtin_fminf(v, __shfl_xor(v, 4));
                        v = __builtin_fminf(v, __shfl_xor(v, 8));
                        v = __builtin_fminf(v, __shfl_xor(v, 16));
                        if ((lane & 31) == 0)
                            rowmin[(size_t) sub * ma_pad + (row_tile + r) * TILE + (g & 3) + 8 * (g >> 2) + 4 * half] = v;
                        rmin[r][g] = INF;
                    }
            }
        }
        if (COLDIR && (((rbi + 1) % rg_blocks) == 0 || rbi + 1 == n_rb)) {
            __syncthreads();
            int rg = (rb * BLOCK_ROWS) / rg_rows;
            int ncols = n_coltiles * TILE;
            for (int i = tid; i < ncols; i += 256) {
                colmin[(size_t) rg * mb_pad + col_tile0 * TILE + i] = key2f(cmin_s[i]);
                cmin_s[i] = 0xffffffffu;
            }
            // the next iteration's first __syncthreads() orders these resets before any new atomicMin
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
// 3a. per query: best filtered value over groups, candidate groups within the margin -> work items (or dense flag)
// table[g][q_pad] ; for the row direction the values are S = d2 - |q|^2, for the column direction d2 (a constant
// offset per query does not matter).
constexpr int MAXC = 8;
struct RerankCounters { unsigned n_items; unsigned n_dense; unsigned overflow; unsigned pad; };

__global__ void rerank_select(const float* __restrict__ table, int n_groups, int nq, int q_pad,
                              const float* __restrict__ nQ, const unsigned* __restrict__ maxnorm_q_bits,
                              const unsigned* __restrict__ maxnorm_t_bits,
                              int table_is_d2,
                              unsigned long long* __restrict__ best, uint2* __restrict__ items, unsigned cap_items,
                              unsigned* __restrict__ dense, RerankCounters* __restrict__ cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    best[i] = ~0ull;
    float nq2 = nQ[i];
    if (!(nq2 < 3.0e38f)) return;   // invalid query: stays unmatched
    float m = __uint_as_float(0x7f800000u);
    for (int g = 0; g < n_groups; ++g) m = fminf(m, table[(size_t) g * q_pad + i]);
    if (!(m < 3.0e38f)) return;     // no valid train row at all
    // proven bound on |filtered - true| for every pair of this query (DESIGN.md "matcher margin"):
    //   fma chain of 34 products + norm rounding, gamma_n = n*u/(1-n*u), u = 2^-24; safety factor 2
    double u = 5.9604644775390625e-8;
    double g36 = 36 * u / (1 - 36 * u);
    double nt = (double) __uint_as_float(*maxnorm_t_bits);
    double nqm = (double) __uint_as_float(*maxnorm_q_bits);
    double sq = sqrt((double) nq2) + sqrt(nt);
    double eps = 2.0 * g36 * sq * sq + 2.0 * g36 * (nqm + nt);
    // relative slack: canonical distances of two rows whose true d2 differ by < 1e-5 relative may tie or swap
    double d2 = fmax(table_is_d2 ? (double) m : (double) m + (double) nq2, 0.0);
    double margin = 2.0 * eps + 1e-5 * d2 + 4.0 * u * fabs((double) m) + 1e-30;   // last terms: float rounding of thr
    float thr = (float) ((double) m + margin);
    if (thr < m) thr = m;
    int nc = 0;
    for (int g = 0; g < n_groups; ++g) nc += table[(size_t) g * q_pad + i] <= thr ? 1 : 0;
    if (nc > MAXC) {
        unsigned p = atomicAdd(&cnt->n_dense, 1u);
        dense[p] = (unsigned) i;
        return;
    }
    unsigned p = atomicAdd(&cnt->n_items, (unsigned) nc);
    if (p + nc > cap_items) { atomicExch(&cnt->overflow, 1u); return; }
    for (int g = 0; g < n_groups; ++g)
        if (table[(size_t) g * q_pad + i] <= thr) items[p++] = make_uint2((unsigned) i, (unsigned) g);
}

// 3b. one wave per (query, group) item: exact distances to the group's train rows.
__global__ __launch_bounds__(256) void rerank_items(const float* __restrict__ Q, const float* __restrict__ T,
                                                    const float* __restrict__ nT, int nt, int group_size,
                                                    int block, int nblocks, const uint2* __restrict__ items,
                                                    const RerankCounters* __restrict__ cnt,
                                                    unsigned long long* __restrict__ best) {
    unsigned n_items = cnt->n_items;
    int lane = threadIdx.x & 63;
    for (unsigned it = blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += gridDim.x * 4) {
        uint2 w = items[it];
        float q[33];
        const float* qp = Q + (size_t) w.x * 33;
#pragma unroll
        for (int k = 0; k < 33; ++k) q[k] = qp[k];
        int j0 = (int) w.y * group_size, j1 = min(nt, j0 + group_size);
        unsigned long long bk = ~0ull;
        for (int j = j0 + lane; j < j1; j += 64) {
            if (!(nT[j] < 3.0e38f)) continue;    // invalid train row never matches
            float t[33];
            const float* tp = T + (size_t) j * 33;
#pragma unroll
            for (int k = 0; k < 33; ++k) t[k] = tp[k];
            float d = exact_l2(q, t);
            if (!(d < 3.4028234663852886e38f)) continue;   // batchDistance keeps only d < FLT_MAX
            unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(j, block, nblocks);
            bk = key < bk ? key : bk;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long ok = __shfl_xor(bk, o);
            bk = ok < bk ? ok : bk;
        }
        if (lane == 0 && bk != ~0ull) atomicMin(&best[w.x], bk);
    }
}

// 3c. dense fallback: queries with more than MAXC candidate groups are matched by plain exact brute force.
__global__ __launch_bounds__(256) void rerank_dense(const float* __restrict__ Q, const float* __restrict__ T,
                                                    const float* __restrict__ nT, int nt, int block, int nblocks,
                                                    const unsigned* __restrict__ dense,
                                                    const RerankCounters* __restrict__ cnt,
                                                    unsigned long long* __restrict__ best) {
    __shared__ float Ts[64 * 33];
    __shared__ float nTs[64];
    unsigned n_dense = cnt->n_dense;
    for (unsigned base = blockIdx.x * 256; base < n_dense; base += gridDim.x * 256) {
        unsigned di = base + threadIdx.x;
        bool act = di < n_dense;
        unsigned qi = act ? dense[di] : 0;
        float q[33];
#pragma unroll
        for (int k = 0; k < 33; ++k) q[k] = act ? Q[(size_t) qi * 33 + k] : 0.f;
        unsigned long long bk = ~0ull;
        for (int j0 = 0; j0 < nt; j0 += 64) {
            __syncthreads();
            int nj = min(64, nt - j0);
            for (int i = threadIdx.x; i < nj * 33; i += 256) Ts[i] = T[(size_t) j0 * 33 + i];
            if (threadIdx.x < nj) nTs[threadIdx.x] = nT[j0 + threadIdx.x];
            __syncthreads();
            if (act) {
                for (int jj = 0; jj < nj; ++jj) {
                    if (!(nTs[jj] < 3.0e38f)) continue;
                    float d = exact_l2(q, Ts + jj * 33);
                    if (!(d < 3.4028234663852886e38f)) continue;
                    unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(j0 + jj, block, nblocks);
                    bk = key < bk ? key : bk;
                }
            }
        }
        if (act && bk != ~0ull) atomicMin(&best[qi], bk);
        __syncthreads();
    }
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

// choose a group size (multiple of `base`, divides 4096) so that the table stays below ~3 GB
int pick_group(size_t rows_pad, size_t other_pad, int base) {
    int g = base;
    while (g < 4096 && (other_pad / g) * rows_pad * 4 > (size_t) 3 << 30) g *= 2;
    return g;
}

int run_rerank(lgr_ctx* ctx, const float* table, int n_groups, int group_size, const float* Q, int nq, int q_pad,
               const float* nQ, const unsigned* maxq, const float* T, int nt, const float* nT, const unsigned* maxt,
               int block, int slot_best, int table_is_d2, int32_t* d_idx, float* d_dist) {
    unsigned long long* best;
    LGR_TRY(lgr_ws_t(ctx, slot_best, (size_t) q_pad, &best));
    unsigned cap_items = (unsigned) q_pad * 4u + 1024u;
    uint2* items;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ITEMS, (size_t) cap_items, &items));
    unsigned* dense;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_DENSE, (size_t) q_pad, &dense));
    RerankCounters* cnt;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, 64, (char**) &cnt));
    cnt = (RerankCounters*) ((char*) cnt + 32);   // first 32 bytes hold the two max-norm words
    LGR_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(RerankCounters), ctx->stream));
    int nblocks = (nt + block - 1) / block;
    rerank_select<<<cdiv(nq, 256), 256, 0, ctx->stream>>>(table, n_groups, nq, q_pad, nQ, maxq, maxt, table_is_d2, best,
                                                           items, cap_items, dense, cnt);
    int grid = ctx->n_cu * 8;
    rerank_items<<<grid, 256, 0, ctx->stream>>>(Q, T, nT, nt, group_size, block, nblocks, items, cnt, best);
    rerank_dense<<<ctx->n_cu * 2, 256, 0, ctx->stream>>>(Q, T, nT, nt, block, nblocks, dense, cnt, best);
    // item-list overflow (pathological: > 4 candidate groups per query on average): redo everything densely
    RerankCounters* h;
    LGR_TRY(lgr_pinned(ctx, sizeof(RerankCounters), (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, cnt, sizeof(RerankCounters), hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h->overflow) {
        // mark every query dense and rerun the brute-force fallback (correct, slow; never seen on FPFH data)
        std::vector<unsigned> all(nq);
        for (int i = 0; i < nq; ++i) all[i] = (unsigned) i;
        LGR_HIP(ctx, hipMemcpyAsync(dense, all.data(), (size_t) nq * 4, hipMemcpyHostToDevice, ctx->stream));
        RerankCounters hc{0u, (unsigned) nq, 0u, 0u};
        LGR_HIP(ctx, hipMemcpyAsync(cnt, &hc, sizeof hc, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rerank_dense<<<ctx->n_cu * 2, 256, 0, ctx->stream>>>(Q, T, nT, nt, block, nblocks, dense, cnt, best);
    }
    rerank_finalize<<<cdiv(nq, 256), 256, 0, ctx->stream>>>(best, nq, block, nblocks, d_idx, d_dist);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

}  // namespace

// statistics of the last match call (candidates per query etc.), for bench/diagnostics
struct lgr_match_stats { unsigned items_ab, dense_ab, items_ba, dense_ba; int sub_cols, rg_rows; };
static lgr_match_stats g_last_stats;
extern "C" int lgr_match_last_stats(unsigned* out6) {
    out6[0] = g_last_stats.items_ab; out6[1] = g_last_stats.dense_ab; out6[2] = g_last_stats.items_ba;
    out6[3] = g_last_stats.dense_ba; out6[4] = (unsigned) g_last_stats.sub_cols; out6[5] = (unsigned) g_last_stats.rg_rows;
    return LGR_OK;
}

static int match_impl(lgr_ctx* ctx, const float* d_a, int ma, const float* d_b, int mb, int block,
                      int32_t* d_ab_idx, float* d_ab_dist, int32_t* d_ba_idx, float* d_ba_dist) {
    LGR_CHECK(ctx, ctx && d_a && d_b && d_ab_idx && d_ab_dist, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, ma >= 0 && mb >= 0 && block > 0, LGR_ERR_INVALID_ARG);
    bool both = d_ba_idx != nullptr;
    if (both) LGR_CHECK(ctx, d_ba_dist != nullptr, LGR_ERR_INVALID_ARG);
    if (ma == 0 && mb == 0) return LGR_OK;
    if (ma == 0 || mb == 0) {   // nothing to match against: every query unmatched
        if (ma) { LGR_HIP(ctx, hipMemsetAsync(d_ab_idx, 0xff, (size_t) ma * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ab_dist, 0, (size_t) ma * 4, ctx->stream)); }
        if (mb && both) { LGR_HIP(ctx, hipMemsetAsync(d_ba_idx, 0xff, (size_t) mb * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ba_dist, 0, (size_t) mb * 4, ctx->stream)); }
        return LGR_OK;
    }
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    int ma_pad = pad_to(ma, PAD), mb_pad = pad_to(mb, PAD);
    float *Ap, *Bp, *nA, *nB;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_AP, (size_t) ma_pad / TILE * KK * 64, &Ap));
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BP, (size_t) mb_pad / TILE * KK * 64, &Bp));
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_NA, (size_t) ma_pad, &nA));
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_NB, (size_t) mb_pad, &nB));
    unsigned* maxn;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, 64, (char**) &maxn));
    LGR_HIP(ctx, hipMemsetAsync(maxn, 0, 32, ctx->stream));
    match_prep<<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, ma, ma_pad, 0, Ap, nA, maxn + 0);
    match_prep<<<cdiv(mb_pad, 256), 256, 0, ctx->stream>>>(d_b, mb, mb_pad, 1, Bp, nB, maxn + 1);

    int sub_cols = pick_group((size_t) ma_pad, (size_t) mb_pad, 1024);
    int rg_rows = pick_group((size_t) mb_pad, (size_t) ma_pad, 1024);
    int n_sub = cdiv(mb_pad, sub_cols), n_rg = cdiv(ma_pad, rg_rows);
    float *rowmin, *colmin = nullptr;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ROWMIN, (size_t) n_sub * ma_pad, &rowmin));
    if (both) LGR_TRY(lgr_ws_t(ctx, WS_MATCH_COLMIN, (size_t) n_rg * mb_pad, &colmin));
    int n_cc = cdiv(mb_pad, CHUNK_COLS), n_sr = cdiv(ma_pad, SUPER_ROWS);
    if (both)
        match_mfma<true><<<n_cc * n_sr, 256, 0, ctx->stream>>>(Ap, Bp, nA, ma_pad, mb_pad, sub_cols, rg_rows, rowmin, colmin, n_cc, n_sr);
    else
        match_mfma<false><<<n_cc * n_sr, 256, 0, ctx->stream>>>(Ap, Bp, nA, ma_pad, mb_pad, sub_cols, rg_rows, rowmin, colmin, n_cc, n_sr);
    LGR_HIP(ctx, hipGetLastError());

    LGR_TRY(run_rerank(ctx, rowmin, n_sub, sub_cols, d_a, ma, ma_pad, nA, maxn + 0, d_b, mb, nB, maxn + 1, block,
                       WS_MATCH_BEST_A, 0, d_ab_idx, d_ab_dist));
    RerankCounters* h = (RerankCounters*) ctx->pinned;
    g_last_stats.items_ab = h->n_items; g_last_stats.dense_ab = h->n_dense;
    g_last_stats.sub_cols = sub_cols; g_last_stats.rg_rows = rg_rows;
    if (both) {
        LGR_TRY(run_rerank(ctx, colmin, n_rg, rg_rows, d_b, mb, mb_pad, nB, maxn + 1, d_a, ma, nA, maxn + 0, block,
                           WS_MATCH_BEST_B, 1, d_ba_idx, d_ba_dist));
        h = (RerankCounters*) ctx->pinned;
        g_last_stats.items_ba = h->n_items; g_last_stats.dense_ba = h->n_dense;
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
    LGR_CHECK(ctx, d_ba_idx && d_ba_dist, LGR_ERR_INVALID_ARG);
    return match_impl(ctx, d_a33, ma, d_b33, mb, block, d_ab_idx, d_ab_dist, d_ba_idx, d_ba_dist);
}

extern "C" int lgr_match_bf(lgr_ctx* ctx, const float* q33, int mq, const float* t33, int mt, int block,
                            int32_t* idx, float* dist) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (q33 || mq == 0) && (t33 || mt == 0) && idx && dist && mq >= 0 && mt >= 0, LGR_ERR_INVALID_ARG);
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
