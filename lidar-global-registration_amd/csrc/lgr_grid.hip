// lgr_grid.hip -- bounding boxes, uniform grid build, exact k-NN tables, smoothed densities (gfx950).
//
// include/common.h:266-280 calculateBoundingBox; src/common.cpp:531-547 calculateSmoothedDensities and :202-208
// calculatePointCloudDensity.  The reference queries pcl::KdTreeFLANN; here a counting-sorted uniform grid is built
// per cloud (rocPRIM radix sort = plumbing) and queried by the hand-written kernels in lgr_grid.cuh.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>

#include "lgr_grid.cuh"
#include "lgr_knn_wave.cuh"

namespace {

__device__ __forceinline__ unsigned fkey(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float fkey_inv(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// out[0..2] true min keys, [3..5] true max keys (finite points only), [6..8] quirk min, [9..11] quirk max
__global__ void bbox_kernel(const float* __restrict__ pts, int n, unsigned* __restrict__ out) {
    float tmn[3] = {INFINITY, INFINITY, INFINITY}, tmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float qmn[3] = {3.4028234663852886e38f, 3.4028234663852886e38f, 3.4028234663852886e38f};
    float qmx[3] = {1.17549435e-38f, 1.17549435e-38f, 1.17549435e-38f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float p[3] = {pts[(size_t) i * 12], pts[(size_t) i * 12 + 1], pts[(size_t) i * 12 + 2]};
        bool fin = lgr_finite3(p[0], p[1], p[2]);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (fin) { tmn[a] = fminf(tmn[a], p[a]); tmx[a] = fmaxf(tmx[a], p[a]); }
            qmn[a] = (p[a] < qmn[a]) ? p[a] : qmn[a];   // std::min(mn, p)
            qmx[a] = (qmx[a] < p[a]) ? p[a] : qmx[a];   // std::max(mx, p)
        }
    }
    // wave reduce, then one LDS stage per block, then 12 atomics per BLOCK (the grid is small)
    __shared__ float sh[4][12];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            tmn[a] = fminf(tmn[a], __shfl_xor(tmn[a], o)); tmx[a] = fmaxf(tmx[a], __shfl_xor(tmx[a], o));
            qmn[a] = fminf(qmn[a], __shfl_xor(qmn[a], o)); qmx[a] = fmaxf(qmx[a], __shfl_xor(qmx[a], o));
        }
        if ((threadIdx.x & 63) == 0) {
            int w = threadIdx.x >> 6;
            sh[w][a] = tmn[a]; sh[w][3 + a] = tmx[a]; sh[w][6 + a] = qmn[a]; sh[w][9 + a] = qmx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        int k = threadIdx.x;
        bool is_min = (k < 3) || (k >= 6 && k < 9);
        float v = sh[0][k];
        for (int w = 1; w < (int) (blockDim.x >> 6); ++w) v = is_min ? fminf(v, sh[w][k]) : fmaxf(v, sh[w][k]);
        if (is_min) atomicMin(&out[k], fkey(v)); else atomicMax(&out[k], fkey(v));
    }
}

__global__ void cell_keys(const float* __restrict__ pts, int n, float ox, float oy, float oz, float h, int dx, int dy, int dz,
                          unsigned invalid_key, unsigned* __restrict__ keys, int* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = pts[(size_t) i * 12], y = pts[(size_t) i * 12 + 1], z = pts[(size_t) i * 12 + 2];
    unsigned k = invalid_key;
    if (lgr_finite3(x, y, z)) {
        int cx = min(max(lgr_cellc(x, ox, h), 0), dx - 1), cy = min(max(lgr_cellc(y, oy, h), 0), dy - 1), cz = min(max(lgr_cellc(z, oz, h), 0), dz - 1);
        k = (unsigned) ((cz * dy + cy) * dx + cx);
    }
    keys[i] = k; vals[i] = i;
}

// cell_start[c] = number of sorted keys below c, for every cell c in [0, ncell] (cell_start[ncell] = number of valid points; invalid
// points carry the key ncell and sort last).  The dense table can have 100+ cells per point (a surface scan in its 3-D bounding box), so
// it is WRITTEN ONCE, straight from the sorted keys, instead of memset + per-cell counts + a scan over all cells (three more passes
// over up to 1 GB): a coarse table (one entry per CSF_CELLS cells) is filled from the keys, then every workgroup of cell_start_fill
// finds the few keys of its CSF_CELLS cells between two coarse entries and writes its piece of the table.
constexpr int CSF_T = 256, CSF_SHIFT = 10, CSF_CELLS = 1 << CSF_SHIFT;   // 4 cells per thread
static_assert(CSF_CELLS == 4 * CSF_T, "one int4 store per thread");
// coarse[b] = number of keys below b * CSF_CELLS, b in [0, nblk].  Keys driven: lane <-> boundary i between keys i-1 and i; the
// coarse entries in (key[i-1] >> shift, key[i] >> shift] all equal i and are filled by the whole wave (gaps can be long).
__global__ __launch_bounds__(256) void coarse_start_fill(const unsigned* __restrict__ keys, int n, int nblk, int* __restrict__ coarse) {
    const int lane = threadIdx.x & 63;
    const int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    const int i = base + lane;
    int lo = 1, hi = 0;
    if (i <= n) {
        lo = i == 0 ? 0 : (int) (keys[i - 1] >> CSF_SHIFT) + 1;
        hi = i == n ? nblk : (int) (keys[i] >> CSF_SHIFT);
    }
    unsigned long long todo = __ballot(lo <= hi);
    while (todo) {
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int lo_j = __builtin_amdgcn_readlane(lo, j), hi_j = __builtin_amdgcn_readlane(hi, j);
        for (int c = lo_j + lane; c <= hi_j; c += 64) coarse[c] = base + j;
    }
}
__global__ __launch_bounds__(CSF_T) void cell_start_fill(const unsigned* __restrict__ keys, const int* __restrict__ coarse, unsigned ncell, int* __restrict__ start) {
    const int k0 = coarse[blockIdx.x], k1 = coarse[blockIdx.x + 1];   // the keys of this workgroup's cells: [k0, k1)
    const unsigned c = (unsigned) blockIdx.x * CSF_CELLS + 4u * threadIdx.x;
    if (c > ncell) return;
    int lo = k0, hi = k1;                                             // first key >= c
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < c) lo = mid + 1; else hi = mid; }
    int4 v;
    v.x = lo;
    while (lo < k1 && keys[lo] < c + 1u) ++lo;
    v.y = lo;
    while (lo < k1 && keys[lo] < c + 2u) ++lo;
    v.z = lo;
    while (lo < k1 && keys[lo] < c + 3u) ++lo;
    v.w = lo;
    if (c + 3u <= ncell) {
        *reinterpret_cast<int4*>(start + c) = v;
    } else {
        start[c] = v.x;
        if (c + 1u <= ncell) start[c + 1] = v.y;
        if (c + 2u <= ncell) start[c + 2] = v.z;
    }
}

__global__ void gather_points(const float* __restrict__ pts, const int* __restrict__ vals, int nvalid,
                              float4* __restrict__ pxyz, float4* __restrict__ pnrm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvalid) return;
    int o = vals[i];
    const float4* p = reinterpret_cast<const float4*>(pts + (size_t) o * 12);
    float4 a = p[0], b = p[1], c = p[2];
    pxyz[i] = make_float4(a.x, a.y, a.z, __int_as_float(o));
    pnrm[i] = make_float4(b.x, b.y, b.z, c.y);
}

constexpr int KNN_BLOCK = 128;

// mode 0: full table idx/d2 [nq][k];  mode 1: dk[i] = sqrt(d2[k-1]) (NaN if fewer), nn1[i] = idx[1] (i if fewer)
// by_grid (the queries are the grid's own points): 1 = thread t takes the point at sorted position t, so a wave's queries
// sit in one or two cells and walk the same rings; 2 = second launch in the original order for the points that are not in
// the grid (non-finite); 0 = queries in the order given.
template <int MODE>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(GridDev g, const float* __restrict__ q, int nq, int k,
                                                        int32_t* __restrict__ idx, float* __restrict__ d2, int by_grid) {
    extern __shared__ float smem[];
    float* sd = smem;
    int* si = (int*) (smem + (size_t) k * KNN_BLOCK);
    int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
    bool active = true;
    if (by_grid == 1) {
        i = lgr_xcd_tile(blockIdx.x, cdiv_dev(g.n, KNN_BLOCK)) * KNN_BLOCK + threadIdx.x;   // grid order, one contiguous range per XCD
        active = i < g.n;
        i = active ? __float_as_int(g.pxyz[i].w) : 0;
    } else if (i >= nq) { active = false; i = 0; }
    float x = 0.f, y = 0.f, z = 0.f;
    if (active) { x = q[(size_t) i * 12]; y = q[(size_t) i * 12 + 1]; z = q[(size_t) i * 12 + 2]; }
    if (by_grid == 2 && lgr_finite3(x, y, z)) active = false;   // done by the grid-ordered launch
    KnnList<KNN_BLOCK> L;
    L.init(sd, si, k, threadIdx.x);
    if (active && lgr_finite3(x, y, z) && g.n > 0) lgr_knn_query(g, x, y, z, L);
    if (MODE == 0) {
        // The lists leave through the workgroup: a wave takes one query's row at a time and its lanes write the k entries of that row --
        // one contiguous k * 4-byte segment per store instruction.  (A thread writing its own row touched 64 cache lines with 4 bytes each
        // per instruction, 2 k instructions per wave: the two 40-NN tables of the cluster filter are 320 MB of such stores per cloud.)
        // (each wave writes its own 64 queries: row ids and counts travel through v_readlane, no extra LDS -- one more KB would cost the
        // k = 40 launch its fourth workgroup per CU -- and no workgroup barrier)
        const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
        const int my_row = active ? i : -1, my_cnt = L.count;
        for (int r = 0; r < 64; ++r) {
            const int row = __builtin_amdgcn_readlane(my_row, r);
            if (row < 0) continue;
            const int cnt = __builtin_amdgcn_readlane(my_cnt, r);
            for (int j = lane; j < k; j += 64) {
                idx[(size_t) row * k + j] = j < cnt ? si[j * KNN_BLOCK + wbase + r] : -1;
                if (d2) d2[(size_t) row * k + j] = j < cnt ? sd[j * KNN_BLOCK + wbase + r] : INFINITY;
            }
        }
    } else if (active) {
        d2[i] = L.count >= k ? __builtin_sqrtf(L.dist(k - 1)) : __uint_as_float(0x7fc00000u);
        idx[i] = L.count >= 2 ? L.index(1) : i;
    }
}

// The same tables from the wave-per-query search (lgr_knn_wave.cuh): a wave answers WK_Q consecutive queries, one after the
// other, carrying the k-th distance of one as the first guess of the next.  by_grid as above.
constexpr int WK_WAVES = 4, WK_Q = 32;
template <int MODE, int KPL>
__device__ __forceinline__ void knn_wave_emit(const WaveKnn<KPL>& W, int m, int i, int k, int32_t* __restrict__ idx, float* __restrict__ d2) {
    const int lane = threadIdx.x & 63;
    if (MODE == 0) {
        const int mk = min(m, k);
#pragma unroll
        for (int j = 0; j < KPL; ++j)
            if (m > 0 && W.rank[j] < mk) {
                idx[(size_t) i * k + W.rank[j]] = (int) (unsigned) W.key[j];
                if (d2) d2[(size_t) i * k + W.rank[j]] = wk_key_d2(W.key[j]);
            }
        for (int j = mk + lane; j < k; j += 64) {
            idx[(size_t) i * k + j] = -1;
            if (d2) d2[(size_t) i * k + j] = INFINITY;
        }
    } else {
        if (m >= k) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) if (W.rank[j] == k - 1) d2[i] = __builtin_sqrtf(wk_key_d2(W.key[j]));
        } else if (lane == 0) d2[i] = __uint_as_float(0x7fc00000u);
        if (m >= 2) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) if (W.rank[j] == 1) idx[i] = (int) (unsigned) W.key[j];
        } else if (lane == 0) idx[i] = i;
    }
}

// by_grid 0 / 2: queries in the order given, one after the other (lgr_wave_knn)
template <int MODE, int KPL>
__global__ __launch_bounds__(64 * WK_WAVES) void knn_wave_kernel(GridDev g, const float* __restrict__ q, int nq, int k,
                                                                  int32_t* __restrict__ idx, float* __restrict__ d2, int by_grid, float r2_init) {
    __shared__ unsigned long long sbuf[WK_WAVES][WaveKnn<KPL>::BUF];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t0 = (blockIdx.x * WK_WAVES + wv) * WK_Q;
    // the wave's queries: lane l holds query t0 + l
    int qi = -1;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (lane < WK_Q && t0 + lane < nq) {
        qi = t0 + lane;
        qx = q[(size_t) qi * 12]; qy = q[(size_t) qi * 12 + 1]; qz = q[(size_t) qi * 12 + 2];
        if (by_grid == 2 && lgr_finite3(qx, qy, qz)) qi = -1;   // done by the grid-ordered launch
    }
    float guess = r2_init;
    WaveKnn<KPL> W;
    const int nloc = min(WK_Q, nq - t0);
    for (int l = 0; l < nloc; ++l) {
        const int i = __builtin_amdgcn_readlane(qi, l);
        if (i < 0) continue;
        const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qx), l));
        const float y = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qy), l));
        const float z = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qz), l));
        W.m = 0;
        if (lgr_finite3(x, y, z) && g.n > 0) lgr_wave_knn<KPL>(g, x, y, z, k, guess, sbuf[wv], W);
        knn_wave_emit<MODE, KPL>(W, W.m, i, k, idx, d2);
    }
}

// by_grid 1: the grid's own points in sorted order, 64 per wave, cell by cell (lgr_wave_knn_tile)
template <int MODE, int KPL>
__global__ __launch_bounds__(64 * WK_WAVES) void knn_tile_kernel(GridDev g, int k, int32_t* __restrict__ idx, float* __restrict__ d2, float r2_init) {
    __shared__ unsigned long long sbuf[WK_WAVES][WaveKnn<KPL>::BUF];
    __shared__ int srow[WK_WAVES][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_tiles = cdiv_dev(g.n, WK_WAVES * 64);
    const int tile = lgr_xcd_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int t = (tile * WK_WAVES + wv) * 64 + lane;
    int qi = -1;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (t < g.n) {
        const float4 p = g.pxyz[t];
        qi = __float_as_int(p.w); px = p.x; py = p.y; pz = p.z;
    }
    float guess = r2_init;
    lgr_wave_knn_tile<KPL>(g, qi >= 0, px, py, pz, k, guess, sbuf[wv], srow[wv], [&](int l, const WaveKnn<KPL>& W) {
        knn_wave_emit<MODE, KPL>(W, W.m, __builtin_amdgcn_readlane(qi, l), k, idx, d2);
    });
}

// first threshold of a wave's first query: the radius that holds 1.25 k points at `ppc` points per cell of a surface
static float wk_r2_init(const GridDev& g, int k, float ppc) { return g.h * g.h * 1.25f * (float) k / (3.14159265f * ppc); }

template <int MODE>
static void knn_wave_launch(lgr_ctx* ctx, const GridDev& g, const float* d_q, int nq, int k, int32_t* d_idx, float* d_d2, int by_grid, float ppc) {
    const float r2i = wk_r2_init(g, k, ppc);
    const int T = 64 * WK_WAVES;
    if (by_grid == 1) {
        if (g.n <= 0) return;
        const int grid = lgr_xcd_grid(cdiv(g.n, WK_WAVES * 64));
        if (k <= 40) knn_tile_kernel<MODE, 1><<<grid, T, 0, ctx->stream>>>(g, k, d_idx, d_d2, r2i);
        else if (k <= 96) knn_tile_kernel<MODE, 2><<<grid, T, 0, ctx->stream>>>(g, k, d_idx, d_d2, r2i);
        else knn_tile_kernel<MODE, 4><<<grid, T, 0, ctx->stream>>>(g, k, d_idx, d_d2, r2i);
        return;
    }
    if (nq <= 0) return;
    const int grid = cdiv(nq, WK_WAVES * WK_Q);
    if (k <= 40) knn_wave_kernel<MODE, 1><<<grid, T, 0, ctx->stream>>>(g, d_q, nq, k, d_idx, d_d2, by_grid, r2i);
    else if (k <= 96) knn_wave_kernel<MODE, 2><<<grid, T, 0, ctx->stream>>>(g, d_q, nq, k, d_idx, d_d2, by_grid, r2i);
    else knn_wave_kernel<MODE, 4><<<grid, T, 0, ctx->stream>>>(g, d_q, nq, k, d_idx, d_d2, by_grid, r2i);
}

__global__ void density_min(const float* __restrict__ dk, const int32_t* __restrict__ nn1, int n, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = dk[i], b = dk[nn1[i]];
    out[i] = (b < a) ? b : a;   // std::min(a, b)
}

}  // namespace

// the reduction only: *d_keys12 = 12 order-preserving keys on the device (key -> float: lgr_bbox_key_inv, lgr_internal.h), valid until the
// context's next bounding box; no host synchronisation
int lgr_bbox_launch(lgr_ctx* ctx, const float* d_pts, int n, const unsigned** d_keys12) {
    unsigned* d;
    LGR_TRY(lgr_ws_t(ctx, WS_GRID_MISC, 16, &d));
    unsigned init[12];
    unsigned pinf = 0x7f800000u | 0x80000000u;         // fkey(+inf)
    unsigned ninf = ~0xff800000u;                       // fkey(-inf)
    float fmx = 3.4028234663852886e38f, fmn = 1.17549435e-38f;
    unsigned kfmx, kfmn;
    memcpy(&kfmx, &fmx, 4); kfmx |= 0x80000000u;
    memcpy(&kfmn, &fmn, 4); kfmn |= 0x80000000u;
    for (int a = 0; a < 3; ++a) { init[a] = pinf; init[3 + a] = ninf; init[6 + a] = kfmx; init[9 + a] = kfmn; }
    LGR_HIP(ctx, hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    if (n > 0) bbox_kernel<<<std::min(cdiv(n, 256), ctx->n_cu), 256, 0, ctx->stream>>>(d_pts, n, d);
    LGR_HIP(ctx, hipGetLastError());
    *d_keys12 = d;
    return LGR_OK;
}

int lgr_bbox_host(lgr_ctx* ctx, const float* d_pts, int n, float* out12) {
    const unsigned* d;
    LGR_TRY(lgr_bbox_launch(ctx, d_pts, n, &d));
    unsigned* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, d, 48, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 12; ++i) out12[i] = fkey_inv(h[i]);
    return LGR_OK;
}

extern "C" int lgr_bbox_dev(lgr_ctx* ctx, const float* d_pts, int n, float* d_min3_max3) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_pts || n == 0) && d_min3_max3 && n >= 0, LGR_ERR_INVALID_ARG);
    float bb[12];
    LGR_TRY(lgr_bbox_host(ctx, d_pts, n, bb));
    LGR_HIP(ctx, hipMemcpyAsync(d_min3_max3, bb + 6, 24, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // bb is a stack buffer
    return LGR_OK;
}

int lgr_grid_build(lgr_ctx* ctx, int sb, const float* d_pts, int n, float h, float target, GridDev* out) {
    float bb[12];
    LGR_TRY(lgr_bbox_host(ctx, d_pts, n, bb));
    float mn[3] = {bb[0], bb[1], bb[2]}, mx[3] = {bb[3], bb[4], bb[5]};
    if (!(mn[0] <= mx[0])) { for (int a = 0; a < 3; ++a) { mn[a] = 0.f; mx[a] = 0.f; } }
    if (!(h > 0.f)) {
        float e[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        std::sort(e, e + 3);
        float area = std::max(e[2] * e[1], 1e-12f);
        h = std::sqrt(area * target / (float) std::max(n, 1));
        h = std::max(h, std::max(e[2] / 1024.f, 1e-6f));
    }
    int dim[3];
    for (;;) {
        double cells = 1;
        for (int a = 0; a < 3; ++a) {
            dim[a] = (int) std::floor((mx[a] - mn[a]) / h) + 1;   // same float expression as lgr_cellc
            if (dim[a] < 1) dim[a] = 1;
            cells *= dim[a];
        }
        if (cells <= 256e6) break;
        h *= 2.f;
    }
    int ncell = dim[0] * dim[1] * dim[2];
    unsigned *keys, *keys2;
    int *vals, *vals2, *start;
    float4 *pxyz, *pnrm;
    LGR_TRY(lgr_ws_t(ctx, sb + 0, (size_t) n + 1, &keys));
    LGR_TRY(lgr_ws_t(ctx, sb + 1, (size_t) n + 1, &vals));
    LGR_TRY(lgr_ws_t(ctx, sb + 2, (size_t) n + 1, &keys2));
    LGR_TRY(lgr_ws_t(ctx, sb + 3, (size_t) n + 1, &vals2));
    LGR_TRY(lgr_ws_t(ctx, sb + 4, (size_t) ncell + 2, &start));
    LGR_TRY(lgr_ws_t(ctx, sb + 5, (size_t) n + 1, &pxyz));
    LGR_TRY(lgr_ws_t(ctx, sb + 6, (size_t) n + 1, &pnrm));
    int nvalid = 0;
    if (n > 0) {
        cell_keys<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_pts, n, mn[0], mn[1], mn[2], h, dim[0], dim[1], dim[2], (unsigned) ncell, keys, vals);
        int bits = 1;
        while (((unsigned long long) 1 << bits) <= (unsigned long long) ncell) ++bits;
        LGR_TRY(lgr_sort_pairs_u32(ctx, keys, keys2, vals, vals2, (size_t) n, 0, bits));
        const int nblk = (int) (((size_t) ncell + 1 + CSF_CELLS - 1) >> CSF_SHIFT);   // workgroups of cell_start_fill; cells 0..ncell
        int* coarse;
        LGR_TRY(lgr_ws_t(ctx, WS_GRID_MISC, (size_t) nblk + 1 + 64, &coarse));
        coarse += 64;   // (the first 256 bytes of the slot belong to lgr_bbox_host)
        coarse_start_fill<<<cdiv(n + 1, 256), 256, 0, ctx->stream>>>(keys2, n, nblk, coarse);
        cell_start_fill<<<nblk, CSF_T, 0, ctx->stream>>>(keys2, coarse, (unsigned) ncell, start);
        int* h_n;
        LGR_TRY(lgr_pinned(ctx, 64, (void**) &h_n));
        LGR_HIP(ctx, hipMemcpyAsync(h_n, start + ncell, 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        nvalid = *h_n;
        if (nvalid > 0) gather_points<<<cdiv(nvalid, 256), 256, 0, ctx->stream>>>(d_pts, vals2, nvalid, pxyz, pnrm);
    } else {
        LGR_HIP(ctx, hipMemsetAsync(start, 0, ((size_t) ncell + 2) * 4, ctx->stream));
    }
    LGR_HIP(ctx, hipGetLastError());
    out->ox = mn[0]; out->oy = mn[1]; out->oz = mn[2]; out->h = h;
    out->dx = dim[0]; out->dy = dim[1]; out->dz = dim[2];
    out->n = nvalid; out->cell_start = start; out->pxyz = pxyz; out->pnrm = pnrm;
    return LGR_OK;
}

extern "C" int lgr_knn_dev(lgr_ctx* ctx, const float* d_q, int nq, const float* d_pts, int n, int k, int32_t* d_idx, float* d_d2) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_d2 != nullptr, LGR_ERR_INVALID_ARG);
    return lgr_knn_lists(ctx, d_q, nq, d_pts, n, k, d_idx, d_d2);
}

// d_d2 == nullptr: the index lists alone (the cluster filter tests membership only)
int lgr_knn_lists(lgr_ctx* ctx, const float* d_q, int nq, const float* d_pts, int n, int k, int32_t* d_idx, float* d_d2) {
    LGR_CHECK(ctx, nq >= 0 && n >= 0 && k >= 1 && k <= 128 && (d_q || nq == 0) && (d_pts || n == 0) && d_idx, LGR_ERR_INVALID_ARG);
    if (nq == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    GridDev g;
    // cell size for about 0.35 k points per cell, at least 4 (the 40-NN tables of the cluster filter at 1M points, wave search: 5 / 8 / 14
    // points per cell -> 45.5 / 45.0 / 44.9 ms per pair with `matching: cluster`; the lists do not depend on it)
    const float ppc = std::max(4.f, 0.35f * (float) k);
    LGR_TRY(lgr_grid_build(ctx, WS_GRID_A, d_pts, n, 0.f, ppc, &g));
    size_t sm = (size_t) k * KNN_BLOCK * 8;
    // below k = 16 the per-thread heaps win (k = 2: 0.16 against 0.96 ms at 1M): a wave per query leaves most lanes without a candidate
    const bool heap = k < 16;
    if (!heap) {
        if (d_q == d_pts && nq == n) {
            knn_wave_launch<0>(ctx, g, d_q, nq, k, d_idx, d_d2, 1, ppc);
            if (g.n < nq) knn_wave_launch<0>(ctx, g, d_q, nq, k, d_idx, d_d2, 2, ppc);
        } else {
            knn_wave_launch<0>(ctx, g, d_q, nq, k, d_idx, d_d2, 0, ppc);
        }
    } else if (d_q == d_pts && nq == n) {
        if (g.n > 0) knn_kernel<0><<<lgr_xcd_grid(cdiv(g.n, KNN_BLOCK)), KNN_BLOCK, sm, ctx->stream>>>(g, d_q, nq, k, d_idx, d_d2, 1);
        if (g.n < nq) knn_kernel<0><<<cdiv(nq, KNN_BLOCK), KNN_BLOCK, sm, ctx->stream>>>(g, d_q, nq, k, d_idx, d_d2, 2);
    } else {
        knn_kernel<0><<<cdiv(nq, KNN_BLOCK), KNN_BLOCK, sm, ctx->stream>>>(g, d_q, nq, k, d_idx, d_d2, 0);
    }
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

// src/common.cpp:531-547
extern "C" int lgr_smoothed_densities_dev(lgr_ctx* ctx, const float* d_pts, int n, int k, float* d_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_pts && d_out && n > 1 && k >= 2 && k <= 128, LGR_ERR_INVALID_ARG);   // rassert(pcd->size() > 1 && k >= 2)
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    GridDev g;
    LGR_TRY(lgr_grid_build(ctx, WS_GRID_A, d_pts, n, 0.f, 4.f, &g));
    float* dk; int32_t* nn1;
    LGR_TRY(lgr_ws_t(ctx, WS_DENS_A, (size_t) n, &dk));
    LGR_TRY(lgr_ws_t(ctx, WS_DENS_B, (size_t) n, &nn1));
    size_t sm = (size_t) k * KNN_BLOCK * 8;
    const bool heap = k < 16;
    if (!heap) {
        knn_wave_launch<1>(ctx, g, d_pts, n, k, nn1, dk, 1, 4.f);
        if (g.n < n) knn_wave_launch<1>(ctx, g, d_pts, n, k, nn1, dk, 2, 4.f);
    } else {
    if (g.n > 0) knn_kernel<1><<<lgr_xcd_grid(cdiv(g.n, KNN_BLOCK)), KNN_BLOCK, sm, ctx->stream>>>(g, d_pts, n, k, nn1, dk, 1);
    if (g.n < n) knn_kernel<1><<<cdiv(n, KNN_BLOCK), KNN_BLOCK, sm, ctx->stream>>>(g, d_pts, n, k, nn1, dk, 2);
    }
    density_min<<<cdiv(n, 256), 256, 0, ctx->stream>>>(dk, nn1, n, d_out);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

extern "C" int lgr_smoothed_densities(lgr_ctx* ctx, const float* pts, int n, int k, float* out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, pts && out && n > 1, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *dp, *dout;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) n * 12, &dp));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) n, &dout));
    LGR_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t) n * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_smoothed_densities_dev(ctx, dp, n, k, dout));
    LGR_HIP(ctx, hipMemcpyAsync(out, dout, (size_t) n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}
