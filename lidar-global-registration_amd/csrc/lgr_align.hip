// lgr_align.hip -- match filters (OneSided / LeftToRight / Cluster), correspondence search and alignPointClouds.
//
//   include/matching.h:395-411, 428-453, 492-550  -> lgr_filter_dev
//   src/correspondence_search.cpp:4-15 + include/matching.h:148-262 (keypoint 'any', single scale) -> lgr_correspondences*
//   src/alignment.cpp:72-109 (RANSAC branch; CSV side effects dropped)                          -> lgr_align*
#include <rocprim/device/device_scan.hpp>

#include <chrono>
#include <cmath>
#include <thread>

#include <climits>

#include "lgr_internal.h"

namespace {

// threshold = std::min(std::max(thr_s[i], thr_t[j]), distance_thr)
__device__ __forceinline__ float corr_threshold(float a, float b, float dthr) {
    float m = (a < b) ? b : a;          // std::max(a, b)
    return (dthr < m) ? dthr : m;       // std::min(m, dthr)
}

// calculateCorrespondenceDistance (include/matching.h:524-550) with randomness = 1: of the k spatial neighbours of i that
// have a match, the share whose match is NOT among the k spatial neighbours of j.  The neighbour list of j is read once
// into registers (KMAX ints), so the k x k membership test touches no memory (it was 1600 dependent global loads per
// correspondence and direction: 44 ms at 1M in the cluster mode).
template <int KMAX>
__device__ __forceinline__ float cluster_distance(int i, int j, int k, const int32_t* __restrict__ knn_a, const int32_t* __restrict__ knn_b,
                                                   const int32_t* __restrict__ ab_idx) {
    int nb[KMAX];
#pragma unroll
    for (int b = 0; b < KMAX; ++b) nb[b] = b < k ? knn_b[(size_t) j * k + b] : -2;   // -2 never equals a match index
    int consistent = 0, pairs = 0;
    // eight neighbours per round: their indices, then their matches, are requested together (index -> match is a dependent gather; one
    // neighbour per round was 2 k memory round trips in a row per correspondence and direction: 2.5 ms at 1M, k = 40)
    for (int a0 = 0; a0 < k; a0 += 8) {
        int in[8], mt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) in[u] = a0 + u < k ? knn_a[(size_t) i * k + a0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) mt[u] = in[u] >= 0 ? ab_idx[in[u]] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (mt[u] < 0) continue;
            bool hit = false;
#pragma unroll
            for (int b = 0; b < KMAX; ++b) hit = hit || (nb[b] == mt[u]);
            consistent += hit ? 1 : 0;
            pairs++;
        }
    }
    if (pairs == 0) return 0.f;
    return 1.f - (float) consistent / (float) pairs;
}

template <int KMAX>
__global__ void filter_flags(int matching_id, int ns, const int32_t* __restrict__ ij, const float* __restrict__ dij,
                             const int32_t* __restrict__ ji, const float* __restrict__ dji,
                             const int32_t* __restrict__ knn_s, const int32_t* __restrict__ knn_t, int k,
                             int* __restrict__ flags, float* __restrict__ dist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    int j = ij[i];
    int keep = 0;
    float d = 0.f;
    if (j >= 0) {
        if (matching_id == LGR_MATCH_ONE_SIDED) { keep = 1; d = dij[i]; }
        else if (matching_id == LGR_MATCH_LR) { keep = ji[j] == i ? 1 : 0; d = dji[j]; }   // reverse-direction distance (:444)
        else {
            float di = cluster_distance<KMAX>(i, j, k, knn_s, knn_t, ij);
            float dj = cluster_distance<KMAX>(j, i, k, knn_t, knn_s, ji);
            keep = (di < 0.95f && dj < 0.95f) ? 1 : 0;      // MATCHING_CLUSTER_THRESHOLD include/common.h:52
            d = (di < dj) ? dj : di;                         // std::max(distance_i, distance_j)
        }
    }
    flags[i] = keep; dist[i] = d;
}

__global__ void filter_emit(int ns, const int32_t* __restrict__ ij, const int* __restrict__ flags, const int* __restrict__ pos,
                            const float* __restrict__ dist, const float* __restrict__ thr_s, const float* __restrict__ thr_t,
                            float distance_thr, lgr_corr* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns || !flags[i]) return;
    int j = ij[i];
    lgr_corr c;
    c.index_query = i; c.index_match = j; c.distance = dist[i];
    c.threshold = corr_threshold(thr_s[i], thr_t[j], distance_thr);
    out[pos[i]] = c;
}

__global__ void invalidate_nan_rows(const float* __restrict__ feat, int m, int32_t* __restrict__ idx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    bool ok = true;
    for (int k = 0; k < 33; ++k) ok = ok && (fabsf(feat[(size_t) i * 33 + k]) <= 3.4028234663852886e38f);
    if (!ok) idx[i] = -1;
}

}  // namespace

// the filter's per-cloud tables: smoothed densities (thresholds) of both key-point clouds and, for the cluster filter, their
// k-NN lists.  They depend on the clouds only, not on the matches.
struct FilterTables { float *thr_s = nullptr, *thr_t = nullptr; int32_t *knn_s = nullptr, *knn_t = nullptr; };
static int filter_tables_alloc(lgr_ctx* ctx, int matching_id, int ns, int nt, int cluster_k, FilterTables* ft) {
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_MISC, (size_t) ns + nt + 16, &ft->thr_s));
    ft->thr_t = ft->thr_s + ns;
    if (matching_id == LGR_MATCH_CLUSTER) {
        LGR_TRY(lgr_ws_t(ctx, WS_PIPE_KNN_S, (size_t) ns * cluster_k, &ft->knn_s));
        LGR_TRY(lgr_ws_t(ctx, WS_PIPE_KNN_T, (size_t) nt * cluster_k, &ft->knn_t));
    }
    return LGR_OK;
}
static int filter_cloud_tables(lgr_ctx* cx, const float* pts, int n, int cluster_k, float* thr, int32_t* knn) {
    LGR_TRY(lgr_smoothed_densities_dev(cx, pts, n, 2, thr));   // calculateSmoothedDensities(kps) (include/matching.h:396-397 etc.)
    if (knn) {
        LGR_TRY(lgr_knn_lists(cx, pts, n, pts, n, cluster_k, knn, nullptr));   // (membership tests only: no distance table)
    }
    return LGR_OK;
}
static int filter_core(lgr_ctx* ctx, int matching_id, int ns, const int32_t* d_ij_idx, const float* d_ij_dist, const int32_t* d_ji_idx, const float* d_ji_dist,
                       float distance_thr, int cluster_k, const FilterTables& ft, lgr_corr* d_out, int* n_out);

extern "C" int lgr_filter_dev(lgr_ctx* ctx, int matching_id, const float* d_src, int ns, const float* d_tgt, int nt,
                              const int32_t* d_ij_idx, const float* d_ij_dist, const int32_t* d_ji_idx, const float* d_ji_dist,
                              float distance_thr, int cluster_k, lgr_corr* d_out, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && d_ij_idx && d_ij_dist && d_out && n_out && ns > 1 && nt > 1, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, matching_id == LGR_MATCH_LR || matching_id == LGR_MATCH_ONE_SIDED || matching_id == LGR_MATCH_CLUSTER, LGR_ERR_INVALID_ARG);
    if (matching_id != LGR_MATCH_ONE_SIDED) LGR_CHECK(ctx, d_ji_idx && d_ji_dist, LGR_ERR_INVALID_ARG);
    if (matching_id == LGR_MATCH_CLUSTER) LGR_CHECK(ctx, cluster_k >= 1 && cluster_k <= 64, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    FilterTables ft;
    LGR_TRY(filter_tables_alloc(ctx, matching_id, ns, nt, cluster_k, &ft));
    // the two clouds' tables side by side on the two contexts
    LGR_TRY(lgr_run_pair(ctx, [&](lgr_ctx* cx) { return filter_cloud_tables(cx, d_src, ns, cluster_k, ft.thr_s, ft.knn_s); },
                         [&](lgr_ctx* cx) { return filter_cloud_tables(cx, d_tgt, nt, cluster_k, ft.thr_t, ft.knn_t); }));
    return filter_core(ctx, matching_id, ns, d_ij_idx, d_ij_dist, d_ji_idx, d_ji_dist, distance_thr, cluster_k, ft, d_out, n_out);
}

static int filter_core(lgr_ctx* ctx, int matching_id, int ns, const int32_t* d_ij_idx, const float* d_ij_dist, const int32_t* d_ji_idx, const float* d_ji_dist,
                       float distance_thr, int cluster_k, const FilterTables& ft, lgr_corr* d_out, int* n_out) {
    float *thr_s = ft.thr_s, *thr_t = ft.thr_t;
    int32_t *knn_s = ft.knn_s, *knn_t = ft.knn_t;
    *n_out = 0;
    int *flags, *pos;
    float* dist;
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_FLAGS, (size_t) ns * 3 + 16, &flags));
    pos = flags + ns; dist = (float*) (pos + ns);
    if (matching_id != LGR_MATCH_CLUSTER) filter_flags<1><<<cdiv(ns, 128), 128, 0, ctx->stream>>>(matching_id, ns, d_ij_idx, d_ij_dist, d_ji_idx, d_ji_dist, knn_s, knn_t, cluster_k, flags, dist);
    else if (cluster_k <= 40) filter_flags<40><<<cdiv(ns, 128), 128, 0, ctx->stream>>>(matching_id, ns, d_ij_idx, d_ij_dist, d_ji_idx, d_ji_dist, knn_s, knn_t, cluster_k, flags, dist);
    else filter_flags<64><<<cdiv(ns, 128), 128, 0, ctx->stream>>>(matching_id, ns, d_ij_idx, d_ij_dist, d_ji_idx, d_ji_dist, knn_s, knn_t, cluster_k, flags, dist);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, flags, pos, 0, (size_t) ns, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, flags, pos, 0, (size_t) ns, rocprim::plus<int>(), ctx->stream));
    filter_emit<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(ns, d_ij_idx, flags, pos, dist, thr_s, thr_t, distance_thr, d_out);
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, pos + (ns - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, flags + (ns - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = h[0] + h[1];
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

static void tick(lgr_ctx* ctx, int i) { (void) hipEventRecord(ctx->ev[i], ctx->stream); }

// ---- multi-scale matching (feature_radius unset): include/matching.h:176-262 (initialize) and :264-352
// (match_multiscale).  The heavy stages (5-NN, down-sampling chain, normals, FPFH, brute-force matching per level) run on
// the device; the per-key-point level assignment (log2f/sqrtf of the reference's host arithmetic, level pruning) and
// the proximity vote over the <= nr_scales matches of a key point run on the host between them.
struct MsSide {
    int n_kps = 0;
    float iss_radius = 0.f;
    int min_l2 = INT_MAX, max_l2 = INT_MIN;
    std::vector<std::vector<int>> lists;   // per scale: key-point indices
    std::vector<size_t> feat_off;          // per scale: row offset into the feature buffer
    float* feat = nullptr;                 // device, sum(rows) x 33
    int32_t* d_lists = nullptr;            // device copy of the lists, concatenated like feat_off
    std::vector<float> xyz;                // host copy of the key points (3 floats each) for the vote
};

__global__ void gather_rows12_kernel(const float* __restrict__ pts, const int32_t* __restrict__ idx, int m, float* __restrict__ out) {
    size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) m * 12) return;
    out[e] = pts[(size_t) idx[e / 12] * 12 + e % 12];
}

static int ms_initialize(lgr_ctx* ctx, MsSide& st, int side, const float* d_pcd, int n, const float* d_kps, int n_kps, float iss_radius,
                         const lgr_params* p, const float* vp, float* ms) {
    st.n_kps = n_kps; st.iss_radius = iss_radius;
    const int k = 5;
    LGR_CHECK(ctx, n >= k, LGR_ERR_INVALID_ARG);
    // :180-188 density of every key point = distance to its 4th neighbour in the full cloud
    int32_t* d_nn;
    float* d_d2;
    LGR_TRY(lgr_ws_t(ctx, WS_MS_KNN_I, (size_t) n_kps * k, &d_nn));
    LGR_TRY(lgr_ws_t(ctx, WS_MS_KNN_D, (size_t) n_kps * k, &d_d2));
    LGR_TRY(lgr_knn_dev(ctx, d_kps, n_kps, d_pcd, n, k, d_nn, d_d2));
    std::vector<float> d2((size_t) n_kps * k);
    std::vector<float> kp((size_t) n_kps * 12);
    LGR_HIP(ctx, hipMemcpyAsync(d2.data(), d_d2, d2.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(kp.data(), d_kps, kp.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    st.xyz.resize((size_t) n_kps * 3);
    for (int i = 0; i < n_kps; ++i) for (int a = 0; a < 3; ++a) st.xyz[3 * (size_t) i + a] = kp[12 * (size_t) i + a];
    std::vector<int> level(n_kps);
    for (int i = 0; i < n_kps; ++i) {
        float density = sqrtf(d2[(size_t) i * k + (k - 1)]);
        float feature_radius = sqrtf((float) p->feature_nr_points * density * density / M_PI);
        level[i] = (int) std::floor(std::log2(feature_radius) / std::log2(p->scale_factor));
        st.min_l2 = std::min(level[i], st.min_l2);
        st.max_l2 = std::max(level[i], st.max_l2);
    }
    // :189-208 levels with too few points are dropped, key points clamped into the remaining range
    LGR_CHECK(ctx, (long long) st.max_l2 - st.min_l2 < 64, LGR_ERR_INVALID_ARG);   // non-finite densities (duplicate-only clouds)
    std::vector<int> count(st.max_l2 - st.min_l2 + 1, 0);
    for (int v : level) count[v - st.min_l2]++;
    const int max_nr = *std::max_element(count.begin(), count.end());
    size_t front = 0, back = count.size();
    while (10 * count[front] < max_nr) { ++front; st.min_l2++; }
    while (1000 * count[back - 1] < max_nr) { --back; st.max_l2--; }
    for (int& v : level) v = std::min(std::max(v, st.min_l2), st.max_l2);
    const int nr_scales = st.max_l2 - st.min_l2 + 1;
    st.lists.assign(nr_scales, {});
    for (int i = 0; i < n_kps; ++i)
        for (int j = level[i]; j <= st.max_l2; ++j) st.lists[j - st.min_l2].push_back(i);
    st.feat_off.assign(nr_scales + 1, 0);
    for (int i = 0; i < nr_scales; ++i) st.feat_off[i + 1] = st.feat_off[i] + st.lists[i].size();
    const size_t rows = st.feat_off[nr_scales];
    LGR_TRY(lgr_ws_t(ctx, side == 0 ? WS_MS_FEAT_S : WS_MS_FEAT_T, rows * 33 + 1, &st.feat));
    LGR_TRY(lgr_ws_t(ctx, side == 0 ? WS_MS_LIST_S : WS_MS_LIST_T, rows + 1, &st.d_lists));
    for (int i = 0; i < nr_scales; ++i)
        if (!st.lists[i].empty())
            LGR_HIP(ctx, hipMemcpyAsync(st.d_lists + st.feat_off[i], st.lists[i].data(), st.lists[i].size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // :228-261 per scale: down-sample the previous level's cloud, normals, FPFH of the level's key points
    float *bufA, *bufB, *sub;
    LGR_TRY(lgr_ws_t(ctx, side == 0 ? WS_PIPE_SURF_S : WS_PIPE_SURF_T, (size_t) n * 12, &bufA));
    LGR_TRY(lgr_ws_t(ctx, WS_MS_SURF2, (size_t) n * 12, &bufB));
    LGR_TRY(lgr_ws_t(ctx, WS_MS_SUB, (size_t) std::max(n_kps, 1) * 12, &sub));
    const float* in = d_pcd;
    int n_in = n;
    for (int i = 0; i < nr_scales; ++i) {
        float search_radius = powf(p->scale_factor, (float) (st.min_l2 + i));
        float voxel = sqrtf(M_PI * search_radius * search_radius / (float) p->feature_nr_points);
        float* out = (i & 1) ? bufB : bufA;
        int nd = 0;
        tick(ctx, 0);
        LGR_TRY(lgr_downsample_dev(ctx, in, n_in, voxel, out, &nd));
        tick(ctx, 1);
        LGR_TRY(lgr_normals_knn_dev(ctx, out, nd, nullptr, 0, p->normal_nr_points, vp, p->normals_available));
        tick(ctx, 2);
        const int m = (int) st.lists[i].size();
        if (m) {
            gather_rows12_kernel<<<cdiv((long long) m * 12, 256), 256, 0, ctx->stream>>>(d_kps, st.d_lists + st.feat_off[i], m, sub);
            LGR_TRY(lgr_fpfh_dev(ctx, sub, m, out, nd, search_radius, st.feat + st.feat_off[i] * 33));
        }
        tick(ctx, 3);
        LGR_HIP(ctx, hipEventSynchronize(ctx->ev[3]));
        float t;
        for (int s = 0; s < 3; ++s) { (void) hipEventElapsedTime(&t, ctx->ev[s], ctx->ev[s + 1]); ms[s] += t; }
        in = out; n_in = nd;
    }
    return LGR_OK;
}

// :316-349 one match per query: the candidate (one per level) with the largest proximity-weighted support among the
// candidates, ties by the smaller descriptor distance
static void ms_vote(const MsSide& tr, const std::vector<std::vector<int>>& mi, const std::vector<std::vector<float>>& md,
                    std::vector<int32_t>& out_idx, std::vector<float>& out_dist) {
    const int nq = (int) mi.size();
    for (int i = 0; i < nq; ++i) {
        const std::vector<int>& m = mi[i];
        float best_c = 0.f, best_d = 0.f;
        int best = -1;
        for (size_t m1 = 0; m1 < m.size(); ++m1) {
            float cnt = 0.f;
            for (size_t m2 = m1; m2 < m.size(); ++m2) {
                const float* a = tr.xyz.data() + 3 * (size_t) m[m1];
                const float* b = tr.xyz.data() + 3 * (size_t) m[m2];
                float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
                float dist_l2 = std::sqrt((dx * dx + dy * dy) + dz * dz);
                if (dist_l2 < 32 * tr.iss_radius) cnt += tr.iss_radius / std::max(dist_l2, tr.iss_radius);
            }
            if (cnt > best_c || (cnt == best_c && md[i][m1] < best_d)) { best_c = cnt; best_d = md[i][m1]; best = (int) m1; }
        }
        out_idx[i] = best >= 0 ? m[best] : -1;
        out_dist[i] = best >= 0 ? md[i][best] : 0.f;
    }
}

// inverse of a column-major 4x4 by Gauss-Jordan with partial pivoting in double, rounded to float: the canonical stand-in for
// Eigen's Matrix4f::inverse() (include/matching.h:296), same as the oracle's orc_inverse4
static void inverse4(const float* m16, float* out16) {
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m16[4 * c + r]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (piv != col) for (int c = 0; c < 8; ++c) std::swap(a[piv][c], a[col][c]);
        double d = a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            double f = a[r][col];
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out16[4 * c + r] = (float) a[r][4 + c];
}

// the matcher dispatch of match_multiscale (include/matching.h:294-312): guess -> matchLocal in both directions (the inverse guess
// for train -> query, :296), else bf -> matchBF (one MFMA pass serves both directions), else matchFLANN
static int match_dispatch(lgr_ctx* ctx, const lgr_params* p, const float* a_pts, const float* fa, int ma, const float* b_pts, const float* fb, int mb,
                          bool need_ba, int32_t* ab_i, float* ab_d, int32_t* ba_i, float* ba_d) {
    if (p->has_guess) {
        LGR_TRY(lgr_match_local_dev(ctx, a_pts, ma, b_pts, mb, fa, fb, p->guess, p->match_search_radius, ab_i, ab_d));
        if (need_ba) {
            float inv[16];
            inverse4(p->guess, inv);
            LGR_TRY(lgr_match_local_dev(ctx, b_pts, mb, a_pts, ma, fb, fa, inv, p->match_search_radius, ba_i, ba_d));
        }
    } else if (p->use_bfmatcher) {
        if (need_ba) LGR_TRY(lgr_match_bf2_dev(ctx, fa, ma, fb, mb, p->bf_block_size, ab_i, ab_d, ba_i, ba_d));
        else LGR_TRY(lgr_match_bf_dev(ctx, fa, ma, fb, mb, p->bf_block_size, ab_i, ab_d));
    } else {
        LGR_TRY(lgr_match_flann_dev(ctx, fa, ma, fb, mb, ab_i, ab_d));
        if (need_ba) LGR_TRY(lgr_match_flann_dev(ctx, fb, mb, fa, ma, ba_i, ba_d));
    }
    return LGR_OK;
}

static int ms_match_tables(lgr_ctx* ctx, const float* const* clouds, const int* sizes, const float* const* kclouds, const int* ksizes,
                           const lgr_params* p, int32_t* d_ij, float* d_dij, int32_t* d_ji, float* d_dji, float* ms) {
    MsSide st[2];
    for (int c = 0; c < 2; ++c) {
        const float* vp = c == 0 ? (p->has_vp_src ? p->vp_src : nullptr) : (p->has_vp_tgt ? p->vp_tgt : nullptr);
        LGR_TRY(ms_initialize(ctx, st[c], c, clouds[c], sizes[c], kclouds[c], ksizes[c], c == 0 ? p->iss_radius_src : p->iss_radius_tgt, p, vp, ms));
    }
    tick(ctx, 4);
    const bool need_ji = p->matching_id != LGR_MATCH_ONE_SIDED;
    const int ns = ksizes[0], nt = ksizes[1];
    std::vector<std::vector<int>> mi_ij(ns), mi_ji(need_ji ? nt : 0);
    std::vector<std::vector<float>> md_ij(ns), md_ji(need_ji ? nt : 0);
    const int lo = std::max(st[0].min_l2, st[1].min_l2), hi = std::min(st[0].max_l2, st[1].max_l2);
    for (int level = lo; level <= hi; ++level) {
        const int ia = level - st[0].min_l2, ib = level - st[1].min_l2;
        const std::vector<int>& la = st[0].lists[ia];
        const std::vector<int>& lb = st[1].lists[ib];
        const int ma = (int) la.size(), mb = (int) lb.size();
        if (ma == 0 || mb == 0) continue;
        int32_t* r;
        LGR_TRY(lgr_ws_t(ctx, WS_MS_RES, (size_t) 2 * (ma + mb) + 4, &r));
        int32_t *ab_i = r, *ba_i = r + ma;
        float *ab_d = (float*) (r + ma + mb), *ba_d = ab_d + ma;
        const float* fa = st[0].feat + st[0].feat_off[ia] * 33;
        const float* fb = st[1].feat + st[1].feat_off[ib] * 33;
        const float *pa = nullptr, *pb = nullptr;
        if (p->has_guess) {   // matchLocal works on the level's key-point sub-clouds (kps_multiscale, include/matching.h:243)
            float* sub;
            LGR_TRY(lgr_ws_t(ctx, WS_MS_SUB, (size_t) (ma + mb) * 12 + 4, &sub));
            gather_rows12_kernel<<<cdiv((long long) ma * 12, 256), 256, 0, ctx->stream>>>(kclouds[0], st[0].d_lists + st[0].feat_off[ia], ma, sub);
            gather_rows12_kernel<<<cdiv((long long) mb * 12, 256), 256, 0, ctx->stream>>>(kclouds[1], st[1].d_lists + st[1].feat_off[ib], mb, sub + (size_t) ma * 12);
            pa = sub; pb = sub + (size_t) ma * 12;
        }
        LGR_TRY(match_dispatch(ctx, p, pa, fa, ma, pb, fb, mb, need_ji, ab_i, ab_d, ba_i, ba_d));
        std::vector<int32_t> h((size_t) 2 * (ma + mb));
        LGR_HIP(ctx, hipMemcpyAsync(h.data(), r, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int32_t *h_ab = h.data(), *h_ba = h.data() + ma;
        const float *h_abd = (const float*) (h.data() + ma + mb), *h_bad = h_abd + ma;
        for (int i = 0; i < ma; ++i)
            if (h_ab[i] >= 0) { mi_ij[la[i]].push_back(lb[h_ab[i]]); md_ij[la[i]].push_back(h_abd[i]); }
        if (need_ji)
            for (int j = 0; j < mb; ++j)
                if (h_ba[j] >= 0) { mi_ji[lb[j]].push_back(la[h_ba[j]]); md_ji[lb[j]].push_back(h_bad[j]); }
    }
    std::vector<int32_t> ij(ns), ji(need_ji ? nt : 0);
    std::vector<float> dij(ns), dji(need_ji ? nt : 0);
    ms_vote(st[1], mi_ij, md_ij, ij, dij);
    if (need_ji) ms_vote(st[0], mi_ji, md_ji, ji, dji);
    LGR_HIP(ctx, hipMemcpyAsync(d_ij, ij.data(), (size_t) ns * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(d_dij, dij.data(), (size_t) ns * 4, hipMemcpyHostToDevice, ctx->stream));
    if (need_ji) {
        LGR_HIP(ctx, hipMemcpyAsync(d_ji, ji.data(), (size_t) nt * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(d_dji, dji.data(), (size_t) nt * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host staging vectors go out of scope
    return LGR_OK;
}

// key-point cloud = pcd[kps_indices] (pcl::copyPointCloud, include/matching.h:167); 12 floats per point
__global__ void gather_points_kernel(const float* __restrict__ pts, const int32_t* __restrict__ idx, int m, float* __restrict__ out) {
    size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) m * 12) return;
    out[e] = pts[(size_t) idx[e / 12] * 12 + e % 12];
}
// finalize (include/matching.h:150-160): local key-point indices -> indices of the clouds
__global__ void finalize_kernel(lgr_corr* __restrict__ corr, int n, const int32_t* __restrict__ ks, const int32_t* __restrict__ kt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    corr[i].index_query = ks[corr[i].index_query];
    corr[i].index_match = kt[corr[i].index_match];
}

extern "C" int lgr_correspondences_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_params* p,
                                       lgr_corr* d_out, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_src || ns == 0) && (d_tgt || nt == 0) && p && n_out && ns >= 0 && nt >= 0, LGR_ERR_INVALID_ARG);
    if (ns < 2 || nt < 2) { *n_out = 0; return LGR_OK; }   // nothing to match (the reference ends with an empty correspondence list)
    LGR_CHECK(ctx, d_out != nullptr, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, p->randomness == 1, LGR_ERR_UNSUPPORTED);        // data/test.yaml:14 "currently only 1 is supported"
    LGR_CHECK(ctx, p->feature_nr_points > 0 && p->normal_nr_points >= 1 && p->normal_nr_points <= 64 && p->bf_block_size > 0 && p->scale_factor > 1.f,
              LGR_ERR_INVALID_ARG);
    // checked before any stage runs (this entry point builds the filter tables itself, so lgr_filter_dev's checks do not cover it):
    // the cluster filter keeps at most 64 spatial neighbours per point (filter_flags<64>)
    LGR_CHECK(ctx, p->matching_id == LGR_MATCH_LR || p->matching_id == LGR_MATCH_ONE_SIDED || p->matching_id == LGR_MATCH_CLUSTER, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, p->matching_id != LGR_MATCH_CLUSTER || (p->cluster_k >= 1 && p->cluster_k <= 64), LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    // include/matching.h:172,230-231: radius quantised to a power of scale_factor; voxel from feature_nr_points
    // (feature_radius unset, i.e. <= 0 here: the multi-scale path below, include/matching.h:176-208)
    const bool multiscale = !(p->feature_radius > 0.f);
    float search_radius = 0.f, voxel = 0.f;
    if (!multiscale) {
        int log2_radius = (int) std::floor(std::log2(p->feature_radius) / std::log2(p->scale_factor));
        search_radius = powf(p->scale_factor, (float) log2_radius);
        voxel = sqrtf(M_PI * search_radius * search_radius / (float) p->feature_nr_points);
    }
    const float* clouds[2] = {d_src, d_tgt};
    int sizes[2] = {ns, nt};
    // key points (src/correspondence_search.cpp:8-11): every point, or the ISS detections.  kps = pcd[kps_indices]
    // (include/matching.h:167); every later stage works on the key-point clouds and the indices are mapped back at the
    // end (finalize, include/matching.h:150-160).
    const bool iss = p->keypoint_id == LGR_KEYPOINT_ISS;
    LGR_CHECK(ctx, p->keypoint_id == LGR_KEYPOINT_ANY || iss, LGR_ERR_UNSUPPORTED);
    const float* kclouds[2] = {d_src, d_tgt};
    int ksizes[2] = {ns, nt};
    int32_t* kidx[2] = {nullptr, nullptr};
    if (iss) {
        for (int c = 0; c < 2; ++c) {
            float* kp;
            LGR_TRY(lgr_ws_t(ctx, c == 0 ? WS_PIPE_KIDX_S : WS_PIPE_KIDX_T, (size_t) sizes[c] + 1, &kidx[c]));
            int m = 0;
            LGR_TRY(lgr_iss_keypoints_dev(ctx, clouds[c], sizes[c], c == 0 ? p->iss_radius_src : p->iss_radius_tgt, 0.975f, 0.975f, 4, kidx[c], &m));
            LGR_TRY(lgr_ws_t(ctx, c == 0 ? WS_PIPE_KPS_S : WS_PIPE_KPS_T, (size_t) std::max(m, 1) * 12, &kp));
            if (m) gather_points_kernel<<<cdiv((long long) m * 12, 256), 256, 0, ctx->stream>>>(clouds[c], kidx[c], m, kp);
            kclouds[c] = kp; ksizes[c] = m;
        }
        if (ksizes[0] == 0 || ksizes[1] == 0) return LGR_OK;
    }
    const int ns_all = ns, nt_all = nt;
    (void) ns_all; (void) nt_all;
    ns = ksizes[0]; nt = ksizes[1];
    float* feat[2];
    float* surf[2];
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_FEAT_S, (size_t) ns * 33, &feat[0]));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_FEAT_T, (size_t) nt * 33, &feat[1]));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_SURF_S, (size_t) sizes[0] * 12, &surf[0]));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_SURF_T, (size_t) sizes[1] * 12, &surf[1]));
    float ms[3] = {0, 0, 0};
    int32_t *ij, *ji;
    float *dij, *dji;
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_IJ, (size_t) ns, &ij));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_JI, (size_t) nt, &ji));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_DIJ, (size_t) ns, &dij));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_DJI, (size_t) nt, &dji));
    // The filter's per-cloud tables do not depend on the matches: a third context computes them from a host thread of its own
    // while the feature stages and the matcher run (their sorts and k-NN kernels fill the matcher's low-occupancy phases).
    FilterTables ftab;
    LGR_TRY(filter_tables_alloc(ctx, p->matching_id, ns, nt, p->cluster_k, &ftab));
    LGR_TRY(lgr_ctx_aux2(ctx));
    auto cloud_tables = [&](lgr_ctx* cx) -> int {
        LGR_TRY(filter_cloud_tables(cx, kclouds[0], ns, p->cluster_k, ftab.thr_s, ftab.knn_s));
        return filter_cloud_tables(cx, kclouds[1], nt, p->cluster_k, ftab.thr_t, ftab.knn_t);
    };
    lgr_helper_guard tables_guard{ctx->aux2, false};   // every exit path waits for the helper (which always drains its stream)
    if (ctx->opt.helper_contexts) {
        LGR_HIP(ctx, hipEventRecord(ctx->aux2_ev, ctx->stream));
        LGR_HIP(ctx, hipStreamWaitEvent(ctx->aux2->stream, ctx->aux2_ev, 0));
        LGR_TRY(lgr_helper_post(ctx, ctx->aux2, lgr_aux_job(ctx->aux2, cloud_tables)));
        tables_guard.armed = true;
    } else {
        const int rc = cloud_tables(ctx->aux2);        // same stream, this thread
        if (rc != LGR_OK) { ctx->err = ctx->aux2->err; return rc; }
    }
    if (multiscale) {
        LGR_TRY(ms_match_tables(ctx, clouds, sizes, kclouds, ksizes, p, ij, dij, ji, dji, ms));
    } else {
    // The two clouds' feature stages are independent until the matcher: the source cloud runs on this context, the target cloud
    // on a second context (own stream and workspace) driven by a second host thread, so that the ~20 host read-backs per cloud
    // (voxel counts, grid extents) and the short sort / scan launches of one cloud hide behind the other cloud's kernels.
    // Results cannot depend on it (disjoint outputs; every kernel is deterministic).
    auto cloud_features = [&](lgr_ctx* cx, int c, float* out_ms) -> int {
        LGR_HIP(cx, hipSetDevice(cx->device));
        int nd = 0;
        tick(cx, 0);
        LGR_TRY(lgr_downsample_dev(cx, clouds[c], sizes[c], voxel, surf[c], &nd));                       // :234
        tick(cx, 1);
        const float* vp = c == 0 ? (p->has_vp_src ? p->vp_src : nullptr) : (p->has_vp_tgt ? p->vp_tgt : nullptr);
        LGR_TRY(lgr_normals_knn_dev(cx, surf[c], nd, nullptr, 0, p->normal_nr_points, vp, p->normals_available));   // :235
        tick(cx, 2);
        // :243-246 re-estimates the normals of the key-point COPY; FPFH reads only the surface normals
        // (include/common.h:329), so that step has no observable effect and is not executed.
        LGR_TRY(lgr_fpfh_dev(cx, kclouds[c], ksizes[c], surf[c], nd, search_radius, feat[c]));            // :248
        tick(cx, 3);
        LGR_HIP(cx, hipEventSynchronize(cx->ev[3]));
        float t;
        for (int s = 0; s < 3; ++s) { (void) hipEventElapsedTime(&t, cx->ev[s], cx->ev[s + 1]); out_ms[s] += t; }
        return LGR_OK;
    };
    float ms_t[3] = {0, 0, 0};
    const auto t_feat0 = std::chrono::steady_clock::now();
    // The source cloud is done first (this thread): the matcher's query-side half (clustering, assignment, sort) runs right behind
    // its features, under the target cloud's feature kernels on the second context.
    const bool both_dirs = p->matching_id != LGR_MATCH_ONE_SIDED;
    const bool prepare_query = p->use_bfmatcher && !p->has_guess;   // match_dispatch: the brute-force matcher will be called
    struct PrepGuard { lgr_ctx* c; ~PrepGuard() { lgr_match_prepare_cancel(c); } } prep_guard{ctx};
    LGR_TRY(lgr_run_pair(ctx,
        [&](lgr_ctx* cx) {
            LGR_TRY(cloud_features(cx, 0, ms));
            return prepare_query ? lgr_match_prepare(cx, feat[0], ns, nt, both_dirs) : (int) LGR_OK;
        },
        [&](lgr_ctx* cx) { return cloud_features(cx, 1, ms_t); }));
    {
        // the two clouds overlap in wall time: report the wall time of the feature stages, split in proportion to the stage times
        // the two streams measured (each of which includes the other stream's interleaved kernels)
        const float wall = 1e3f * std::chrono::duration<float>(std::chrono::steady_clock::now() - t_feat0).count();
        float sum = 0.f;
        for (int s = 0; s < 3; ++s) { ms[s] += ms_t[s]; sum += ms[s]; }
        if (sum > 0.f) for (int s = 0; s < 3; ++s) ms[s] *= wall / sum;
    }
    tick(ctx, 4);
    LGR_TRY(match_dispatch(ctx, p, kclouds[0], feat[0], ns, kclouds[1], feat[1], nt, p->matching_id != LGR_MATCH_ONE_SIDED, ij, dij, ji, dji));
    }
    tick(ctx, 5);
    if (tables_guard.armed) {
        const int rc_tab = tables_guard.wait();
        (void) hipSetDevice(ctx->device);
        if (rc_tab != LGR_OK) { ctx->err = ctx->aux2->err; return rc_tab; }
    }
    LGR_TRY(filter_core(ctx, p->matching_id, ns, ij, dij, ji, dji, p->distance_thr, p->cluster_k, ftab, d_out, n_out));
    if (iss && *n_out) finalize_kernel<<<cdiv(*n_out, 256), 256, 0, ctx->stream>>>(d_out, *n_out, kidx[0], kidx[1]);
    tick(ctx, 6);
    LGR_HIP(ctx, hipEventSynchronize(ctx->ev[6]));
    float t;
    ctx->stage_ms[0] = ms[0]; ctx->stage_ms[1] = ms[1]; ctx->stage_ms[2] = ms[2];
    (void) hipEventElapsedTime(&t, ctx->ev[4], ctx->ev[5]); ctx->stage_ms[3] = t;
    (void) hipEventElapsedTime(&t, ctx->ev[5], ctx->ev[6]); ctx->stage_ms[4] = t;
    return LGR_OK;
}

extern "C" int lgr_correspondences(lgr_ctx* ctx, const float* src, int ns, const float* tgt, int nt, const lgr_params* p, lgr_corr* out, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (src || ns == 0) && (tgt || nt == 0) && p && n_out && ns >= 0 && nt >= 0, LGR_ERR_INVALID_ARG);
    if (ns < 2 || nt < 2) { *n_out = 0; return LGR_OK; }
    LGR_CHECK(ctx, out != nullptr, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *ds, *dt;
    lgr_corr* dc;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) ns * 12, &ds));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) nt * 12, &dt));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_CORR, (size_t) ns + 1, &dc));
    LGR_HIP(ctx, hipMemcpyAsync(ds, src, (size_t) ns * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(dt, tgt, (size_t) nt * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_correspondences_dev(ctx, ds, ns, dt, nt, p, dc, n_out));
    if (*n_out) LGR_HIP(ctx, hipMemcpyAsync(out, dc, (size_t) *n_out * 16, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}

extern "C" int lgr_align_dev(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_params* p, lgr_result* res) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_src || ns == 0) && (d_tgt || nt == 0) && p && res && ns >= 0 && nt >= 0, LGR_ERR_INVALID_ARG);
    // alignTeaser throws in the reference (src/alignment.cpp:40)
    LGR_CHECK(ctx, p->alignment_id == LGR_ALIGN_RANSAC || p->alignment_id == LGR_ALIGN_GROR, LGR_ERR_UNSUPPORTED);
    LGR_CHECK(ctx, p->n_samples >= 3 && p->n_samples <= 8, LGR_ERR_UNSUPPORTED);   // (lgr_ransac.hip instantiates its kernels for 3..8)
    if (ns < 2 || nt < 2) {
        // a cloud without two points gives no correspondences; the reference then leaves the identity, not converged
        // (selectCorrespondences refuses fewer than n_samples, src/sac_prerejective_omp.cpp:36-42)
        memset(res, 0, sizeof(*res));
        for (int i = 0; i < 16; ++i) res->transformation[i] = (i % 5 == 0) ? 1.f : 0.f;
        return LGR_OK;
    }
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    auto t0 = std::chrono::steady_clock::now();
    lgr_corr* dc;
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_CORR, (size_t) ns + 1, &dc));
    int c = 0;
    LGR_TRY(lgr_correspondences_dev(ctx, d_src, ns, d_tgt, nt, p, dc, &c));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double time_cs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    tick(ctx, 7);
    ctx->corr_trusted = true;                  // dc was produced by lgr_correspondences_dev: no range check (lgr_check_corr)
    int rc_te;
    if (p->alignment_id == LGR_ALIGN_GROR) {   // src/alignment.cpp:21-35: resolution = distance_thr, K_optimal = 800
        rc_te = lgr_gror_dev(ctx, d_src, ns, d_tgt, nt, dc, c, p->distance_thr, 800, res, nullptr);
    } else {
        rc_te = lgr_ransac_dev(ctx, d_src, ns, d_tgt, nt, dc, c, p, res, nullptr);
    }
    ctx->corr_trusted = false;
    LGR_TRY(rc_te);
    tick(ctx, 8);
    LGR_HIP(ctx, hipEventSynchronize(ctx->ev[8]));
    float t;
    (void) hipEventElapsedTime(&t, ctx->ev[7], ctx->ev[8]);
    ctx->stage_ms[5] = t;
    res->time_cs = time_cs;
    res->n_correspondences = c;
    for (int i = 0; i < 12; ++i) res->stage_ms[i] = ctx->stage_ms[i];
    return LGR_OK;
}

extern "C" int lgr_align(lgr_ctx* ctx, const float* src, int ns, const float* tgt, int nt, const lgr_params* p, lgr_result* res) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (src || ns == 0) && (tgt || nt == 0) && p && res && ns >= 0 && nt >= 0, LGR_ERR_INVALID_ARG);
    if (ns < 2 || nt < 2) return lgr_align_dev(ctx, nullptr, 0, nullptr, 0, p, res);   // identity, not converged
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *ds, *dt;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) ns * 12, &ds));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) nt * 12, &dt));
    LGR_HIP(ctx, hipMemcpyAsync(ds, src, (size_t) ns * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(dt, tgt, (size_t) nt * 48, hipMemcpyHostToDevice, ctx->stream));
    return lgr_align_dev(ctx, ds, ns, dt, nt, p, res);
}
