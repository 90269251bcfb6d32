// lgr_align.hip -- match filters (OneSided / LeftToRight / Cluster), correspondence search and alignPointClouds.
//
//   include/matching.h:395-411, 428-453, 492-550  -> lgr_filter_dev
//   src/correspondence_search.cpp:4-15 + include/matching.h:148-262 (keypoint 'any', single scale) -> lgr_correspondences*
//   src/alignment.cpp:72-109 (RANSAC branch; CSV side effects dropped)                          -> lgr_align*
#include <rocprim/device/device_scan.hpp>

#include <chrono>
#include <cmath>

#include "lgr_internal.h"

namespace {

// threshold = std::min(std::max(thr_s[i], thr_t[j]), distance_thr)
__device__ __forceinline__ float corr_threshold(float a, float b, float dthr) {
    float m = (a < b) ? b : a;          // std::max(a, b)
    return (dthr < m) ? dthr : m;       // std::min(m, dthr)
}

// calculateCorrespondenceDistance (include/matching.h:524-550) with randomness = 1
__device__ __forceinline__ float cluster_distance(int i, int j, int k, const int32_t* __restrict__ knn_a, const int32_t* __restrict__ knn_b,
                                                   const int32_t* __restrict__ ab_idx) {
    const int32_t* nb = knn_b + (size_t) j * k;
    int consistent = 0, pairs = 0;
    for (int a = 0; a < k; ++a) {
        int in = knn_a[(size_t) i * k + a];
        if (in < 0) continue;
        int mt = ab_idx[in];
        if (mt < 0) continue;
        bool hit = false;
        for (int b = 0; b < k; ++b) hit = hit || (nb[b] == mt);
        consistent += hit ? 1 : 0;
        pairs++;
    }
    if (pairs == 0) return 0.f;
    return 1.f - (float) consistent / (float) pairs;
}

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
            float di = cluster_distance(i, j, k, knn_s, knn_t, ij);
            float dj = cluster_distance(j, i, k, knn_t, knn_s, ji);
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

extern "C" int lgr_filter_dev(lgr_ctx* ctx, int matching_id, const float* d_src, int ns, const float* d_tgt, int nt,
                              const int32_t* d_ij_idx, const float* d_ij_dist, const int32_t* d_ji_idx, const float* d_ji_dist,
                              float distance_thr, int cluster_k, lgr_corr* d_out, int* n_out) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && d_ij_idx && d_ij_dist && d_out && n_out && ns > 1 && nt > 1, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, matching_id == LGR_MATCH_LR || matching_id == LGR_MATCH_ONE_SIDED || matching_id == LGR_MATCH_CLUSTER, LGR_ERR_INVALID_ARG);
    if (matching_id != LGR_MATCH_ONE_SIDED) LGR_CHECK(ctx, d_ji_idx && d_ji_dist, LGR_ERR_INVALID_ARG);
    if (matching_id == LGR_MATCH_CLUSTER) LGR_CHECK(ctx, cluster_k >= 1 && cluster_k <= 64, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    // thresholds: calculateSmoothedDensities(kps) of both clouds (include/matching.h:396-397 etc.)
    float *thr_s, *thr_t;
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_MISC, (size_t) ns + nt + 16, &thr_s));
    thr_t = thr_s + ns;
    LGR_TRY(lgr_smoothed_densities_dev(ctx, d_src, ns, 2, thr_s));
    LGR_TRY(lgr_smoothed_densities_dev(ctx, d_tgt, nt, 2, thr_t));
    int32_t *knn_s = nullptr, *knn_t = nullptr;
    if (matching_id == LGR_MATCH_CLUSTER) {
        float* d2;
        LGR_TRY(lgr_ws_t(ctx, WS_PIPE_KNN_S, (size_t) ns * cluster_k, &knn_s));
        LGR_TRY(lgr_ws_t(ctx, WS_PIPE_KNN_T, (size_t) nt * cluster_k, &knn_t));
        LGR_TRY(lgr_ws_t(ctx, WS_DENS_C, (size_t) std::max(ns, nt) * cluster_k, &d2));
        LGR_TRY(lgr_knn_dev(ctx, d_src, ns, d_src, ns, cluster_k, knn_s, d2));
        LGR_TRY(lgr_knn_dev(ctx, d_tgt, nt, d_tgt, nt, cluster_k, knn_t, d2));
    }
    int *flags, *pos;
    float* dist;
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_FLAGS, (size_t) ns * 3 + 16, &flags));
    pos = flags + ns; dist = (float*) (pos + ns);
    filter_flags<<<cdiv(ns, 128), 128, 0, ctx->stream>>>(matching_id, ns, d_ij_idx, d_ij_dist, d_ji_idx, d_ji_dist, knn_s, knn_t, cluster_k, flags, dist);
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
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && p && d_out && n_out && ns > 1 && nt > 1, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, p->feature_radius > 0.f, LGR_ERR_UNSUPPORTED);   // multi-scale matching: SURVEY 8f "next"
    LGR_CHECK(ctx, p->randomness == 1, LGR_ERR_UNSUPPORTED);        // data/test.yaml:14 "currently only 1 is supported"
    LGR_CHECK(ctx, p->feature_nr_points > 0 && p->normal_nr_points >= 1 && p->normal_nr_points <= 64 && p->bf_block_size > 0 && p->scale_factor > 1.f,
              LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    // include/matching.h:172,230-231: radius quantised to a power of scale_factor; voxel from feature_nr_points
    int log2_radius = (int) std::floor(std::log2(p->feature_radius) / std::log2(p->scale_factor));
    float search_radius = powf(p->scale_factor, (float) log2_radius);
    float voxel = sqrtf(M_PI * search_radius * search_radius / (float) p->feature_nr_points);
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
    for (int c = 0; c < 2; ++c) {
        int nd = 0;
        tick(ctx, 0);
        LGR_TRY(lgr_downsample_dev(ctx, clouds[c], sizes[c], voxel, surf[c], &nd));                       // :234
        tick(ctx, 1);
        const float* vp = c == 0 ? (p->has_vp_src ? p->vp_src : nullptr) : (p->has_vp_tgt ? p->vp_tgt : nullptr);
        LGR_TRY(lgr_normals_knn_dev(ctx, surf[c], nd, nullptr, 0, p->normal_nr_points, vp, p->normals_available));   // :235
        tick(ctx, 2);
        // :243-246 re-estimates the normals of the key-point COPY; FPFH reads only the surface normals
        // (include/common.h:329), so that step has no observable effect and is not executed.
        LGR_TRY(lgr_fpfh_dev(ctx, kclouds[c], ksizes[c], surf[c], nd, search_radius, feat[c]));            // :248
        tick(ctx, 3);
        LGR_HIP(ctx, hipEventSynchronize(ctx->ev[3]));
        float t;
        for (int s = 0; s < 3; ++s) { (void) hipEventElapsedTime(&t, ctx->ev[s], ctx->ev[s + 1]); ms[s] += t; }
    }
    int32_t *ij, *ji;
    float *dij, *dji;
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_IJ, (size_t) ns, &ij));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_JI, (size_t) nt, &ji));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_DIJ, (size_t) ns, &dij));
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_DJI, (size_t) nt, &dji));
    tick(ctx, 4);
    if (p->matching_id == LGR_MATCH_ONE_SIDED) LGR_TRY(lgr_match_bf_dev(ctx, feat[0], ns, feat[1], nt, p->bf_block_size, ij, dij));
    else LGR_TRY(lgr_match_bf2_dev(ctx, feat[0], ns, feat[1], nt, p->bf_block_size, ij, dij, ji, dji));
    tick(ctx, 5);
    LGR_TRY(lgr_filter_dev(ctx, p->matching_id, kclouds[0], ns, kclouds[1], nt, ij, dij, ji, dji, p->distance_thr, p->cluster_k, d_out, n_out));
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
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, src && tgt && p && out && n_out && ns > 1 && nt > 1, LGR_ERR_INVALID_ARG);
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
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_src && d_tgt && p && res && ns > 1 && nt > 1, LGR_ERR_INVALID_ARG);
    // alignTeaser throws in the reference (src/alignment.cpp:40)
    LGR_CHECK(ctx, p->alignment_id == LGR_ALIGN_RANSAC || p->alignment_id == LGR_ALIGN_GROR, LGR_ERR_UNSUPPORTED);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    auto t0 = std::chrono::steady_clock::now();
    lgr_corr* dc;
    LGR_TRY(lgr_ws_t(ctx, WS_PIPE_CORR, (size_t) ns + 1, &dc));
    int c = 0;
    LGR_TRY(lgr_correspondences_dev(ctx, d_src, ns, d_tgt, nt, p, dc, &c));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double time_cs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    tick(ctx, 7);
    if (p->alignment_id == LGR_ALIGN_GROR) {   // src/alignment.cpp:21-35: resolution = distance_thr, K_optimal = 800
        LGR_TRY(lgr_gror_dev(ctx, d_src, ns, d_tgt, nt, dc, c, p->distance_thr, 800, res, nullptr));
    } else {
        LGR_TRY(lgr_ransac_dev(ctx, d_src, ns, d_tgt, nt, dc, c, p, res, nullptr));
    }
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
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, src && tgt && p && res && ns > 1 && nt > 1, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *ds, *dt;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) ns * 12, &ds));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) nt * 12, &dt));
    LGR_HIP(ctx, hipMemcpyAsync(ds, src, (size_t) ns * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(dt, tgt, (size_t) nt * 48, hipMemcpyHostToDevice, ctx->stream));
    return lgr_align_dev(ctx, ds, ns, dt, nt, p, res);
}
