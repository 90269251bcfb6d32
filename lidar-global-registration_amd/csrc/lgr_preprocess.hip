// lgr_preprocess.hip -- loader preprocessing for gfx950: the steps of loadPointClouds
// (reference src/common.cpp:429-470) between the PLY reader and the hot path's boundary:
//   filterDuplicatePoints (:417-427)  exact (x, y, z) duplicates removed, first occurrence kept
//   intensity = 1 (:446-451)          the weight accumulated by the voxel grid
//   voxel = 2 * calculatePointCloudDensity (:453-456, :202-208: 0.8 quantile of the smoothed 8-NN densities)
//   downsamplePointCloud in place, estimateNormalsPoints(30)
// Device: three stable radix sorts (z, y, x bit patterns; -0 folded into +0, points with a NaN coordinate never equal)
// give the duplicate runs with their lowest index first; flags -> scan -> compaction in input order.  The quantile is
// element k of the radix-sorted densities.  The reference's std::unordered_set / unordered_map output ORDER is
// reproduced on request by the host entry point (replay through the same containers), like lgr_downsample does.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <functional>
#include <unordered_set>

#include "lgr_internal.h"

namespace {

__device__ __forceinline__ unsigned eq_bits(float v) {   // bit pattern with -0 folded into +0 (PointEqual uses ==)
    unsigned b = __float_as_uint(v);
    return b == 0x80000000u ? 0u : b;
}
__global__ void dd_keys_kernel(const float* __restrict__ pts, int n, int axis, const int* __restrict__ order, unsigned* __restrict__ keys,
                               int* __restrict__ vals) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int i = order ? order[t] : t;
    keys[t] = eq_bits(pts[(size_t) i * 12 + axis]);
    vals[t] = i;
}
// sorted by (x, y, z) bits, ties in index order: a point is dropped iff it equals its predecessor and has no NaN
__global__ void dd_flags_kernel(const float* __restrict__ pts, const int* __restrict__ order, int n, int* __restrict__ keep) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int i = order[t];
    const float* p = pts + (size_t) i * 12;
    bool dup = false;
    if (t > 0) {
        const float* q = pts + (size_t) order[t - 1] * 12;
        dup = p[0] == q[0] && p[1] == q[1] && p[2] == q[2];   // false as soon as a coordinate is NaN
    }
    keep[i] = dup ? 0 : 1;
}
__global__ void dd_compact_kernel(const float* __restrict__ pts, const int* __restrict__ keep, const int* __restrict__ pos, int n,
                                  float* __restrict__ out, int* __restrict__ n_out) {
    size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) n * 12) return;
    int i = (int) (e / 12), f = (int) (e % 12);
    if (keep[i]) out[(size_t) pos[i] * 12 + f] = f == 8 ? 1.0f : pts[e];   // field 8 = intensity := 1
    if (e == (size_t) n * 12 - 1) *n_out = pos[i] + keep[i];
}

int dedupe_dev(lgr_ctx* ctx, const float* d_pts, int n, float* d_out, int* n_out) {
    unsigned *k0, *k1;
    int *v0, *v1, *keep, *pos;
    LGR_TRY(lgr_ws_t(ctx, WS_DS_KEYS, (size_t) n + 1, &k0));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_KEYS2, (size_t) n + 1, &k1));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_VALS, (size_t) n + 1, &v0));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_VALS2, (size_t) n + 1, &v1));
    LGR_TRY(lgr_ws_t(ctx, WS_DS_FLAGS, (size_t) 2 * n + 8, &keep));
    pos = keep + n;
    int* d_n = pos + n;
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, (size_t) n, 0, 32, ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    const int* order = nullptr;
    for (int axis = 2; axis >= 0; --axis) {   // least significant key first; every pass is stable
        dd_keys_kernel<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_pts, n, axis, order, k0, v0);
        LGR_HIP(ctx, rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, (size_t) n, 0, 32, ctx->stream));
        // the sorted indices are the input order of the next pass: keep them apart from the buffers that pass writes
        LGR_HIP(ctx, hipMemcpyAsync(pos, v1, (size_t) n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        order = pos;
    }
    dd_flags_kernel<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_pts, order, n, keep);
    size_t sb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, sb, keep, v0, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    void* stmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, std::max(sb, tb), &stmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(stmp, sb, keep, v0, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    dd_compact_kernel<<<cdiv((long long) n * 12, 256), 256, 0, ctx->stream>>>(d_pts, keep, v0, n, d_out, d_n);
    LGR_HIP(ctx, hipGetLastError());
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = *h;
    return LGR_OK;
}

// src/common.cpp:202-208 calculatePointCloudDensity(pcd, quantile): element k of the sorted smoothed densities
int cloud_density_dev(lgr_ctx* ctx, const float* d_pts, int n, float quantile, float* out) {
    float *dens, *sorted;
    LGR_TRY(lgr_ws_t(ctx, WS_DENS_C, (size_t) 2 * n + 2, &dens));
    sorted = dens + n;
    LGR_TRY(lgr_smoothed_densities_dev(ctx, d_pts, n, 8, dens));
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::radix_sort_keys(nullptr, tb, dens, sorted, (size_t) n, 0, 32, ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::radix_sort_keys(tmp, tb, dens, sorted, (size_t) n, 0, 32, ctx->stream));
    int k = std::max(std::min((int) (quantile * (float) n - 1), n - 1), 0);
    float* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, sorted + k, 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = *h;
    return LGR_OK;
}

struct HPt { float v[12]; };
struct HPtHash {   // include/common.h:202-210 + include/utils.h:28-32
    size_t operator()(const HPt& p) const {
        size_t seed = 0;
        for (int a = 0; a < 3; ++a) seed ^= std::hash<float>()(p.v[a]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};
struct HPtEq {
    bool operator()(const HPt& a, const HPt& b) const { return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2]; }
};

}  // namespace

extern "C" int lgr_cloud_density_dev(lgr_ctx* ctx, const float* d_pts, int n, float quantile, float* out_host) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_pts && out_host && n > 1 && quantile >= 0.f && quantile <= 1.f, LGR_ERR_INVALID_ARG);   // rasserts :203, :532
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    return cloud_density_dev(ctx, d_pts, n, quantile, out_host);
}

extern "C" int lgr_dedupe_dev(lgr_ctx* ctx, const float* d_pts, int n, float* d_out, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_pts || n == 0) && (d_out || n == 0) && n_out && n >= 0 && d_pts != d_out, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    if (n == 0) return LGR_OK;
    return dedupe_dev(ctx, d_pts, n, d_out, n_out);
}

extern "C" int lgr_preprocess_dev(lgr_ctx* ctx, const float* d_pts, int n, const float* vp3, int normals_available, float* d_out, int* n_out,
                                  float* voxel_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, d_pts && d_out && n_out && n > 1 && d_pts != d_out, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float* u;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_E, (size_t) n * 12, &u));
    int m = 0;
    LGR_TRY(dedupe_dev(ctx, d_pts, n, u, &m));
    LGR_CHECK(ctx, m > 1, LGR_ERR_INVALID_ARG);
    float density = 0.f;
    LGR_TRY(cloud_density_dev(ctx, u, m, 0.8f, &density));
    const float voxel = 2 * density;   // FINE_VOXEL_SIZE_COEFFICIENT, include/common.h:59
    LGR_TRY(lgr_downsample_dev(ctx, u, m, voxel, d_out, n_out));
    LGR_TRY(lgr_normals_knn_dev(ctx, d_out, *n_out, nullptr, 0, 30, vp3, normals_available));
    if (voxel_out) *voxel_out = voxel;
    return LGR_OK;
}

extern "C" int lgr_preprocess(lgr_ctx* ctx, const float* pts, int n, const float* vp3, int normals_available, int order, float* out, int* n_out,
                              float* voxel_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, pts && out && n_out && n > 1, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, order == LGR_ORDER_REFERENCE || order == LGR_ORDER_CANONICAL, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *dp, *dout;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) n * 12, &dp));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_D, (size_t) n * 12, &dout));
    if (order == LGR_ORDER_CANONICAL) {
        LGR_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t) n * 48, hipMemcpyHostToDevice, ctx->stream));
        LGR_TRY(lgr_preprocess_dev(ctx, dp, n, vp3, normals_available, dout, n_out, voxel_out));
        LGR_HIP(ctx, hipMemcpyAsync(out, dout, (size_t) *n_out * 48, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return LGR_OK;
    }
    // reference order: the set's iteration order after inserting the points in input order (filterDuplicatePoints), then
    // the unordered_map order of the voxel grid (lgr_downsample with LGR_ORDER_REFERENCE)
    std::vector<float> u;
    {
        std::unordered_set<HPt, HPtHash, HPtEq> set;
        set.reserve(n);
        for (int i = 0; i < n; ++i) {
            HPt p;
            memcpy(p.v, pts + 12 * (size_t) i, 48);
            set.insert(p);
        }
        u.resize(set.size() * 12);
        size_t o = 0;
        for (const HPt& p : set) { memcpy(u.data() + 12 * o, p.v, 48); u[12 * o + 8] = 1.f; ++o; }
    }
    const int m = (int) (u.size() / 12);
    LGR_CHECK(ctx, m > 1, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipMemcpyAsync(dp, u.data(), (size_t) m * 48, hipMemcpyHostToDevice, ctx->stream));
    float density = 0.f;
    LGR_TRY(cloud_density_dev(ctx, dp, m, 0.8f, &density));
    const float voxel = 2 * density;
    std::vector<float> ds((size_t) m * 12);
    int nd = 0;
    LGR_TRY(lgr_downsample(ctx, u.data(), m, voxel, LGR_ORDER_REFERENCE, ds.data(), &nd));
    LGR_HIP(ctx, hipMemcpyAsync(dout, ds.data(), (size_t) nd * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_normals_knn_dev(ctx, dout, nd, nullptr, 0, 30, vp3, normals_available));
    LGR_HIP(ctx, hipMemcpyAsync(out, dout, (size_t) nd * 48, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = nd;
    if (voxel_out) *voxel_out = voxel;
    return LGR_OK;
}
