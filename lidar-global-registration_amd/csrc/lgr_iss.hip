// lgr_iss.hip -- ISS key points for gfx950.
//
// Replaces detectKeyPoints with keypoint_id = iss (reference src/common.cpp:657-691): ISSKeypoint3DDebug =
// pcl::ISSKeypoint3D with salient radius = non-maxima radius = iss_radius, gamma21 = gamma32 = 0.975,
// min_neighbors = 4, no border estimation; indices ascending.  Canonical choices (neighbour order, Jacobi eigenvalues
// in double) as stated in oracle/src/orc_iss.cpp; results are bit-identical to that oracle.
//
// Kernels (uniform grid with cell = 1.001 r, points sorted by (cell, index)):
//   iss_saliency : one lane per point, the 27 neighbouring cells scanned in (z, y, x, index) order, scatter matrix and
//                  eigenvalues in double -> third eigenvalue (0 when rejected) and neighbour count, by original index
//   iss_nms      : one lane per candidate, same scan, "no neighbour has a larger third eigenvalue" -> flags
//   scan + emit  : ascending key-point indices
#include <rocprim/device/device_scan.hpp>

#include "lgr_grid.cuh"
#include "lgr_internal.h"
#include "lgr_math.cuh"

namespace {

__global__ void iss_saliency_kernel(GridDev g, float r2, double gamma21, double gamma32, int min_nb, double* __restrict__ third,
                                    int* __restrict__ nnb) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.n) return;
    float4 P = g.pxyz[t];
    int i = __float_as_int(P.w);
    double S[6] = {0, 0, 0, 0, 0, 0};
    int k = 0;
    lgr_visit27(g, P.x, P.y, P.z, [&](int, float4 Q) {
        if (!(lgr_dist2(P.x, P.y, P.z, Q.x, Q.y, Q.z) < r2)) return;
        ++k;
        double dx = (double) Q.x - (double) P.x, dy = (double) Q.y - (double) P.y, dz = (double) Q.z - (double) P.z;
        S[0] += dx * dx; S[1] += dx * dy; S[2] += dx * dz; S[3] += dy * dy; S[4] += dy * dz; S[5] += dz * dz;
    });
    nnb[i] = k;
    double out = 0.0;
    if (k >= min_nb) {
        double ev[3];
        lgr_eigvals3d(S, ev);
        double e1 = ev[2], e2 = ev[1], e3 = ev[0];
        bool fin = isfinite(e1) && isfinite(e2) && isfinite(e3);
        if (fin && !(e3 < 0) && e2 / e1 < gamma21 && e3 / e2 < gamma32) out = e3;
    }
    third[i] = out;
}

__global__ void iss_nms_kernel(GridDev g, float r2, int min_nb, const double* __restrict__ third, const int* __restrict__ nnb,
                               int* __restrict__ flags) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.n) return;
    float4 P = g.pxyz[t];
    int i = __float_as_int(P.w);
    double mine = third[i];
    int f = 0;
    if (mine > 0.0 && nnb[i] >= min_nb) {
        bool is_max = true;
        lgr_visit27(g, P.x, P.y, P.z, [&](int, float4 Q) {
            if (!(lgr_dist2(P.x, P.y, P.z, Q.x, Q.y, Q.z) < r2)) return;
            if (mine < third[__float_as_int(Q.w)]) is_max = false;
        });
        f = is_max ? 1 : 0;
    }
    flags[i] = f;
}

__global__ void iss_emit_kernel(const int* __restrict__ flags, const int* __restrict__ pos, int n, int32_t* __restrict__ idx, int* __restrict__ n_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) idx[pos[i]] = i;
    if (i == n - 1) *n_out = pos[i] + flags[i];
}

}  // namespace

extern "C" int lgr_iss_keypoints_dev(lgr_ctx* ctx, const float* d_pts, int n, float radius, float gamma21, float gamma32, int min_neighbors,
                                     int32_t* d_idx, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_pts || n == 0) && (d_idx || n == 0) && n_out && n >= 0, LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, radius > 0.f && gamma21 > 0.f && gamma32 > 0.f && min_neighbors > 0, LGR_ERR_INVALID_ARG);   // iss_debug.cpp:98-121
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    if (n == 0) return LGR_OK;
    GridDev g;
    LGR_TRY(lgr_grid_build(ctx, WS_GRID_C, d_pts, n, radius * 1.001f, 0.f, &g));
    double* third;
    LGR_TRY(lgr_ws_t(ctx, WS_DENS_A, (size_t) n + 1, &third));
    int* ib;
    LGR_TRY(lgr_ws_t(ctx, WS_DENS_B, (size_t) 3 * n + 16, &ib));
    int *nnb = ib, *flags = ib + n, *pos = ib + 2 * (size_t) n, *d_n = ib + 3 * (size_t) n;
    // points with non-finite coordinates are not in the grid: no neighbours, never a key point
    LGR_HIP(ctx, hipMemsetAsync(third, 0, (size_t) n * 8, ctx->stream));
    LGR_HIP(ctx, hipMemsetAsync(ib, 0, ((size_t) 2 * n) * 4, ctx->stream));
    const float r2 = radius * radius;
    if (g.n > 0) {
        iss_saliency_kernel<<<cdiv(g.n, 128), 128, 0, ctx->stream>>>(g, r2, (double) gamma21, (double) gamma32, min_neighbors, third, nnb);
        iss_nms_kernel<<<cdiv(g.n, 128), 128, 0, ctx->stream>>>(g, r2, min_neighbors, third, nnb, flags);
    }
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, flags, pos, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, flags, pos, 0, (size_t) n, rocprim::plus<int>(), ctx->stream));
    iss_emit_kernel<<<cdiv(n, 256), 256, 0, ctx->stream>>>(flags, pos, n, d_idx, d_n);
    LGR_HIP(ctx, hipGetLastError());
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = *h;
    return LGR_OK;
}

extern "C" int lgr_iss_keypoints(lgr_ctx* ctx, const float* pts, int n, float radius, float gamma21, float gamma32, int min_neighbors,
                                 int32_t* idx, int* n_out) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (pts || n == 0) && (idx || n == 0) && n_out && n >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float* dp;
    int32_t* di;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) n * 12 + 1, &dp));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) n + 1, &di));
    if (n) LGR_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t) n * 48, hipMemcpyHostToDevice, ctx->stream));
    LGR_TRY(lgr_iss_keypoints_dev(ctx, dp, n, radius, gamma21, gamma32, min_neighbors, di, n_out));
    if (*n_out) LGR_HIP(ctx, hipMemcpyAsync(idx, di, (size_t) *n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}
