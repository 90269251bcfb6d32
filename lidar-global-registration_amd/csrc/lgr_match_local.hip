// lgr_match_local.hip -- the two other matchers of match_multiscale's dispatch (include/matching.h:294-312) for gfx950:
//
//   matchFLANN<FPFH> (:565-592)  exact nearest neighbour in descriptor space = the row the brute-force kernel finds; only the
//                                reported distance follows FLANN's L2_Simple order (sequential sum of squares, then sqrt).
//   matchLocal<FPFH> (:637-678)  descriptor nearest neighbour restricted to the train points within match_search_radius of
//                                guess * query point.
//
// matchLocal is a guided refinement step (the caller already has a pose): with a finite radius a query sees a few hundred
// train points, so one thread per query walks the 27 cells of a uniform grid (cell = 1.001 * radius) and compares descriptors
// row by row -- HBM / L2 bound gathers of 132-byte rows, no matrix cores.  With the reference test's FLT_MAX radius
// (tests/flann_bf_matcher.h:71) every train point qualifies and the kernel degenerates into a plain tiled brute force.
#include <algorithm>
#include <cmath>

#include "lgr_grid.cuh"
#include "lgr_internal.h"

namespace {

__device__ __forceinline__ bool row_finite33(const float* __restrict__ r) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 33; ++j) ok = ok && fabsf(r[j]) <= 3.4028234663852886e38f;
    return ok;
}

// pcl::L2_Norm_SQR / FLANN L2_Simple: result += diff * diff, sequentially from dimension 0
__device__ __forceinline__ float l2sqr33_seq(const float* q, const float* __restrict__ t) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 33; ++j) { float d = q[j] - t[j]; s += d * d; }
    return s;
}

// distance of the match the brute-force kernel found, in FLANN's order (include/matching.h:586-588: std::sqrt of the squared
// distance nearestKSearch returns)
__global__ void flann_dist_kernel(const float* __restrict__ q33, const float* __restrict__ t33, const int32_t* __restrict__ idx, int mq, float* __restrict__ dist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mq) return;
    int j = idx[i];
    if (j < 0) { dist[i] = 0.f; return; }
    float q[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) q[k] = q33[(size_t) i * 33 + k];
    dist[i] = __builtin_sqrtf(l2sqr33_seq(q, t33 + (size_t) j * 33));
}

struct Best { float d, s; int j; };
__device__ __forceinline__ void offer(Best& b, float d, float s, int j) {
    if (b.j < 0 || d < b.d || (d == b.d && (s < b.s || (s == b.s && j < b.j)))) { b.d = d; b.s = s; b.j = j; }
}

constexpr int LB = 128;
// GRID: candidates = the 27 cells around the moved query point; otherwise every train point, staged 32 rows at a time in LDS
template <bool GRID>
__global__ __launch_bounds__(LB) void local_kernel(GridDev g, const float* __restrict__ qpts, int mq, const float* __restrict__ tpts, int mt,
                                                   const float* __restrict__ q33, const float* __restrict__ t33, const float* __restrict__ G /* 16, device */,
                                                   float r2, int32_t* __restrict__ idx, float* __restrict__ dist) {
    __shared__ float tile[32 * 36];
    const int i = blockIdx.x * LB + threadIdx.x;
    float q[33];
    bool valid = i < mq;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 33; ++k) q[k] = q33[(size_t) i * 33 + k];
        valid = row_finite33(q);
    }
    float x = 0.f, y = 0.f, z = 0.f;
    if (valid) {
        const float px = qpts[(size_t) i * 12], py = qpts[(size_t) i * 12 + 1], pz = qpts[(size_t) i * 12 + 2];
        // pcl::detail::Transformer::se3: x * c0 + (y * c1 + (z * c2 + c3))
        x = G[0] * px + (G[4] * py + (G[8] * pz + G[12]));
        y = G[1] * px + (G[5] * py + (G[9] * pz + G[13]));
        z = G[2] * px + (G[6] * py + (G[10] * pz + G[14]));
        valid = lgr_finite3(x, y, z);
    }
    Best b{0.f, 0.f, -1};
    if (GRID) {
        if (valid)
            lgr_visit27(g, x, y, z, [&](int, float4 Q) {
                const float s = lgr_dist2(x, y, z, Q.x, Q.y, Q.z);
                if (!(s < r2)) return;
                const int j = __float_as_int(Q.w);
                const float* t = t33 + (size_t) j * 33;
                if (!row_finite33(t)) return;
                offer(b, __builtin_sqrtf(l2sqr33_seq(q, t)), s, j);
            });
    } else {
        for (int j0 = 0; j0 < mt; j0 += 32) {
            const int nj = min(32, mt - j0);
            __syncthreads();
            for (int e = threadIdx.x; e < nj * 36; e += LB) {
                const int r = e / 36, c = e % 36;
                tile[e] = c < 33 ? t33[(size_t) (j0 + r) * 33 + c] : tpts[(size_t) (j0 + r) * 12 + (c - 33)];
            }
            __syncthreads();
            if (!valid) continue;
            for (int r = 0; r < nj; ++r) {
                const float* t = tile + r * 36;
                const float tx = t[33], ty = t[34], tz = t[35];
                if (!lgr_finite3(tx, ty, tz)) continue;
                const float s = lgr_dist2(x, y, z, tx, ty, tz);
                if (!(s < r2)) continue;
                if (!row_finite33(t)) continue;
                offer(b, __builtin_sqrtf(l2sqr33_seq(q, t)), s, j0 + r);
            }
        }
    }
    if (i < mq) { idx[i] = b.j; dist[i] = b.j >= 0 ? b.d : 0.f; }
}

}  // namespace

extern "C" int lgr_match_flann_dev(lgr_ctx* ctx, const float* d_q33, int mq, const float* d_t33, int mt, int32_t* d_idx, float* d_dist) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    // a single train block: the tie rule of the BF kernel is then "lowest index", the oracle's choice for FLANN's unspecified order
    LGR_TRY(lgr_match_bf_dev(ctx, d_q33, mq, d_t33, mt, std::max(mt, 1), d_idx, d_dist));
    if (mq > 0 && mt > 0) flann_dist_kernel<<<cdiv(mq, 128), 128, 0, ctx->stream>>>(d_q33, d_t33, d_idx, mq, d_dist);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

extern "C" int lgr_match_local_dev(lgr_ctx* ctx, const float* d_qpts, int mq, const float* d_tpts, int mt, const float* d_q33, const float* d_t33,
                                   const float guess16[16], float radius, int32_t* d_idx, float* d_dist) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_qpts || mq == 0) && (d_tpts || mt == 0) && (d_q33 || mq == 0) && (d_t33 || mt == 0) && (d_idx || mq == 0) && (d_dist || mq == 0) &&
                   guess16 && mq >= 0 && mt >= 0 && radius >= 0.f, LGR_ERR_INVALID_ARG);
    if (mq == 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    if (mt == 0) {
        LGR_HIP(ctx, hipMemsetAsync(d_idx, 0xff, (size_t) mq * 4, ctx->stream));
        LGR_HIP(ctx, hipMemsetAsync(d_dist, 0, (size_t) mq * 4, ctx->stream));
        return LGR_OK;
    }
    float* dG;
    LGR_TRY(lgr_ws_t(ctx, WS_LOCAL_G, 64, &dG));
    LGR_HIP(ctx, hipMemcpyAsync(dG, guess16, 64, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // guess16 is the caller's
    const float r2 = radius * radius;
    GridDev g{};
    const bool use_grid = std::isfinite(r2) && radius > 0.f;
    if (use_grid) {
        LGR_TRY(lgr_grid_build(ctx, WS_GRID_C, d_tpts, mt, radius * 1.001f, 0.f, &g));
        local_kernel<true><<<cdiv(mq, LB), LB, 0, ctx->stream>>>(g, d_qpts, mq, d_tpts, mt, d_q33, d_t33, dG, r2, d_idx, d_dist);
    } else {
        local_kernel<false><<<cdiv(mq, LB), LB, 0, ctx->stream>>>(g, d_qpts, mq, d_tpts, mt, d_q33, d_t33, dG, r2, d_idx, d_dist);
    }
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

extern "C" int lgr_match_flann(lgr_ctx* ctx, const float* q33, int mq, const float* t33, int mt, int32_t* idx, float* dist) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
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
    LGR_TRY(lgr_match_flann_dev(ctx, dq, mq, dt, mt, di, dd));
    if (mq) {
        LGR_HIP(ctx, hipMemcpyAsync(idx, di, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(dist, dd, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}

extern "C" int lgr_match_local(lgr_ctx* ctx, const float* qpts, int mq, const float* tpts, int mt, const float* q33, const float* t33,
                               const float guess16[16], float radius, int32_t* idx, float* dist) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (qpts || mq == 0) && (tpts || mt == 0) && (q33 || mq == 0) && (t33 || mt == 0) && (idx || mq == 0) && (dist || mq == 0) &&
                   guess16 && mq >= 0 && mt >= 0, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    float *dqp, *dtp, *dq, *dt, *dd;
    int32_t* di;
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_A, (size_t) mq * 33 + 1, &dq));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_B, (size_t) mt * 33 + 1, &dt));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_C, (size_t) mq + 1, &di));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_D, (size_t) mq + 1, &dd));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_E, (size_t) mq * 12 + 1, &dqp));
    LGR_TRY(lgr_ws_t(ctx, WS_HOST_F, (size_t) mt * 12 + 1, &dtp));
    if (mq) {
        LGR_HIP(ctx, hipMemcpyAsync(dq, q33, (size_t) mq * 132, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(dqp, qpts, (size_t) mq * 48, hipMemcpyHostToDevice, ctx->stream));
    }
    if (mt) {
        LGR_HIP(ctx, hipMemcpyAsync(dt, t33, (size_t) mt * 132, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(dtp, tpts, (size_t) mt * 48, hipMemcpyHostToDevice, ctx->stream));
    }
    LGR_TRY(lgr_match_local_dev(ctx, dqp, mq, dtp, mt, dq, dt, guess16, radius, di, dd));
    if (mq) {
        LGR_HIP(ctx, hipMemcpyAsync(idx, di, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(dist, dd, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}
