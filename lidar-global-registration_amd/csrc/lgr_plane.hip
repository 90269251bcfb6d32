// lgr_plane.hip -- closest-plane metric for gfx950.
//
// Replaces buildClosestPlaneInliers / ClosestPlaneMetricEstimator (reference src/metric.cpp:10-53, :181-199) as the
// RANSAC loop uses it: always on a sparse 1 % subset of the source cloud (sparse = true,
// src/sac_prerejective_omp.cpp:109).  For every subset point: transform, nearest target point within
// 2 x inlier_threshold (inlier_threshold = calculatePointCloudDensity(tgt)), distance to its tangent plane, inlier when
// below the threshold; metric = score / (0.01 |src|).
//
// Canonical choices shared with the oracle (oracle/src/orc_ransac.cpp, plane_eval): the reference draws the subset
// from the thread's mt19937 stream, here draw j of hypothesis `counter` is Philox4x32-10(counter, j / 4, 0x5A17, 0)[j % 4]
// >> 1, idx = r % n with the reference's linear probing over `visited` -- done with atomicOr on a per-workgroup bitmap,
// because the SET linear probing ends with does not depend on the insertion order; score and squared-error sums are
// 2^-32 fixed-point integers (order free, exact), so any number of lanes can accumulate them.
//
// One 256-thread workgroup per hypothesis (persistent grid): phase A claims the subset, phase B evaluates it
// (27-cell scan of a uniform grid over the target with cell = 1.001 x radius) and releases the bits.
#include <algorithm>

#include "lgr_grid.cuh"
#include "lgr_internal.h"
#include "lgr_math.cuh"

namespace {

__device__ __forceinline__ void philox4(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned out[4]) { lgr_philox4(seed, c0, c1, c2, c3, out); }

constexpr int PB = 256;

__global__ __launch_bounds__(PB) void plane_kernel(GridDev g, const float* __restrict__ src, int ns, int n_sp, float thr, float r2,
                                                   unsigned long long seed, const float* __restrict__ Ts, const int* __restrict__ list, int nh,
                                                   unsigned counter_base, int score_id, unsigned* __restrict__ visited_all,
                                                   int* __restrict__ claimed_all, int* __restrict__ cnt_out, float* __restrict__ metric_out,
                                                   float* __restrict__ rmse_out, int2* __restrict__ pairs_out, int* __restrict__ n_pairs,
                                                   float best_prev, int record_prev, const float* __restrict__ factor, lgr_plane_dyn dyn) {
    // (device-driven schedule: extents and gate values are the loop's state at the time this launch runs)
    if (dyn.nh) nh = min(nh, *dyn.nh);
    if (dyn.counter_base) counter_base = (unsigned) *dyn.counter_base;
    if (dyn.best_prev) best_prev = *dyn.best_prev;
    if (dyn.record_prev) record_prev = *dyn.record_prev;
    __shared__ long long s_sc[PB / 64], s_sq[PB / 64];
    __shared__ int s_cnt[PB / 64];
    __shared__ long long g_sc[2][PB / 64];
    __shared__ int g_cnt[2][PB / 64];
    const int words = (ns + 31) / 32;
    unsigned* visited = visited_all + (size_t) blockIdx.x * words;
    int* claimed = claimed_all + (size_t) blockIdx.x * n_sp;
    const int tid = threadIdx.x;
    for (int h = blockIdx.x; h < nh; h += gridDim.x) {
        const int off = list ? list[h] : h;
        const float* T = Ts + (size_t) off * 16;
        const unsigned counter = counter_base + (unsigned) off;
        // phase A: the subset (linear probing; the bitmap is all zero on entry)
        for (int j4 = tid; j4 * 4 < n_sp; j4 += PB) {   // one Philox block = four draws
            unsigned w[4];
            philox4(seed, counter, (unsigned) j4, 0x5A17u, 0u, w);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * j4 + q;
                if (j >= n_sp) break;
                int idx = (int) ((w[q] >> 1) % (unsigned) ns);
                for (;;) {
                    unsigned bit = 1u << (idx & 31);
                    unsigned old = atomicOr(&visited[idx >> 5], bit);
                    if (!(old & bit)) break;
                    idx = idx + 1 == ns ? 0 : idx + 1;
                }
                claimed[j] = idx;
            }
        }
        __syncthreads();
        // phase B: evaluate and release
        float Tr[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) Tr[k] = T[k];
        long long sc = 0, sq = 0;
        int cnt = 0;
        // Gate (RANSAC batches only: best_prev > 0 or record_prev < INT_MAX): every score is <= 1, so after j points the metric cannot
        // exceed (score so far + points left) / n_sp.  Once that bound (x the hypothesis' correspondence metric under `combination`)
        // is below the best metric of the EARLIER batches and the inlier count cannot reach their record either, the hypothesis can
        // neither become the best (strict >) nor a record: the rest of its subset is only released.  Checked every GATE_ITERS x 256
        // points on exact block totals.  WHICH points form a prefix claimed[0 .. j) depends on how the linear probing of phase A resolved
        // its collisions (thread timing), so whether and when a hypothesis is abandoned -- and the partial cnt / metric it then reports --
        // can differ between runs; the bound holds for ANY prefix of the subset, so the best hypothesis, the records and every value a
        // caller sees for a hypothesis that was NOT abandoned are run independent.  The partial values an abandoned
        // hypothesis writes stay below best_prev / record_prev (that is the gate's condition), so the batch reduction can pick neither.
        constexpr int GATE_ITERS = 4;
        const bool gated = !pairs_out && !rmse_out && (best_prev > 0.f || record_prev < 0x7fffffff);
        const float fac = (gated && factor) ? factor[h] : 1.f;
        int j_stop = n_sp, chunk = 0;
        for (int j0 = 0; j0 < n_sp; j0 += GATE_ITERS * PB, ++chunk) {
#pragma unroll 1
          for (int q = 0; q < GATE_ITERS; ++q) {
            const int j = j0 + q * PB + tid;
            if (j >= n_sp) break;
            const int idx = claimed[j];
            atomicAnd(&visited[idx >> 5], ~(1u << (idx & 31)));
            const float* s = src + (size_t) idx * 12;
            const float sx = s[0], sy = s[1], sz = s[2];
            // Eigen Matrix4f * Vector4f on SSE: ((c0 x + c1 y) + c2 z) + c3
            const float px = ((Tr[0] * sx + Tr[4] * sy) + Tr[8] * sz) + Tr[12];
            const float py = ((Tr[1] * sx + Tr[5] * sy) + Tr[9] * sz) + Tr[13];
            const float pz = ((Tr[2] * sx + Tr[6] * sy) + Tr[10] * sz) + Tr[14];
            if (!lgr_finite3(px, py, pz)) continue;
            int nn = -1, nn_t = -1;
            float best = 0.f;
            // the 27 cells around the moved point, (z, y) row by row (three x-cells are contiguous in memory); four candidate loads are
            // issued before the first compare -- one load per candidate, each waited for, was ~150 dependent round trips per point
            auto offer = [&](int t, const float4& Q) {
                float d2 = lgr_dist2(px, py, pz, Q.x, Q.y, Q.z);
                if (!(d2 < r2)) return;
                int qi = __float_as_int(Q.w);
                if (nn < 0 || d2 < best || (d2 == best && qi < nn)) { nn = qi; nn_t = t; best = d2; }
            };
            {
                const int cx = lgr_cellc(px, g.ox, g.h), cy = lgr_cellc(py, g.oy, g.h), cz = lgr_cellc(pz, g.oz, g.h);
                const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
                if (x0 <= x1) {
                    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dz - 1); ++z)
                        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dy - 1); ++y) {
                            const size_t cr = ((size_t) z * g.dy + y) * g.dx;
                            const int b = g.cell_start[cr + x0], e = g.cell_start[cr + x1 + 1];
                            for (int t = b; t < e; t += 4) {
                                const int last = e - 1;
                                const float4 q0 = g.pxyz[t], q1 = g.pxyz[min(t + 1, last)], q2 = g.pxyz[min(t + 2, last)], q3 = g.pxyz[min(t + 3, last)];
                                offer(t, q0);
                                if (t + 1 < e) offer(t + 1, q1);
                                if (t + 2 < e) offer(t + 2, q2);
                                if (t + 3 < e) offer(t + 3, q3);
                            }
                        }
                }
            }
            if (nn < 0) continue;
            const float4 Q = g.pxyz[nn_t], N = g.pnrm[nn_t];
            const float dist = fabsf((N.x * (Q.x - px) + N.y * (Q.y - py)) + N.z * (Q.z - pz));
            if (!(dist < thr)) continue;
            ++cnt;
            float value = 1.f;
            if (score_id == LGR_SCORE_MAE) value = fabsf(dist - thr) / thr;
            else if (score_id == LGR_SCORE_MSE) value = (dist - thr) * (dist - thr) / (thr * thr);
            else if (score_id == LGR_SCORE_EXP) value = lgr_expf(-dist * dist / (2 * thr * thr));
            sc += (long long) ((double) value * 4294967296.0);
            const float rel = dist / thr;
            sq += (long long) ((double) (rel * rel) * 4294967296.0);
            if (pairs_out) { int p = atomicAdd(n_pairs, 1); pairs_out[p] = make_int2(idx, nn); }
          }
          const int done_pts = min(n_sp, j0 + GATE_ITERS * PB);
          if (gated && done_pts < n_sp) {   // (uniform over the workgroup)
            long long wsc = sc;
            int wcnt = cnt;
            for (int o = 32; o > 0; o >>= 1) { wsc += __shfl_xor(wsc, o); wcnt += __shfl_xor(wcnt, o); }
            const int pp = chunk & 1;      // double buffered: one barrier per check
            if ((tid & 63) == 0) { g_sc[pp][tid >> 6] = wsc; g_cnt[pp][tid >> 6] = wcnt; }
            __syncthreads();
            long long tsc = 0;
            int tcnt = 0;
#pragma unroll
            for (int w = 0; w < PB / 64; ++w) { tsc += g_sc[pp][w]; tcnt += g_cnt[pp][w]; }
            const int left = n_sp - done_pts;
            const float m_max = (float) (((double) tsc / 4294967296.0 + (double) left) / (0.01 * (double) (float) ns)) * fac * 1.00001f;
            if (m_max < best_prev && tcnt + left < record_prev) { j_stop = done_pts; break; }
          }
        }
        for (int j = j_stop + tid; j < n_sp; j += PB) {   // gated out: release the rest of the subset
            const int idx = claimed[j];
            atomicAnd(&visited[idx >> 5], ~(1u << (idx & 31)));
        }
        for (int o = 32; o > 0; o >>= 1) { sc += __shfl_xor(sc, o); sq += __shfl_xor(sq, o); cnt += __shfl_xor(cnt, o); }
        __syncthreads();   // also: every bit of this hypothesis is released before the next one starts claiming
        if ((tid & 63) == 0) { s_sc[tid >> 6] = sc; s_sq[tid >> 6] = sq; s_cnt[tid >> 6] = cnt; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < PB / 64; ++w) { sc += s_sc[w]; sq += s_sq[w]; cnt += s_cnt[w]; }
            const float score = (float) ((double) sc / 4294967296.0);
            cnt_out[h] = cnt;
            metric_out[h] = (float) ((double) score / (0.01 * (double) (float) ns));   // score / (SPARSE_POINTS_FRACTION * src.size())
            if (rmse_out) rmse_out[h] = cnt ? thr * (float) sqrt((double) sq / 4294967296.0 / (double) cnt) : 3.4028234663852886e38f;
        }
        __syncthreads();
    }
}

}  // namespace

int lgr_plane_setup(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, uint64_t seed, lgr_plane_dev* out) {
    LGR_CHECK(ctx, ns > 0 && nt > 1, LGR_ERR_INVALID_ARG);
    float density = 0.f;
    LGR_TRY(lgr_cloud_density_dev(ctx, d_tgt, nt, 0.8f, &density));   // ClosestPlaneMetricEstimator::setTargetCloud
    out->thr = density;
    const float radius = 2 * out->thr;                                 // DIST_TO_PLANE_COEFFICIENT * inlier_threshold
    out->r2 = radius * radius;
    LGR_CHECK(ctx, radius > 0.f, LGR_ERR_INVALID_ARG);
    LGR_TRY(lgr_grid_build(ctx, WS_GRID_C, d_tgt, nt, radius * 1.001f, 0.f, &out->g));
    out->n_sp = (int) (0.01 * (float) ns);
    out->ns = ns; out->d_src = d_src; out->seed = seed;
    out->n_wg = std::max(1, 8 * ctx->n_cu);   // resident workgroups: the evaluation is a chain of dependent gathers, more waves in flight hide them
    const size_t words = (size_t) (ns + 31) / 32;
    LGR_TRY(lgr_ws_t(ctx, WS_PLANE_VISITED, (size_t) out->n_wg * words + 1, &out->visited));
    LGR_TRY(lgr_ws_t(ctx, WS_PLANE_CLAIMED, (size_t) out->n_wg * std::max(out->n_sp, 1) + 1, &out->claimed));
    LGR_HIP(ctx, hipMemsetAsync(out->visited, 0, (size_t) out->n_wg * words * 4, ctx->stream));
    return LGR_OK;
}

int lgr_plane_eval(lgr_ctx* ctx, const lgr_plane_dev& pd, const float* d_Ts, const int* d_list, int nh, unsigned counter_base, int score_id,
                   int* d_cnt, float* d_metric, float* d_rmse, int2* d_pairs, int* d_n_pairs, float best_prev, int record_prev, const float* d_factor,
                   const lgr_plane_dyn* dyn) {
    if (nh <= 0) return LGR_OK;
    if (d_pairs) LGR_HIP(ctx, hipMemsetAsync(d_n_pairs, 0, 4, ctx->stream));
    int grid = std::min(nh, pd.n_wg);
    const lgr_plane_dyn none{nullptr, nullptr, nullptr, nullptr};
    plane_kernel<<<grid, PB, 0, ctx->stream>>>(pd.g, pd.d_src, pd.ns, pd.n_sp, pd.thr, pd.r2, pd.seed, d_Ts, d_list, nh, counter_base, score_id,
                                               pd.visited, pd.claimed, d_cnt, d_metric, d_rmse, d_pairs, d_n_pairs, best_prev, record_prev, d_factor, dyn ? *dyn : none);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}
