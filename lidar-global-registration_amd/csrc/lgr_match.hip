// lgr_match.hip -- brute-force FPFH matching (both directions from one MFMA pass) for gfx950.
//
// Replaces include/matching.h:594-634 matchBF<FPFH> (cv::BFMatcher(NORM_L2)::knnMatch, k = 1) and the cross-block
// merge src/common.cpp:517-529.  Result contract (bit-exact with the oracle): for every valid query row, the train
// row minimising the CANONICAL distance d = sqrtf(normL2Sqr) -- OpenCV 4.5.1 SSE lane order, see exact_l2() -- with
// ties broken "highest bf block, then lowest index inside the block"; NaN rows never match.
//
// Structure (DESIGN.md "matcher"):
//   1. cluster      : 16 k-means centres of the descriptors (Lloyd on a sample, on the device); every row is assigned
//                     to its nearest centre and both sets are sorted by (cluster, leaf, distance to the cluster centre) -- the
//                     last key makes every 32 / 128 / 256-row piece of a leaf a thin radial shell about that centre, which
//                     the skipping passes use as a second lower bound (| |a - c| - |b - c| | <= |a - b|: mask_kernel, match_mfma).
//   2. pack         : MFMA operands, K = 34: A' = [-2(a - c_p), 1] for a in cluster p, and one column set per cluster,
//                     B'(p) = [b - c_p, |b - c_p|^2].  Distances are translation invariant, so for a row of cluster p
//                     S = A'.B'(p) = |b - c_p|^2 - 2 (a - c_p).(b - c_p) = d2 - |a - c_p|^2 -- and its rounding error
//                     scales with (|a - c_p| + |b - c_p|)^2, i.e. it is tiny exactly for the near pairs that matter
//                     (FPFH data is full of near-duplicate "flat surface" rows far from the global mean).
//   3. match_mfma   : the brute-force contraction on v_mfma_f32_32x32x2_f32 with a fused epilogue that keeps only
//                     min_b d2~ per (row, column group) and min_a d2~ per (column, row group), d2~ = S + |a'|^2.
//                     FILTER only.  (f16 formats, the default: two-term f16 splits of the operands on v_mfma_f32_32x32x16_f16,
//                     K = 96 or 112, under the same kind of proven bound; bound-based tile skipping in masked passes: section 3b.)
//   4. rerank_*     : per query, a group is a candidate when its lower bound (value - proven error) does not exceed
//                     the smallest upper bound; candidate groups are rescanned with the exact canonical distance and
//                     a packed 64-bit atomicMin applies the reference's tie rules (order independent).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include "lgr_internal.h"

#include "lgr_match_common.cuh"
#include "lgr_match_cluster.cuh"
#include "lgr_match_pack.cuh"
#include "lgr_match_mfma.cuh"
#include "lgr_match_sweep.cuh"
#include "lgr_match_bounds.cuh"
#include "lgr_match_rerank.cuh"

// diagnostics of a context's last match call (lgr_match_stats / mcheck live in the context: two contexts driven from one host thread
// keep separate figures, and the helper thread of the second rerank direction writes into the same object as the caller)
extern "C" int lgr_match_last_stats(lgr_ctx* ctx, unsigned* out6) {
    if (!ctx || !out6) return LGR_ERR_INVALID_ARG;
    const lgr_match_stats& s = ctx->mstats;
    out6[0] = s.items_ab; out6[1] = s.dense_ab; out6[2] = s.items_ba;
    out6[3] = s.dense_ba; out6[4] = (unsigned) s.sub_cols; out6[5] = (unsigned) s.rg_rows;
    return LGR_OK;
}
// fraction of the (row block x column stage) tiles of the last match call that the MFMA passes computed (1 = dense): every (row block,
// stage) counted ONCE, however many passes touched it (a stage that straddles two leaves is computed whole by each pass that schedules
// one of them) -- never above 1
extern "C" int lgr_match_last_work(lgr_ctx* ctx, double* executed_fraction) {
    if (!ctx || !executed_fraction) return LGR_ERR_INVALID_ARG;
    *executed_fraction = ctx->mstats.stages_all > 0 ? ctx->mstats.stages_unique / ctx->mstats.stages_all : 1.0;
    return LGR_OK;
}
// the same with every pass's stages summed: the work that was ISSUED (what the bench prices MFMA FLOP with); >= lgr_match_last_work
// out2[1]: the same as a number of (row, column) element pairs of the PADDED operands (stages x 256 rows x 128 columns) -- what the MFMA
// FLOP count is made of (clusters are padded to whole row blocks / column tiles: a few per cent more than Mq x Mt)
extern "C" int lgr_match_last_issued(lgr_ctx* ctx, double* out2) {
    if (!ctx || !out2) return LGR_ERR_INVALID_ARG;
    out2[0] = ctx->mstats.stages_all > 0 ? ctx->mstats.stages_done / ctx->mstats.stages_all : 1.0;
    out2[1] = ctx->mstats.stages_done * (double) BLOCK_ROWS * (double) STAGE_COLS;
    return LGR_OK;
}
extern "C" int lgr_match_last_pairs(lgr_ctx* ctx, unsigned* out2) {
    if (!ctx || !out2) return LGR_ERR_INVALID_ARG;
    out2[0] = ctx->mstats.pairs_ab; out2[1] = ctx->mstats.pairs_ba;
    return LGR_OK;
}
extern "C" int lgr_match_last_shell(lgr_ctx* ctx, double* tiles_skipped) {
    if (!ctx || !tiles_skipped) return LGR_ERR_INVALID_ARG;
    *tiles_skipped = ctx->mstats.shell_skipped;
    return LGR_OK;
}
// (row block, leaf) pairs of the last pruned match call whose lower bound is zero -- never excludable -- and pairs with a finite bound at all:
// what lgr_match_options.auto_dense decides on
extern "C" int lgr_match_last_lbstats(lgr_ctx* ctx, double* out2) {
    if (!ctx || !out2) return LGR_ERR_INVALID_ARG;
    out2[0] = ctx->mstats.lb_zero; out2[1] = ctx->mstats.lb_finite;
    return LGR_OK;
}
// irregular rows of the last match call that went through the exact side scan: [query side, train side, 1 = the lane gave up]
extern "C" int lgr_match_last_irregular(lgr_ctx* ctx, unsigned* out3) {
    if (!ctx || !out3) return LGR_ERR_INVALID_ARG;
    out3[0] = ctx->mstats.irr_a; out3[1] = ctx->mstats.irr_b; out3[2] = ctx->mstats.irr_gave_up;
    return LGR_OK;
}
extern "C" int lgr_match_last_coarse(lgr_ctx* ctx, double* out2) {
    if (!ctx || !out2) return LGR_ERR_INVALID_ARG;
    out2[0] = ctx->mstats.coarse_tested; out2[1] = ctx->mstats.coarse_rejected;
    return LGR_OK;
}
// MFMA operand format of the last match call: 2 / 1 = f16-split operands on v_mfma_f32_32x32x16_f16 (rotated: 192, plain: 224 FLOP per
// pair), 0 = f32 operands on v_mfma_f32_32x32x2_f32 (68 FLOP per pair)
extern "C" int lgr_match_last_format(lgr_ctx* ctx, int* f16) {
    if (!ctx || !f16) return LGR_ERR_INVALID_ARG;
    *f16 = ctx->mstats.f16;
    return LGR_OK;
}

// lgr_match_options.self_check (tests): worst |filtered - exact| / eps over the sampled table entries of the last call, per direction
// (rows, columns); -1 when the check did not run.  A proven bound: must be <= 1.
extern "C" int lgr_match_last_check(lgr_ctx* ctx, double* out2) {
    if (!ctx || !out2) return LGR_ERR_INVALID_ARG;
    out2[0] = ctx->mcheck[0]; out2[1] = ctx->mcheck[1];
    return LGR_OK;
}

// environment: debug dumps only (LGR_MATCH_DEBUG); everything that selects a path is an lgr_match_options field
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Orthonormal basis for the box bounds: principal axes of the k-means sample (both sets).  Covariance on the device,
// cyclic Jacobi on the host (33 x 33), rows of V = eigenvectors, mu = sample mean.  box_bounds == 2: raw coordinates.
// `meanwhile` (optional): enqueues other work of the same stream after the covariance has been requested -- the host Jacobi
// (~0.3 ms) then runs while the device executes it instead of leaving the device idle.
template <class F>
static int box_basis(lgr_ctx* ctx, bool raw, const float* smp, const int* smp_ok, int ns, float* d_basis /* [34][33] + 1: V rows, then mu */, F&& meanwhile) {
    std::vector<float> h(34 * 33 + 1, 0.f);
    if (raw) LGR_TRY(meanwhile());
    // The covariance runs on the context's third stream: `meanwhile` (the ~45 short launches of the Lloyd steps, which everything after the
    // clustering waits for) then does not queue behind it.  Prepared ahead of the call, both compete with the other cloud's feature kernels for
    // the device, and the covariance -- 1.6 ms there, in FRONT of the Lloyd chain -- kept the chain on the critical path (round 3 timeline).
    hipStream_t sC = ctx->stream;
    if (!raw) {
        LGR_TRY(lgr_ctx_stream3(ctx, &sC));
        if (sC != ctx->stream) {
            LGR_HIP(ctx, hipEventRecord(ctx->ev[28], ctx->stream));   // (the sample rows come from km_sample on the context's stream)
            LGR_HIP(ctx, hipStreamWaitEvent(sC, ctx->ev[28], 0));
        }
        const int nb = cdiv(ns, COV_ROWS);
        float* part;
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_DENSE, (size_t) nb * (34 * 33 + 1) + 64, &part));   // scratch: the rerank buffers are not live yet
        LGR_HIP(ctx, hipMemsetAsync(part, 0, (size_t) nb * (34 * 33 + 1) * 4, sC));   // the a > b product slots are never written
        cov_kernel<<<nb, COV_THREADS, 0, sC>>>(smp, smp_ok, ns, part);
        cov_reduce<<<cdiv(34 * 33 + 1, 256), 256, 0, sC>>>(part, nb, d_basis);
        float* hp;
        LGR_TRY(lgr_pinned(ctx, (34 * 33 + 1) * 4, (void**) &hp));
        LGR_HIP(ctx, hipMemcpyAsync(hp, d_basis, (34 * 33 + 1) * 4, hipMemcpyDeviceToHost, sC));
        LGR_HIP(ctx, hipEventRecord(ctx->ev[30], sC));
        const int rc_m = meanwhile();
        LGR_HIP(ctx, hipEventSynchronize(ctx->ev[30]));   // (also when `meanwhile` failed: nothing of this call may stay queued on sC)
        LGR_TRY(rc_m);
        memcpy(h.data(), hp, (34 * 33 + 1) * 4);
    }
    const double n = h[34 * 33];
    std::vector<double> C(33 * 33, 0.0), mu(33, 0.0), Vd(33 * 33, 0.0);
    for (int i = 0; i < 33; ++i) Vd[i * 33 + i] = 1.0;
    if (!raw && n >= 2) {
        for (int k = 0; k < 33; ++k) mu[k] = h[k] / n;
        for (int a = 0; a < 33; ++a)
            for (int b = a; b < 33; ++b) {
                double c = h[33 + a * 33 + b] / n - mu[a] * mu[b];
                C[a * 33 + b] = c; C[b * 33 + a] = c;
            }
        // cyclic Jacobi; rows of Vd become the eigenvectors.  Whatever it converges to, Vd stays a product of plane
        // rotations, i.e. orthonormal -- which is all the bound needs.
        double tr = 0.0;
        for (int i = 0; i < 33; ++i) tr += std::fabs(C[i * 33 + i]);
        for (int sweep = 0; sweep < 12; ++sweep) {
            double off = 0.0;   // converged: the off-diagonal part is at rounding level (more sweeps would only cost host time)
            for (int p = 0; p < 32; ++p)
                for (int q = p + 1; q < 33; ++q) off = std::max(off, std::fabs(C[p * 33 + q]));
            if (off <= 1e-13 * tr) break;
            for (int p = 0; p < 32; ++p)
                for (int q = p + 1; q < 33; ++q) {
                    double apq = C[p * 33 + q];
                    if (std::fabs(apq) < 1e-300) continue;
                    double theta = (C[q * 33 + q] - C[p * 33 + p]) / (2.0 * apq);
                    double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                    for (int k = 0; k < 33; ++k) {
                        double ckp = C[k * 33 + p], ckq = C[k * 33 + q];
                        C[k * 33 + p] = c * ckp - sn * ckq; C[k * 33 + q] = sn * ckp + c * ckq;
                    }
                    for (int k = 0; k < 33; ++k) {
                        double cpk = C[p * 33 + k], cqk = C[q * 33 + k];
                        C[p * 33 + k] = c * cpk - sn * cqk; C[q * 33 + k] = sn * cpk + c * cqk;
                    }
                    for (int k = 0; k < 33; ++k) {
                        double vpk = Vd[p * 33 + k], vqk = Vd[q * 33 + k];
                        Vd[p * 33 + k] = c * vpk - sn * vqk; Vd[q * 33 + k] = sn * vpk + c * vqk;
                    }
                }
        }
    }
    for (int i = 0; i < 33 * 33; ++i) h[i] = (float) Vd[i];
    for (int k = 0; k < 33; ++k) h[33 * 33 + k] = (float) mu[k];
    LGR_HIP(ctx, hipMemcpyAsync(d_basis, h.data(), 34 * 33 * 4, hipMemcpyHostToDevice, sC));
    LGR_HIP(ctx, hipStreamSynchronize(sC));   // h goes out of scope; the basis is in place before anything the host enqueues from here on
    return LGR_OK;
}

// Clustering of one call: centres, leaves, the box basis.  lgr_match_prepare (below) computes it from the query side ahead of the
// call, while the train side's descriptors are still being computed; match_impl consumes it.
struct MatchPrep {
    bool armed = false;            // prepared ahead of the call and not yet consumed
    const float* d_a = nullptr;    // what it was prepared for
    int ma = 0, mb = 0;
    bool both = false;
    lgr_match_options mopt{};
    int sub = 1, rg_rows = BLOCK_ROWS, ns = 0;
    float *cen = nullptr, *cen2 = nullptr, *smp = nullptr, *basis = nullptr;   // in WS_MATCH_MISC
    int* smp_ok = nullptr;
    IrrRef* irr = nullptr;         // the block-sum consensus (km_consensus / km_sample), or nullptr: irregular-row lane off
    bool basis_ready = false;
    Side A;
};
static void match_prep_free(void* p) { delete (MatchPrep*) p; }
static MatchPrep* match_prep_of(lgr_ctx* ctx) {
    if (!ctx->match_prep) { ctx->match_prep = new MatchPrep(); ctx->match_prep_free = match_prep_free; }
    return (MatchPrep*) ctx->match_prep;
}

// ---- 1. k-means centres on a sample: KCL clusters, then `sub` leaves inside every cluster; the basis of the box bounds from the
// same sample.  d_b == nullptr: the train side does not exist yet, the sample comes from the query side alone (any centres are
// valid; the two sides of a registration pair are scans of the same scene).  basis_side_by_side: covariance + host Jacobi on the
// second context while the Lloyd steps run on this one (only when the second context is free).
static int match_cluster(lgr_ctx* ctx, const float* d_a, int ma, const float* d_b, int mb, bool both, bool basis_side_by_side, MatchPrep* P) {
    const lgr_match_options& mo = ctx->mopt;
    // leaves per cluster: about 1024 rows per leaf on the larger side.  lgr_match_options (ctx->mopt: leaves / prune / near)
    // override the automatic choices (tests force the skipping path on small inputs); results never depend on them.
    int sub = 1;
    while (sub < SUBMAX && (long long) KCL * sub * 1024 < std::max(ma, mb)) sub *= 2;
    if (mo.leaves > 0) sub = mo.leaves;
    sub = std::min(SUBMAX, std::max(1, sub));
    const int n_leaves = KCL * sub;
    const int ns = 2 * KM_SAMPLE;
    char* misc;
    size_t off = 8192;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t) 255; return o; };
    const size_t o_cen2 = carve((size_t) MAXLEAF * 33 * 4);
    const size_t o_smp = carve((size_t) ns * 33 * 4), o_ok = carve((size_t) ns * 4), o_label = carve((size_t) ns * 4);
    const size_t o_cbuf = carve((size_t) 2 * KCL * 33 * 4), o_basis = carve((size_t) (34 * 33 + 64) * 4);
    const size_t o_skeys = carve((size_t) 2 * ns * 4), o_svals = carve((size_t) ns * 4), o_sidx = carve((size_t) ns * 4), o_coff = carve(256);
    const size_t o_zero = off;   // zeroed once per call: largest sample magnitude, one set of level-1 sums per Lloyd step, the level-2 sums
    const size_t o_kmax = carve(256), o_irr = carve(sizeof(IrrRef)), o_acc1 = carve((size_t) KM_ITERS * KCL * sizeof(KmAcc)), o_acc2 = carve((size_t) MAXLEAF * sizeof(KmAcc));
    const size_t zero_bytes = off - o_zero;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, off, &misc));
    float* cen = (float*) (misc + 256);                 // [KCL][33]
    float* cen2 = (float*) (misc + o_cen2);             // [n_leaves][33]
    float* smp = (float*) (misc + o_smp);
    int* smp_ok = (int*) (misc + o_ok);
    int* label = (int*) (misc + o_label);
    float* cbuf = (float*) (misc + o_cbuf);             // two scratch copies of the level-1 centres (read one, write the other)
    float* basis = (float*) (misc + o_basis);           // V [33][33], mu [33], count; + 8: rmax2
    unsigned* skeys = (unsigned*) (misc + o_skeys);      // [2][ns]: level-1 labels as sort keys, sorted copy
    int* svals = (int*) (misc + o_svals);
    int* sidx = (int*) (misc + o_sidx);                  // sample indices in cluster order
    int* coff = (int*) (misc + o_coff);                  // [KCL + 1]
    unsigned* kmax = (unsigned*) (misc + o_kmax);
    KmAcc* acc1 = (KmAcc*) (misc + o_acc1);
    KmAcc* acc2 = (KmAcc*) (misc + o_acc2);
    LGR_HIP(ctx, hipMemsetAsync(misc + o_zero, 0, zero_bytes, ctx->stream));
    // irregular rows: a vote over the sample rows finds the consensus of the block sums; km_sample leaves rows off it out of the sample
    IrrRef* irr = mo.irregular_rows ? (IrrRef*) (misc + o_irr) : nullptr;
    if (irr) km_consensus<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(d_a, ma, d_b, mb, KM_SAMPLE, irr);
    km_sample<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(d_a, ma, d_b, mb, KM_SAMPLE, smp, smp_ok, kmax, irr);
    auto lloyd = [&](lgr_ctx* cx) -> int {
        km_init<<<1, 64, 0, cx->stream>>>(smp, smp_ok, ns, cbuf);
        for (int it = 0; it <= KM_ITERS; ++it) {   // Lloyd with order-free integer sums; the last launch labels with the final centres
            const bool last = it == KM_ITERS;
            km1_step<<<cdiv(ns, KM1_THREADS), KM1_THREADS, 0, cx->stream>>>(smp, smp_ok, ns, kmax, cbuf + (it & 1) * KCL * 33, it ? acc1 + (size_t) (it - 1) * KCL : nullptr,
                                                                            last ? nullptr : acc1 + (size_t) it * KCL, last ? cen : cbuf + ((it + 1) & 1) * KCL * 33, label);
        }
        // the samples in cluster order (stable: sample order inside a cluster)
        km_label_keys<<<cdiv(ns, 256), 256, 0, cx->stream>>>(label, ns, skeys, svals);
        LGR_TRY(lgr_sort_pairs_u32(cx, skeys, skeys + ns, svals, sidx, (size_t) ns, 0, 5));
        km_cluster_offsets<<<1, 64, 0, cx->stream>>>(skeys + ns, ns, coff);
        km2_init<<<KCL, 64, 0, cx->stream>>>(smp, sidx, coff, cen, sub, cen2);
        if (sub > 1) {
            for (int it = 0; it < KM2_ITERS; ++it) {
                km2_step<<<dim3(KM2_PIECES, KCL), KM2_THREADS, 0, cx->stream>>>(smp, sidx, coff, kmax, cen2, sub, acc2);
                km2_finalize<<<n_leaves, 64, 0, cx->stream>>>(acc2, kmax, cen2);
            }
        }
        LGR_HIP(cx, hipGetLastError());
        return LGR_OK;
    };
    const bool want_basis = mo.box_bounds != 0;
    if (want_basis && basis_side_by_side) {
        LGR_TRY(lgr_run_pair(ctx, lloyd, [&](lgr_ctx* cx) { return box_basis(cx, mo.box_bounds == 2, smp, smp_ok, ns, basis, []() { return (int) LGR_OK; }); }));
    } else if (want_basis) {
        LGR_TRY(box_basis(ctx, mo.box_bounds == 2, smp, smp_ok, ns, basis, [&]() { return lloyd(ctx); }));   // Jacobi on the host under the Lloyd steps
    } else {
        LGR_TRY(lloyd(ctx));
    }
    auto pick_group = [](size_t q_count, size_t t_count) {   // table [t/g][q] floats kept under ~6 GB
        int g = 1024;
        while (g < 4096 && (t_count / g + 1) * q_count * 4 > ((size_t) 6 << 30)) g *= 2;
        return g;
    };
    int rg_rows = both ? pick_group((size_t) mb, (size_t) ma) : BLOCK_ROWS;   // row groups (column direction table)
    if (ma <= 65536) rg_rows = BLOCK_ROWS;                                     // small inputs: keep the cluster padding small
    P->d_a = d_a; P->ma = ma; P->mb = mb; P->both = both; P->mopt = mo;
    P->sub = sub; P->rg_rows = rg_rows; P->ns = ns;
    P->cen = cen; P->cen2 = cen2; P->smp = smp; P->smp_ok = smp_ok; P->basis = basis; P->basis_ready = want_basis; P->irr = irr;
    return LGR_OK;
}

// Clustering for a coming lgr_match_bf*_dev(ctx, d_a, ma, <train side of mb rows>, both directions or not) from the query side's
// descriptors alone (a chain of ~60 short launches and one host Jacobi), so that the caller can run it while the train side's
// descriptors are still being computed on another context.  Consumed by the next matcher call on this context when
// (d_a, ma, mb, both, options) agree; lgr_match_prepare_cancel drops it (every exit path of the caller).
int lgr_match_prepare(lgr_ctx* ctx, const float* d_a, int ma, int mb, bool both) {
    MatchPrep* P = match_prep_of(ctx);
    P->armed = false;
    if (!d_a || ma <= 0 || mb <= 0) return LGR_OK;
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    LGR_TRY(match_cluster(ctx, d_a, ma, nullptr, mb, both, false, P));
    P->armed = true;
    return LGR_OK;
}
void lgr_match_prepare_cancel(lgr_ctx* ctx) {
    if (ctx && ctx->match_prep) ((MatchPrep*) ctx->match_prep)->armed = false;
}

static int match_impl(lgr_ctx* ctx, const float* d_a, int ma, const float* d_b, int mb, int block,
                      int32_t* d_ab_idx, float* d_ab_dist, int32_t* d_ba_idx, float* d_ba_dist) {
    LGR_CHECK(ctx, ctx && (d_a || ma == 0) && (d_b || mb == 0) && (d_ab_idx || ma == 0) && (d_ab_dist || ma == 0), LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, ma >= 0 && mb >= 0 && block > 0, LGR_ERR_INVALID_ARG);
    bool both = d_ba_idx != nullptr && mb > 0;
    if (d_ba_idx) LGR_CHECK(ctx, d_ba_dist != nullptr, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    lgr_match_stats& g_last_stats = ctx->mstats;
    double* const g_last_check = ctx->mcheck;
    memset(&g_last_stats, 0, sizeof g_last_stats);
    g_last_check[0] = g_last_check[1] = -1;
    ctx->mfma_timed = 0;
    // default result: unmatched
    if (ma) { LGR_HIP(ctx, hipMemsetAsync(d_ab_idx, 0xff, (size_t) ma * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ab_dist, 0, (size_t) ma * 4, ctx->stream)); }
    if (mb && d_ba_idx) { LGR_HIP(ctx, hipMemsetAsync(d_ba_idx, 0xff, (size_t) mb * 4, ctx->stream)); LGR_HIP(ctx, hipMemsetAsync(d_ba_dist, 0, (size_t) mb * 4, ctx->stream)); }
    if (ma == 0 || mb == 0) return LGR_OK;

    const lgr_match_options& mo = ctx->mopt;
    const int prune_mode = mo.prune;   // -1 auto, 0 off, 1 on
    const int near_t = mo.near > 0 ? mo.near : NEAR_T;
    const bool prune = prune_mode == 1 || (prune_mode != 0 && (double) ma * mb >= 65536.0 * 65536.0);

    // ---- 1. + 2. clustering, then assign / sort / place both sides -- unless the query side was prepared ahead of this call
    MatchPrep* P = match_prep_of(ctx);
    const bool prepared = P->armed && P->d_a == d_a && P->ma == ma && P->mb == mb && P->both == both && memcmp(&P->mopt, &mo, sizeof mo) == 0;
    P->armed = false;
    Side B;
    if (!prepared) LGR_TRY(match_cluster(ctx, d_a, ma, d_b, mb, both, true, P));
    // the two sides are independent (assign, sort, two host read-backs each): side by side on the two contexts
    auto build_sides = [&](const IrrRef* irr) {
        return lgr_run_pair(ctx, [&](lgr_ctx* cx) { return build_side(cx, d_a, ma, P->cen, P->cen2, P->sub, 1, P->rg_rows, WS_MATCH_NA, WS_MATCH_AP, 0, irr, &P->A); },
                            [&](lgr_ctx* cx) { return build_side(cx, d_b, mb, P->cen, P->cen2, P->sub, TILE, PAD, WS_MATCH_NB, WS_MATCH_BP, 1, irr, &B); });
    };
    LGR_TRY(build_sides(P->irr));
    // the irregular-row lane gives up when a side has more such rows than its list holds, or when they are all a side has (the centres then
    // come from the other side alone and nothing is left for the filter): both sides again with every finite row in the operands
    if (P->A.n_irr > IRR_CAP || B.n_irr > IRR_CAP || ((P->A.n_irr || B.n_irr) && (P->A.n_valid == 0 || B.n_valid == 0))) {
        g_last_stats.irr_gave_up = 1u;
        LGR_TRY(build_sides(nullptr));
    }
    const Side& A = P->A;
    g_last_stats.irr_a = (unsigned) A.n_irr; g_last_stats.irr_b = (unsigned) B.n_irr;
    const int sub = P->sub, n_leaves = KCL * sub, rg_rows = P->rg_rows, ns = P->ns;
    float *const cen = P->cen, *const cen2 = P->cen2, *const smp = P->smp;
    int* const smp_ok = P->smp_ok;
    (void) smp; (void) smp_ok; (void) ns;
    char* misc;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, 8192, &misc));   // (grown by match_cluster; the first 8 KB hold small per-call scalars)
    if (A.n_valid == 0 || B.n_valid == 0) return LGR_OK;
    const int ma_pad = A.n_pad, mb_pad = B.n_pad;
    g_last_stats.rg_rows = rg_rows;
    const int ta = ma_pad / TILE, tb = mb_pad / TILE;
    const int n_rb = ma_pad / BLOCK_ROWS, n_stage_total = mb_pad / STAGE_COLS;

    // The rows in sorted order (exact rerank, box bounds) and the bounding boxes of the row blocks / leaves need nothing of the operand
    // packing below: four launches on the second stream, which run under the packing passes (joined before section 4).
    float *sortedA = nullptr, *sortedB;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_SORTED_B, (size_t) mb_pad * 33, &sortedB));
    if (both || prune) LGR_TRY(lgr_ws_t(ctx, WS_MATCH_SORTED_A, (size_t) ma_pad * 33, &sortedA));
    const bool boxes = prune && mo.box_bounds != 0;
    float *boxA = nullptr, *boxBt = nullptr;
    unsigned* rmax2 = nullptr;
    if (boxes) {
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BOX, (size_t) n_rb * 66 + (size_t) n_leaves * 66 + 64, &boxA));
        boxBt = boxA + (size_t) n_rb * 66;
        rmax2 = (unsigned*) (boxBt + (size_t) n_leaves * 66);
    }
    // Whatever makes this call return early from here on (an allocation failure, a HIP error): kernels queued on the helper streams read the
    // caller's d_a / d_b and write this call's buffers, so those streams are drained before the caller gets its buffers back -- and before the
    // device's turn goes to another context (lgr_turn records its hand-over event on ctx->stream only).  On the normal path both streams have
    // been joined into ctx->stream long before, and the synchronisation returns at once.
    struct DrainOnExit {
        hipStream_t s, own;
        ~DrainOnExit() { if (s && s != own) (void) hipStreamSynchronize(s); }
    };
    LGR_TRY(lgr_ctx_aux(ctx));
    DrainOnExit drain_aux{ctx->aux->stream, ctx->stream};   // box_kernel / gather_rows_kernel (launch_sorted_copies)
    {
        hipStream_t s2 = ctx->aux->stream;
        LGR_HIP(ctx, hipEventRecord(ctx->aux_ev, ctx->stream));
        LGR_HIP(ctx, hipStreamWaitEvent(s2, ctx->aux_ev, 0));
        if (boxes) {   // first (the bounds wait for them), reading the rows through the placement
            const float* basis = P->basis;   // V [33][33], mu [33] (match_cluster)
            LGR_HIP(ctx, hipMemsetAsync(rmax2, 0, 4, s2));
            box_kernel<<<n_rb, 256, 0, s2>>>(d_a, A.perm, nullptr, n_rb, basis, basis + 33 * 33, 0, boxA, rmax2);
            box_kernel<<<n_leaves, 256, 0, s2>>>(d_b, B.perm, B.leaf_start, n_leaves, basis, basis + 33 * 33, 1, boxBt, rmax2);
            LGR_HIP(ctx, hipEventRecord(ctx->ev[30], s2));
        }
        LGR_HIP(ctx, hipGetLastError());
    }
    // The sorted copies are for the exact rerank (and the f32 ball bounds): with the f16 formats nothing before the MFMA passes reads them, so
    // they are gathered UNDER pass 0 (launch_sorted_copies, called where that pass is queued: the set-up phase in front of it is bound by
    // HBM -- 3.1 GB of column operands -- and the passes are not).
    bool sorted_launched = false;
    auto launch_sorted_copies = [&](bool behind_main) -> int {
        if (sorted_launched) return LGR_OK;
        sorted_launched = true;
        hipStream_t s2 = ctx->aux->stream;
        if (behind_main) {   // not before the work queued on the main stream so far
            LGR_HIP(ctx, hipEventRecord(ctx->ev[26], ctx->stream));
            LGR_HIP(ctx, hipStreamWaitEvent(s2, ctx->ev[26], 0));
        }
        gather_rows_kernel<<<cdiv((long long) mb_pad * 33, 256), 256, 0, s2>>>(d_b, B.perm, mb_pad, sortedB);
        if (sortedA) gather_rows_kernel<<<cdiv((long long) ma_pad * 33, 256), 256, 0, s2>>>(d_a, A.perm, ma_pad, sortedA);
        LGR_HIP(ctx, hipGetLastError());
        LGR_HIP(ctx, hipEventRecord(ctx->aux_ev, s2));
        return LGR_OK;
    };

    // ---- 3. pack operands, group maxima, stage -> leaf map
    const bool f16 = mo.operand_format != 0;
    g_last_stats.f16 = f16 ? 1 : 0;
    EpsExtra ex{0.f, 0.f, 1.f};
    float c_scale = 1.f, out_scale = 1.f;
    F16Scale sc{1.f, 1.f, {1.f, 1.f, 1.f}};
    bool rot = false;
    float *nAp, *nBp;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_NORMS, (size_t) ma_pad + (size_t) KCL * mb_pad + 2 * ((size_t) ta + (size_t) KCL * tb) + 64, &nAp));
    nBp = nAp + ma_pad;
    // radial shells of the 32-row / 32-column tiles about the centres they are packed against (written by the f16 packing kernels)
    float2* const shellA = (float2*) (nBp + (size_t) KCL * mb_pad);
    float2* const shellB = shellA + ta;
    unsigned* d_max = (unsigned*) (misc + 128);   // [2]: norm overflow flag of the f32 packing (the f16 statistics come from assign_kernel)
    LGR_HIP(ctx, hipMemsetAsync(d_max, 0, 12, ctx->stream));
    bool force_dense = false;
    if (f16) {
        // the power-of-two scale 2^s puts the largest operand (2 |a'| 2^s, |b'| 2^s) just under 2^15; the largest |x - c|^2 and the
        // largest energy of the three coordinates the rotated 30-D format would drop come from assign_kernel (build_side)
        const float r2 = std::max(A.nstat_n2, B.nstat_n2), drop2 = std::max(A.nstat_drop, B.nstat_drop);
        force_dense = A.nstat_ovf || B.nstat_ovf;
        // Rotated format (FMT_F16R) when what it drops is negligible: with u the dropped coordinates of a row relative to a
        // centre, d2 = d2_30 + |u_a - u_b|^2 and 0 <= |u_a - u_b|^2 <= 4 max |u|^2 -- that bound joins the absolute error
        // term, so the choice below only trades speed.  FPFH rows: every block sums to 100 -> max |u|^2 ~ 1e-7.
        const int rot_env = mo.operand_format < 0 ? -1 : (mo.operand_format == 2 ? 1 : 0);
        rot = rot_env >= 0 ? rot_env != 0 : (4.0 * (double) drop2 <= 1e-8 * (double) r2);
        if (env_int("LGR_MATCH_DEBUG", 0))
            fprintf(stderr, "[lgr] operand statistics: max |x - c|^2 %.6g, dropped energy %.6g (A %.6g, B %.6g) -> %s; irregular rows %d + %d\n", (double) r2, (double) drop2,
                    (double) A.nstat_drop, (double) B.nstat_drop, rot ? "rotated" : "plain", A.n_irr, B.n_irr);
        double R = std::sqrt((double) r2);
        int sexp = R > 0 ? (int) std::floor(std::log2(16384.0 / R)) : 14;
        sexp = std::max(-40, std::min(14, sexp));
        double N_max = (double) r2 * std::ldexp(1.0, 2 * sexp);
        int e1 = N_max > 32768.0 ? (int) std::ceil(std::log2(N_max / 32768.0)) : 0;
        e1 = std::min(15, std::max(0, e1));
        sc.s_mul = (float) std::ldexp(1.0, sexp);
        sc.inv_s2 = (float) std::ldexp(1.0, -2 * sexp);
        sc.a_norm[0] = (float) std::ldexp(1.0, e1);
        sc.a_norm[1] = (float) std::ldexp(1.0, std::max(0, e1 - 11));
        sc.a_norm[2] = (float) std::ldexp(1.0, std::max(0, e1 - 22));
        c_scale = (float) std::ldexp(1.0, 2 * sexp);
        out_scale = sc.inv_s2;
        // error terms of the split: elements whose second half would be an f16 subnormal may be flushed
        // (tau per element, linear term); three-term norm expansion (absolute term); the 112-product f32 accumulation
        // chain and the 2^-22 split residual are covered by doubling the quadratic term
        const double tau = std::ldexp(1.0, -14 - sexp);
        ex.lin = (float) (12.0 * tau);
        ex.abs = (float) (2.0 * std::ldexp(1.0, -14) * (sc.a_norm[2] + sc.a_norm[1] / 2048.0 + sc.a_norm[0] / 4194304.0) * (double) sc.inv_s2 * 1.01);   // both norms
        ex.quad = 2.f;
        if (rot) {
            // Helmert coordinates are computed in f32: |dy| <= 10.4 u |x'| per vector (prefix sums of <= 11 terms, one
            // rounded constant) -> 20.8 u (x + y)^2 on d2, 0.13 of the unit 4 g40 (x + y)^2; the dropped energy is absolute
            ex.quad = 2.2f;
            ex.abs = (float) ((double) ex.abs + 4.0 * (double) drop2 * 1.0001);
        }
    }
    g_last_stats.f16 = f16 ? (rot ? 2 : 1) : 0;
    const int KS = !f16 ? OpFmt<FMT_F32>::KS : rot ? OpFmt<FMT_F16R>::KS : OpFmt<FMT_F16>::KS;
    const size_t frag_bytes = f16 ? sizeof(f16x8) : sizeof(float);
    const size_t a_op_bytes = (size_t) ta * KS * 64 * frag_bytes;
    const size_t bset_stride = (size_t) tb * KS * 64;   // fragments per column set
    char *Aop, *Bop;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ROWMIN, a_op_bytes + 256, &Aop));
    const size_t b_op_bytes = KCL * bset_stride * frag_bytes;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_COLMIN, b_op_bytes + 256, &Bop));
    const int n_rg = cdiv(ma_pad, rg_rows);
    // column groups of the row-minimum table: a leaf, cut into pieces of at most GROUP_COLS columns (k-means leaves of
    // near-duplicate descriptors can hold tens of thousands of rows; the exact rerank scans a whole group per item)
    std::vector<int> h_group_start, h_tiles(2 * (size_t) tb);   // [tile] -> group, [tb + tile] -> leaf
    int group_cols = GROUP_COLS;   // larger pieces for very large inputs: keep the table [groups][ma_pad] under ~24 GB
    while (group_cols < 65536 && ((size_t) mb_pad / group_cols + n_leaves) * (size_t) ma_pad * 4 > ((size_t) 24 << 30)) group_cols *= 2;
    for (int l = 0; l < n_leaves; ++l)
        for (int s0 = B.h_leaf_start[l]; s0 < B.h_leaf_start[l + 1]; s0 += group_cols) {
            int s1 = std::min(B.h_leaf_start[l + 1], s0 + group_cols), g = (int) h_group_start.size();
            h_group_start.push_back(s0);
            for (int t = s0 / TILE; t < s1 / TILE; ++t) { h_tiles[t] = g; h_tiles[tb + t] = l; }
        }
    const int n_groups = (int) h_group_start.size();
    g_last_stats.sub_cols = n_groups;
    h_group_start.push_back(mb_pad);
    float *gmaxB, *gmaxA;
    int *cl_of_rg, *tile_group, *tile_leaf, *group_start;
    int *group_leaf, *leaf_g0;   // [n_groups]: leaf of a group; [n_leaves + 1]: first group of a leaf (the skipping schedule's tables)
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BEST_B, (size_t) KCL * n_groups + 2 * (size_t) n_rg + 2 * (size_t) tb + 2 * (size_t) n_groups + n_leaves + 72, &gmaxB));
    gmaxA = gmaxB + (size_t) KCL * n_groups;
    cl_of_rg = (int*) (gmaxA + n_rg);
    tile_group = cl_of_rg + n_rg;
    tile_leaf = tile_group + tb;
    group_start = tile_leaf + tb;
    group_leaf = group_start + n_groups + 1;
    leaf_g0 = group_leaf + n_groups;
    {
        // every small host-built table of the call goes up HERE, before the operand packing is enqueued: the one host wait they need
        // (the staging vectors go out of scope) then falls on the short placement kernels, not behind the packing
        std::vector<int> h(n_rg), hgl(n_groups), hl(n_leaves + 1, n_groups);
        for (int g = 0; g < n_rg; ++g) h[g] = A.h_blkcl[(size_t) g * (rg_rows / BLOCK_ROWS)];
        for (int g = 0; g < n_groups; ++g) hgl[g] = h_tiles[tb + h_group_start[g] / TILE];
        for (int g = n_groups - 1; g >= 0; --g) hl[hgl[g]] = g;                       // first group of every leaf that has one
        for (int l = n_leaves - 1; l >= 0; --l) hl[l] = std::min(hl[l], hl[l + 1]);   // empty leaves: g0 == g1
        LGR_HIP(ctx, hipMemcpyAsync(cl_of_rg, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(tile_group, h_tiles.data(), h_tiles.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(group_start, h_group_start.data(), h_group_start.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(group_leaf, hgl.data(), hgl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(leaf_g0, hl.data(), hl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // The column (train-side) operands -- 16 sets, 3 GB at 1M rows, the longest piece of the set-up -- are packed on a third stream:
    // nothing before the first MFMA pass reads them (the bounds of pass 0 need the ROW operands, the packed leaf centres and the
    // boxes), so that chain runs beside the packing instead of behind it.  sB is joined in launch_mfma.
    // (It starts BEHIND the row operands' packing: side by side the two packing kernels share the HBM write bandwidth and the row
    // operands, which the bounds wait for, take as long as the 3 GB of column sets.)
    hipStream_t sB = ctx->stream;
    if (f16) LGR_TRY(lgr_ctx_stream3(ctx, &sB));
    DrainOnExit drain_sB{sB, ctx->stream};   // (the third stream: column operands, table initialisation)
    auto fork_b = [&]() -> int {
        if (sB != ctx->stream) {
            LGR_HIP(ctx, hipEventRecord(ctx->ev3, ctx->stream));
            LGR_HIP(ctx, hipStreamWaitEvent(sB, ctx->ev3, 0));
        }
        return LGR_OK;
    };
    const int n_cc = cdiv(mb_pad, CHUNK_COLS);
    // ---- (3b ahead of time) the skipping bookkeeping's workspace, cleared on the third stream, and the leaf centres packed as train rows for the
    // ball bounds on the matrix cores -- both need nothing of the operand packing below and used to sit on the critical chain behind it
    // (a 12 MB memset and two short launches: 0.36 ms between pack16 and lb_mfma_kernel in the round-5 timeline)
    char* pb = nullptr;
    size_t poff = 0;
    auto pcarve = [&](size_t bytes) { size_t o = poff; poff += (bytes + 255) & ~(size_t) 255; return o; };
    const size_t o_lb = pcarve((size_t) n_rb * n_leaves * 4), o_done = pcarve((size_t) n_rb * n_leaves), o_sched = pcarve((size_t) n_rb * n_leaves);
    const size_t o_touched = pcarve((size_t) n_rb * n_leaves);   // (inside the range cleared below)
    const size_t o_mask = pcarve((size_t) n_rb * n_cc * 4), o_macc = pcarve((size_t) n_rb * n_cc * 4), o_urb = pcarve((size_t) n_rb * 4), o_urt = pcarve((size_t) n_rb * (BLOCK_ROWS / TILE) * 4), o_ul = pcarve((size_t) MAXLEAF * 4);
    const size_t o_stats = pcarve(sizeof(MaskStats)), o_lbpart = pcarve((size_t) n_rb * sizeof(uint2));
    const size_t o_lfst = pcarve((size_t) MAXLEAF * 4), o_llst = pcarve((size_t) MAXLEAF * 4);   // first / last column stage of every leaf (mask_sparse_kernel)
    const size_t o_mchk = pcarve(mo.self_check ? (size_t) 2 * n_rb * n_cc * 4 + 256 : 0);         // self_check: mask_kernel's masks beside the sparse kernel's
    const size_t o_ust = pcarve((size_t) n_stage_total * 4), o_uct = pcarve((size_t) tb * 4);
    const size_t o_cr = pcarve((size_t) n_rb * n_groups), o_cc = pcarve((size_t) n_leaves * n_rg);
    const size_t o_smax = pcarve((size_t) KCL * n_stage_total * 4), o_ccnt = pcarve(32);
    const size_t o_urow = pcarve((size_t) ma_pad * 4), o_ucolv = pcarve((size_t) mb_pad * 4);
    const size_t o_ssh = pcarve((size_t) KCL * n_stage_total * 8), o_rsh = pcarve((size_t) n_rb * 8);
    const size_t o_cperm = pcarve((size_t) (n_leaves + TILE) * 4), o_cnrm = pcarve((size_t) KCL * (n_leaves + TILE) * 4);
    const size_t o_cop = pcarve((size_t) KCL * ((n_leaves + TILE) / TILE) * 7 * 64 * sizeof(f16x8));   // the leaf centres as packed train rows
    if (prune) LGR_TRY(lgr_ws_t(ctx, WS_MATCH_PRUNE, poff, &pb));
    const int n_cpad = pad_to(n_leaves, TILE);
    const size_t cset_stride = (size_t) (n_cpad / TILE) * KS * 64;
    bool early_b = false;   // ev[27] on sB: workspace cleared (+ centres packed, f16 formats)
    if (prune) {
        if (sB != ctx->stream) {
            LGR_HIP(ctx, hipEventRecord(ctx->ev[31], ctx->stream));
            LGR_HIP(ctx, hipStreamWaitEvent(sB, ctx->ev[31], 0));
        }
        LGR_HIP(ctx, hipMemsetAsync(pb + o_done, 0, o_cr - o_done, sB));   // done, sched, touched, masks, bounds, stats
        if (f16) {
            int* cperm = (int*) (pb + o_cperm);
            centre_perm_kernel<<<cdiv(n_cpad, 256), 256, 0, sB>>>(n_leaves, n_cpad, cperm);
            if (rot) pack16_kernel<true><<<cdiv(n_cpad, 256), 256, 0, sB>>>(cen2, cperm, n_cpad, 1, cen, nullptr, sc, (_Float16*) (pb + o_cop), (float*) (pb + o_cnrm), nullptr);
            else pack16_kernel<false><<<cdiv(n_cpad, 256), 256, 0, sB>>>(cen2, cperm, n_cpad, 1, cen, nullptr, sc, (_Float16*) (pb + o_cop), (float*) (pb + o_cnrm), nullptr);
        }
        if (sB != ctx->stream) { LGR_HIP(ctx, hipEventRecord(ctx->ev[27], sB)); early_b = true; }
    }
    if (!f16) {
        pack_kernel<<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, (float*) Aop, nAp, d_max + 2);
        pack_kernel<<<dim3(cdiv(mb_pad, 256), KCL), 256, 0, ctx->stream>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, (float*) Bop, nBp, d_max + 2);
        unsigned* h_ovf;
        LGR_TRY(lgr_pinned(ctx, 64, (void**) &h_ovf));
        LGR_HIP(ctx, hipMemcpyAsync(h_ovf, d_max + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        force_dense = h_ovf[0] != 0u;
    } else {
        if (rot) pack16_kernel<true><<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, sc, (_Float16*) Aop, nAp, shellA);
        else pack16_kernel<false><<<cdiv(ma_pad, 256), 256, 0, ctx->stream>>>(d_a, A.perm, ma_pad, 0, cen, A.blkcl, sc, (_Float16*) Aop, nAp, shellA);
    }
    // the column operands (f16 formats: 3 GB at 1M, HBM-write bound) and their group maxima, on the third stream behind the row operands.  (Round 5
    // measured the alternatives: behind lb_mfma_kernel -- which then runs alone in 0.18 ms instead of 0.85 -- box_lb_kernel crawls beside the
    // packing instead (0.07 -> 0.92 ms); behind the near kernels, with pass 0's stage selection from assign_kernel's distances so that
    // mask_kernel need not wait for the column norms: mask_kernel crawls (0.4 -> 1.04 ms) and pass 0 starts 0.1 ms later than with this order.)
    bool b_packed = !f16;
    auto pack_b = [&]() -> int {
        if (b_packed) return LGR_OK;
        b_packed = true;
        LGR_TRY(fork_b());
        if (rot) pack16_kernel<true><<<cdiv(mb_pad, 256), 256, 0, sB>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, sc, (_Float16*) Bop, nBp, shellB);
        else pack16_kernel<false><<<cdiv(mb_pad, 256), 256, 0, sB>>>(d_b, B.perm, mb_pad, 1, cen, nullptr, sc, (_Float16*) Bop, nBp, shellB);
        group_max_kernel<<<dim3(n_groups, KCL), 256, 0, sB>>>(nBp, mb_pad, 0, group_start, gmaxB);
        return LGR_OK;
    };
    if (!f16) group_max_kernel<<<dim3(n_groups, KCL), 256, 0, sB>>>(nBp, mb_pad, 0, group_start, gmaxB);
    else LGR_TRY(pack_b());   // (right behind the row operands: 1 ms of HBM writes that everything up to pass 0 crawls beside -- started later, pass 0 starts later)
    group_max_kernel<<<dim3(n_rg, 1), 256, 0, ctx->stream>>>(nAp, ma_pad, rg_rows, nullptr, gmaxA);
    bool b_joined = sB == ctx->stream, b_recorded = false;
    // (the event is recorded behind the last PRODUCER on sB -- record_b, called where the set-up has been enqueued -- not where the first reader
    //  joins: by then sB also holds pass 0's init_tables_sparse_kernel, which mask_kernel is meant to run beside, not behind; it has an event of its own)
    auto record_b = [&]() -> int {
        if (!b_joined && !b_recorded) LGR_HIP(ctx, hipEventRecord(ctx->ev3, sB));
        b_recorded = true;
        return LGR_OK;
    };
    auto join_b = [&]() -> int {   // everything that reads the column operands, their norms or maxima comes after this
        if (!b_joined) {
            LGR_TRY(record_b());
            LGR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev3, 0));
        }
        b_joined = true;
        return LGR_OK;
    };
    // (sortedA / sortedB / the boxes: forked onto the second stream above; joined where they are first read)
    bool sorted_joined = false;
    auto join_sorted = [&]() -> int {
        LGR_TRY(launch_sorted_copies(false));   // (a reader in front of the passes: the f32 ball bounds)
        if (!sorted_joined) LGR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_ev, 0));
        sorted_joined = true;
        return LGR_OK;
    };

    // ---- 4. MFMA passes into the two minimum tables (+inf initialised)
    int *rowmin, *colmin = nullptr;
    const size_t tab_floats = (size_t) n_groups * ma_pad + (both ? (size_t) n_rg * mb_pad : 0);
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_BEST_A, tab_floats + 2 * ((size_t) ma + mb) + 64, &rowmin));
    if (both) colmin = rowmin + (size_t) n_groups * ma_pad;
    unsigned long long* bestA = (unsigned long long*) (rowmin + tab_floats + (tab_floats & 1));
    unsigned long long* bestB = bestA + ma;
    fill_u64<<<cdiv(ma + mb, 256), 256, 0, sB>>>(bestA, ma + mb, ~0ull);   // (the exact rerank's tables: joined with the column operands)
    // dense mode: +inf everywhere; skipping mode: init_tables_sparse_kernel covers what each pass computes (lgr_match_options.poison_tables, tests: the
    // rest is filled with 0 -- the most harmful value a stale entry could have -- to show that nothing reads it)
    if (!prune) LGR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t) rowmin, 0x7f800000, tab_floats, ctx->stream));
    else if (mo.poison_tables) LGR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t) rowmin, 0, tab_floats, ctx->stream));
    // work items of the persistent MFMA kernel: one row group (the owner of its column minima) x one column chunk
    const int item_rb = both ? std::min(rg_rows / BLOCK_ROWS, 16) : 4;
    const int n_ir = cdiv(n_rb, item_rb), ccx = cdiv(n_cc, 8), n_flags = 8 * ccx * n_ir;
    int* ibuf;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ITEMS2, (size_t) 4 * n_flags + 128, &ibuf));
    int *iflags = ibuf, *ipos = ibuf + n_flags;
    int2* ilist = (int2*) (ibuf + 2 * (size_t) n_flags);
    int* xcd_start = ibuf + 4 * (size_t) n_flags;   // [9]
    int* xcd_ctr = xcd_start + 16;                   // [8]
    const int mfma_grid = 8 * (LGR_MM_OCC / 2) * std::max(1, ctx->n_cu / 8);   // resident workgroups: LGR_MM_OCC / 2 per CU
    // final pass as sweep + listed tiles (lgr_match_sweep.cuh): the list, its counter, the per-column thresholds as bf16
    const unsigned kept_cap = mo.kept_cap > 0 ? (unsigned) mo.kept_cap : (16u << 20);
    uint2* kept = nullptr;
    unsigned long long* kept_count = nullptr;
    unsigned short* ucol16 = nullptr;
    bool split_used = false;
    if (f16 && rot && prune && mo.coarse_rejection != 0 && mo.split_sweep != 0) {
        char* kb;
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_KEPT, (size_t) kept_cap * sizeof(uint2) + (size_t) mb_pad * 2 + 512, &kb));
        kept = (uint2*) kb;
        kept_count = (unsigned long long*) (kb + (size_t) kept_cap * sizeof(uint2));
        ucol16 = (unsigned short*) (kb + (size_t) kept_cap * sizeof(uint2) + 256);
    }
    const unsigned long long* cur_pass_stages = nullptr;   // device: the stage count of the pass being launched (mask_kernel's statistics)
    // final pass as sweep + listed tiles: the tables are initialised behind the sweep, for the (row block, leaf) pairs the list touches (touched_kernel)
    std::function<int()> init_touched;   // set by the pass loop for the pass it applies to
    auto launch_mfma = [&](const unsigned* mask, CoarseArgs ca, bool allow_split = true) -> int {
        items_flag_kernel<<<cdiv(n_flags, 256), 256, 0, ctx->stream>>>(mask, n_rb, n_cc, item_rb, n_ir, ccx, iflags);
        size_t sb = 0;
        LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, sb, iflags, ipos, 0, (size_t) n_flags, rocprim::plus<int>(), ctx->stream));
        void* stmp;
        LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, sb, &stmp));
        LGR_HIP(ctx, rocprim::exclusive_scan(stmp, sb, iflags, ipos, 0, (size_t) n_flags, rocprim::plus<int>(), ctx->stream));
        items_emit_kernel<<<cdiv(n_flags, 256), 256, 0, ctx->stream>>>(iflags, ipos, item_rb, n_ir, ccx, ilist, xcd_start);
        LGR_HIP(ctx, hipMemsetAsync(xcd_ctr, 0, 32, ctx->stream));
        LGR_CHECK(ctx, ctx->mfma_timed < 8, LGR_ERR_INVALID_ARG);
        LGR_TRY(join_b());
        (void) hipEventRecord(ctx->ev[9 + 2 * ctx->mfma_timed], ctx->stream);
#define LGR_MFMA_ARGS bset_stride, c_scale, out_scale, A.blkcl, nAp, ma_pad, mb_pad, rg_rows, tile_group, mask, rowmin, colmin, n_cc, item_rb, ilist, xcd_start, xcd_ctr, ca
        if (f16 && rot && ca.u_rb && kept && allow_split) {
            // the coarse sweep appends the tiles it keeps to a list, a second kernel finishes them
            split_used = true;
            LGR_HIP(ctx, hipMemsetAsync(kept_count, 0, 8, ctx->stream));
            if (ca.u_colv) ucol_pack_kernel<<<cdiv(mb_pad, 256), 256, 0, ctx->stream>>>(ca.u_colv, mb_pad, c_scale, ucol16);
            // Descriptors the bounds cannot separate (structureless rows: every stage is scheduled) gain nothing from the coarse sweep -- nearly every
            // tile passes it and is then computed a second time in full.  The device decides from the pass's stage count (no host round trip: a
            // synchronisation here cost the pass 2.6 ms): above half of all stages the work list goes to the plain six-step kernel, otherwise to
            // the sweep; the other kernel finds an empty list.
            int* xs_sweep = xcd_start + 32;
            int* xs_plain = xcd_start + 48;
            pass_select_kernel<<<1, 16, 0, ctx->stream>>>(mo.coarse_rejection == 2 ? nullptr : cur_pass_stages, 0.5 * g_last_stats.stages_all, xcd_start, xs_sweep, xs_plain);
            const int sweep_grid = 8 * (SW_OCC / 2) * std::max(1, ctx->n_cu / 8);
            match_sweep<<<sweep_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, bset_stride, c_scale, A.blkcl, ma_pad, mb_pad, rg_rows, mask, n_cc,
                                                              item_rb, ilist, xs_sweep, xcd_ctr, ca, ca.u_colv ? ucol16 : nullptr, kept, kept_count, kept_cap);
            if (init_touched) LGR_TRY(init_touched());
            const int tiles_grid = 8 * std::max(1, ctx->n_cu);
            if (both) match_tiles<true><<<tiles_grid, 64 * TL_WAVES, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, bset_stride, out_scale, A.blkcl, ma_pad, mb_pad,
                                                                                       rg_rows, tile_group, rowmin, colmin, kept, kept_count, kept_cap);
            else match_tiles<false><<<tiles_grid, 64 * TL_WAVES, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, bset_stride, out_scale, A.blkcl, ma_pad, mb_pad,
                                                                                   rg_rows, tile_group, rowmin, colmin, kept, kept_count, kept_cap);
            {
                const CoarseArgs none{};
                const CoarseArgs& ca = none;
                int* const xcd_start_real = xcd_start;
                int* xcd_start = xs_plain;   // (LGR_MFMA_ARGS names xcd_start and ca)
                (void) xcd_start_real;
                if (both) match_mfma<true, FMT_F16R, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
                else match_mfma<false, FMT_F16R, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
            }
        } else if (f16 && rot && ca.u_rb) {
            if (both) match_mfma<true, FMT_F16R, true><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
            else match_mfma<false, FMT_F16R, true><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
        } else if (f16 && rot) {
            if (both) match_mfma<true, FMT_F16R, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
            else match_mfma<false, FMT_F16R, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
        } else if (f16) {
            if (both) match_mfma<true, FMT_F16, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
            else match_mfma<false, FMT_F16, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const f16x8*) Aop, (const f16x8*) Bop, LGR_MFMA_ARGS);
        } else {
            if (both) match_mfma<true, FMT_F32, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const float*) Aop, (const float*) Bop, LGR_MFMA_ARGS);
            else match_mfma<false, FMT_F32, false><<<mfma_grid, NTHR, 0, ctx->stream>>>((const float*) Aop, (const float*) Bop, LGR_MFMA_ARGS);
        }
#undef LGR_MFMA_ARGS
        (void) hipEventRecord(ctx->ev[10 + 2 * ctx->mfma_timed], ctx->stream);
        ctx->mfma_timed += 1;
        LGR_HIP(ctx, hipGetLastError());
#ifdef EXP_PROF
        {
            LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            unsigned long long hp[16];
            (void) hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_prof), sizeof hp);
            int hx[9];
            (void) hipMemcpy(hx, xcd_start, sizeof hx, hipMemcpyDeviceToHost);
            fprintf(stderr, "[lgr] prof launch %d (10 ns ticks): prologue %llu stages %llu (barrier+dma wait %llu, - %llu) colflush %llu wg_total %llu | wgs %llu visits %llu rowflush %llu | items %d (per xcd %d %d %d %d %d %d %d %d)\n",
                    ctx->mfma_timed - 1, hp[0], hp[1], hp[2], hp[5], hp[3], hp[4], hp[8], hp[9], hp[10], hx[8], hx[1] - hx[0], hx[2] - hx[1], hx[3] - hx[2],
                    hx[4] - hx[3], hx[5] - hx[4], hx[6] - hx[5], hx[7] - hx[6], hx[8] - hx[7]);
            unsigned long long z[16] = {0};
            (void) hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z);
        }
#endif
        return LGR_OK;
    };
    g_last_stats.stages_all = (double) n_rb * n_stage_total;
    CompView comp_rows{nullptr, 0, nullptr}, comp_cols{nullptr, 0, nullptr};
    const uint8_t *chk_done = nullptr, *chk_sched = nullptr;
    const float* chk_lb = nullptr;
    const unsigned* chk_ustage = nullptr;
    const float* chk_uq_rows = nullptr;
    const float* chk_uq_cols = nullptr;
    if (!prune) {
        LGR_TRY(launch_sorted_copies(true));
        LGR_TRY(launch_mfma(nullptr, CoarseArgs{}));
        g_last_stats.stages_done = g_last_stats.stages_unique = g_last_stats.stages_all;
    } else {
        // section 3b: lower bounds, pass 1 (nearest tiles), upper bounds, pass 2 (everything the bounds cannot exclude)
        float* LBsq = (float*) (pb + o_lb);
        uint8_t* done = (uint8_t*) (pb + o_done);
        uint8_t* sched = (uint8_t*) (pb + o_sched);
        uint8_t* touched = (uint8_t*) (pb + o_touched);
        const uint8_t* sched_final = sched;   // what the final pass left computed: its schedule, or -- sweep + listed tiles -- the pairs the list touched
        unsigned* mask = (unsigned*) (pb + o_mask);
        float* u_rb = (float*) (pb + o_urb);
        float* u_rt = (float*) (pb + o_urt);
        float* u_row = (float*) (pb + o_urow);   // every row's own upper bound (the sweep's per-row thresholds)
        float* u_colv = (float*) (pb + o_ucolv); // ... and every column's (the per-element re-test of the tiles the sweep keeps)
        unsigned* u_leaf = (unsigned*) (pb + o_ul);
        MaskStats* mstats = (MaskStats*) (pb + o_stats);
        unsigned* u_stage = (unsigned*) (pb + o_ust);
        unsigned* u_ct = (unsigned*) (pb + o_uct);
        // ball bounds: on the matrix cores from the packed operands (f16 formats; the leaf centres were packed as train rows ahead of the operand
        // packing, on the third stream), with packed FMAs from the sorted rows otherwise
        if (early_b) LGR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[27], 0));   // the cleared workspace, the packed centres
        auto launch_lb = [&](lgr_ctx* cx) -> int {
            if (!f16) {
                lb_kernel<<<n_rb, 256, 0, cx->stream>>>(sortedA, A.perm, cen2, B.r2max, B.leaf_count, n_leaves, LBsq);
            } else {
                float* nC = (float*) (pb + o_cnrm);
                f16x8* Cop = (f16x8*) (pb + o_cop);
                if (rot) lb_mfma_kernel<OpFmt<FMT_F16R>::KS><<<n_rb, LBM_THREADS, 0, cx->stream>>>((const f16x8*) Aop, Cop, cset_stride, out_scale, A.blkcl, nAp, nC, ex,
                                                                                                  B.r2max, B.leaf_count, n_leaves, n_cpad, LBsq);
                else lb_mfma_kernel<OpFmt<FMT_F16>::KS><<<n_rb, LBM_THREADS, 0, cx->stream>>>((const f16x8*) Aop, Cop, cset_stride, out_scale, A.blkcl, nAp, nC, ex,
                                                                                              B.r2max, B.leaf_count, n_leaves, n_cpad, LBsq);
            }
            LGR_HIP(cx, hipGetLastError());
            return (int) LGR_OK;
        };
        if (!f16) LGR_TRY(join_sorted());   // lb_kernel reads sortedA
        LGR_TRY(launch_lb(ctx));
        const bool colstage = both && mo.column_stage != 0;
        // coarse rejection inside match_mfma (rotated format, passes with upper bounds): thresholds from u_rb / u_stage
        const bool coarse = f16 && rot && mo.coarse_rejection != 0;
        float* smaxB = (float*) (pb + o_smax);
        unsigned long long* coarse_cnt = (unsigned long long*) (pb + o_ccnt);
        CoarseArgs ca_on{};
        if (coarse) {
            // stage shells and stage maxima from the tiles' shells (behind the packing, on its stream)
            shell_reduce_kernel<<<cdiv((long long) KCL * n_stage_total, 256), 256, 0, sB>>>(shellB, KCL, tb, STAGE_TILES, (float2*) (pb + o_ssh), smaxB);
            LGR_HIP(ctx, hipMemsetAsync(coarse_cnt, 0, 32, ctx->stream));
            const double c_quad = 9.5367477e-6 * (double) ex.quad * 1.00001;            // eps (group_eps)
            const double d11 = std::ldexp(1.0, -11) * (1.0 + std::ldexp(1.0, -9));      // delta: 2^-11 (x^2 + y^2) + 2^-10 x y
            ca_on.xmax = gmaxA; ca_on.ymax = smaxB; ca_on.n_stage_total = n_stage_total;
            ca_on.quad = (float) ((c_quad + d11) * 1.000001);                          // 2^-11 (x^2 + y^2) = 2^-11 (x + y)^2 - 2^-10 x y
            ca_on.cross = (float) (2.0 * d11 * 1.000001);                               // 2^-9 x y (a1.b2 and a2.b1) - 2^-10 x y
            ca_on.lin = (float) (2.0 * (double) ex.lin * 1.00001 + 1e-30);
            ca_on.abs = (float) (((double) ex.abs * 1.00001 + 2.0 * (double) sc.a_norm[0] * std::ldexp(1.0, -25) * (double) sc.inv_s2) * 1.000001 + 1e-12);
            ca_on.cnt = coarse_cnt;
            chk_uq_rows = u_row; chk_uq_cols = both ? u_colv : nullptr;
        }
        // shell bound of the masked passes that have upper bounds (with the coarse rejection: the same "an entry may miss what lies above
        // the U^2 of its row and column" contract, and the same upper-bound tables)
        ShellArgs shell{};
        if (coarse && mo.shell_bound != 0) {
            shell_reduce_kernel<<<cdiv(n_rb, 256), 256, 0, ctx->stream>>>(shellA, 1, ta, BLOCK_ROWS / TILE, (float2*) (pb + o_rsh), nullptr);
            shell.rshA = (const float2*) (pb + o_rsh); shell.sshB = (const float2*) (pb + o_ssh);
            shell.blkcl = A.blkcl; shell.u_rb = u_rb; shell.cols = both ? 1 : 0;
            ca_on.rt_shell = shellA; ca_on.ct_shell = shellB;   // ... and per tile for the test inside the coarse sweep
        }
        ShellArgs shell0 = shell;   // pass 0: the stages of overlapping shells only
        shell0.u_rb = nullptr;
        uint8_t* comp_r = (uint8_t*) (pb + o_cr);
        uint8_t* comp_c = (uint8_t*) (pb + o_cc);
        auto build_comp = [&]() {
            comp_rows_kernel<<<cdiv((long long) n_rb * n_groups, 256), 256, 0, ctx->stream>>>(done, sched_final, group_leaf, n_rb, n_leaves, n_groups, comp_r);
            if (both) comp_cols_kernel<<<cdiv((long long) n_leaves * n_rg, 256), 256, 0, ctx->stream>>>(done, sched_final, n_rb, n_leaves, n_rg, rg_rows / BLOCK_ROWS, comp_c);
        };
        chk_done = done; chk_sched = sched;   // (chk_sched: re-pointed to the touched pairs below when the final pass defers its initialisation) chk_lb = LBsq; chk_ustage = colstage ? u_stage : nullptr;
        comp_rows = CompView{comp_r, n_groups, nullptr};
        comp_cols = CompView{comp_c, n_rg, tile_leaf};
        // do the bounds separate anything?  (zero / finite lower bounds: counted by box_lb_kernel where it writes the final bounds, by lb_stats_kernel
        // without boxes; near_kernel: when nearly every lower bound is zero, pass 0 takes everything)
        unsigned long long* lbstat = &mstats->stages[5];   // [5] zero, [6] finite lower bounds (MaskStats slots the passes do not use)
        uint2* lb_part = boxes ? (uint2*) (pb + o_lbpart) : nullptr;   // (box_lb_kernel's counts per row block; summed by near_kernel)
        if (boxes) {
            LGR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[30], 0));
            box_lb_kernel<<<n_rb, 256, 0, ctx->stream>>>(boxA, boxBt, n_leaves, rmax2, LBsq, lb_part);
        } else {
            lb_stats_kernel<<<std::min(cdiv((long long) n_rb * n_leaves, 1024), 1024), 256, 0, ctx->stream>>>(LBsq, (size_t) n_rb * n_leaves, lbstat);
        }
        // pass 0: the NEAR_T nearest leaves of every row block and the NEAR_T nearest row blocks of every leaf
        const float widen_frac = mo.auto_dense ? LGR_AUTO_DENSE_FRAC : 0.f;
        auto launch_near = [&](int n_vec, int len, size_t vs, size_t es) -> int {
            if (len <= NEAR_LDS_MAX) {
                if ((size_t) len * 4 > 64 * 1024)
                    LGR_HIP(ctx, hipFuncSetAttribute((const void*) near_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, len * 4));
                near_kernel<true><<<n_vec, NEAR_THREADS, (size_t) len * 4, ctx->stream>>>(near_t, LBsq, n_vec, len, vs, es, sched, vs, es, lbstat, lb_part, n_rb, widen_frac);
            } else {
                near_kernel<false><<<n_vec, NEAR_THREADS, 0, ctx->stream>>>(near_t, LBsq, n_vec, len, vs, es, sched, vs, es, lbstat, lb_part, n_rb, widen_frac);
            }
            return LGR_OK;
        };
        LGR_TRY(launch_near(n_rb, n_leaves, (size_t) n_leaves, 1));
        LGR_TRY(launch_near(n_leaves, n_rb, 1, (size_t) n_leaves));
        // passes 1..: tiles within beta * U of the bounds known so far; the last pass (beta = 1) takes everything the
        // bounds cannot exclude
        static const float betas[] = {LGR_PRUNE_BETAS};
        const int n_beta = (int) (sizeof betas / sizeof betas[0]);
        // (an intermediate sweeping pass whose tile list overflows would go unrepaired: the repair below re-runs the LAST pass's mask only, and
        //  every launch resets the list's counter -- seen as wrong matches with -DLGR_PRUNE_BETAS=0.5f,1.0f at pass-0 widths 6 and 12)
        static_assert(sizeof betas / sizeof betas[0] == 1, "LGR_PRUNE_BETAS: one final pass only (lgr_match_common.cuh)");
        LGR_TRY(record_b());   // sB: the column operands, their maxima and shells, the rerank's tables -- everything a reader of join_b() waits for
        for (int pass = 0; pass <= n_beta; ++pass) {
            if (pass > 0) {
                build_comp();
                row_u_kernel<<<n_rb, BLOCK_ROWS, (size_t) (n_groups + 8) * 4, ctx->stream>>>((const float*) rowmin, n_groups, ma_pad, A.perm, nAp, A.blkcl, gmaxB, ex, comp_rows, u_rb, u_rt, coarse ? u_row : nullptr);
                if (both) {
                    LGR_HIP(ctx, hipMemsetAsync(u_leaf, 0, (size_t) MAXLEAF * 4, ctx->stream));
                    LGR_HIP(ctx, hipMemsetAsync(u_stage, 0, (size_t) n_stage_total * 4, ctx->stream));
                    col_u_kernel<<<cdiv(mb_pad, 256), 256, (size_t) (n_rg + 8) * 4, ctx->stream>>>((const float*) colmin, n_rg, mb_pad, B.perm, nBp, gmaxA, cl_of_rg, tile_leaf, ex, comp_cols, u_leaf,
                                                                                                   (colstage || coarse) ? u_stage : nullptr, coarse ? u_ct : nullptr, coarse ? u_colv : nullptr);
                }
                float bsq = betas[pass - 1] * betas[pass - 1];
                sched_kernel<<<cdiv((long long) n_rb * n_leaves, 256), 256, 0, ctx->stream>>>(both ? 1 : 0, bsq, LBsq, u_rb, u_leaf, n_rb, n_leaves,
                                                                                            colstage && pass == n_beta ? 1 : 0, pass == 1 && shell0.rshA ? 1 : 0, done, sched);
            }
            // the table initialisation of the pass needs the schedule only: on the third stream, beside mask_kernel (0.27 + 0.28 ms in a row
            // in front of pass 0, 0.30 + 0.37 between the passes); the MFMA launch waits for both
            // (the final pass as sweep + listed tiles: initialised behind the sweep, for the pairs its list touches -- touched_kernel)
            const bool defer_init = pass == n_beta && pass > 0 && f16 && rot && coarse && kept != nullptr;
            const bool init_aside = sB != ctx->stream && !defer_init;
            if (init_aside) {
                LGR_HIP(ctx, hipEventRecord(ctx->ev[28], ctx->stream));
                LGR_HIP(ctx, hipStreamWaitEvent(sB, ctx->ev[28], 0));
            }
            if (!defer_init) {
                init_tables_sparse_kernel<<<cdiv((long long) n_rb * n_leaves, 256), 256, 0, init_aside ? sB : ctx->stream>>>(sched, done, n_rb, n_leaves, leaf_g0, group_start, rg_rows / BLOCK_ROWS, rowmin,
                                                                                                        (size_t) ma_pad, colmin, (size_t) mb_pad);
                init_touched = nullptr;
            } else {
                sched_final = touched;
                init_touched = [&]() -> int {   // (called by launch_mfma between match_sweep and match_tiles, on the context's stream)
                    LGR_HIP(ctx, hipMemsetAsync(touched, 0, (size_t) n_rb * n_leaves, ctx->stream));
                    touched_kernel<<<4 * std::max(1, ctx->n_cu), 256, 0, ctx->stream>>>(kept, kept_count, kept_cap, xcd_start + 48, sched, tile_leaf, n_leaves, (size_t) n_rb * n_leaves, touched);
                    init_tables_sparse_kernel<<<cdiv((long long) n_rb * n_leaves, 256), 256, 0, ctx->stream>>>(touched, done, n_rb, n_leaves, leaf_g0, group_start, rg_rows / BLOCK_ROWS, rowmin,
                                                                                               (size_t) ma_pad, colmin, (size_t) mb_pad);
                    return LGR_OK;
                };
            }
            if (init_aside) LGR_HIP(ctx, hipEventRecord(ctx->ev[29], sB));
            if (pass == 0 && shell0.rshA) LGR_TRY(join_b());   // (the stage shells come from the column norms, written by the packing)
            {
                // the pass's stage masks from the scheduled (row block, leaf) pairs (mask_sparse_kernel; 0.41 + 0.30 -> 2 x ~0.1 ms at 1M)
                int* lfst = (int*) (pb + o_lfst);
                int* llst = (int*) (pb + o_llst);
                if (pass == 0) {
                    LGR_HIP(ctx, hipMemsetAsync(lfst, 0x7f, (size_t) MAXLEAF * 4, ctx->stream));
                    LGR_HIP(ctx, hipMemsetAsync(llst, 0xff, (size_t) MAXLEAF * 4, ctx->stream));
                    leaf_stage_range_kernel<<<cdiv(tb, 256), 256, 0, ctx->stream>>>(tile_leaf, tb, n_leaves, lfst, llst);
                }
                const long long n_pairs_m = (long long) n_rb * n_cc;
                LGR_HIP(ctx, hipMemsetAsync(mask, 0, (size_t) n_pairs_m * 4, ctx->stream));
                mask_sparse_kernel<<<cdiv((long long) n_rb * n_leaves, 256), 256, 0, ctx->stream>>>(sched, lfst, llst, n_rb, n_cc, n_leaves, n_stage_total, LBsq, u_stage,
                                                                                                    pass > 0 ? shell : shell0, mask);
                if (mo.self_check) {   // (tests) mask_kernel's masks must be the same words; its bookkeeping goes to scratch
                    unsigned* chk = (unsigned*) (pb + o_mchk);
                    unsigned* chk_acc = chk + n_pairs_m;
                    unsigned* n_diff = chk_acc + n_pairs_m;
                    MaskStats* scratch_stats = (MaskStats*) (pb + o_lbpart);   // (box_lb_kernel's partial counts: read by the near kernels, long done)
                    LGR_HIP(ctx, hipMemsetAsync(chk_acc, 0, (size_t) n_pairs_m * 4 + 64, ctx->stream));
                    mask_kernel<<<std::min(cdiv(n_pairs_m * 32, 256), 4096), 256, 0, ctx->stream>>>(pass, sched, tile_leaf, n_rb, n_cc, n_leaves, n_stage_total, LBsq, u_stage,
                                                                                                     pass > 0 ? shell : shell0, chk, chk_acc, scratch_stats);
                    mask_compare_kernel<<<cdiv(n_pairs_m, 256), 256, 0, ctx->stream>>>(mask, chk, n_pairs_m, n_diff);
                    unsigned h_diff = 0;
                    LGR_HIP(ctx, hipMemcpyAsync(&h_diff, n_diff, 4, hipMemcpyDeviceToHost, ctx->stream));
                    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if (h_diff) { ctx->err = "matcher self-check: the sparse stage masks differ from mask_kernel's"; return LGR_ERR_HIP; }
                }
                mask_stats_kernel<<<std::min(cdiv(n_pairs_m, 256), 256), 256, 0, ctx->stream>>>(pass, mask, (unsigned*) (pb + o_macc), n_pairs_m, mstats);
            }
            if (init_aside) LGR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[29], 0));
            CoarseArgs ca = ca_on;
            if (coarse && pass > 0) { ca.u_rb = u_rb; ca.u_rt = u_rt; ca.u_row = u_row; ca.u_stage = both ? u_stage : nullptr; ca.u_ct = u_ct; ca.u_colv = both ? u_colv : nullptr; ca.n_ct_total = tb; }
            cur_pass_stages = &mstats->stages[pass];
            if (pass == 0) LGR_TRY(launch_sorted_copies(true));
            LGR_TRY(launch_mfma(mask, ca));
        }
        build_comp();   // final state for the rerank scans
        chk_sched = sched_final;
        init_touched = nullptr;   // (its captures end with this block)
        MaskStats* hs;
        LGR_TRY(lgr_pinned(ctx, 256, (void**) &hs));
        LGR_HIP(ctx, hipMemcpyAsync(hs, mstats, sizeof(MaskStats), hipMemcpyDeviceToHost, ctx->stream));
        unsigned long long* h_cc = (unsigned long long*) ((char*) hs + 128);
        h_cc[0] = h_cc[1] = h_cc[2] = 0ull;
        if (coarse) LGR_HIP(ctx, hipMemcpyAsync(h_cc, coarse_cnt, 24, hipMemcpyDeviceToHost, ctx->stream));
        unsigned long long* h_kept = (unsigned long long*) ((char*) hs + 192);
        h_kept[0] = 0ull;
        if (split_used) LGR_HIP(ctx, hipMemcpyAsync(h_kept, kept_count, 8, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (split_used && h_kept[0] > (unsigned long long) kept_cap) {
            // The sweep kept more tiles than the list holds (descriptors without structure): the last pass again on the fused kernel.  The
            // tables only ever take minima, so what the listed tiles already contributed stays valid.  (Statistics: the fused launch's.)
            LGR_HIP(ctx, hipMemsetAsync(coarse_cnt, 0, 32, ctx->stream));
            CoarseArgs ca = ca_on;
            ca.u_rb = u_rb; ca.u_rt = u_rt; ca.u_row = u_row; ca.u_stage = both ? u_stage : nullptr; ca.u_ct = u_ct; ca.u_colv = both ? u_colv : nullptr; ca.n_ct_total = tb;
            // (touched_kernel saw the overflow: it marked, and init_tables_sparse_kernel initialised, every scheduled pair -- the fused kernel finds its tables ready)
            LGR_TRY(launch_mfma(mask, ca, false));
            LGR_HIP(ctx, hipMemcpyAsync(h_cc, coarse_cnt, 24, hipMemcpyDeviceToHost, ctx->stream));
            LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        g_last_stats.coarse_tested = (double) h_cc[0];
        g_last_stats.coarse_rejected = (double) h_cc[1];
        g_last_stats.shell_skipped = (double) h_cc[2];
        static_assert(sizeof betas / sizeof betas[0] + 1 <= 7, "MaskStats: slot 7 holds the stages counted once");
        g_last_stats.stages_done = 0;
        for (int k = 0; k <= n_beta; ++k) g_last_stats.stages_done += (double) hs->stages[k];
        g_last_stats.stages_unique = (double) hs->stages[7];
        g_last_stats.lb_zero = (double) hs->stages[5]; g_last_stats.lb_finite = (double) hs->stages[6];
        if (env_int("LGR_MATCH_DEBUG", 0)) {
            fprintf(stderr, "[lgr] stages per pass:");
            for (int k = 0; k <= n_beta; ++k) fprintf(stderr, " %llu", hs->stages[k]);
            fprintf(stderr, " of %.0f (n_rb %d n_cc %d item_rb %d leaves %d)\n", g_last_stats.stages_all, n_rb, n_cc, item_rb, n_leaves);
            // what the schedule asks for at leaf granularity (the stages computed above also cover the neighbours' boundary tiles)
            std::vector<uint8_t> hd((size_t) n_rb * n_leaves), hsch((size_t) n_rb * n_leaves);
            LGR_HIP(ctx, hipMemcpy(hd.data(), done, hd.size(), hipMemcpyDeviceToHost));
            LGR_HIP(ctx, hipMemcpy(hsch.data(), sched, hsch.size(), hipMemcpyDeviceToHost));
            double need_cols = 0;
            for (int rb = 0; rb < n_rb; ++rb)
                for (int l = 0; l < n_leaves; ++l)
                    if (hd[(size_t) rb * n_leaves + l] | hsch[(size_t) rb * n_leaves + l]) need_cols += B.h_leaf_start[l + 1] - B.h_leaf_start[l];
            fprintf(stderr, "[lgr] scheduled (row block, leaf) pairs cover %.4f of the tiles; computed stages %.4f\n",
                    need_cols / ((double) n_rb * mb_pad), g_last_stats.stages_done / g_last_stats.stages_all);
            // which criterion asked for the final-pass tiles (hsch = the last pass): the block's rows, the leaf's columns, or both
            std::vector<float> hlb((size_t) n_rb * n_leaves), hurb(n_rb);
            std::vector<unsigned> hul(MAXLEAF);
            LGR_HIP(ctx, hipMemcpy(hlb.data(), LBsq, hlb.size() * 4, hipMemcpyDeviceToHost));
            LGR_HIP(ctx, hipMemcpy(hurb.data(), u_rb, hurb.size() * 4, hipMemcpyDeviceToHost));
            LGR_HIP(ctx, hipMemcpy(hul.data(), u_leaf, hul.size() * 4, hipMemcpyDeviceToHost));
            double by_rows = 0, by_cols = 0, by_both = 0;
            for (int rb = 0; rb < n_rb; ++rb)
                for (int l = 0; l < n_leaves; ++l) {
                    if (!hsch[(size_t) rb * n_leaves + l]) continue;
                    float lb = hlb[(size_t) rb * n_leaves + l], ug;
                    memcpy(&ug, &hul[l], 4);
                    bool r = hurb[rb] >= 0.f && lb <= hurb[rb] * 1.00001f + 1e-12f, c = lb <= ug * 1.00001f + 1e-12f;
                    double w = B.h_leaf_start[l + 1] - B.h_leaf_start[l];
                    (r && c ? by_both : r ? by_rows : by_cols) += w;
                }
            const double tot = (double) n_rb * mb_pad;
            fprintf(stderr, "[lgr] final pass by criterion: rows only %.4f, columns only %.4f, both %.4f of the tiles\n", by_rows / tot, by_cols / tot, by_both / tot);
            if (env_int("LGR_MATCH_DEBUG", 0) >= 2) {
                // how full are the sweep's VISITS?  A visit = one row block against one 128-tile column chunk: the stages its mask holds, of 32.  Every visit pays
                // the A fragments, the thresholds of its tile slots, the start of the DMA ring and two barriers before its first MFMA.
                std::vector<unsigned> hm((size_t) n_rb * n_cc);
                LGR_HIP(ctx, hipMemcpy(hm.data(), mask, hm.size() * 4, hipMemcpyDeviceToHost));
                double visits = 0, stages_ = 0, hist[6] = {0, 0, 0, 0, 0, 0}, runs = 0;   // visits holding 1-2, 3-4, 5-8, 9-16, 17-24, 25-32 stages
                for (unsigned m : hm) {
                    if (!m) continue;
                    const int n = __builtin_popcount(m);
                    visits += 1; stages_ += n; runs += __builtin_popcount(m & ~(m << 1));
                    hist[n <= 2 ? 0 : n <= 4 ? 1 : n <= 8 ? 2 : n <= 16 ? 3 : n <= 24 ? 4 : 5] += 1;
                }
                fprintf(stderr, "[lgr] last pass: %.0f visits (row block x chunk) of %zu, %.2f stages per visit in %.2f runs; visits by stages 1-2: %.3g, 3-4: %.3g, 5-8: %.3g, 9-16: %.3g, 17-24: %.3g, 25-32: %.3g\n",
                        visits, hm.size(), stages_ / std::max(visits, 1.0), runs / std::max(visits, 1.0), hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
            }
            if (split_used && env_int("LGR_MATCH_DEBUG", 0) >= 2 && h_kept[0] <= (unsigned long long) kept_cap) {
                // how are the tiles the sweep keeps distributed over the (row block, stage) pairs -- 32 tile slots each?  (Round 5, 900 k points, scene
                // seed 571: 16 M kept tiles, 78 % of them in pairs that keep more than half of their slots -- blobs of near-duplicate descriptors.
                // Flagging such stages from the list and giving them to the plain six-step kernel as a whole: 40.8 -> 38.0 ms for that scene, nothing
                // for the others; what those blobs needed was pass 0 taking every zero lower bound (near_kernel): 26.1 ms, 0.6 M kept tiles.)
                const size_t nk = (size_t) h_kept[0];
                std::vector<uint2> hk(nk);
                if (nk) LGR_HIP(ctx, hipMemcpy(hk.data(), kept, nk * sizeof(uint2), hipMemcpyDeviceToHost));
                std::vector<unsigned long long> key(nk);
                for (size_t i = 0; i < nk; ++i) key[i] = ((unsigned long long) (hk[i].x / (BLOCK_ROWS / TILE)) << 32) | (hk[i].y / STAGE_TILES);
                std::sort(key.begin(), key.end());
                double hist[6] = {0, 0, 0, 0, 0, 0}, pairs_ = 0;   // kept tiles in pairs holding 1-2, 3-4, 5-8, 9-16, 17-24, 25-32 of them
                for (size_t i = 0; i < nk;) {
                    size_t j = i;
                    while (j < nk && key[j] == key[i]) ++j;
                    const size_t n = j - i;
                    hist[n <= 2 ? 0 : n <= 4 ? 1 : n <= 8 ? 2 : n <= 16 ? 3 : n <= 24 ? 4 : 5] += (double) n;
                    pairs_ += 1; i = j;
                }
                fprintf(stderr, "[lgr] kept tiles %zu in %.0f (row block, stage) pairs; tiles by the pair's count 1-2: %.3g, 3-4: %.3g, 5-8: %.3g, 9-16: %.3g, 17-24: %.3g, 25-32: %.3g\n",
                        nk, pairs_, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
            }
            if (coarse && both && env_int("LGR_MATCH_DEBUG", 0) >= 2) {
                // What would homogeneous tiles be worth?  (Round 5, bench pair: scheduled pairs 1.57e8 tiles, tile maxima in the present order 1.03e8,
                // element level 6.8e7 -- rows and columns sorted by U would lose the radial shells, which take 88 M tile slots to 54 M tested, for at
                // most a third fewer; and the leaf's bound tested per tile inside the sweep took 54.4 M tested tiles to 47.4 M for 0.1 ms and four
                // spilled VGPRs: the (row block, leaf) bounds themselves are what limits the final pass, not the granularity of the upper bounds.)  32 x 32 tiles of the scheduled (row block, leaf) pairs that ANY element-level criterion
                // needs (LB^2 <= U^2 of the row or of the column), counted (i) per (row block, leaf) as scheduled, (ii) per tile with the present
                // row / column order (tile maxima), (iii) as if rows and columns were sorted by U inside their block / leaf (the fraction of
                // rows and of columns that need the pair).
                std::vector<float> hur(ma_pad), huc(mb_pad);
                LGR_HIP(ctx, hipMemcpy(hur.data(), u_row, hur.size() * 4, hipMemcpyDeviceToHost));
                LGR_HIP(ctx, hipMemcpy(huc.data(), u_colv, huc.size() * 4, hipMemcpyDeviceToHost));
                std::vector<std::vector<float>> rs(n_rb), cs(n_leaves), rts(n_rb), cts(n_leaves);
                for (int rb = 0; rb < n_rb; ++rb) {
                    for (int r = 0; r < BLOCK_ROWS; ++r) rs[rb].push_back(std::max(hur[(size_t) rb * BLOCK_ROWS + r], 0.f));
                    for (int t = 0; t < BLOCK_ROWS / TILE; ++t) rts[rb].push_back(*std::max_element(rs[rb].begin() + t * TILE, rs[rb].begin() + (t + 1) * TILE));
                    std::sort(rs[rb].begin(), rs[rb].end()); std::sort(rts[rb].begin(), rts[rb].end());
                }
                for (int l = 0; l < n_leaves; ++l) {
                    for (int c = B.h_leaf_start[l]; c < B.h_leaf_start[l + 1]; ++c) cs[l].push_back(std::max(huc[c], 0.f));
                    for (size_t t = 0; t + TILE <= cs[l].size(); t += TILE) cts[l].push_back(*std::max_element(cs[l].begin() + t, cs[l].begin() + t + TILE));
                    std::sort(cs[l].begin(), cs[l].end()); std::sort(cts[l].begin(), cts[l].end());
                }
                auto frac_ge = [](const std::vector<float>& v, float x) { return v.empty() ? 0.0 : (double) (v.end() - std::lower_bound(v.begin(), v.end(), x)) / (double) v.size(); };
                double t_sched = 0, t_tile = 0, t_ideal = 0, t_rows_ideal = 0, t_cols_ideal = 0;
                for (int rb = 0; rb < n_rb; ++rb)
                    for (int l = 0; l < n_leaves; ++l) {
                        if (!hsch[(size_t) rb * n_leaves + l]) continue;
                        const float lb = hlb[(size_t) rb * n_leaves + l] / 1.00001f;
                        const double tiles = 8.0 * (double) cts[l].size();
                        const double pa = frac_ge(rs[rb], lb), pb = frac_ge(cs[l], lb), ta_ = frac_ge(rts[rb], lb), tb_ = frac_ge(cts[l], lb);
                        t_sched += tiles;
                        t_tile += tiles * (1.0 - (1.0 - ta_) * (1.0 - tb_));
                        t_ideal += tiles * (1.0 - (1.0 - pa) * (1.0 - pb));
                        t_rows_ideal += tiles * pa; t_cols_ideal += tiles * pb;
                    }
                fprintf(stderr, "[lgr] final pass, 32 x 32 tiles by granularity of the criterion: scheduled pairs %.3g, tile maxima (present order) %.3g, element level (U-sorted tiles) %.3g "
                                "(rows alone %.3g, columns alone %.3g)\n", t_sched, t_tile, t_ideal, t_rows_ideal, t_cols_ideal);
            }
            if (coarse) {
                // how loose are the tile-level maxima the sweep's shell test uses?  quantiles of the rows' own bounds and of (tile max / tile median)
                std::vector<float> hu(ma_pad);
                LGR_HIP(ctx, hipMemcpy(hu.data(), u_row, hu.size() * 4, hipMemcpyDeviceToHost));
                std::vector<float> all, ratio;
                for (int t = 0; t < ma_pad / TILE; ++t) {
                    std::vector<float> v;
                    for (int r = 0; r < TILE; ++r) if (hu[(size_t) t * TILE + r] > 0.f) v.push_back(hu[(size_t) t * TILE + r]);
                    if (v.size() < 8) continue;
                    std::sort(v.begin(), v.end());
                    ratio.push_back(v.back() / v[v.size() / 2]);
                    all.insert(all.end(), v.begin(), v.end());
                }
                std::sort(all.begin(), all.end()); std::sort(ratio.begin(), ratio.end());
                auto q = [](const std::vector<float>& v, double f) { return v.empty() ? 0.f : v[(size_t) (f * (v.size() - 1))]; };
                fprintf(stderr, "[lgr] row bounds U^2: q10 %.3g q50 %.3g q90 %.3g q99 %.3g max %.3g; tile max / tile median: q10 %.2f q50 %.2f q90 %.2f q99 %.2f\n",
                        q(all, 0.1), q(all, 0.5), q(all, 0.9), q(all, 0.99), q(all, 1.0), q(ratio, 0.1), q(ratio, 0.5), q(ratio, 0.9), q(ratio, 0.99));
            }
        }
    }
    LGR_HIP(ctx, hipGetLastError());

    LGR_TRY(join_sorted());   // the self-check and the exact rerank read the sorted rows
    // irregular rows: every pair they are part of, both roles (irregular_scan); the exact rerank below never sees them
    {
        const int nb_a = (ma + block - 1) / block, nb_b = (mb + block - 1) / block;
        if (B.n_irr) irregular_scan<<<cdiv(ma, 256), 256, 0, ctx->stream>>>(d_a, A.valid, ma, d_b, B.irr_list, B.n_irr, block, nb_a, nb_b, bestA, both ? bestB : nullptr);
        if (A.n_irr) irregular_scan<<<cdiv(mb, 256), 256, 0, ctx->stream>>>(d_b, B.valid, mb, d_a, A.irr_list, A.n_irr, block, nb_b, nb_a, both ? bestB : nullptr, bestA);
    }
    if (mo.self_check && sortedA) {
        unsigned* d_worst = (unsigned*) (misc + 192);
        LGR_HIP(ctx, hipMemsetAsync(d_worst, 0, 8, ctx->stream));
        const int stride = 37;
        check_kernel<true><<<cdiv(ma_pad, stride), 256, (size_t) (n_groups + 8) * 4, ctx->stream>>>(
            (const float*) rowmin, n_groups, ma_pad, 0, group_start, sortedA, A.perm, sortedB, B.perm, mb_pad, nAp, A.blkcl, nullptr, gmaxB, nullptr,
            ex, comp_rows, stride, nullptr, nullptr, 0, nullptr, nullptr, chk_uq_rows, chk_uq_cols, d_worst);
        if (both)
            check_kernel<false><<<cdiv(mb_pad, stride), 256, (size_t) (n_rg + 8) * 4, ctx->stream>>>(
                (const float*) colmin, n_rg, mb_pad, rg_rows, nullptr, sortedB, B.perm, sortedA, A.perm, ma_pad, nullptr, nullptr, nBp, gmaxA, cl_of_rg,
                ex, comp_cols, stride, chk_done, chk_sched, n_leaves, chk_lb, chk_ustage, chk_uq_rows, chk_uq_cols, d_worst + 1);
        unsigned* hw;
        LGR_TRY(lgr_pinned(ctx, 64, (void**) &hw));
        LGR_HIP(ctx, hipMemcpyAsync(hw, d_worst, 8, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float r0, r1;
        memcpy(&r0, hw, 4); memcpy(&r1, hw + 1, 4);
        g_last_check[0] = r0; g_last_check[1] = r1;
    }

    // ---- 5. exact rerank
    // (the MFMA re-filter of the rerank items needs the f16 operand formats and the padded train copies)
    RefilterArgs ra{(const f16x8*) Aop, (const f16x8*) Bop, bset_stride, out_scale, (f16 && mo.rerank_refilter) ? KS : 0, A.blkcl, mo.pair_cap};
    lgr_match_stats* st = &g_last_stats;   // (in the context: the helper thread of the other direction writes its own fields)
    auto rerank_ab = [&](lgr_ctx* cx) {
        return run_rerank<true>(cx, ex, comp_rows, (const float*) rowmin, n_groups, 0, group_start, d_a, A, nAp, nullptr, gmaxB, nullptr, d_b, sortedB, B, block, bestA,
                                d_ab_idx, d_ab_dist, &st->items_ab, &st->dense_ab, force_dense, ra, &st->pairs_ab);
    };
    auto rerank_ba = [&](lgr_ctx* cx) {
        return run_rerank<false>(cx, ex, comp_cols, (const float*) colmin, n_rg, rg_rows, nullptr, d_b, B, nullptr, nBp, gmaxA, cl_of_rg, d_a, sortedA, A, block, bestB,
                                 d_ba_idx, d_ba_dist, &st->items_ba, &st->dense_ba, force_dense, ra, &st->pairs_ba);
    };
    if (both) LGR_TRY(lgr_run_pair(ctx, rerank_ab, rerank_ba));   // the two directions' exact reranks are independent
    else LGR_TRY(rerank_ab(ctx));
    return LGR_OK;
}

// duration of the match_mfma launch(es) of the last match call in ms (hipEvents on the ctx stream); -1 if none
extern "C" int lgr_match_last_kernel_ms(lgr_ctx* ctx, float* ms) {
    if (!ctx || !ms) return LGR_ERR_INVALID_ARG;
    *ms = -1.f;
    if (!ctx->mfma_timed) return LGR_OK;
    *ms = 0.f;
    for (int k = 0; k < ctx->mfma_timed; ++k) {   // one event pair per masked pass
        float t = 0.f;
        LGR_HIP(ctx, hipEventSynchronize(ctx->ev[10 + 2 * k]));
        LGR_HIP(ctx, hipEventElapsedTime(&t, ctx->ev[9 + 2 * k], ctx->ev[10 + 2 * k]));
        if (env_int("LGR_MATCH_DEBUG", 0)) fprintf(stderr, "[lgr] match_mfma pass %d: %.2f ms\n", k, t);
        *ms += t;
    }
    return LGR_OK;
}

extern "C" int lgr_match_bf_dev(lgr_ctx* ctx, const float* d_q33, int mq, const float* d_t33, int mt, int block,
                                int32_t* d_idx, float* d_dist) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    return match_impl(ctx, d_q33, mq, d_t33, mt, block, d_idx, d_dist, nullptr, nullptr);
}

extern "C" int lgr_match_bf2_dev(lgr_ctx* ctx, const float* d_a33, int ma, const float* d_b33, int mb, int block,
                                 int32_t* d_ab_idx, float* d_ab_dist, int32_t* d_ba_idx, float* d_ba_dist) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, (d_ba_idx && d_ba_dist) || mb == 0, LGR_ERR_INVALID_ARG);
    return match_impl(ctx, d_a33, ma, d_b33, mb, block, d_ab_idx, d_ab_dist, d_ba_idx, d_ba_dist);
}

extern "C" int lgr_match_bf(lgr_ctx* ctx, const float* q33, int mq, const float* t33, int mt, int block,
                            int32_t* idx, float* dist) {
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
    LGR_TRY(lgr_match_bf_dev(ctx, dq, mq, dt, mt, block, di, dd));
    if (mq) {
        LGR_HIP(ctx, hipMemcpyAsync(idx, di, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
        LGR_HIP(ctx, hipMemcpyAsync(dist, dd, (size_t) mq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LGR_OK;
}
