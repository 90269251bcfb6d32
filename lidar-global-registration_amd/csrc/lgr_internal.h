// lgr_internal.h -- internals shared by the HIP translation units of liblgr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lgr.h"

#define LGR_WAVE 64
#define LGR_STREAM_OWN_LOW ((void*) (intptr_t) -2)   // internal: an own non-blocking stream of the lowest priority (helper contexts)

struct lgr_buf {
    void* p = nullptr;
    size_t cap = 0;
};

// statistics of a context's last match call (bench / diagnostics: lgr_match_last_*): candidate (query, group) items and
// dense-fallback queries per direction, group counts, the column stages the MFMA passes executed out of all (row block, stage) pairs
struct lgr_match_stats { unsigned items_ab, dense_ab, items_ba, dense_ba; int sub_cols, rg_rows; double stages_done, stages_all, stages_unique; int f16; double coarse_tested, coarse_rejected; unsigned pairs_ab, pairs_ba; double shell_skipped; double lb_zero, lb_finite; unsigned irr_a, irr_b, irr_gave_up; };

// The one persistent helper host thread of an internal context (lgr_ctx::aux / aux2): started on first use, parked on a condition
// variable between jobs, joined when the context is destroyed.  One job at a time: post, then wait.
struct lgr_helper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, busy = false, quit = false, started = false;
    int rc = 0;
};

// Per-thread/GPU context: stream, error string, named workspace buffers (grown on demand, reused across calls).
struct lgr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    lgr_buf ws[112];
    void* pinned = nullptr;  // small pinned host scratch for read-backs
    size_t pinned_cap = 0;
    hipEvent_t ev[32];       // 0..8 stage timers (lgr_align), 9.. pairs around the match_mfma passes
    float stage_ms[12];
    int n_cu = 256;
    int mfma_timed = 0;
    lgr_match_options mopt{-1, 0, 0, -1, 1, 1, 1, 1, -1, 0, 0, 1, 1, 0, 1, 1};   // lgr_match_default_options
    bool corr_trusted = false;                  // set by lgr_align* around its own RANSAC / GROR call: the correspondences came from the pipeline itself
    void* match_prep = nullptr;                 // the matcher's clustering / prepared query side (lgr_match.hip: MatchPrep)
    void (*match_prep_free)(void*) = nullptr;
    lgr_ctx* aux = nullptr;      // second context (own stream + workspace, same device): the target cloud's feature stages run on it
                                 // from a second host thread while this one does the source cloud (lgr_align.hip)
    hipEvent_t aux_ev = nullptr;
    lgr_ctx* aux2 = nullptr;     // third context: the match filter's per-cloud tables (densities, cluster k-NN lists) are computed on it
                                 // while the matcher runs on this one (lgr_correspondences_dev)
    hipEvent_t aux2_ev = nullptr;
    hipStream_t stream3 = nullptr;  // matcher: the column operands are packed on it while the bounds of pass 0 are computed on `stream` (lgr_ctx_stream3)
    hipEvent_t ev3 = nullptr;
    lgr_ctx_options opt{1, 0, LGR_ARITH_FAST, 0, 0, {0, 0, 0}};   // lgr_ctx_default_options
    lgr_helper* helper = nullptr;               // of an internal context: the host thread that drives it (opt.helper_contexts)
    lgr_match_stats mstats{};                   // lgr_match_last_*: the last match call of THIS context
    double mcheck[2] = {-1, -1};
    bool internal = false;                      // an aux / aux2 context: works inside its owner's turn (lgr_turn)
    int turn_depth = 0;                         // public entry points call each other: only the outermost one takes the device's turn
    unsigned long long turn_id = 0;             // identity for the turn hand-over (never reused, unlike the address)
};

// Contexts of ONE device (in one process) take turns: the outermost public entry point on a context holds the device's turn for the
// duration of the call, and the first call of a context after ANOTHER context's makes its stream wait for everything that context had
// queued (an event recorded when the turn was given back).  So the device never executes two contexts' work side by side -- what
// happens INSIDE a context (its helper contexts and streams) is unaffected.  lgr_ctx_options.concurrent_contexts = 1 opts a context
// out (include/lgr.h says when that is advisable).  Internal contexts never take a turn: they work inside their owner's.
struct lgr_turn {
    lgr_ctx* c = nullptr;
    bool held = false;
    explicit lgr_turn(lgr_ctx* ctx);
    ~lgr_turn();
    lgr_turn(const lgr_turn&) = delete;
    lgr_turn& operator=(const lgr_turn&) = delete;
};

int lgr_fail(lgr_ctx* ctx, int code, const char* what, const char* file, int line);

#define LGR_HIP(ctx, call)                                                                 \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            char b__[512];                                                                 \
            snprintf(b__, sizeof b__, "%s -> %s", #call, hipGetErrorString(e__));          \
            return lgr_fail(ctx, e__ == hipErrorOutOfMemory ? LGR_ERR_OOM : LGR_ERR_HIP, b__, __FILE__, __LINE__); \
        }                                                                                  \
    } while (0)

#define LGR_CHECK(ctx, cond, code)                                                         \
    do {                                                                                   \
        if (!(cond)) return lgr_fail(ctx, code, #cond, __FILE__, __LINE__);                \
    } while (0)

#define LGR_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != LGR_OK) return rc__; \
    } while (0)

// workspace slot ids
enum {
    WS_MATCH_AP = 0, WS_MATCH_BP, WS_MATCH_NA, WS_MATCH_NB, WS_MATCH_ROWMIN, WS_MATCH_COLMIN, WS_MATCH_ITEMS,
    WS_MATCH_BEST_A, WS_MATCH_BEST_B, WS_MATCH_MISC, WS_MATCH_DENSE, WS_MATCH_SORTED_A, WS_MATCH_SORTED_B, WS_MATCH_PRUNE, WS_MATCH_ITEMS2, WS_MATCH_NORMS, WS_MATCH_PAIRS, WS_MATCH_KEPT,
    // three independent uniform grids (7 slots each: keys, vals, keys2, vals2, start, xyz, nrm) + shared sort temp
    WS_GRID_A, WS_GRID_B = WS_GRID_A + 7, WS_GRID_C = WS_GRID_B + 7, WS_GRID_TMP = WS_GRID_C + 7, WS_GRID_MISC,
    WS_DS_KEYS, WS_DS_VALS, WS_DS_KEYS2, WS_DS_VALS2, WS_DS_FLAGS, WS_DS_MISC,
    WS_SPFH, WS_KP_ORDER, WS_DENS_A, WS_DENS_B, WS_DENS_C,
    WS_RANSAC_T, WS_RANSAC_STATS, WS_RANSAC_PACK, WS_RANSAC_LIST, WS_RANSAC_HIST, WS_RANSAC_MISC, WS_RANSAC_MASK, WS_RANSAC_MASKT,
    WS_PIPE_SURF_S, WS_PIPE_SURF_T, WS_PIPE_FEAT_S, WS_PIPE_FEAT_T, WS_PIPE_IJ, WS_PIPE_JI, WS_PIPE_DIJ, WS_PIPE_DJI,
    WS_PIPE_CORR, WS_PIPE_KNN_S, WS_PIPE_KNN_T, WS_PIPE_FLAGS, WS_PIPE_MISC, WS_PIPE_KIDX_S, WS_PIPE_KIDX_T, WS_PIPE_KPS_S, WS_PIPE_KPS_T,
    WS_MS_KNN_I, WS_MS_KNN_D, WS_MS_LIST_S, WS_MS_LIST_T, WS_MS_SUB, WS_MS_FEAT_S, WS_MS_FEAT_T, WS_MS_SURF2, WS_MS_RES, WS_PLANE_VISITED, WS_PLANE_CLAIMED, WS_PLANE_OUT, WS_LOCAL_G, WS_SORT_TMP, WS_MATCH_BOX, WS_RANSAC_GHIST,
    WS_HOST_A, WS_HOST_B, WS_HOST_C, WS_HOST_D, WS_HOST_E, WS_HOST_F,
    WS_COUNT
};
static_assert(WS_COUNT <= 112, "grow lgr_ctx::ws");

// returns device pointer of at least `bytes` (contents undefined unless kept); grows with 25% slack
int lgr_ws(lgr_ctx* ctx, int slot, size_t bytes, void** out);
int lgr_pinned(lgr_ctx* ctx, size_t bytes, void** out);

template <class T>
static inline int lgr_ws_t(lgr_ctx* ctx, int slot, size_t count, T** out) {
    void* p = nullptr;
    int rc = lgr_ws(ctx, slot, count * sizeof(T), &p);
    *out = (T*) p;
    return rc;
}

static inline int cdiv(long long a, long long b) { return (int) ((a + b - 1) / b); }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- uniform grid over a cloud (lgr_grid.hip); device view in lgr_grid.cuh ----
struct GridDev {
    float ox, oy, oz, h;
    int dx, dy, dz;
    int n;                      // number of valid (finite) points = entries of the sorted arrays
    const int* cell_start;      // dx*dy*dz + 1
    const float4* pxyz;         // sorted by (cell, original index): x, y, z, bits(original index)
    const float4* pnrm;         // sorted: nx, ny, nz, curvature
};
// h <= 0: automatic cell (about `target` points per occupied cell, 2-D manifold heuristic)
int lgr_grid_build(lgr_ctx* ctx, int slot_base, const float* d_pts, int n, float h, float target, GridDev* out);
// both bounding boxes of a cloud: out12 (host) = true min3, true max3 (finite points only; +-inf when empty),
// reference-quirk min3, max3 (include/common.h:266-280)
int lgr_bbox_host(lgr_ctx* ctx, const float* d_pts, int n, float* out12);
// the same twelve values left on the device as order-preserving integer keys (no host synchronisation); lgr_bbox_key_inv turns a key back
int lgr_bbox_launch(lgr_ctx* ctx, const float* d_pts, int n, const unsigned** d_keys12);
__host__ __device__ inline float lgr_bbox_key_inv(unsigned k) {
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
// lgr_knn_dev with an optional distance table (d_d2 == nullptr: index lists only)
int lgr_knn_lists(lgr_ctx* ctx, const float* d_q, int nq, const float* d_pts, int n, int k, int32_t* d_idx, float* d_d2);

// Two independent pieces of host-driven GPU work side by side: fa(ctx) on this context, fb(ctx->aux) on the second context (own
// stream and workspace, same device; created on first use) from a second host thread.  Everything enqueued on ctx->stream before
// the call is visible to both; the call returns when both have finished and the aux stream has drained, so the caller simply goes
// on using ctx->stream.  The pieces must write disjoint outputs; buffers they allocate belong to the context they ran on.
// What it buys: the host read-backs (counts, extents) and short launches of one piece hide behind the other piece's kernels.
int lgr_ctx_aux(lgr_ctx* ctx);   // makes sure ctx->aux exists
int lgr_ctx_aux2(lgr_ctx* ctx);  // makes sure ctx->aux2 exists
int lgr_ctx_stream3(lgr_ctx* ctx, hipStream_t* out);   // a third stream (ctx->stream itself when opt.helper_contexts == 0)
// post `job` to the helper thread of the internal context `aux` (started on first use; a thread that cannot be started is an error
// code, never an exception across the C ABI); lgr_helper_wait blocks until it has run and returns its status.  Every post must be
// followed by exactly one wait -- also on the caller's error paths (lgr_helper_guard).
int lgr_helper_post(lgr_ctx* owner, lgr_ctx* aux, std::function<int()> job);
int lgr_helper_wait(lgr_ctx* aux);
struct lgr_helper_guard {   // waits for a posted job on every exit path of the scope
    lgr_ctx* aux = nullptr;
    bool armed = false;
    int wait() { armed = false; return lgr_helper_wait(aux); }
    ~lgr_helper_guard() { if (armed) (void) lgr_helper_wait(aux); }
};
// a job for an internal context: run f(aux), then ALWAYS drain the aux stream (also when f failed: kernels it had already enqueued may
// still be writing caller-visible buffers, and the caller is about to return an error and let go of them)
template <class F>
static inline std::function<int()> lgr_aux_job(lgr_ctx* aux, F&& f) {
    return [aux, &f]() -> int {
        int rc = hipSetDevice(aux->device) == hipSuccess ? f(aux) : (int) LGR_ERR_HIP;
        const hipError_t e = hipStreamSynchronize(aux->stream);
        if (rc == LGR_OK && e != hipSuccess) rc = lgr_fail(aux, LGR_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__);
        return rc;
    };
}
template <class FA, class FB>
static inline int lgr_run_pair(lgr_ctx* ctx, FA&& fa, FB&& fb) {
    LGR_TRY(lgr_ctx_aux(ctx));
    lgr_ctx* ax = ctx->aux;
    if (!ctx->opt.helper_contexts) {   // one host thread, one stream (ax->stream == ctx->stream): the second context is a second workspace
        LGR_TRY(fa(ctx));
        const int rc = fb(ax);
        if (rc != LGR_OK) ctx->err = ax->err;
        return rc;
    }
    LGR_HIP(ctx, hipEventRecord(ctx->aux_ev, ctx->stream));
    LGR_HIP(ctx, hipStreamWaitEvent(ax->stream, ctx->aux_ev, 0));
    lgr_helper_guard g{ax, false};
    LGR_TRY(lgr_helper_post(ctx, ax, lgr_aux_job(ax, fb)));
    g.armed = true;
    const int rc_a = fa(ctx);
    const int rc_b = g.wait();
    (void) hipSetDevice(ctx->device);
    if (rc_b != LGR_OK) { ctx->err = ax->err; return rc_b; }
    return rc_a;
}

// caller-supplied correspondences (public lgr_ransac* / lgr_gror* / lgr_evaluate* / lgr_refit_svd entry points): LGR_ERR_INVALID_ARG
// when an index_query is outside [0, ns) or an index_match outside [0, nt) -- checked on the device BEFORE any kernel gathers
// points through them (an out-of-range gather is a GPU memory fault, not an error code).  One tiny launch + a 4-byte read-back;
// skipped when ctx->corr_trusted (lgr_ransac.hip).
int lgr_check_corr(lgr_ctx* ctx, const lgr_corr* d_corr, int c, int ns, int nt);
// lgr_sort.hip: stable radix sort of (key, 32-bit value) pairs, out of place (in and out must differ; the input is preserved), on
// ctx->stream.  u32: the key bits [begin_bit, end_bit); u64: the bit ranges (shift, width) listed from the least significant up --
// bits outside the ranges must be equal in all keys.
int lgr_sort_pairs_u32(lgr_ctx* ctx, const unsigned* kin, unsigned* kout, const int* vin, int* vout, size_t n, int begin_bit, int end_bit);
int lgr_sort_pairs_u64(lgr_ctx* ctx, const unsigned long long* kin, unsigned long long* kout, const int* vin, int* vout, size_t n,
                       const int* shifts, const int* widths, int n_ranges);
// lgr_match.hip: the query-side half of a coming brute-force match (clustering + assignment / sort / placement of d_a33), run ahead
// of the call while the train side's descriptors are still being computed; consumed by the next lgr_match_bf*_dev on the same
// (d_a33, ma, mb, both directions) or dropped by lgr_match_prepare_cancel.
int lgr_match_prepare(lgr_ctx* ctx, const float* d_a33, int ma, int mb, bool both);
void lgr_match_prepare_cancel(lgr_ctx* ctx);

// ---- closest-plane metric on the device (lgr_plane.hip) ----
struct lgr_plane_dev {
    GridDev g;                  // uniform grid over the target (cell = 1.001 * radius), with normals
    float thr, r2;              // inlier threshold (target cloud density), squared search radius (2 * thr)
    int n_sp, ns;               // sparse subset size = (int) (0.01 * |src|), |src|
    const float* d_src;
    uint64_t seed;
    unsigned* visited;          // [n_wg][(ns + 31) / 32] claim bitmaps, all zero between hypotheses
    int* claimed;               // [n_wg][n_sp]
    int n_wg;
};
int lgr_plane_setup(lgr_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt, uint64_t seed, lgr_plane_dev* out);
// hypothesis h in [0, nh): transform d_Ts + 16 * off, Philox counter = counter_base + off, off = d_list ? d_list[h] : h.
// d_rmse / d_pairs (+ d_n_pairs) optional; pairs = (source index, nearest target index) of the inliers, unordered.
// Gate (RANSAC batches): best_prev = best metric of the earlier batches, record_prev = their record inlier count (INT_MAX when the plane
// counts are not the records), d_factor (optional) = per-hypothesis factor of the metric (combination: the correspondence metric);
// a hypothesis whose upper bounds fall below both is abandoned -- it can be neither the best nor a record.  0 / INT_MAX / NULL: no gate.
// dyn (the device-driven RANSAC schedule, round 5): the number of hypotheses, the counter base and the gate's two values are READ ON THE DEVICE from
// these words when the launch runs (nh is then only an upper bound for the grid); a null member keeps the host's value.
struct lgr_plane_dyn { const int* nh; const int* counter_base; const float* best_prev; const int* record_prev; };
int lgr_plane_eval(lgr_ctx* ctx, const lgr_plane_dev& pd, const float* d_Ts, const int* d_list, int nh, unsigned counter_base, int score_id,
                   int* d_cnt, float* d_metric, float* d_rmse, int2* d_pairs, int* d_n_pairs, float best_prev = 0.f, int record_prev = 0x7fffffff,
                   const float* d_factor = nullptr, const lgr_plane_dyn* dyn = nullptr);

