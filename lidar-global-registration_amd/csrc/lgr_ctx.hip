// lgr_ctx.hip -- context, workspace, error plumbing of liblgr_hip.so.
#include <vector>
#include "lgr_internal.h"

#include <atomic>

// ---- the device's turn (lgr_turn, lgr_internal.h) ----
namespace {
struct DevTurn {
    std::mutex mu;                 // held by the host thread inside the outermost public entry point of a context
    unsigned long long last = 0;   // turn_id of the context that held it last
    hipEvent_t done = nullptr;     // recorded on that context's stream when it gave the turn back
    bool recorded = false;
};
constexpr int MAX_DEV = 64;
DevTurn g_turn[MAX_DEV];
std::atomic<unsigned long long> g_next_turn_id{1};
}  // namespace

lgr_turn::lgr_turn(lgr_ctx* ctx) : c(ctx) {
    if (!c) return;
    if (c->turn_depth++ > 0) return;   // an entry point called from another one of the same context
    if (c->internal || c->opt.concurrent_contexts || c->device < 0 || c->device >= MAX_DEV) return;
    DevTurn& t = g_turn[c->device];
    t.mu.lock();
    held = true;
    if (t.recorded && t.last != c->turn_id) {   // another context worked last: everything it queued comes first
        (void) hipSetDevice(c->device);
        (void) hipStreamWaitEvent(c->stream, t.done, 0);
    }
}

lgr_turn::~lgr_turn() {
    if (!c) return;
    c->turn_depth--;
    if (!held) return;
    DevTurn& t = g_turn[c->device];
    // (helper streams have been drained or joined into c->stream by the time a public entry point returns: lgr_aux_job, join_b)
    (void) hipSetDevice(c->device);
    if (!t.done && hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) t.done = nullptr;
    t.recorded = t.done && hipEventRecord(t.done, c->stream) == hipSuccess;
    t.last = c->turn_id;
    t.mu.unlock();
}

int lgr_fail(lgr_ctx* ctx, int code, const char* what, const char* file, int line) {
    if (ctx) {
        char b[768];
        snprintf(b, sizeof b, "[lgr %d] %s (%s:%d)", code, what, file, line);
        ctx->err = b;
    }
    return code;
}

int lgr_ws(lgr_ctx* ctx, int slot, size_t bytes, void** out) {
    lgr_buf& b = ctx->ws[slot];
    if (bytes == 0) bytes = 16;
    if (b.cap < bytes) {
        // growing a buffer: everything enqueued so far may still use the old one
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (b.p) LGR_HIP(ctx, hipFree(b.p));
        b.p = nullptr; b.cap = 0;
        // slack: a half for buffers under 256 MB, a quarter above (a job of pairs of different sizes -- bench.py --job -- re-allocates whenever a pair
        // needs more than any before it, and a hipFree + hipMalloc of the large buffers costs tens of milliseconds)
        size_t want = bytes + (bytes < ((size_t) 256 << 20) ? bytes / 2 : bytes / 4) + 256;
        want = (want + 255) & ~(size_t) 255;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) {
            (void) hipGetLastError();
            want = (bytes + 255) & ~(size_t) 255;
            LGR_HIP(ctx, hipMalloc(&b.p, want));
        }
        b.cap = want;
        if (getenv("LGR_WS_DEBUG")) fprintf(stderr, "[lgr] workspace slot %d of context %p grows to %.1f MB (asked %.1f MB)\n", slot, (void*) ctx, want / 1048576.0, bytes / 1048576.0);
    }
    *out = b.p;
    return LGR_OK;
}

int lgr_pinned(lgr_ctx* ctx, size_t bytes, void** out) {
    if (ctx->pinned_cap < bytes) {
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pinned) LGR_HIP(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr; ctx->pinned_cap = 0;
        size_t want = bytes < 4096 ? 4096 : bytes;
        LGR_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_cap = want;
    }
    *out = ctx->pinned;
    return LGR_OK;
}

extern "C" int lgr_version(void) { return LGR_VERSION; }

extern "C" int lgr_ctx_create(int device, void* stream, lgr_ctx** out) {
    if (!out) return LGR_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LGR_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return LGR_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return LGR_ERR_NO_DEVICE;
    lgr_ctx* c = new lgr_ctx();
    c->device = device;
    if (stream != LGR_STREAM_OWN && stream != LGR_STREAM_OWN_LOW) { c->stream = (hipStream_t) stream; c->own_stream = false; }
    else {
        // LGR_STREAM_OWN_LOW (the internal helper contexts): lowest stream priority, so that the caller's stream -- which carries the
        // chains of short dependent launches (clustering, bounds, masks) -- gets compute units first when both have work queued
        int lo = 0, hi = 0;
        (void) hipDeviceGetStreamPriorityRange(&lo, &hi);
        const hipError_t e = stream == LGR_STREAM_OWN_LOW ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, lo)
                                                          : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return LGR_ERR_HIP; }
        c->own_stream = true;
    }
    for (int i = 0; i < 32; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) { delete c; return LGR_ERR_HIP; }
    for (int i = 0; i < 12; ++i) c->stage_ms[i] = 0.f;
    c->turn_id = g_next_turn_id++;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    *out = c;
    return LGR_OK;
}

// internal contexts: own non-blocking stream + helper thread (opt.helper_contexts), or the parent's stream (workspace only)
static int make_internal(lgr_ctx* ctx, lgr_ctx** out, hipEvent_t* ev) {
    if (*out) return LGR_OK;
    // (LGR_STREAM_OWN_LOW was measured for the helpers: -0.1 ms under `lr`, but the cluster filter's second 40-NN table then slides from the
    // feature phase under pass 0 of the matcher and costs it 2 ms: plain priority.  Round 5: a helper stream with every 16th / 8th compute unit
    // masked out (hipExtStreamCreateWithCUMask), so that the caller's chains of short launches always find room beside the helper's big feature
    // kernels: 33.1 / 29.5 ms per pair instead of 25.4 -- launches on a masked queue take far longer to start: dropped)
    LGR_CHECK(ctx, lgr_ctx_create(ctx->device, ctx->opt.helper_contexts ? LGR_STREAM_OWN : (void*) ctx->stream, out) == LGR_OK, LGR_ERR_HIP);
    (*out)->opt = ctx->opt;
    (*out)->mopt = ctx->mopt;
    (*out)->internal = true;
    if (!*ev) LGR_HIP(ctx, hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return LGR_OK;
}
int lgr_ctx_stream3(lgr_ctx* ctx, hipStream_t* out) {
    if (!ctx->ev3) LGR_HIP(ctx, hipEventCreateWithFlags(&ctx->ev3, hipEventDisableTiming));
    if (!ctx->opt.helper_contexts) { *out = ctx->stream; return LGR_OK; }
    if (!ctx->stream3) LGR_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
    *out = ctx->stream3;
    return LGR_OK;
}
int lgr_ctx_aux(lgr_ctx* ctx) { return make_internal(ctx, &ctx->aux, &ctx->aux_ev); }
int lgr_ctx_aux2(lgr_ctx* ctx) { return make_internal(ctx, &ctx->aux2, &ctx->aux2_ev); }

static void helper_main(lgr_helper* h, int device) {
    (void) hipSetDevice(device);
    std::unique_lock<std::mutex> lk(h->mu);
    for (;;) {
        h->cv.wait(lk, [h] { return h->has_job || h->quit; });
        if (h->quit) return;
        std::function<int()> job = std::move(h->job);
        h->has_job = false;
        lk.unlock();
        int rc;
        try { rc = job(); } catch (...) { rc = LGR_ERR_HIP; }   // nothing may unwind out of a helper thread
        lk.lock();
        h->rc = rc;
        h->busy = false;
        h->cv.notify_all();
    }
}

int lgr_helper_post(lgr_ctx* owner, lgr_ctx* aux, std::function<int()> job) {
    if (!aux->helper) aux->helper = new (std::nothrow) lgr_helper();
    lgr_helper* h = aux->helper;
    LGR_CHECK(owner, h != nullptr, LGR_ERR_OOM);
    if (!h->started) {
        try { h->th = std::thread(helper_main, h, aux->device); }
        catch (...) { return lgr_fail(owner, LGR_ERR_HIP, "could not start the helper host thread (std::system_error)", __FILE__, __LINE__); }
        h->started = true;
    }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (h->busy) return lgr_fail(owner, LGR_ERR_INVALID_ARG, "helper context busy (a context is not re-entrant)", __FILE__, __LINE__);
        h->job = std::move(job);
        h->has_job = true;
        h->busy = true;
    }
    h->cv.notify_all();
    return LGR_OK;
}

int lgr_helper_wait(lgr_ctx* aux) {
    lgr_helper* h = aux->helper;
    if (!h) return LGR_OK;
    std::unique_lock<std::mutex> lk(h->mu);
    h->cv.wait(lk, [h] { return !h->busy; });
    return h->rc;
}

static void helper_stop(lgr_ctx* c) {
    lgr_helper* h = c->helper;
    if (!h) return;
    if (h->started) {
        { std::lock_guard<std::mutex> lk(h->mu); h->quit = true; }
        h->cv.notify_all();
        if (h->th.joinable()) h->th.join();
    }
    delete h;
    c->helper = nullptr;
}

static void drop_internal(lgr_ctx* ctx) {
    if (ctx->match_prep) lgr_match_prepare_cancel(ctx);
    if (ctx->aux) { (void) lgr_ctx_destroy(ctx->aux); ctx->aux = nullptr; }
    if (ctx->aux2) { (void) lgr_ctx_destroy(ctx->aux2); ctx->aux2 = nullptr; }
}

extern "C" int lgr_ctx_destroy(lgr_ctx* ctx) {
    if (!ctx) return LGR_OK;
    (void) hipSetDevice(ctx->device);
    helper_stop(ctx);
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->match_prep && ctx->match_prep_free) { ctx->match_prep_free(ctx->match_prep); ctx->match_prep = nullptr; }
    drop_internal(ctx);
    if (ctx->aux_ev) { (void) hipEventDestroy(ctx->aux_ev); ctx->aux_ev = nullptr; }
    if (ctx->stream3) { (void) hipStreamSynchronize(ctx->stream3); (void) hipStreamDestroy(ctx->stream3); ctx->stream3 = nullptr; }
    if (ctx->ev3) { (void) hipEventDestroy(ctx->ev3); ctx->ev3 = nullptr; }
    if (ctx->aux2_ev) { (void) hipEventDestroy(ctx->aux2_ev); ctx->aux2_ev = nullptr; }
    for (int i = 0; i < WS_COUNT; ++i)
        if (ctx->ws[i].p) (void) hipFree(ctx->ws[i].p);
    if (ctx->pinned) (void) hipHostFree(ctx->pinned);
    for (int i = 0; i < 32; ++i) (void) hipEventDestroy(ctx->ev[i]);
    if (ctx->own_stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
    return LGR_OK;
}

extern "C" void lgr_ctx_default_options(lgr_ctx_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->helper_contexts = 1;
}

extern "C" int lgr_ctx_set_options(lgr_ctx* ctx, const lgr_ctx_options* opt) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    lgr_ctx_options o;
    lgr_ctx_default_options(&o);
    if (opt) o = *opt;
    LGR_CHECK(ctx, (o.helper_contexts == 0 || o.helper_contexts == 1) && (o.concurrent_contexts == 0 || o.concurrent_contexts == 1), LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, (o.arithmetic == LGR_ARITH_FAST || o.arithmetic == LGR_ARITH_PCL) && (o.pcl_neighbour_cap == 0 || o.pcl_neighbour_cap == 64 || o.pcl_neighbour_cap == 1024), LGR_ERR_INVALID_ARG);
    LGR_CHECK(ctx, o.ransac_schedule >= 0 && o.ransac_schedule <= 2, LGR_ERR_INVALID_ARG);
    if (o.helper_contexts != ctx->opt.helper_contexts) {
        // the internal contexts are bound to a stream when they are created: drop them (workspaces included), they come back on
        // first use with the stream the new setting asks for
        LGR_HIP(ctx, hipSetDevice(ctx->device));
        LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drop_internal(ctx);
    }
    ctx->opt = o;
    // the internal contexts run whole stages of the path (the target cloud's features on ctx->aux): they follow their owner's options
    std::function<void(lgr_ctx*)> push = [&](lgr_ctx* c) { if (!c) return; c->opt = o; push(c->aux); push(c->aux2); };
    push(ctx->aux); push(ctx->aux2);
    return LGR_OK;
}

extern "C" int lgr_ctx_get_options(lgr_ctx* ctx, lgr_ctx_options* opt) {
    if (!ctx || !opt) return LGR_ERR_INVALID_ARG;
    *opt = ctx->opt;
    return LGR_OK;
}

static int helper_threads(const lgr_ctx* c) {
    if (!c) return 0;
    return ((c->helper && c->helper->started) ? 1 : 0) + helper_threads(c->aux) + helper_threads(c->aux2);
}
extern "C" int lgr_ctx_host_threads(lgr_ctx* ctx, int* n) {
    if (!ctx || !n) return LGR_ERR_INVALID_ARG;
    *n = 1 + helper_threads(ctx->aux) + helper_threads(ctx->aux2);
    return LGR_OK;
}

extern "C" int lgr_ctx_sync(lgr_ctx* ctx) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream3) LGR_HIP(ctx, hipStreamSynchronize(ctx->stream3));   // (joined into `stream` by every successful call; an early error exit may leave work on it)
    for (lgr_ctx* ax : {ctx->aux, ctx->aux2})                              // ... and so for the helper contexts' streams
        if (ax && ax->stream != ctx->stream) LGR_HIP(ctx, hipStreamSynchronize(ax->stream));
    return LGR_OK;
}

extern "C" const char* lgr_last_error(lgr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int lgr_ctx_stage_ms(lgr_ctx* ctx, float* out12) {
    if (!ctx || !out12) return LGR_ERR_INVALID_ARG;
    for (int i = 0; i < 12; ++i) out12[i] = ctx->stage_ms[i];
    return LGR_OK;
}

extern "C" void lgr_match_default_options(lgr_match_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->prune = -1; o->leaves = 0; o->near = 0; o->operand_format = -1; o->box_bounds = 1; o->column_stage = 1;
    o->coarse_rejection = 1; o->rerank_refilter = 1; o->pair_cap = -1; o->poison_tables = 0; o->self_check = 0; o->shell_bound = 1; o->split_sweep = 1; o->auto_dense = 1; o->irregular_rows = 1;
}

// (every internal context below this one, however deep: an internal context that runs a pair of jobs itself owns internal contexts too)
static void propagate_mopt(lgr_ctx* ctx, const lgr_match_options& o) {
    ctx->mopt = o;
    if (ctx->aux) propagate_mopt(ctx->aux, o);
    if (ctx->aux2) propagate_mopt(ctx->aux2, o);
}

extern "C" int lgr_ctx_set_match_options(lgr_ctx* ctx, const lgr_match_options* opt) {
    if (!ctx) return LGR_ERR_INVALID_ARG;
    lgr_match_options o;
    if (!opt) lgr_match_default_options(&o);
    else {
        LGR_CHECK(ctx, opt->prune >= -1 && opt->prune <= 1 && opt->leaves >= 0 && opt->leaves <= 64 && opt->near >= 0 &&
                       opt->operand_format >= -1 && opt->operand_format <= 2 && opt->box_bounds >= 0 && opt->box_bounds <= 2 && opt->kept_cap >= 0 &&
                       (opt->auto_dense == 0 || opt->auto_dense == 1) && (opt->irregular_rows == 0 || opt->irregular_rows == 1), LGR_ERR_INVALID_ARG);
        o = *opt;
    }
    propagate_mopt(ctx, o);
    return LGR_OK;
}

extern "C" int lgr_ctx_get_match_options(lgr_ctx* ctx, lgr_match_options* opt) {
    if (!ctx || !opt) return LGR_ERR_INVALID_ARG;
    *opt = ctx->mopt;
    return LGR_OK;
}

extern "C" int lgr_ctx_workspace_bytes(lgr_ctx* ctx, uint64_t* bytes) {
    if (!ctx || !bytes) return LGR_ERR_INVALID_ARG;
    uint64_t t = 0;
    for (int i = 0; i < WS_COUNT; ++i) t += ctx->ws[i].cap;
    if (ctx->aux) for (int i = 0; i < WS_COUNT; ++i) t += ctx->aux->ws[i].cap;
    if (ctx->aux2) for (int i = 0; i < WS_COUNT; ++i) t += ctx->aux2->ws[i].cap;
    *bytes = t;
    return LGR_OK;
}

// defaults of getParametersFromConfig (src/common.cpp:216-223, 335-413) and include/common.h:38-57
extern "C" void lgr_default_params(lgr_params* p) {
    memset(p, 0, sizeof(*p));
    p->feature_nr_points = 352;
    p->normal_nr_points = 30;
    p->edge_thr_coef = 0.95f;
    p->distance_thr = 0.1f;
    p->feature_radius = 0.25f;
    p->scale_factor = 2.0f;
    p->confidence = 0.999f;
    p->bf_block_size = 10000;
    p->cluster_k = 40;
    p->randomness = 1;
    p->n_samples = 3;
    p->alignment_id = LGR_ALIGN_RANSAC;
    p->keypoint_id = LGR_KEYPOINT_ANY;   // BASELINE configs; the reference's struct default is iss (include/common.h:148)
    p->matching_id = LGR_MATCH_CLUSTER;
    p->metric_id = LGR_METRIC_UNIFORMITY;
    p->score_id = LGR_SCORE_MSE;
    p->max_iterations = 2147483647;
    p->normals_available = 0;
    p->fix_seed = 1;
    p->ransac_batch = 65536;
    p->seed = 566;
    p->use_bfmatcher = 1;                // ALIGNMENT_USE_BFMATCHER include/common.h:41
    for (int i = 0; i < 4; ++i) p->guess[5 * i] = 1.f;
}

// Diagnostics (tools/exp_concurrent_stages.py; not part of include/lgr.h): device pointer and capacity of a named pipeline buffer of the
// context's workspace, as the last lgr_correspondences* / lgr_align* call left it.
extern "C" int lgr_debug_ws(lgr_ctx* ctx, const char* name, void** p, size_t* cap) {
    if (!ctx || !name || !p || !cap) return LGR_ERR_INVALID_ARG;
    static const struct { const char* n; int slot; } T[] = {
        {"surf_s", WS_PIPE_SURF_S}, {"surf_t", WS_PIPE_SURF_T}, {"feat_s", WS_PIPE_FEAT_S}, {"feat_t", WS_PIPE_FEAT_T}, {"ij", WS_PIPE_IJ}, {"ji", WS_PIPE_JI},
        {"dij", WS_PIPE_DIJ}, {"dji", WS_PIPE_DJI}, {"corr", WS_PIPE_CORR}, {"thr", WS_PIPE_MISC}, {"knn_s", WS_PIPE_KNN_S}, {"knn_t", WS_PIPE_KNN_T}};
    for (const auto& e : T)
        if (!strcmp(e.n, name)) { *p = ctx->ws[e.slot].p; *cap = ctx->ws[e.slot].cap; return LGR_OK; }
    return LGR_ERR_INVALID_ARG;
}
