// lgr_internal.h -- internals shared by the HIP translation units of liblgr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/lgr.h"

#define LGR_WAVE 64

struct lgr_buf {
    void* p = nullptr;
    size_t cap = 0;
};

// Per-thread/GPU context: stream, error string, named workspace buffers (grown on demand, reused across calls).
struct lgr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    lgr_buf ws[64];
    void* pinned = nullptr;  // small pinned host scratch for read-backs
    size_t pinned_cap = 0;
    hipEvent_t ev[16];
    float stage_ms[12];
    int n_cu = 256;
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
    WS_MATCH_BEST_A, WS_MATCH_BEST_B, WS_MATCH_MISC, WS_MATCH_DENSE,
    WS_GRID_KEYS, WS_GRID_VALS, WS_GRID_KEYS2, WS_GRID_VALS2, WS_GRID_START, WS_GRID_TMP, WS_GRID_PTS, WS_GRID_MISC,
    WS_GRID2_KEYS, WS_GRID2_VALS, WS_GRID2_KEYS2, WS_GRID2_VALS2, WS_GRID2_START, WS_GRID2_PTS,
    WS_DS_OUT, WS_DS_MISC, WS_SPFH, WS_KNN_IDX, WS_KNN_D2, WS_DENS_A, WS_DENS_B,
    WS_RANSAC_T, WS_RANSAC_FLAGS, WS_RANSAC_STATS, WS_RANSAC_PACK, WS_RANSAC_LIST, WS_RANSAC_HIST, WS_RANSAC_MISC, WS_RANSAC_MASK,
    WS_PIPE_SURF_S, WS_PIPE_SURF_T, WS_PIPE_FEAT_S, WS_PIPE_FEAT_T, WS_PIPE_IJ, WS_PIPE_JI, WS_PIPE_DIJ, WS_PIPE_DJI,
    WS_PIPE_CORR, WS_PIPE_KNN_S, WS_PIPE_KNN_T, WS_PIPE_MISC,
    WS_HOST_A, WS_HOST_B, WS_HOST_C, WS_HOST_D, WS_HOST_E, WS_HOST_F,
    WS_COUNT
};
static_assert(WS_COUNT <= 64, "grow lgr_ctx::ws");

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

// ---- stage functions implemented across translation units (device pointers) ----
struct lgr_grid;  // lgr_grid.hip
