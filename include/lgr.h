/*
 * lgr.h -- C ABI of the MI355X-native global-registration hot path (liblgr_hip.so).
 *
 * Drop-in boundary for aleksandrina-streltsova/lidar-global-registration.  The reference has no FFI layer: its
 * boundary is the C++ header surface include/alignment.h:6-19, include/correspondence_search.h:9-28,
 * include/sac_prerejective_omp.h:21-56 plus the free functions named below.  Each entry point cites the reference
 * interface it replaces; lidar-global-registration_amd/host/ holds the header-only C++ shim that re-exposes the
 * reference names on top of this ABI, and INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.  Every call returns an int status (LGR_OK == 0, negative =
 *     error); nothing throws across the ABI.  "not converged" is NOT an error: see lgr_result.converged.
 *   - point  : 12 floats, pcl::PointXYZINormal layout {x,y,z,1 | nx,ny,nz,0 | intensity,curvature,pad,pad} (48 B)
 *   - fpfh   : 33 floats (pcl::FPFHSignature33, 132 B, row-major M x 33)
 *   - corr   : lgr_corr (include/common.h:120-127 Correspondence), 16 B
 *   - T      : 16 floats COLUMN-major (Eigen::Matrix4f default)
 *   - host entry points (no suffix) borrow caller memory for the duration of the call, upload, run the device
 *     path and download.  *_dev entry points take DEVICE pointers (HIP), enqueue on the context stream and are
 *     asynchronous unless stated; outputs are caller-allocated device buffers.
 *   - one lgr_ctx per host thread / GPU.  A ctx owns its workspace (grown on demand, never inside a timed launch
 *     once warmed up) and is not re-entrant.  Several contexts on ONE device are allowed, but by default they take turns call
 *     by call (lgr_ctx_options.concurrent_contexts): results are then bit-identical to a serial run by construction.
 *   - the HIP extension is mandatory: there is no CPU fallback anywhere behind this ABI.
 */
#ifndef LGR_H
#define LGR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision.  5 (round 5): lgr_ctx_options.arithmetic / pcl_neighbour_cap and lgr_match_options.auto_dense / irregular_rows (former reserved words: a host that zeroes
 * them keeps the default arithmetic and switches auto_dense and the irregular-row lane off), lgr_match_last_lbstats, lgr_match_last_irregular, lgr_selfcheck_philox,
 * lgr_selfcheck_libm added; the DEFAULT arithmetic of the normals and pair features changed to PCL's own sequences (results differ from
 * revision 4 at rounding level).  4 (round 4): lgr_match_options.split_sweep / kept_cap and lgr_ctx_options.concurrent_contexts (former reserved words: a host that
 * zeroes them switches the split off and keeps the contexts exclusive), lgr_match_last_issued, lgr_selfcheck_rcp added.  3 (round 3): lgr_match_options.shell_bound (one of the reserved words: a host that zeroes them would switch the shell
 * bound off), lgr_match_last_shell added.  2 (round 3): lgr_match_last_* take the context, lgr_ctx_options / lgr_ctx_host_threads added;
 * lgr_params grew in revision 1 -> 2 as well (use_bfmatcher, has_guess, match_search_radius, guess).  A host built against another revision must not
 * call in: check lgr_version() == LGR_VERSION once after loading (lgr_amd/capi.py and host/lgr_compat.hpp do). */
#define LGR_VERSION 5

enum {
    LGR_OK = 0,
    LGR_ERR_INVALID_ARG = -1,
    LGR_ERR_NO_DEVICE = -2,
    LGR_ERR_OOM = -3,
    LGR_ERR_HIP = -4,          /* a HIP runtime call failed; see lgr_last_error */
    LGR_ERR_UNSUPPORTED = -5,  /* e.g. n_samples outside 3..8, randomness != 1, alignment teaser (throws in the reference) */
    LGR_ERR_VOXEL_TOO_SMALL = -6
};

enum { LGR_MATCH_LR = 0, LGR_MATCH_ONE_SIDED = 1, LGR_MATCH_CLUSTER = 2 };      /* src/matching.cpp:21-75 */
enum { LGR_METRIC_CORRESPONDENCES = 0, LGR_METRIC_UNIFORMITY = 1, LGR_METRIC_CLOSEST_PLANE = 2, LGR_METRIC_COMBINATION = 3 };   /* src/metric.cpp:272-301 */
enum { LGR_SCORE_CONSTANT = 0, LGR_SCORE_MAE = 1, LGR_SCORE_MSE = 2, LGR_SCORE_EXP = 3 };
enum { LGR_ALIGN_RANSAC = 0, LGR_ALIGN_GROR = 1 };                              /* src/alignment.cpp:92-101 */
enum { LGR_KEYPOINT_ANY = 0, LGR_KEYPOINT_ISS = 1 };                            /* src/common.cpp:657-691 */
enum { LGR_ORDER_REFERENCE = 0, LGR_ORDER_CANONICAL = 1 };                      /* downsample output order */

typedef struct { int32_t index_query, index_match; float distance, threshold; } lgr_corr;

/* mirrors AlignmentParameters (include/common.h:135-163); string ids become enums; optionals become has_* flags */
typedef struct {
    int32_t feature_nr_points;   /* 352 */
    int32_t normal_nr_points;    /* 30 */
    float   edge_thr_coef;       /* 0.95 */
    float   distance_thr;
    float   feature_radius;      /* > 0: single scale; <= 0 = unset: multi-scale (include/matching.h:176-262) */
    float   scale_factor;        /* 2.0 */
    float   confidence;          /* 0.999 */
    int32_t bf_block_size;       /* ALIGNMENT_BLOCK_SIZE (lgr_default_params: 10000); every shipped YAML sets 200000 (data/test.yaml:12) */
    int32_t cluster_k;           /* 40 */
    int32_t randomness;          /* 1 (only 1, as data/test.yaml:14 says) */
    int32_t n_samples;           /* 3 (every shipped config); 3..8 accepted: sampler, polygon test and Umeyama are generic in it,
                                    src/sac_prerejective_omp.cpp:33-77,105-108,220 */
    int32_t alignment_id, matching_id, metric_id, score_id;
    int32_t max_iterations;
    int32_t normals_available;
    int32_t fix_seed;            /* 1: seed = 566 (SEED include/common.h:25); 0: seed field below */
    int32_t has_vp_src, has_vp_tgt;
    float   vp_src[3], vp_tgt[3];
    int32_t ransac_batch;        /* iterations per device batch (deterministic schedule), default 65536 */
    uint64_t seed;
    int32_t keypoint_id;         /* LGR_KEYPOINT_ANY (every point, BASELINE configs) or LGR_KEYPOINT_ISS */
    float   iss_radius_src, iss_radius_tgt;   /* include/common.h:139; salient = non-maxima radius */
    int32_t use_bfmatcher;       /* 1 (ALIGNMENT_USE_BFMATCHER include/common.h:41); 0: matchFLANN (include/matching.h:309) */
    /* include/common.h:158-160: "cannot be set in config, set before alignment steps" */
    int32_t has_guess;           /* 1: matching is matchLocal around guess * p (include/matching.h:294-299) and the guess is the
                                  *    hypothesis RANSAC has to beat (src/sac_prerejective_omp.cpp:134-147) */
    float   match_search_radius;
    float   guess[16];           /* column-major */
} lgr_params;

/* mirrors AlignmentResult (include/common.h:165-174) + diagnostics */
typedef struct {
    float   transformation[16];  /* column-major */
    int32_t iterations;
    int32_t converged;
    int32_t n_inliers;
    float   metric;
    float   best_metric_before_refit;
    int32_t best_iteration;
    int32_t num_rejections;
    int32_t estimated_iters;
    int32_t n_correspondences;
    double  time_cs, time_te;    /* seconds, device-synchronised wall time */
    float   stage_ms[12];        /* 0 downsample 1 normals 2 fpfh 3 match 4 filter 5 ransac 6 refit (hipEvent) */
} lgr_result;

typedef struct lgr_ctx lgr_ctx;

/* How a context uses the HOST and the device queue (never what it returns).  Independent pieces of the path -- the two clouds'
 * feature stages, the two sides / directions of the matcher, the match filter's per-cloud tables -- run side by side on up to two
 * internal contexts, each with a stream of its own and ONE persistent helper host thread (started on first use, parked on a
 * condition variable between calls).  helper_contexts = 0 turns that off: every piece runs on the context's own stream from the
 * calling thread (no extra threads, no extra streams; the internal contexts remain as workspaces only) -- for hosts that give a
 * rank fewer cores than 3, at the price of the overlap (about +20 % per 1M-point pair). */
typedef struct {
    int32_t helper_contexts;      /* 1 (default) / 0 */
    int32_t concurrent_contexts;  /* 0 (default): the contexts of one device take turns call by call -- the device never executes two
                                   * contexts' work side by side (a context's own helper streams are not affected).  1 (EXPERIMENTAL): this
                                   * context does not wait its turn.  Several pairs in flight per GPU bought +5 % throughput at best
                                   * (DESIGN.md section 10).  Every kernel is deterministic and contexts share no state, so results must not
                                   * depend on it, and the -m gpu suite asserts exactly that (tests/test_gpu_concurrent_contexts.py: three
                                   * overlapping contexts bit-equal to the serial run); but in round 3 the builder's MI355X boxes returned
                                   * normals that differed at rounding level between runs when 2-3 contexts worked at once, and rounds 4-5
                                   * could not reproduce that on the units they were given -- not even with the round-3 binary -- so the
                                   * cause is not established.  Do not enable it where bit-reproducibility is a requirement.  One process per
                                   * GPU (the multi-GPU layout) never has two contexts on a device. */
    int32_t arithmetic;           /* LGR_ARITH_FAST (0, default) / LGR_ARITH_PCL (1): see below */
    int32_t pcl_neighbour_cap;    /* LGR_ARITH_PCL: neighbours of a key point sorted at once; 0 default (512), 1024, 64 (tests: drives the shell path).  Never changes results */
    int32_t ransac_schedule;      /* how the RANSAC loop (uniformity / correspondences metrics) is driven; never changes results.
                                   * LGR_RANSAC_SCHEDULE_DEFAULT (0) = LGR_RANSAC_SCHEDULE_CHAIN (1): device-driven chain of launches, six per round;
                                   * LGR_RANSAC_SCHEDULE_RESIDENT (2): ONE resident kernel for the whole loop (a workgroup per CU, phases handed over
                                   * at grid barriers).  The plane metrics always use the chain. */
    int32_t reserved[3];
} lgr_ctx_options;
enum { LGR_RANSAC_SCHEDULE_DEFAULT = 0, LGR_RANSAC_SCHEDULE_CHAIN = 1, LGR_RANSAC_SCHEDULE_RESIDENT = 2 };
/* Arithmetic of the third-party pieces (normals, pair features, FPFH weighting: PCL 1.12.1 behind include/common.h:322-332 and
 * src/common.cpp:644-655).  In BOTH modes the normals are pcl::eigen33's closed form and the pair features use the acosf swap test and the
 * atan2f of the named libm (GNU libc 2.35's float routines restated op for op: csrc/lgr_libm.cuh, pinned against the running libm by
 * tests/test_oracle_libm.py) -- PCL's own sequences since round 5.  The modes differ in the FPFH weighting only:
 *   LGR_ARITH_FAST  one fused multiply-add chain per bin over the neighbours in grid order (what v_mfma_f32_16x16x4_f32 computes), block
 *                   normaliser from the finished bins: rounding-level deviation from PCL (measured: profiles/r5_pcl_order_by_piece_1M.json);
 *   LGR_ARITH_PCL   pcl::FPFHEstimation::weightPointSPFHSignature as written: neighbours by ascending (squared distance, index),
 *                   val = hist * w rounded, float adds, double block sums of the vals (a sort per key point: several ms per 1M-point cloud).
 * Each mode is bit-identical to the oracle's mode of the same name (ORC_ARITH_CANONICAL / ORC_ARITH_PCL). */
enum { LGR_ARITH_FAST = 0, LGR_ARITH_PCL = 1 };

/* How the brute-force matcher runs (NEVER what it returns: every setting gives the same matches and distance bits).  The
 * defaults are the production schedule; the other values exist so that tests can drive every path at small sizes and so that
 * profiles can switch single mechanisms off.  Held by the context: lgr_ctx_set_match_options. */
typedef struct {
    int32_t prune;            /* exact bound-based tile skipping: -1 auto (on from 65536 x 65536 pairs), 0 off (dense), 1 on */
    int32_t leaves;           /* second-level k-means leaves per cluster: 0 auto (about 1024 rows per leaf), else 1 .. 64 */
    int32_t near;             /* pass-0 width: nearest leaves per row block / row blocks per leaf; 0 = default (40) */
    int32_t operand_format;   /* -1 auto (f16 two-term splits, rotated to 30 coordinates when the rows allow it), 0 f32, 1 f16, 2 f16 rotated */
    int32_t box_bounds;       /* bounding-box lower bounds beside the ball bounds: 1 PCA basis (default), 2 raw coordinates, 0 off */
    int32_t column_stage;     /* per-stage column criterion in the final schedule: 1 (default) / 0 */
    int32_t coarse_rejection; /* two-step coarse test in the final MFMA pass: 1 (default: unless the pass schedules more than half of all tiles --
                               * descriptors the bounds cannot separate --, then the plain six-step kernel), 2 (always), 0 (never) */
    int32_t rerank_refilter;  /* MFMA re-filter of the rerank's candidate groups: 1 (default) / 0 (whole-group exact scan) */
    int32_t pair_cap;         /* pairs per rerank item the re-filter may emit before falling back to the group scan: -1 default (8) */
    int32_t poison_tables;    /* diagnostics: fill never-computed minimum-table entries with 0 (nothing may read them) */
    int32_t self_check;       /* diagnostics: device check of the proven filter bound on sampled queries (lgr_match_last_check) */
    int32_t shell_bound;      /* radial shell bound per (row block, column stage) in the passes that have upper bounds: 1 (default) / 0 */
    int32_t split_sweep;      /* final pass of the rotated format as two kernels -- the coarse sweep appends the tiles it keeps to a list, a second
                               * kernel finishes them: 1 (default) / 0 (one fused kernel, round 3) */
    int32_t kept_cap;         /* capacity of that list in tiles: 0 default (16 M); a pass that keeps more is repeated on the fused kernel (tests: force it) */
    int32_t auto_dense;       /* 1 (default): when the bounds can separate (almost) nothing -- >= 90 % of the (row block, leaf) lower bounds are zero:
                               * descriptors without cluster structure -- pass 0 computes everything and the final pass finds an empty schedule, decided on
                               * the device (lgr_match_last_lbstats reports the counts).  0: never */
    int32_t irregular_rows;   /* 1 (default): the few finite rows whose three 11-bin block sums differ from the consensus of the sets (FPFH: an all-zero row of an
                               * isolated point among rows whose blocks sum to 100) are kept out of the MFMA filter -- one of them would cost both sets the
                               * rotated 30-coordinate format -- and matched by an exact side scan instead (lgr_match_last_irregular reports the counts; at
                               * most 1024 per side, else they stay in the filter as before).  0: never */
} lgr_match_options;

/* ---- context ---- */
int  lgr_version(void);
/* device: HIP ordinal.  stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; NULL is HIP's null
 * stream, which is what torch uses by default) or LGR_STREAM_OWN -> the ctx creates and owns a non-blocking stream */
#define LGR_STREAM_OWN ((void*) (intptr_t) -1)
int  lgr_ctx_create(int device, void* stream, lgr_ctx** out);
int  lgr_ctx_destroy(lgr_ctx* ctx);
int  lgr_ctx_sync(lgr_ctx* ctx);
const char* lgr_last_error(lgr_ctx* ctx);
void lgr_default_params(lgr_params* p);                      /* defaults of src/common.cpp:216-223,335-413 */
/* on-device stage timers of the last lgr_align*/
int  lgr_ctx_stage_ms(lgr_ctx* ctx, float* out12);
void lgr_match_default_options(lgr_match_options* opt);
/* opt == NULL restores the defaults.  Applies to every later matcher call of this context, stand-alone or inside lgr_align*. */
int  lgr_ctx_set_match_options(lgr_ctx* ctx, const lgr_match_options* opt);
int  lgr_ctx_get_match_options(lgr_ctx* ctx, lgr_match_options* opt);
/* device bytes the context's workspace currently holds (the sum of its grown-on-demand buffers): what a caller sizes its own
 * HBM budget against when it pushes many pairs of different sizes through one context (src/main.cpp:384-407 loops pairs in
 * one process) */
int  lgr_ctx_workspace_bytes(lgr_ctx* ctx, uint64_t* bytes);
void lgr_ctx_default_options(lgr_ctx_options* opt);
/* opt == NULL restores the defaults.  Waits for the context's queued work; call it between alignments, not during one. */
int  lgr_ctx_set_options(lgr_ctx* ctx, const lgr_ctx_options* opt);
int  lgr_ctx_get_options(lgr_ctx* ctx, lgr_ctx_options* opt);
/* host threads this context drives the device from: 1 (the caller's) + the helper threads it has started so far (at most 2) */
int  lgr_ctx_host_threads(lgr_ctx* ctx, int* n);

/* ---- building block under every grid / voxel / placement step (the reference has no counterpart: its containers are hash maps and
 *      kd-trees): stable LSD radix sort of (key, 32-bit value) pairs on the context's stream, out of place (in != out, input kept).
 *      u32: key bits [begin_bit, end_bit).  u64: bit ranges (shift, width), least significant first; bits outside the ranges must be
 *      equal in all keys.  Exported for tests and for callers that build their own orderings on the device. ---- */
int lgr_sort_pairs_u32_dev(lgr_ctx*, const uint32_t* d_keys_in, uint32_t* d_keys_out, const int32_t* d_vals_in, int32_t* d_vals_out,
                           size_t n, int begin_bit, int end_bit);
int lgr_sort_pairs_u64_dev(lgr_ctx*, const uint64_t* d_keys_in, uint64_t* d_keys_out, const int32_t* d_vals_in, int32_t* d_vals_out,
                           size_t n, const int* shifts, const int* widths, int n_ranges);

/* ---- include/common.h:266-280 calculateBoundingBox ---- */
int lgr_bbox_dev(lgr_ctx*, const float* d_pts, int n, float* d_min3_max3 /* 6 floats */);

/* ---- loader preprocessing: the steps of loadPointClouds (src/common.cpp:429-470) after the PLY reader:
 *      filterDuplicatePoints (:417-427), intensity = 1, voxel grid at 2 x calculatePointCloudDensity (:453-456,
 *      include/common.h:288), estimateNormalsPoints(30).  out holds n points, out != pts.  vp3 NULL -> origin.
 *      order (host entry): LGR_ORDER_REFERENCE reproduces the libstdc++ unordered_set / unordered_map output order. ---- */
int lgr_preprocess(lgr_ctx*, const float* pts, int n, const float* vp3, int normals_available, int order, float* out, int* n_out, float* voxel_out);
int lgr_preprocess_dev(lgr_ctx*, const float* d_pts, int n, const float* vp3 /* host */, int normals_available, float* d_out, int* n_out /* host */,
                       float* voxel_out /* host, or NULL */);
/* src/common.cpp:417-427 filterDuplicatePoints (first occurrence of every exact xyz, input order; intensity := 1 as :446-451) */
int lgr_dedupe_dev(lgr_ctx*, const float* d_pts, int n, float* d_out, int* n_out /* host */);
/* src/common.cpp:202-208 calculatePointCloudDensity(pcd, quantile) */
int lgr_cloud_density_dev(lgr_ctx*, const float* d_pts, int n, float quantile, float* out /* host */);

/* ---- include/common.h:304-310 detectKeyPoints(pcd, parameters, iss_radius) with keypoint_id = iss
 *      (src/common.cpp:657-691: pcl::ISSKeypoint3D, salient = non-max radius = iss_radius, thresholds 0.975,
 *      min_neighbors 4; the reference passes gamma/min_neighbors as constants, they are arguments here).
 *      idx must hold n entries; ascending point indices. ---- */
int lgr_iss_keypoints(lgr_ctx*, const float* pts, int n, float radius, float gamma21, float gamma32, int min_neighbors,
                      int32_t* idx, int* n_out);
int lgr_iss_keypoints_dev(lgr_ctx*, const float* d_pts, int n, float radius, float gamma21, float gamma32, int min_neighbors,
                          int32_t* d_idx, int* n_out /* host */);

/* ---- include/downsample.h:32 downsamplePointCloud(pcd, pcd_down, voxel_size)  (src/downsample.cpp:5-41) ----
 * out may alias the input (the reference passes the same cloud, src/common.cpp:455-456).  n_out <= n.
 * order: LGR_ORDER_REFERENCE reproduces the libstdc++ unordered_map iteration order (host post-pass),
 *        LGR_ORDER_CANONICAL = voxels sorted by (iz,iy,ix). */
int lgr_downsample(lgr_ctx*, const float* pts, int n, float voxel, int order, float* out, int* n_out);
int lgr_downsample_dev(lgr_ctx*, const float* d_pts, int n, float voxel, float* d_out, int* n_out /* host */);

/* ---- src/common.cpp:644-655 estimateNormalsPoints(k, pcd, surface, vp, normals_available) ----
 * writes normal_x/y/z + curvature of pts in place; surf NULL -> pts is its own surface; vp NULL -> origin */
int lgr_normals_knn(lgr_ctx*, float* pts, int n, const float* surf, int ns, int k, const float* vp3, int normals_available);
int lgr_normals_knn_dev(lgr_ctx*, float* d_pts, int n, const float* d_surf, int ns, int k, const float* vp3 /* host */, int normals_available);

/* ---- include/common.h:322-332 estimateFeatures<FPFH>(kps, surface, features, radius, params) ----
 * LGR_ERR_UNSUPPORTED for radius > 1e18 (the weighting kernel's reciprocal is checked for squared distances up to 1e36) and for
 * more than 2^32 / 48 - 2 surface points (32-bit row offsets into the SPFH table). */
int lgr_fpfh(lgr_ctx*, const float* kps, int m, const float* surf, int n, float radius, float* out_m_x_33);
int lgr_fpfh_dev(lgr_ctx*, const float* d_kps, int m, const float* d_surf, int n, float radius, float* d_out);
/* Device self-check of the FPFH weighting kernel's reciprocal (v_rcp_f32 + one Newton step in place of the IEEE division sequence;
 * include/common.h:322-332 -> pcl::FPFHEstimation::weightPointSPFHSignature's 1.0f / dists[idx]): every float whose bit pattern lies in
 * [lo_bits, hi_bits] goes through both; out2[0] = values where they differ (must be 0 on [1e-36, 1e36], the range the kernel uses it on),
 * out2[1] = values tested. */
int lgr_selfcheck_rcp(lgr_ctx*, unsigned lo_bits, unsigned hi_bits, unsigned long long* out2);
/* the named libm's float routines as the device evaluates them (csrc/lgr_libm.cuh: GNU libc 2.35's acosf / atanf / atan2f / sinf / cosf restated op
 * for op), element-wise on host arrays: fn 0 acosf(a), 1 atanf(a), 2 atan2f(a, b), 3 sinf(a), 4 cosf(a) (sinf / cosf: |a| < 120).  A host
 * can compare them with its own libm (tests/test_gpu_pcl_arith.py compares with the oracle's restatement, which is pinned against glibc). */
int lgr_selfcheck_libm(lgr_ctx*, int fn, const float* a, const float* b, long long n, float* out);

/* ---- include/matching.h:373-376 matchBF<FPFH>(query, train, params), randomness = 1 ----
 * idx[i] = matched train row or -1 (invalid / NaN query), dist[i] = L2 distance (sqrt) */
int lgr_match_bf(lgr_ctx*, const float* q33, int mq, const float* t33, int mt, int block, int32_t* idx, float* dist);
int lgr_match_bf_dev(lgr_ctx*, const float* d_q33, int mq, const float* d_t33, int mt, int block, int32_t* d_idx, float* d_dist);
/* both directions in one MFMA pass (what LeftToRight/Cluster matchers need, include/matching.h:431-432,495-496) */
int lgr_match_bf2_dev(lgr_ctx*, const float* d_a33, int ma, const float* d_b33, int mb, int block,
                      int32_t* d_ab_idx, float* d_ab_dist, int32_t* d_ba_idx, float* d_ba_dist);
/* ---- include/matching.h:373-376 matchFLANN<FPFH>(query_features, train_features, parameters), randomness 1 (:565-592):
 *      pcl::KdTreeFLANN is an exact search, so the nearest row is the one matchBF finds (the reference's own test asserts that,
 *      tests/flann_bf_matcher.h:82-83); what differs is the reported distance: sqrt of FLANN's L2_Simple (sequential sum of
 *      squares) instead of OpenCV's lane-ordered norm.  idx = -1 for invalid query rows. ---- */
int lgr_match_flann(lgr_ctx*, const float* q33, int mq, const float* t33, int mt, int32_t* idx, float* dist);
int lgr_match_flann_dev(lgr_ctx*, const float* d_q33, int mq, const float* d_t33, int mt, int32_t* d_idx, float* d_dist);
/* ---- include/matching.h:383-387 matchLocal<FPFH>(query_pcd, train_tree, query_features, train_features, parameters, guess)
 *      (:637-678): for every valid query row, the train row with the nearest descriptor (pcl::L2_Norm: sequential sum, sqrtf)
 *      among the train POINTS within match_search_radius of guess * query point (strict d2 < r*r; FLT_MAX radius = all points);
 *      equal descriptor distances: the spatially nearer point, then the lower index (KNNResult keeps the first, radiusSearch
 *      visits by ascending distance).  guess16: column-major, host. ---- */
int lgr_match_local(lgr_ctx*, const float* query_pts, int mq, const float* train_pts, int mt, const float* q33, const float* t33,
                    const float guess16[16], float match_search_radius, int32_t* idx, float* dist);
int lgr_match_local_dev(lgr_ctx*, const float* d_query_pts, int mq, const float* d_train_pts, int mt, const float* d_q33, const float* d_t33,
                        const float guess16[16] /* host */, float match_search_radius, int32_t* d_idx, float* d_dist);

/* diagnostics of the context's last match call (stand-alone or inside lgr_align*), kept in the context:
 * [items_ab, dense_ab, items_ba, dense_ba, sub_cols, rg_rows] and the duration of
 * its MFMA filter kernel (hipEvents on the ctx stream) -- what bench.py's roofline object is computed from */
int lgr_match_last_stats(lgr_ctx*, unsigned* out6);
int lgr_match_last_kernel_ms(lgr_ctx*, float* ms);
/* fraction of the (256-row block x 128-column stage) tiles the MFMA passes of the last match call computed; the exact
 * bound-based skipping leaves the rest out.  1.0 = dense.  Every (row block, stage) counts once, so the figure is
 * never above 1; lgr_match_last_issued sums the passes (a stage that straddles two leaves may be computed by two passes): the work
 * that was issued, >= the executed fraction. */
int lgr_match_last_work(lgr_ctx*, double* executed_fraction);
int lgr_match_last_issued(lgr_ctx*, double* out2 /* [0] issued fraction, [1] issued (row, column) element pairs of the padded operands */);
/* coarse rejection inside the MFMA filter kernel (rotated format only; lgr_match_options.coarse_rejection = 0 turns it off): 32 x 32 tiles
 * tested after their first two MFMA steps in the last match call, and tiles abandoned there. */
int lgr_match_last_coarse(lgr_ctx*, double* out2);
/* shell test inside the same sweep (lgr_match_options.shell_bound): 32 x 32 tiles of the swept stages a wave left out before any MFMA step,
 * because the radial shells of its rows and of the tile's columns about their cluster centre are farther apart than every upper bound */
int lgr_match_last_shell(lgr_ctx*, double* tiles_skipped);
/* exact rerank (f16 operand formats; lgr_match_options.rerank_refilter = 0 turns it off): (query, train row) pairs the MFMA re-filter of
 * the candidate groups passed on to the exact distance in the last match call, query->train and train->query direction; a
 * count above the pair buffer (8 per candidate group) means that direction fell back to the exact scan of whole groups. */
int lgr_match_last_pairs(lgr_ctx*, unsigned* out2);
/* MFMA operand format of the last match call: 1 = two-term f16 splits on v_mfma_f32_32x32x16_f16, K = 112; 2 = the same on
 * 30 Helmert coordinates, K = 96 (chosen when every 11-bin block of all rows has the same sum, as FPFH rows do); 0 = f32
 * operands on v_mfma_f32_32x32x2_f32 (lgr_match_options.operand_format selects one explicitly).  Results do not depend on it. */
int lgr_match_last_format(lgr_ctx*, int* f16);
/* (row block, leaf) pairs whose lower bound is zero / finite in the last pruned match call (lgr_match_options.auto_dense) */
int lgr_match_last_lbstats(lgr_ctx*, double* out2);
/* irregular rows of the last match call (lgr_match_options.irregular_rows): [0] query side, [1] train side: rows that were matched by the
 * exact side scan instead of the MFMA filter (0 when the lane was off, found no consensus among the block sums, or gave up); [2] = 1 when it
 * gave up (more than 1024 such rows on a side: the call was rebuilt with every finite row in the filter). */
int lgr_match_last_irregular(lgr_ctx*, unsigned* out3);
/* self-check of the matcher's filter bound (lgr_match_options.self_check = 1, test sizes): worst |filtered - exact| / eps over
 * sampled table entries of the last match call, rows then columns; -1 = not run.  Must be <= 1. */
int lgr_match_last_check(lgr_ctx*, double* out2);
/* ---- src/common.cpp:531-547 calculateSmoothedDensities(pcd, k) / :202-208 calculatePointCloudDensity ---- */
int lgr_smoothed_densities(lgr_ctx*, const float* pts, int n, int k, float* out);
int lgr_smoothed_densities_dev(lgr_ctx*, const float* d_pts, int n, int k, float* d_out);
int lgr_knn_dev(lgr_ctx*, const float* d_q, int nq, const float* d_pts, int n, int k, int32_t* d_idx, float* d_d2);

/* ---- include/matching.h:395-411 / 428-453 / 492-550 match_impl of OneSided / LeftToRight / Cluster matcher ---- */
int lgr_filter_dev(lgr_ctx*, int matching_id, const float* d_src, int ns, const float* d_tgt, int nt,
                   const int32_t* d_ij_idx, const float* d_ij_dist, const int32_t* d_ji_idx, const float* d_ji_dist,
                   float distance_thr, int cluster_k, lgr_corr* d_out, int* n_out /* host */);

/* ---- include/correspondence_search.h:14-28 FeatureBasedCorrespondenceSearch::calculateCorrespondences (keypoint any) ---- */
int lgr_correspondences(lgr_ctx*, const float* src, int ns, const float* tgt, int nt, const lgr_params*, lgr_corr* out, int* n_out);
int lgr_correspondences_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_params*, lgr_corr* d_out, int* n_out /* host */);

/* ---- include/sac_prerejective_omp.h:21-56 SampleConsensusPrerejectiveOMP(src,tgt,corrs,params).align() ----
 * final_mask (optional): c bytes, inlier mask of the refit transform */
int lgr_ransac(lgr_ctx*, const float* src, int ns, const float* tgt, int nt, const lgr_corr* corr, int c,
               const lgr_params*, lgr_result*, uint8_t* final_mask);
int lgr_ransac_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                   const lgr_params*, lgr_result* /* host */, uint8_t* d_final_mask);
/* replay mode (SURVEY section 7 "RANSAC RNG"): evaluate n caller-supplied sample tuples (params.n_samples correspondence indices each) */
int lgr_ransac_replay_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                          const lgr_params*, const int32_t* d_triples, int n,
                          uint8_t* d_ok, float* d_T16, int32_t* d_n_inliers, float* d_metric);
/* the on-device sampler alone: triples of iterations [first, first+n) (Philox4x32-10 + selectCorrespondences :33-77) */
int lgr_ransac_samples_dev(lgr_ctx*, uint64_t seed, int first, int n, int n_corr, int32_t* d_triples);
/* ... for any n_samples in 3..8: draw j of an iteration is word j % 4 of Philox(key = seed, counter = (iteration, j / 4, 0, 0)) >> 1 */
int lgr_ransac_samples_n_dev(lgr_ctx*, uint64_t seed, int first, int n, int n_corr, int n_samples, int32_t* d_tuples);
/* one Philox4x32-10 block from the device's generator (the sampler above uses counter = (iteration, 0, 0, 0), key = seed; the closest-plane
 * metric's subsets the full counter): key = (k0 | k1 << 32).  For Random123's known-answer vectors. */
int lgr_selfcheck_philox(lgr_ctx*, uint64_t key, const uint32_t counter4[4], uint32_t out4[4]);
/* src/metric.cpp:125-179 buildInliersAndEstimateMetric for one transform */
int lgr_evaluate_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                     const float T16[16] /* host */, int metric_id, int score_id,
                     uint8_t* d_mask, int* n_inliers, float* rmse, float* metric /* host outs */);

/* ---- ClosestPlaneMetricEstimator::buildInliersAndEstimateMetric (src/metric.cpp:10-53,181-199) of one transform on the
 *      sparse 1 % subset RANSAC uses (src/sac_prerejective_omp.cpp:109); the subset of hypothesis `counter` is defined by
 *      Philox.  threshold = calculatePointCloudDensity(tgt).  pairs (optional, 2 ints per inlier, room for
 *      0.01 * ns pairs): (source index, nearest target index), ascending source index.  In lgr_ransac / lgr_align:
 *      params.metric_id = LGR_METRIC_CLOSEST_PLANE or LGR_METRIC_COMBINATION. ---- */
int lgr_evaluate_plane_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const float T16[16] /* host */, int score_id,
                           uint64_t seed, uint32_t counter, int* n_inliers, float* rmse, float* metric, float* threshold /* or NULL */,
                           int32_t* pairs /* host, or NULL */, int* n_pairs);

/* ---- include/transformation.h:6-7 estimateOptimalRigidTransformation(src, tgt, inliers, T) ---- */
int lgr_refit_svd(lgr_ctx*, const float* src, const float* tgt, int ns, int nt, const lgr_corr* inliers, int n, float T16[16]);
int lgr_refit_svd_dev(lgr_ctx*, const float* d_src, const float* d_tgt, const lgr_corr* d_corr, int c,
                      const uint8_t* d_mask /* NULL: all */, float T16[16] /* host */);

/* ---- include/alignment.h:18-19 alignPointClouds(src, tgt, params) (src/alignment.cpp:72-109, no CSV side effects);
 *      alignRansac (:14-19) is lgr_ransac; alignGror (:21-35) via params.alignment_id ---- */
int lgr_align(lgr_ctx*, const float* src, int ns, const float* tgt, int nt, const lgr_params*, lgr_result*);
int lgr_align_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_params*, lgr_result* /* host */);

/* ---- alignGror(src, tgt, correspondences, parameters) (src/alignment.cpp:21-35) =
 *      pcl::registration::GRORInitialAlignment::computeTransformation (include/gror/ia_gror.hpp:367-415) with
 *      setResolution(distance_thr), setOptimalSelectionNumber(800).  Result: transformation, iterations = 1,
 *      converged = 1 (as alignGror reports), n_inliers = inliers of the refinement (:261-293), metric = size of the
 *      maximum consistent set (best_count_), best_iteration = rows that reached the tight-constraint stage,
 *      estimated_iters = K.  inlier_mask (c bytes) optional.  Tie orders of the three std::sort calls: DESIGN.md. ---- */
int lgr_gror(lgr_ctx*, const float* src, int ns, const float* tgt, int nt, const lgr_corr* corr, int c,
             float resolution, int k_optimal, lgr_result* res, uint8_t* inlier_mask /* host, or NULL */);
int lgr_gror_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                 float resolution, int k_optimal, lgr_result* res /* host */, uint8_t* d_inlier_mask /* or NULL */);
/* node degrees of optimalSelectionBasedOnNodeReliability (include/gror/ia_gror.hpp:126-170), c int32 on the device */
int lgr_gror_node_degree_dev(lgr_ctx*, const float* d_src, const float* d_tgt, const lgr_corr* d_corr, int c,
                             float resolution, int32_t* d_degree);

/* ---- include/hypotheses.h:10-16 (host bookkeeping; compiled out in the reference, SAVE_MULTIPLE_HYPOTHESES false) ---- */
/* include/hypotheses.h:14-16 chooseBestHypothesis(src, tgt, correspondences, params, tns) (src/hypotheses.cpp:50-129), the
 * decision only: the hypothesis with the largest inlier uniformity (identity / index -1 when none is positive) */
int lgr_choose_best_hypothesis_dev(lgr_ctx*, const float* d_src, int ns, const float* d_tgt, int nt, const lgr_corr* d_corr, int c,
                                   const float* tns16 /* host, n x 16 */, int n, float T_out16[16], int* best_index, float* uniformities /* host, n, or NULL */);
int lgr_update_hypotheses(float* tns16, float* metrics, int n, int cap, const float* new_T16, float new_metric, float distance_thr);

#ifdef __cplusplus
}
#endif
#endif
