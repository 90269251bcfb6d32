/*
 * lgr_oracle.h -- C API of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * This library is a CPU restatement of the hot path of
 * aleksandrina-streltsova/lidar-global-registration (reference file:line cited at
 * every function in src/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (liblgr_hip.so) never links or calls it.
 *
 * PARITY STATUS: pinned against the reference's own code where it can be compiled here (oracle/_ref =
 * src/utils.cpp + src/csv_parser.cpp in place: RNG stream, iteration cap, container hashes, CSV tokeniser, number formatting --
 * tests/test_oracle_ref.py), against its golden vectors (KNNResult, tests/knn_result.cpp:30-51) and against its end-to-end
 * acceptance tests (tests/point2plane_distance.cpp:29-96, tests/flann_bf_matcher.h:70-96 -- tests/test_oracle_acceptance.py).
 * "Parity unpinned" for the third-party arithmetic the reference ships no fixtures for and cannot run here (PCL FPFH / normals /
 * polygon rejector / umeyama, OpenCV batchDistance, GROR): restated from the pinned upstream versions and property-tested
 * (tests/test_oracle_*.py).
 *
 * Layouts (all host pointers):
 *   point  : 12 floats = pcl::PointXYZINormal {x,y,z,1 | nx,ny,nz,0 | intensity,curvature,pad,pad}
 *   fpfh   : 33 floats (pcl::FPFHSignature33)
 *   corr   : lgr_orc_corr {int query; int match; float distance; float threshold}  (include/common.h:120-127)
 *   T      : 16 floats, COLUMN-major 4x4 (Eigen::Matrix4f default)
 */
#ifndef LGR_ORACLE_H
#define LGR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t query, match; float distance, threshold; } lgr_orc_corr;

enum { ORC_ORDER_LIBSTDCXX = 0, ORC_ORDER_CANONICAL = 1 };
/* arithmetic of the third-party pieces: the canonical orders the HIP path restates, or PCL 1.12.1's own (measurement only) */
enum { ORC_ARITH_ROUND4 = 0,        /* none of the pieces below: the canonical orders of rounds 1-4 (kept for measurement: tools/pcl_order_report.py) */
       ORC_ARITH_PCL_EIGEN33 = 1,   /* normals: pcl::eigen33 (closed form; atan2f / cosf / sinf of glibc 2.35, src/orc_libm.h) instead of the Jacobi solver */
       ORC_ARITH_PCL_ACOS = 2,      /* pair features: swap test acosf(|a1|) > acosf(|a2|) of glibc 2.35 instead of the comparison of the arguments */
       ORC_ARITH_PCL_W_ORDER = 4,   /* FPFH weighting: neighbours by ascending (d2, index) instead of grid order */
       ORC_ARITH_PCL_W_ROUND = 8,   /* FPFH weighting: val = hist * w rounded, then added, instead of one fused chain per bin */
       ORC_ARITH_PCL_W_NORM = 16,   /* FPFH weighting: block normaliser from the running double sum of the vals instead of the finished bins */
       ORC_ARITH_PCL_ATAN2 = 32,    /* pair features: f1 by atan2f of glibc 2.35 instead of the round-4 polynomial */
       ORC_ARITH_PCL_LIBM = 2 | 32, ORC_ARITH_PCL_WEIGHTING = 4 | 8 | 16,
       ORC_ARITH_CANONICAL = 1 | 2 | 32,   /* DEFAULT since round 5 = what liblgr_hip.so restates by default (LGR_ARITH_FAST): PCL's own normals and pair
                                            * features; only the weighting keeps the grid-order fused chain that runs on the matrix cores */
       ORC_ARITH_PCL = 63 };               /* every piece = PCL 1.12.1's own arithmetic = lgr_ctx_options.arithmetic LGR_ARITH_PCL */
enum { ORC_METRIC_CORRESPONDENCES = 0, ORC_METRIC_UNIFORMITY = 1, ORC_METRIC_CLOSEST_PLANE = 2, ORC_METRIC_COMBINATION = 3 };
enum { ORC_SCORE_CONSTANT = 0, ORC_SCORE_MAE = 1, ORC_SCORE_MSE = 2, ORC_SCORE_EXP = 3 };
enum { ORC_MATCH_LR = 0, ORC_MATCH_ONE_SIDED = 1, ORC_MATCH_CLUSTER = 2 };
enum { ORC_RNG_MT19937_LEMIRE = 0, ORC_RNG_MT19937_REJECT = 1, ORC_RNG_PHILOX = 2 };
enum { ORC_KEYPOINT_ANY = 0, ORC_KEYPOINT_ISS = 1 };

typedef struct {
    /* mirrors AlignmentParameters (include/common.h:135-163), enum ids instead of strings */
    int   feature_nr_points;   /* 352 */
    int   normal_nr_points;    /* 30 */
    float edge_thr_coef;       /* 0.95 */
    float distance_thr;
    float feature_radius;      /* must be > 0 (single scale) */
    float scale_factor;        /* 2.0 */
    float confidence;          /* 0.999 */
    int   bf_block_size;       /* 10000 struct default, 200000 in shipped yaml */
    int   cluster_k;           /* 40 */
    int   n_samples;           /* 3 (only 3 supported) */
    int   matching_id;         /* ORC_MATCH_* */
    int   metric_id;           /* ORC_METRIC_* */
    int   score_id;            /* ORC_SCORE_* */
    int   max_iterations;
    int   normals_available;   /* 0 */
    int   has_vp_src, has_vp_tgt;
    float vp_src[3], vp_tgt[3];
    /* RANSAC schedule */
    int   keypoint_id;         /* ORC_KEYPOINT_* (0 = any: every point is a key point) */
    float iss_radius_src, iss_radius_tgt;
    int   rng_mode;            /* ORC_RNG_* */
    int   n_threads;           /* reference schedule (mt19937 modes): emulated OpenMP team size */
    int   batch_size;          /* philox schedule: iterations per batch */
    uint64_t seed;             /* 566 */
    /* matcher dispatch of match_multiscale (include/matching.h:300-312) + the RANSAC seed (src/sac_prerejective_omp.cpp:134-147) */
    int   use_bfmatcher;       /* 1 (ALIGNMENT_USE_BFMATCHER); 0 -> matchFLANN */
    int   has_guess;           /* 1 -> matchLocal around guess * p, and the guess is RANSAC's first hypothesis */
    float match_search_radius;
    float guess[16];           /* column-major */
} lgr_orc_params;

typedef struct {
    float T[16];               /* column-major */
    int   iterations;          /* ransac_iterations */
    int   converged;
    int   n_inliers;           /* of the final (refit) transform */
    float metric;              /* of the final transform */
    float best_metric_before_refit;
    int   best_iteration;      /* philox schedule: iteration index of the winning hypothesis */
    int   num_rejections;
    int   estimated_iters;
} lgr_orc_result;

void orc_default_params(lgr_orc_params* p);
int  orc_num_threads(void);
void orc_set_num_threads(int n);
void orc_set_arith_mode(int mode);   /* process-wide bit mask; ORC_ARITH_CANONICAL by default */
int  orc_arith_mode(void);
/* src/orc_libm.h: fn 0 acosf(a), 1 atanf(a), 2 atan2f(a, b), 3 sinf(a), 4 cosf(a) (sinf / cosf: |a| < 120) */
int  orc_libm_eval(int fn, const float* a, const float* b, long n, float* out);
long orc_libm_check_range(int fn, uint32_t lo_bits, uint32_t hi_bits);   /* results differing from the running libm's */
long orc_libm_check_atan2(const float* y, const float* x, long n);

/* include/common.h:266-280 (FLT_MIN-initialised max quirk reproduced) */
int orc_bbox(const float* pts, int n, float* mn3, float* mx3);

/* src/downsample.cpp:5-41, include/downsample.h:6-30.  out must hold n points. */
int orc_downsample(const float* pts, int n, float voxel, int order, float* out, int* n_out);

/* src/common.cpp:644-655,593-628 + PCL NormalEstimationOMP (k-NN).  Writes nx,ny,nz,curvature of pts in place.
 * surf==NULL -> surface = pts itself. vp==NULL -> (0,0,0). */
int orc_normals_knn(float* pts, int n, const float* surf, int ns, int k, const float* vp, int normals_available);

/* include/common.h:322-332 -> pcl::FPFHEstimationOMP.  libm_mode=1 uses libm atan2f instead of the canonical polynomial. */
int orc_fpfh(const float* kps, int m, const float* surf, int n, float radius, float* out33, int libm_mode);
/* SPFH rows (n x 33) for every surface point, for stage-wise tests. */
int orc_spfh(const float* surf, int n, float radius, float* out33, int libm_mode);

/* include/matching.h:594-634 + cv::BFMatcher(NORM_L2) knnMatch k=1 + src/common.cpp:517-529. idx=-1: no match */
int orc_match_bf(const float* q33, int mq, const float* t33, int mt, int block, int* idx, float* dist);
/* same, only for the queries listed in qsel (bounded CPU baseline sample) */
int orc_match_bf_subset(const float* q33, const int* qsel, int nsel, const float* t33, int mt, int block, int* idx, float* dist);

/* include/matching.h:565-592 matchFLANN (pcl::KdTreeFLANN<FPFH>, exact search, randomness 1): nearest train row under FLANN's
 * L2_Simple (sequential sum of squared differences), distance = sqrt of it; ties -> lowest index (FLANN: unspecified). */
int orc_match_flann(const float* q33, int mq, const float* t33, int mt, int* idx, float* dist);
/* include/matching.h:637-678 matchLocal: query points moved by guess (pcl::transformPointCloudWithNormals), radius search in the
 * train cloud (strict d2 < r*r, visited by ascending (d2, index)), nearest valid train descriptor by pcl::L2_Norm (sequential
 * sum, sqrtf), KNNResult k = 1 (first of equal distances stays). guess16 column-major. */
int orc_match_local(const float* qpts, int mq, const float* tpts, int mt, const float* q33, const float* t33, const float* guess16,
                    float radius, int* idx, float* dist);
/* inverse of a 4x4 (column-major) by Gauss-Jordan with partial pivoting in double, rounded to float: the canonical stand-in for
 * Eigen's Matrix4f::inverse() (include/matching.h:293) */
void orc_inverse4(const float* m16, float* out16);

/* exact k-NN in 3-D, sorted by (d2, index); idx/d2 are n*k. Used by several stages and by tests. */
int orc_knn(const float* qpts, int nq, const float* pts, int n, int k, int* idx, float* d2);

/* src/common.cpp:531-547 */
int orc_smoothed_densities(const float* pts, int n, int k, float* out);
/* src/common.cpp:202-208 */
int orc_cloud_density(const float* pts, int n, float quantile, float* out);

/* include/matching.h:395-411 / 428-453 / 492-550 : build correspondences from the two directional 1-NN tables.
 * out must hold ns entries. */
int orc_filter(int matching_id, const float* src, int ns, const float* tgt, int nt,
               const int* ij_idx, const float* ij_dist, const int* ji_idx, const float* ji_dist,
               float distance_thr, int cluster_k, lgr_orc_corr* out, int* n_out);

/* loader preprocessing, src/common.cpp:417-427 (filterDuplicatePoints) and :446-459 (weights, 2 x density voxel grid,
 * normals); out holds n points */
int orc_dedupe(const float* pts, int n, int order, float* out, int* n_out);
int orc_preprocess(const float* pts, int n, const float* vp3, int normals_available, int order, float* out, int* n_out, float* voxel_out);

/* src/common.cpp:657-691 detectKeyPoints with keypoint_id = iss (pcl::ISSKeypoint3D, salient = non-max radius,
 * thresholds 0.975, min_neighbors 4): ascending indices.  out_idx holds n ints; third (optional) n doubles */
int orc_iss_keypoints(const float* pts, int n, float radius, float gamma21, float gamma32, int min_neighbors,
                      int* out_idx, int* n_out, double* third);
void orc_eigvals3d(const double S6[6] /* a00 a01 a02 a11 a12 a22 */, double ev3[3] /* ascending */);

/* src/correspondence_search.cpp:4-15 (keypoint 'any' or 'iss'): whole correspondence search */
int orc_correspondences(const float* src, int ns, const float* tgt, int nt, const lgr_orc_params* p,
                        lgr_orc_corr* out, int* n_out, double* stage_seconds /* 8 doubles or NULL */);

/* ---- RANSAC pieces ---- */
/* include/utils.h:13-26: first n outputs of UniformRandIntGenerator(0,INT_MAX,seed) */
int orc_rng_stream(int rng_mode, uint64_t seed, int n, int* out);
/* restatements pinned against the reference's own code in oracle/_ref (tests/test_oracle_ref.py) */
int orc_comb_or_max(int n, int k);                       /* include/utils.h:34-43 */
uint64_t orc_voxel_hash(int ix, int iy, int iz);         /* include/common.h:212-223 over include/utils.h:28-32 semantics */
uint64_t orc_point_hash(float x, float y, float z);      /* include/common.h:202-210 */
/* Philox4x32-10: counter=(iter,0,0,0) key=(seed_lo,seed_hi) -> 4 words */
void orc_philox(uint64_t seed, uint32_t iter, uint32_t out[4]);
void orc_philox_full(uint64_t key, const uint32_t counter4[4], uint32_t out[4]);   /* key = k0 | k1 << 32 */
/* src/sac_prerejective_omp.cpp:33-77 given the three raw draws r[3] (already non-negative) */
void orc_select3(const int r[3], int n_corr, int sample[3]);
#define ORC_MAX_SAMPLES 8   /* n_samples this restatement's local arrays hold (the reference's code is generic in it; 3 everywhere it ships) */
void orc_select_n(const int* r, int nr_samples, int n_corr, int* sample);
void orc_philox_draws(uint64_t seed, uint32_t iter, int nr_samples, int* r);
/* pcl CorrespondenceRejectorPoly::thresholdPolygon (call site src/sac_prerejective_omp.cpp:214) */
int orc_poly_ok(const float* src, const float* tgt, const int sidx[3], const int tidx[3], float edge_thr);
int orc_poly_ok_n(const float* src, const float* tgt, const int* sidx, const int* tidx, int n, float edge_thr);
/* pcl TransformationEstimationSVD (umeyama, no scaling) on 3 pairs (call site :220) */
void orc_umeyama3(const float* src, const float* tgt, const int sidx[3], const int tidx[3], float T[16]);
void orc_umeyama_n(const float* src, const float* tgt, const int* sidx, const int* tidx, int n, float T[16]);
/* src/metric.cpp:125-179 : inlier mask (n_corr bytes), rmse, metric for transform T */
int orc_evaluate(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
                 const float T[16], int metric_id, int score_id,
                 uint8_t* mask, int* n_inl, float* rmse, float* metric);
/* src/metric.cpp:10-53,181-199 closest-plane metric of one transform on the Philox-defined sparse subset (counter = the
 * hypothesis / iteration index); pairs (optional, 2 ints per inlier): (source index, nearest target index), ascending */
int orc_evaluate_plane(const float* src, int ns, const float* tgt, int nt, const float T[16], int score_id, uint64_t seed,
                       uint32_t counter, int* n_inl, float* score, float* rmse, float* metric, float* thr, int* pairs);
/* src/metric.cpp:103-123 */
int orc_estimate_max_iterations(const float* src, const float* tgt, const lgr_orc_corr* corr, int c,
                                const float T[16], float confidence, int nr_samples);
/* src/transformation.cpp:4-38 on the inliers flagged in mask */
int orc_refit(const float* src, const float* tgt, const lgr_orc_corr* corr, int c, const uint8_t* mask, float T[16]);
/* replay mode: evaluate a list of sample triples (3*n ints = correspondence indices, as produced by select3).
 * per hypothesis: ok flag, T (16 floats, identity if rejected), n_inliers, metric */
int orc_replay(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
               const lgr_orc_params* p, const int* triples, int n,
               uint8_t* ok, float* Ts, int* n_inl, float* metric);
/* src/sac_prerejective_omp.cpp:115-314 */
int orc_ransac(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
               const lgr_orc_params* p, lgr_orc_result* res, uint8_t* final_mask /* c bytes or NULL */);
/* src/alignment.cpp:72-109 (RANSAC branch, no CSV side effects) */
int orc_align(const float* src, int ns, const float* tgt, int nt, const lgr_orc_params* p,
              lgr_orc_result* res, lgr_orc_corr* corr_out, int* n_corr_out, double* stage_seconds);

/* include/gror/ia_gror.hpp (BASELINE config 5), driven like alignGror (src/alignment.cpp:21-35) */
int orc_gror_node_degree(const float* src, const float* tgt, const lgr_orc_corr* corr, int c, float resolution, int* degree);
int orc_gror(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
             float resolution, int K_optimal, float T16[16], int* diag8, float* best_angle);

/* src/hypotheses.cpp:50-129 (decision only): index of the hypothesis with the largest inlier uniformity, or -1 */
int orc_choose_best_hypothesis(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
                               const float* tns, int n, float T_out[16], float* uniformities);
/* src/hypotheses.cpp:14-48.  tns: n*16 (col-major) in/out, capacity cap. returns new n */
int orc_update_hypotheses(float* tns, float* metrics, int n, int cap, const float* new_T, float new_metric, float distance_thr);
/* src/analysis.cpp:19-24 */
void orc_rot_trans_diff(const float* T1, const float* T2, float* angle, float* tdist);

/* include/matching.h:44-94 KNNResult<float> exercised by tests/knn_result.cpp */
int orc_knnresult_run(int capacity, const float* dists, const int* indices, int n, int* out_idx, float* out_dist);

/* canonical elementary functions (exposed for property tests vs libm) */
float orc_atan2f(float y, float x);
float orc_logf(float x);
float orc_cbrtf(float x);
float orc_expf(float x);
void  orc_svd3(const float A[9] /*row-major*/, float U[9], float S[3], float V[9]);
void  orc_eig3_smallest(const float C[9] /*row-major symmetric*/, float* eval, float evec[3]);

#ifdef __cplusplus
}
#endif
#endif
