// orc_ransac.cpp -- ORACLE (test infrastructure): prerejective RANSAC, metrics, refit, hypotheses bookkeeping.
// Reference paths relative to /root/reference.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "../lgr_oracle.h"
#include "orc_grid.h"
#include "orc_math.h"

using namespace orc;

extern "C" void orc_default_params(lgr_orc_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->feature_nr_points = 352;   // FEATURE_NR_POINTS include/common.h:56
    p->normal_nr_points = 30;     // NORMAL_NR_POINTS  include/common.h:57
    p->edge_thr_coef = 0.95f;     // ALIGNMENT_EDGE_THR :38
    p->distance_thr = 0.1f;
    p->feature_radius = 0.25f;
    p->scale_factor = 2.0f;       // FEATURES_SCALE_FACTOR :47
    p->confidence = 0.999f;       // ALIGNMENT_CONFIDENCE :39
    p->bf_block_size = 10000;     // ALIGNMENT_BLOCK_SIZE :45
    p->cluster_k = 40;            // MATCHING_CLUSTER_K :53
    p->n_samples = 3;
    p->matching_id = ORC_MATCH_CLUSTER;
    p->metric_id = ORC_METRIC_UNIFORMITY;   // yaml default src/common.cpp:335-413
    p->score_id = ORC_SCORE_MSE;
    p->max_iterations = INT_MAX;
    p->rng_mode = ORC_RNG_PHILOX;
    p->n_threads = 8;
    p->batch_size = 65536;
    p->seed = 566;                // SEED include/common.h:25
    p->use_bfmatcher = 1;         // ALIGNMENT_USE_BFMATCHER :41
    for (int i = 0; i < 4; ++i) p->guess[5 * i] = 1.f;
}

// ---------------------------------------------------------------- RNG
// Philox4x32-10 (Salmon et al. 2011), counter = (iter, 0, 0, 0), key = (seed_lo, seed_hi).
static void philox4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
extern "C" void orc_philox(uint64_t seed, uint32_t iter, uint32_t out[4]) { philox4(seed, iter, 0, 0, 0, out); }
extern "C" void orc_philox_full(uint64_t key, const uint32_t c[4], uint32_t out[4]) { philox4(key, c[0], c[1], c[2], c[3], out); }   // Random123's known-answer vectors
static void philox4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t) seed, k1 = (uint32_t) (seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t) 0xD2511F53u * c0;
        uint64_t p1 = (uint64_t) 0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t) p1;
        uint32_t n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t) p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

namespace {
// include/utils.h:13-26 UniformRandIntGenerator(0, INT_MAX, seed) = uniform_int_distribution<int>(0, INT_MAX) over
// mt19937.  libstdc++ >= 11 (this container): Lemire path -> mt() >> 1, one draw.  libstdc++ <= 10 (the
// reference's CI, g++-9): rejection -> redraw until mt() < 2^31.  SURVEY.md A.7.
struct RefRng {
    std::mt19937 gen;
    std::uniform_int_distribution<int> dist{0, INT_MAX};
    int mode;
    RefRng(int m, uint64_t seed) : gen((std::mt19937::result_type) seed), mode(m) {}
    int operator()() {
        if (mode == ORC_RNG_MT19937_LEMIRE) return dist(gen);   // literal libstdc++-11 code path
        for (;;) { uint32_t v = gen(); if (v < 0x80000000u) return (int) v; }
    }
};
}  // namespace

extern "C" int orc_rng_stream(int rng_mode, uint64_t seed, int n, int* out) {
    if (rng_mode == ORC_RNG_PHILOX) {
        for (int i = 0; i < n; ++i) { uint32_t w[4]; orc_philox(seed, (uint32_t) (i / 3), w); out[i] = (int) (w[i % 3] >> 1); }
        return 0;
    }
    RefRng r(rng_mode, seed);
    for (int i = 0; i < n; ++i) out[i] = r();
    return 0;
}

// src/sac_prerejective_omp.cpp:33-77 selectCorrespondences; r[i] are the raw RNG outputs, one per sample
// (control flow copied literally, including the wrap-around branch).  nr_samples is the reference's
// AlignmentParameters::n_samples (3 in every shipped config; ORC_MAX_SAMPLES bounds the local arrays of this restatement).
extern "C" void orc_select_n(const int* r, int nr_samples, int n_corr, int* sample) {
    int temp_sample;
    for (int i = 0; i < nr_samples; i++) {
        sample[i] = r[i] % n_corr;
        for (int j = 0; j < i; j++) {
            if (sample[i] >= sample[j]) {
                if (sample[i] < n_corr - 1) {
                    sample[i]++;
                    continue;
                } else if (sample[j] == 0) {
                    sample[i] = 1;
                    continue;
                } else {
                    sample[i] = 0;
                }
            }
            temp_sample = sample[i];
            for (int k = i; k > j; k--) sample[k] = sample[k - 1];
            sample[j] = temp_sample;
            break;
        }
    }
}
extern "C" void orc_select3(const int r[3], int n_corr, int sample[3]) { orc_select_n(r, 3, n_corr, sample); }

// The raw draws of Philox iteration `iter`: draw j is word j % 4 of philox4x32-10(key = seed, counter = (iter, j / 4, 0, 0)), top 31 bits
// (for n_samples <= 4 -- every shipped config -- that is the one block orc_philox returns).
extern "C" void orc_philox_draws(uint64_t seed, uint32_t iter, int nr_samples, int* r) {
    uint32_t w[4];
    for (int j = 0; j < nr_samples; ++j) {
        if ((j & 3) == 0) philox4(seed, iter, (uint32_t) (j >> 2), 0, 0, w);
        r[j] = (int) (w[j & 3] >> 1);
    }
}

// pcl::registration::CorrespondenceRejectorPoly::thresholdPolygon, cardinality 3 [3P PCL 1.12.1
// registration/correspondence_rejection_poly.h]; call site src/sac_prerejective_omp.cpp:105-108,214; SURVEY.md A.4.
// computeSquaredDistance = dx*dx + dy*dy + dz*dz (left to right).
// cardinality 2 tests its one edge, any other cardinality every edge i -> (i + 1) % n.
extern "C" int orc_poly_ok_n(const float* src, const float* tgt, const int* sidx, const int* tidx, int n, float edge_thr) {
    float thr2 = edge_thr * edge_thr;
    const int edges = n == 2 ? 1 : n;
    for (int i = 0; i < edges; ++i) {
        int j = (i + 1) % n;
        const float *a = src + 12 * (size_t) sidx[i], *b = src + 12 * (size_t) sidx[j];
        const float *c = tgt + 12 * (size_t) tidx[i], *d = tgt + 12 * (size_t) tidx[j];
        float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        float ds = dx * dx + dy * dy + dz * dz;
        dx = c[0] - d[0]; dy = c[1] - d[1]; dz = c[2] - d[2];
        float dt = dx * dx + dy * dy + dz * dz;
        float sim = ds < dt ? ds / dt : dt / ds;
        if (!(sim >= thr2)) return 0;   // "return dist_ratio >= threshold" fails for NaN too
    }
    return 1;
}
extern "C" int orc_poly_ok(const float* src, const float* tgt, const int sidx[3], const int tidx[3], float edge_thr) {
    return orc_poly_ok_n(src, tgt, sidx, tidx, 3, edge_thr);
}

// pcl::registration::TransformationEstimationSVD::estimateRigidTransformation -> pcl::umeyama(src, tgt, false)
// [3P PCL 1.12.1 common/impl/eigen.hpp == Eigen/src/Geometry/Umeyama.h]; call site :220; SURVEY.md A.5.
//   means; demean; sigma = (1/n) * dst_demean * src_demean^T; SVD; S = (1,1,+-1) by det(U)*det(V); R = U S V^T;
//   t = dst_mean - R*src_mean.  The SVD is the canonical one-sided Jacobi (orc_math.h) instead of Eigen::JacobiSVD
//   (DEVIATION at rounding level; R is unique whenever sigma has rank >= 2).
//   For n points the means and the entries of sigma are left-to-right sums over the points (Eigen's coefficient-based evaluation of a
//   3 x n by n x 3 product; a vectorised build of Eigen may associate the inner sums differently for n >= 4: parity unpinned there, the
//   reference holds no fixture for any n, and n = 3 has a single association).
extern "C" void orc_umeyama_n(const float* src, const float* tgt, const int* sidx, const int* tidx, int n, float T[16]) {
    const float one_over_n = 1.0f / (float) n;
    float sm[3], dm[3], S[3][ORC_MAX_SAMPLES], D[3][ORC_MAX_SAMPLES];
    for (int a = 0; a < 3; ++a) {
        float ss = src[12 * (size_t) sidx[0] + a], ds = tgt[12 * (size_t) tidx[0] + a];
        for (int j = 1; j < n; ++j) { ss += src[12 * (size_t) sidx[j] + a]; ds += tgt[12 * (size_t) tidx[j] + a]; }
        sm[a] = ss * one_over_n;
        dm[a] = ds * one_over_n;
        for (int j = 0; j < n; ++j) { S[a][j] = src[12 * (size_t) sidx[j] + a] - sm[a]; D[a][j] = tgt[12 * (size_t) tidx[j] + a] - dm[a]; }
    }
    float sigma[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = D[i][0] * S[j][0];
            for (int k = 1; k < n; ++k) acc += D[i][k] * S[j][k];
            sigma[3 * i + j] = one_over_n * acc;
        }
    float U[9], Sg[3], V[9];
    c_svd3(sigma, U, Sg, V);
    float sgn = (c_det3(U) * c_det3(V) < 0.f) ? -1.f : 1.f;
    float R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = (U[3 * i + 0] * V[3 * j + 0] + U[3 * i + 1] * V[3 * j + 1]) + (U[3 * i + 2] * sgn) * V[3 * j + 2];
    float t[3];
    for (int i = 0; i < 3; ++i) t[i] = dm[i] - ((R[3 * i + 0] * sm[0] + R[3 * i + 1] * sm[1]) + R[3 * i + 2] * sm[2]);
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * j + i] = R[3 * i + j];   // column-major
        T[12 + i] = t[i];
    }
    T[15] = 1.f;
}
extern "C" void orc_umeyama3(const float* src, const float* tgt, const int sidx[3], const int tidx[3], float T[16]) {
    orc_umeyama_n(src, tgt, sidx, tidx, 3, T);
}

namespace {
// T (column-major) applied to [x,y,z,1]: Eigen Matrix4f * Vector4f on SSE = ((c0*x + c1*y) + c2*z) + c3*1
inline void apply(const float T[16], const float* s, float o[3]) {
    for (int i = 0; i < 3; ++i) o[i] = ((T[i] * s[0] + T[4 + i] * s[1]) + T[8 + i] * s[2]) + T[12 + i];
}
// 4-vector norm of (T*[s,1] - [t,1]) as Eigen reduces it on SSE: (d0 + d2) + (d1 + d3), d3 = 0 (src/metric.cpp:141)
inline float dist4(const float T[16], const float* s, const float* t) {
    float o[3];
    apply(T, s, o);
    float dx = o[0] - t[0], dy = o[1] - t[1], dz = o[2] - t[2];
    return std::sqrt((dx * dx + dz * dz) + (dy * dy + 0.f));
}
// 3-vector block norm (src/metric.cpp:111): sequential
inline float dist3(const float T[16], const float* s, const float* t) {
    float o[3];
    apply(T, s, o);
    float dx = o[0] - t[0], dy = o[1] - t[1], dz = o[2] - t[2];
    return std::sqrt((dx * dx + dy * dy) + dz * dz);
}

// ---- closest-plane metric (src/metric.cpp:10-53 buildClosestPlaneInliers, :181-199 ClosestPlaneMetricEstimator) on the
// sparse subset the RANSAC loop always uses (sparse = true, src/sac_prerejective_omp.cpp:109).
// Canonical choices (the reference draws the subset from the thread's mt19937 stream, so neither the subset nor the
// order of its float sums is reproducible): draw j of hypothesis `counter` = Philox4x32-10(counter, j / 4, 0x5A17, 0)[j % 4]
// >> 1, idx = r % n with the reference's linear probing over `visited` (the resulting SET does not depend on the
// insertion order); score and squared-error sums are accumulated in 2^-32 fixed point (order free, exact).
struct PlaneCtx {
    const float* src = nullptr; int ns = 0;
    const float* tgt = nullptr; int nt = 0;
    Grid g;
    float thr = 0.f, r2 = 0.f;
    int n_sp = 0;
    uint64_t seed = 0;
};
struct PlaneEval { int n_inl; float score; float rmse; float metric; };

void plane_setup(PlaneCtx& pc, const float* src, int ns, const float* tgt, int nt, uint64_t seed) {
    pc.src = src; pc.ns = ns; pc.tgt = tgt; pc.nt = nt; pc.seed = seed;
    float density = 0.f;
    orc_cloud_density(tgt, nt, 0.8f, &density);            // setTargetCloud: inlier_threshold_ = calculatePointCloudDensity(tgt)
    pc.thr = density;
    float radius = 2 * pc.thr;                              // DIST_TO_PLANE_COEFFICIENT * inlier_threshold
    pc.r2 = radius * radius;
    pc.g.build(tgt, nt, radius * 1.001f);
    pc.n_sp = (int) (0.01 * (float) ns);                    // SPARSE_POINTS_FRACTION * src.size()
}

PlaneEval plane_eval(const PlaneCtx& pc, const float T[16], uint32_t counter, int score_id, std::vector<uint32_t>& visited,
                     std::vector<std::pair<int, int>>* list) {
    const int n = pc.ns;
    if ((int) visited.size() < (n + 31) / 32) visited.assign((n + 31) / 32, 0u);
    std::vector<int> picked;
    picked.reserve(pc.n_sp);
    for (int j = 0; j < pc.n_sp; ++j) {
        uint32_t w[4];
        philox4(pc.seed, counter, (uint32_t) (j >> 2), 0x5A17u, 0u, w);
        int idx = (int) ((w[j & 3] >> 1) % (uint32_t) n);
        while ((visited[idx >> 5] >> (idx & 31)) & 1u) idx = (idx + 1) % n;
        visited[idx >> 5] |= 1u << (idx & 31);
        picked.push_back(idx);
    }
    long long sc = 0, sq = 0;
    int cnt = 0;
    if (list) list->clear();
    for (int idx : picked) {
        visited[idx >> 5] &= ~(1u << (idx & 31));
        const float* s = pc.src + 12 * (size_t) idx;
        float pt[3];
        apply(T, s, pt);
        if (!finite3(pt)) continue;
        int nn = -1;
        float best = 0.f;
        pc.g.visit27(pt, [&](int q) {
            float d2 = dist2(pt, pc.tgt + 12 * (size_t) q);
            if (!(d2 < pc.r2)) return;
            if (nn < 0 || d2 < best || (d2 == best && q < nn)) { nn = q; best = d2; }
        });
        if (nn < 0) continue;
        const float* q = pc.tgt + 12 * (size_t) nn;
        float dist = std::fabs((q[4] * (q[0] - pt[0]) + q[5] * (q[1] - pt[1])) + q[6] * (q[2] - pt[2]));
        if (!(dist < pc.thr)) continue;
        ++cnt;
        float thr = pc.thr, value = 1.f;
        switch (score_id) {
            case ORC_SCORE_CONSTANT: value = 1.f; break;
            case ORC_SCORE_MAE: value = std::fabs(dist - thr) / thr; break;
            case ORC_SCORE_MSE: value = (dist - thr) * (dist - thr) / (thr * thr); break;
            case ORC_SCORE_EXP: value = c_expf(-dist * dist / (2 * thr * thr)); break;
        }
        sc += (long long) ((double) value * 4294967296.0);
        float rel = dist / thr;
        sq += (long long) ((double) (rel * rel) * 4294967296.0);
        if (list) list->push_back({idx, nn});
    }
    if (list) std::sort(list->begin(), list->end());
    PlaneEval e;
    e.n_inl = cnt;
    e.score = (float) ((double) sc / 4294967296.0);
    e.rmse = cnt ? pc.thr * (float) std::sqrt((double) sq / 4294967296.0 / (double) cnt) : std::numeric_limits<float>::max();
    e.metric = (float) ((double) e.score / (0.01 * (double) (float) pc.ns));   // score / (SPARSE_POINTS_FRACTION * src.size())
    return e;
}

struct Eval { int n_inl; float rmse; float metric; };

// src/metric.cpp:125-165 (buildInliers + calculateScore :55-81) and :167-179 + src/analysis.cpp:95-130 (uniformity)
Eval evaluate(const float* src, const float* tgt, const lgr_orc_corr* corr, int c, const float T[16],
              int metric_id, int score_id, const float* bbmin, const float* bbmax, uint8_t* mask,
              std::vector<int>& hist /* 3*100*100 scratch */, const PlaneCtx* pc = nullptr, uint32_t counter = 0,
              std::vector<uint32_t>* visited = nullptr, std::vector<std::pair<int, int>>* plane_list = nullptr) {
    Eval e{0, 0.f, 0.f};
    if (metric_id == ORC_METRIC_CLOSEST_PLANE) {           // inliers, rmse and metric all come from the plane test
        PlaneEval pe = plane_eval(*pc, T, counter, score_id, *visited, plane_list);
        if (mask) std::memset(mask, 0, c);
        e.n_inl = pe.n_inl; e.rmse = pe.rmse; e.metric = pe.metric;
        return e;
    }
    const bool combination = metric_id == ORC_METRIC_COMBINATION;
    const int plane_score_id = score_id;                   // the closest-plane member gets the configured score function,
    if (combination) score_id = ORC_SCORE_CONSTANT;        // the member CorrespondencesMetricEstimator is default-constructed (include/metric.h:191-192)
    float rmse = 0.f, score = 0.f;
    bool uni = metric_id == ORC_METRIC_UNIFORMITY;
    if (uni) std::fill(hist.begin(), hist.end(), 0);
    for (int i = 0; i < c; ++i) {
        const float* s = src + 12 * (size_t) corr[i].query;
        const float* t = tgt + 12 * (size_t) corr[i].match;
        float dist = dist4(T, s, t);
        float thr = corr[i].threshold;
        bool in = dist < thr;
        if (mask) mask[i] = in ? 1 : 0;
        if (!in) continue;
        e.n_inl++;
        rmse += dist * dist;
        if (uni) {
            int bin[3];
            for (int a = 0; a < 3; ++a)
            {
                float fb = std::min(std::floor((s[a] - bbmin[a]) / (bbmax[a] - bbmin[a]) * 100), 100 - 1.f);
                bin[a] = (fb >= 0.f) ? (int) fb : 0;   // NaN / negative would index out of bounds in the reference (UB); pinned to bin 0
            }
            for (int k = 0; k < 3; ++k) hist[(k * 100 + bin[(k + 1) % 3]) * 100 + bin[(k + 2) % 3]]++;
        } else {
            float value = 1.f;
            switch (score_id) {
                case ORC_SCORE_CONSTANT: value = 1.f; break;
                case ORC_SCORE_MAE: value = std::fabs(dist - thr) / thr; break;
                case ORC_SCORE_MSE: value = (dist - thr) * (dist - thr) / (thr * thr); break;
                case ORC_SCORE_EXP: value = c_expf(-dist * dist / (2 * thr * thr)); break;
            }
            score += value;
        }
    }
    e.rmse = e.n_inl ? std::sqrt(rmse / static_cast<float>(e.n_inl)) : std::numeric_limits<float>::max();
    if (uni) {
        if (e.n_inl == 0) { e.metric = 0.f; return e; }
        float entropy[3] = {0.f, 0.f, 0.f};
        float n = (float) e.n_inl;
        for (int k = 0; k < 3; ++k) {
            for (int b = 0; b < 10000; ++b) {
                float p = (float) hist[k * 10000 + b] / n;
                if (p == 0.f) continue;
                entropy[k] -= p * c_logf(p);
            }
            entropy[k] /= 9.210340371976184f;   // std::log((float) (N_BINS * N_BINS))
        }
        e.metric = c_cbrtf(entropy[0] * entropy[1] * entropy[2]);
    } else {
        e.metric = score / (float) c;
    }
    if (combination) {                                      // src/metric.cpp:239-249: metric_cs * metric_cp
        PlaneEval pe = plane_eval(*pc, T, counter, plane_score_id, *visited, nullptr);
        e.metric = e.metric * pe.metric;
    }
    return e;
}

// src/metric.cpp:103-123 MetricEstimator::estimateMaxIterations
int est_max_iter(const float* src, const float* tgt, const lgr_orc_corr* corr, int c, const float T[16],
                 float confidence, int nr_samples) {
    int count = 0;
    for (int i = 0; i < c; ++i) {
        float e = dist3(T, src + 12 * (size_t) corr[i].query, tgt + 12 * (size_t) corr[i].match);
        if (e < corr[i].threshold) count++;
    }
    float frac = (float) count / (float) c;
    frac /= 4.f;
    if (frac <= 0.0 || std::log(1.0 - std::pow(frac, nr_samples)) >= 0.0) return INT_MAX;
    double iterations = std::log(1.0 - confidence) / std::log(1.0 - std::pow(frac, nr_samples));
    return static_cast<int>(std::min((double) INT_MAX, iterations));
}

// src/transformation.cpp:4-38 estimateOptimalRigidTransformation; JacobiSVD replaced by the canonical SVD.
void refit(const float* src, const float* tgt, const lgr_orc_corr* corr, int c, const uint8_t* mask, float T[16]) {
    float cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
    int n = 0;
    for (int i = 0; i < c; ++i) {
        if (!mask[i]) continue;
        const float* p = src + 12 * (size_t) corr[i].query;
        const float* q = tgt + 12 * (size_t) corr[i].match;
        for (int a = 0; a < 3; ++a) { cs[a] += p[a]; ct[a] += q[a]; }
        ++n;
    }
    for (int a = 0; a < 3; ++a) { cs[a] /= (float) n; ct[a] /= (float) n; }
    float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < c; ++i) {
        if (!mask[i]) continue;
        const float* p = src + 12 * (size_t) corr[i].query;
        const float* q = tgt + 12 * (size_t) corr[i].match;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) H[3 * a + b] += (p[a] - cs[a]) * (q[b] - ct[b]);
    }
    float U[9], Sg[3], V[9];
    c_svd3(H, U, Sg, V);
    float R[9];
    auto mulVUt = [&](const float* Vm) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                R[3 * i + j] = (Vm[3 * i + 0] * U[3 * j + 0] + Vm[3 * i + 1] * U[3 * j + 1]) + Vm[3 * i + 2] * U[3 * j + 2];
    };
    mulVUt(V);
    if (c_det3(R) < 0.f) {
        float V2[9];
        for (int i = 0; i < 9; ++i) V2[i] = V[i];
        V2[2] = -V2[2]; V2[5] = -V2[5]; V2[8] = -V2[8];
        mulVUt(V2);
    }
    float t[3];
    for (int i = 0; i < 3; ++i) t[i] = ct[i] - ((R[3 * i + 0] * cs[0] + R[3 * i + 1] * cs[1]) + R[3 * i + 2] * cs[2]);
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * j + i] = R[3 * i + j];
        T[12 + i] = t[i];
    }
    T[15] = 1.f;
}

// include/utils.h:34-43 calculateCombinationOrMax<int>
int comb_or_max(int n, int k) {
    double result = 1.0;
    for (int i = 0; i < k; ++i) { result *= n - i; result /= i + 1; }
    int mx = INT_MAX;
    return result > mx ? mx : (int) result;
}

}  // namespace
extern "C" int orc_comb_or_max(int n, int k) { return comb_or_max(n, k); }   // pinned against oracle/_ref (tests/test_oracle_ref.py)
namespace {

struct Hyp { bool ok; float T[16]; };
inline Hyp make_hyp(const float* src, const float* tgt, const lgr_orc_corr* corr, const int* sample, float edge_thr, int n = 3) {
    Hyp h;
    int sidx[ORC_MAX_SAMPLES], tidx[ORC_MAX_SAMPLES];
    for (int j = 0; j < n; ++j) { sidx[j] = corr[sample[j]].query; tidx[j] = corr[sample[j]].match; }  // buildIndices :17-31
    h.ok = orc_poly_ok_n(src, tgt, sidx, tidx, n, edge_thr) != 0;
    if (h.ok) orc_umeyama_n(src, tgt, sidx, tidx, n, h.T);
    else { for (int i = 0; i < 16; ++i) h.T[i] = (i % 5 == 0) ? 1.f : 0.f; }
    return h;
}
}  // namespace

extern "C" int orc_evaluate(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
                            const float T[16], int metric_id, int score_id,
                            uint8_t* mask, int* n_inl, float* rmse, float* metric) {
    (void) nt;
    if (metric_id == ORC_METRIC_CLOSEST_PLANE || metric_id == ORC_METRIC_COMBINATION) return -7;   // use orc_evaluate_plane
    float mn[3], mx[3];
    orc_bbox(src, ns, mn, mx);   // UniformityMetricEstimator::setSourceCloud src/metric.cpp:167-170
    std::vector<int> hist(30000);
    Eval e = evaluate(src, tgt, corr, c, T, metric_id, score_id, mn, mx, mask, hist);
    *n_inl = e.n_inl; *rmse = e.rmse; *metric = e.metric;
    return 0;
}

// closest-plane metric of one transform (stage-level parity tests): count, score, rmse, metric_cp, threshold used
extern "C" int orc_evaluate_plane(const float* src, int ns, const float* tgt, int nt, const float T[16], int score_id, uint64_t seed,
                                  uint32_t counter, int* n_inl, float* score, float* rmse, float* metric, float* thr, int* pairs /* 2*n_sp or NULL */) {
    if (ns < 1 || nt < 2) return -2;
    PlaneCtx pc;
    plane_setup(pc, src, ns, tgt, nt, seed);
    std::vector<uint32_t> visited;
    std::vector<std::pair<int, int>> list;
    PlaneEval e = plane_eval(pc, T, counter, score_id, visited, pairs ? &list : nullptr);
    *n_inl = e.n_inl; *score = e.score; *rmse = e.rmse; *metric = e.metric; *thr = pc.thr;
    if (pairs) for (size_t i = 0; i < list.size(); ++i) { pairs[2 * i] = list[i].first; pairs[2 * i + 1] = list[i].second; }
    return 0;
}

extern "C" int orc_estimate_max_iterations(const float* src, const float* tgt, const lgr_orc_corr* corr, int c,
                                           const float T[16], float confidence, int nr_samples) {
    return est_max_iter(src, tgt, corr, c, T, confidence, nr_samples);
}

extern "C" int orc_refit(const float* src, const float* tgt, const lgr_orc_corr* corr, int c, const uint8_t* mask, float T[16]) {
    refit(src, tgt, corr, c, mask, T);
    return 0;
}

extern "C" int orc_replay(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
                          const lgr_orc_params* p, const int* triples, int n,
                          uint8_t* ok, float* Ts, int* n_inl, float* metric) {
    (void) nt;
    float mn[3], mx[3];
    orc_bbox(src, ns, mn, mx);
#pragma omp parallel
    {
        std::vector<int> hist(30000);
#pragma omp for schedule(dynamic, 8)
        for (int i = 0; i < n; ++i) {
            Hyp h = make_hyp(src, tgt, corr, triples + (size_t) p->n_samples * i, p->edge_thr_coef, p->n_samples);
            ok[i] = h.ok;
            std::memcpy(Ts + 16 * (size_t) i, h.T, 64);
            if (h.ok) {
                Eval e = evaluate(src, tgt, corr, c, h.T, p->metric_id, p->score_id, mn, mx, nullptr, hist);
                n_inl[i] = e.n_inl; metric[i] = e.metric;
            } else { n_inl[i] = 0; metric[i] = 0.f; }
        }
    }
    return 0;
}

// src/sac_prerejective_omp.cpp:115-314 SampleConsensusPrerejectiveOMP::align.
//  * rng_mode MT19937_*: the reference schedule -- n_threads emulated OpenMP threads, static schedule, thread t owns
//    iterations [t*chunk, ...), RNG seeded SEED + t, thread-local early-out `ransac_iterations * nthreads >= iters_local`
//    (:197), per-thread best merged in thread order (:242-256).
//  * rng_mode PHILOX: the deterministic device schedule -- iteration i draws Philox(seed, i); iterations are processed
//    in batches of batch_size; after each batch the record inlier set (largest count, ties -> lowest i) updates the
//    bound via estimateMaxIterations, the best hypothesis is the max metric (strict '>', ties -> lowest i).  This is
//    what the HIP path implements; it is order-independent by construction.
extern "C" int orc_ransac(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
                          const lgr_orc_params* p, lgr_orc_result* res, uint8_t* final_mask) {
    (void) nt;
    std::memset(res, 0, sizeof(*res));
    if (p->n_samples < 3 || p->n_samples > ORC_MAX_SAMPLES) return -1;
    const int nsmp = p->n_samples;
    for (int i = 0; i < 16; ++i) res->T[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (c < nsmp) { res->converged = 0; return 0; }   // selectCorrespondences refuses (:36-42)
    float mn[3], mx[3];
    orc_bbox(src, ns, mn, mx);
    const bool plane = p->metric_id == ORC_METRIC_CLOSEST_PLANE || p->metric_id == ORC_METRIC_COMBINATION;
    PlaneCtx pc;
    if (plane) {
        if (nt < 2) return -2;
        plane_setup(pc, src, ns, tgt, nt, p->seed);
    }
    const int MIN_NR_INLIERS = 10, MIN_NR_FINAL_INLIERS = 20;
    const double MIN_INLIER_RATE = 0.15;
    int max_iterations = std::min(comb_or_max(c, p->n_samples), p->max_iterations);
    int estimated_iters = max_iterations;
    float final_T[16];
    for (int i = 0; i < 16; ++i) final_T[i] = (i % 5 == 0) ? 1.f : 0.f;
    float final_metric = 0.f;
    int ransac_iterations = 0, num_rejections = 0, best_iter = -1;

    if (p->has_guess) {
        // :134-147: the guess is the hypothesis to beat (final_tn / final_metric).  Its inliers seed only the GLOBAL largest_inlier_set,
        // which the loop never reads (the threads start from empty local sets, :177), so the iteration bound is unaffected.
        // The reference evaluates it with a fresh generator (:138); here the plane metrics' subset counter is 0xFFFFFFFD.
        std::vector<int> hist(30000);
        std::vector<uint32_t> visited;
        Eval e = evaluate(src, tgt, corr, c, p->guess, p->metric_id, p->score_id, mn, mx, nullptr, hist, &pc, 0xFFFFFFFDu, &visited);
        std::memcpy(final_T, p->guess, 64);
        final_metric = e.metric;
    }

    if (p->rng_mode == ORC_RNG_PHILOX) {
        int bound = max_iterations, done = 0, largest = 0;
        int batch = std::max(1, p->batch_size);
        std::vector<float> Ts; std::vector<int> ninl; std::vector<float> met; std::vector<uint8_t> ok;
        while (done < bound) {
            int nb = std::min(batch, max_iterations - done);
            Ts.resize((size_t) nb * 16); ninl.assign(nb, 0); met.assign(nb, 0.f); ok.assign(nb, 0);
#pragma omp parallel
            {
                std::vector<int> hist(30000);
                std::vector<uint32_t> visited;
#pragma omp for schedule(dynamic, 8)
                for (int b = 0; b < nb; ++b) {
                    int r[ORC_MAX_SAMPLES], sample[ORC_MAX_SAMPLES];
                    orc_philox_draws(p->seed, (uint32_t) (done + b), nsmp, r);
                    orc_select_n(r, nsmp, c, sample);
                    Hyp h = make_hyp(src, tgt, corr, sample, p->edge_thr_coef, nsmp);
                    ok[b] = h.ok;
                    std::memcpy(&Ts[(size_t) b * 16], h.T, 64);
                    if (!h.ok) continue;
                    Eval e = evaluate(src, tgt, corr, c, h.T, p->metric_id, p->score_id, mn, mx, nullptr, hist, &pc, (uint32_t) (done + b), &visited);
                    ninl[b] = e.n_inl; met[b] = e.metric;
                }
            }
            int rec = -1;
            for (int b = 0; b < nb; ++b) {
                if (!ok[b]) { num_rejections++; continue; }
                if (ninl[b] < MIN_NR_INLIERS) continue;
                if (rec < 0 || ninl[b] > ninl[rec]) rec = b;
                if (final_metric < met[b]) { final_metric = met[b]; std::memcpy(final_T, &Ts[(size_t) b * 16], 64); best_iter = done + b; }
            }
            if (rec >= 0 && ninl[rec] > largest) {
                largest = ninl[rec];
                bound = std::min(bound, est_max_iter(src, tgt, corr, c, &Ts[(size_t) rec * 16], p->confidence, p->n_samples));
            }
            done += nb;
            if (done >= max_iterations) break;
        }
        ransac_iterations = done;
        estimated_iters = bound;
    } else {
        int T = std::max(1, p->n_threads);
        // OpenMP static schedule without chunk: iterations divided into T contiguous chunks, sizes differ by <= 1
        int q = max_iterations / T, r = max_iterations % T;
        struct Local { float T[16]; float metric; int iters_local; int largest; int its; int rej; };
        std::vector<Local> loc(T);
#pragma omp parallel for schedule(dynamic, 1)
        for (int t = 0; t < T; ++t) {
            Local& L = loc[t];
            for (int i = 0; i < 16; ++i) L.T[i] = (i % 5 == 0) ? 1.f : 0.f;
            L.metric = 0.f; L.iters_local = max_iterations; L.largest = 0; L.its = 0; L.rej = 0;
            int lo = t * q + std::min(t, r), cnt = q + (t < r ? 1 : 0);
            RefRng rng(p->rng_mode, p->seed + (uint64_t) t);
            std::vector<int> hist(30000);
            std::vector<uint32_t> visited;
            for (int i = lo; i < lo + cnt; ++i) {
                if ((long long) L.its * T >= L.iters_local) continue;
                ++L.its;
                int rr[ORC_MAX_SAMPLES], sample[ORC_MAX_SAMPLES];
                for (int j = 0; j < nsmp; ++j) rr[j] = rng();
                orc_select_n(rr, nsmp, c, sample);
                Hyp h = make_hyp(src, tgt, corr, sample, p->edge_thr_coef, nsmp);
                if (!h.ok) { ++L.rej; continue; }
                Eval e = evaluate(src, tgt, corr, c, h.T, p->metric_id, p->score_id, mn, mx, nullptr, hist, &pc, (uint32_t) i, &visited);
                if (e.n_inl < MIN_NR_INLIERS) continue;
                if (L.largest < e.n_inl) {
                    L.largest = e.n_inl;
                    L.iters_local = std::min(est_max_iter(src, tgt, corr, c, h.T, p->confidence, p->n_samples), L.iters_local);
                }
                if (L.metric < e.metric) { std::memcpy(L.T, h.T, 64); L.metric = e.metric; }
            }
        }
        for (int t = 0; t < T; ++t) {   // critical section, merged in thread order
            if (loc[t].iters_local < estimated_iters) estimated_iters = loc[t].iters_local;
            if (final_metric < loc[t].metric) { final_metric = loc[t].metric; std::memcpy(final_T, loc[t].T, 64); }
            ransac_iterations += loc[t].its; num_rejections += loc[t].rej;
        }
    }

    // :265-296 final re-estimation
    // The two evaluations of the final block use a fresh generator in the reference (:270); here: counters 0xFFFFFFFE / 0xFFFFFFFF.
    std::vector<uint8_t> mask(c);
    std::vector<int> hist(30000);
    std::vector<uint32_t> visited;
    std::vector<std::pair<int, int>> plane_list;
    Eval e = evaluate(src, tgt, corr, c, final_T, p->metric_id, p->score_id, mn, mx, mask.data(), hist, &pc, 0xFFFFFFFEu, &visited, &plane_list);
    bool enough = e.n_inl > MIN_NR_FINAL_INLIERS || (float) e.n_inl > MIN_INLIER_RATE * (float) c;
    float min_tol = p->metric_id == ORC_METRIC_UNIFORMITY ? 0.3f : 0.0f;   // include/metric.h:97-99 / 73-75
    bool converged = enough && e.metric > min_tol;
    float Tn[16];
    if (e.n_inl > 0) {
        if (p->metric_id == ORC_METRIC_CLOSEST_PLANE) {
            // the inliers handed to estimateOptimalRigidTransformation are the (source point, nearest target point) pairs
            // of the plane test (canonical order: ascending source index)
            std::vector<lgr_orc_corr> pl(plane_list.size());
            std::vector<uint8_t> all(plane_list.size(), 1);
            for (size_t i = 0; i < pl.size(); ++i) pl[i] = lgr_orc_corr{plane_list[i].first, plane_list[i].second, 0.f, pc.thr};
            refit(src, tgt, pl.data(), (int) pl.size(), all.data(), Tn);
        } else refit(src, tgt, corr, c, mask.data(), Tn);
    } else { for (int i = 0; i < 16; ++i) Tn[i] = std::numeric_limits<float>::quiet_NaN(); }   // 0/0 centroids in the reference
    Eval e2 = evaluate(src, tgt, corr, c, Tn, p->metric_id, p->score_id, mn, mx, mask.data(), hist, &pc, 0xFFFFFFFFu, &visited, nullptr);
    if (final_mask) std::memcpy(final_mask, mask.data(), c);
    std::memcpy(res->T, Tn, 64);
    res->iterations = ransac_iterations;
    res->converged = converged ? 1 : 0;
    res->n_inliers = e2.n_inl;
    res->metric = e2.metric;
    res->best_metric_before_refit = final_metric;
    res->best_iteration = best_iter;
    res->num_rejections = num_rejections;
    res->estimated_iters = estimated_iters;
    return 0;
}

extern "C" int orc_align(const float* src, int ns, const float* tgt, int nt, const lgr_orc_params* p,
                         lgr_orc_result* res, lgr_orc_corr* corr_out, int* n_corr_out, double* st) {
    std::vector<lgr_orc_corr> corr((size_t) ns);
    int c = 0;
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc = orc_correspondences(src, ns, tgt, nt, p, corr.data(), &c, t);
    if (rc) return rc;
    double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    rc = orc_ransac(src, ns, tgt, nt, corr.data(), c, p, res, nullptr);
    double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    t[5] = t1 - t0;
    if (corr_out) std::memcpy(corr_out, corr.data(), sizeof(lgr_orc_corr) * (size_t) c);
    if (n_corr_out) *n_corr_out = c;
    if (st) for (int i = 0; i < 8; ++i) st[i] = t[i];
    return rc;
}

// ---------------------------------------------------------------- hypotheses (dead path in the reference build)
// src/analysis.cpp:19-24 calculateRotationAndTranslationDifferences: angle of R1^-1 R2 (Eigen::AngleAxisf: via
// quaternion, angle = 2*atan2(|vec|, |w|)); here angle = 2*atan2(sqrt(x^2+y^2+z^2), |w|) of the unit quaternion of
// R1^T R2 computed in double (host-side bookkeeping, not a parity-critical path).
extern "C" void orc_rot_trans_diff(const float* T1, const float* T2, float* angle, float* tdist) {
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double) T1[4 * i + k] * (double) T2[4 * j + k];   // (R1^T R2)_ij, col-major
            R[3 * i + j] = s;
        }
    double tr = R[0] + R[4] + R[8];
    double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    double sn = 0.5 * std::sqrt(vx * vx + vy * vy + vz * vz), cs = 0.5 * (tr - 1.0);
    *angle = (float) std::atan2(sn, cs);
    double dx = (double) T1[12] - T2[12], dy = (double) T1[13] - T2[13], dz = (double) T1[14] - T2[14];
    *tdist = (float) std::sqrt(dx * dx + dy * dy + dz * dz);
}

// src/hypotheses.cpp:50-129 chooseBestHypothesis, the decision only: the candidate with the largest uniformity of its
// correspondence inliers (strict >, identity when none is positive); the hypotheses.csv side output (areas, overlaps) is
// not reproduced.  tns: n column-major 4x4.  Returns the index chosen or -1.
extern "C" int orc_choose_best_hypothesis(const float* src, int ns, const float* tgt, int nt, const lgr_orc_corr* corr, int c,
                                          const float* tns, int n, float T_out[16], float* uniformities) {
    (void) nt;
    float mn[3], mx[3];
    orc_bbox(src, ns, mn, mx);   // calculateCorrespondenceUniformity(src, inliers) computes the bounding box of src itself
    std::vector<int> hist(30000);
    float best = 0.f;
    int best_i = -1;
    for (int i = 0; i < 16; ++i) T_out[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int i = 0; i < n; ++i) {
        Eval e = evaluate(src, tgt, corr, c, tns + 16 * (size_t) i, ORC_METRIC_UNIFORMITY, ORC_SCORE_MSE, mn, mx, nullptr, hist);
        if (uniformities) uniformities[i] = e.metric;
        if (e.metric > best) { best = e.metric; best_i = i; std::memcpy(T_out, tns + 16 * (size_t) i, 64); }
    }
    return best_i;
}

// src/hypotheses.cpp:14-48 updateHypotheses (MIN_ANGLE pi/9, MIN_DISTANCE_COEF 20, MIN_METRIC_COEF 0.1)
extern "C" int orc_update_hypotheses(float* tns, float* metrics, int n, int cap, const float* new_T, float new_metric, float distance_thr) {
    std::vector<std::vector<float>> T(n, std::vector<float>(16));
    std::vector<float> M(metrics, metrics + n);
    for (int i = 0; i < n; ++i) std::memcpy(T[i].data(), tns + 16 * (size_t) i, 64);
    float best = n == 0 ? 0.f : *std::max_element(M.begin(), M.end());
    auto flush = [&]() {
        int m = (int) T.size();
        if (m > cap) return -1;
        for (int i = 0; i < m; ++i) { std::memcpy(tns + 16 * (size_t) i, T[i].data(), 64); metrics[i] = M[i]; }
        return m;
    };
    if (new_metric < 0.1 * best) return flush();
    std::vector<int> similar;
    for (int i = (int) T.size() - 1; i >= 0; --i) {
        float r, t;
        orc_rot_trans_diff(new_T, T[i].data(), &r, &t);
        bool is_similar = r < (M_PI / 9) && t < 20 * distance_thr;
        if (is_similar) similar.push_back(i);
        if (is_similar && M[i] > new_metric) return flush();
    }
    for (int idx : similar) { T.erase(T.begin() + idx); M.erase(M.begin() + idx); }   // descending indices
    T.emplace_back(new_T, new_T + 16);
    M.push_back(new_metric);
    if (new_metric > best) {
        for (int i = (int) T.size() - 1; i >= 0; --i)
            if (M[i] < 0.1 * new_metric) { T.erase(T.begin() + i); M.erase(M.begin() + i); }
    }
    return flush();
}

// include/matching.h:44-94 KNNResult<T>::addPoint, restated; driven by tests/knn_result.cpp:30-51 golden sequence.
extern "C" int orc_knnresult_run(int capacity, const float* dists, const int* indices, int n, int* out_idx, float* out_dist) {
    int count = 0;
    std::vector<int> idx; std::vector<float> ds;
    for (int t = 0; t < n; ++t) {
        float dist = dists[t]; int index = indices[t];
        if (count < capacity) { idx.resize(count + 1); ds.resize(count + 1); }
        int i;
        for (i = count; i > 0; --i) {
            if (ds[i - 1] > dist) {
                if (i < capacity) { ds[i] = ds[i - 1]; idx[i] = idx[i - 1]; }
            } else break;
        }
        if (i < capacity) { ds[i] = dist; idx[i] = index; }
        if (count < capacity) count++;
    }
    for (int i = 0; i < count; ++i) { out_idx[i] = idx[i]; out_dist[i] = ds[i]; }
    return count;
}

extern "C" float orc_atan2f(float y, float x) { return c_atan2f(y, x); }
extern "C" float orc_logf(float x) { return c_logf(x); }
extern "C" float orc_cbrtf(float x) { return c_cbrtf(x); }
extern "C" float orc_expf(float x) { return c_expf(x); }
extern "C" void orc_svd3(const float A[9], float U[9], float S[3], float V[9]) { c_svd3(A, U, S, V); }
extern "C" void orc_eig3_smallest(const float C[9], float* eval, float evec[3]) {
    float U[9], S[3], V[9];
    c_svd3(C, U, S, V);
    *eval = S[2]; evec[0] = V[2]; evec[1] = V[5]; evec[2] = V[8];
}
