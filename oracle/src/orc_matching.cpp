// orc_matching.cpp -- ORACLE (test infrastructure): brute-force descriptor matching, densities, match filters,
// whole correspondence search.  Reference paths relative to /root/reference.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <unordered_set>
#include <vector>

#include "../lgr_oracle.h"
#include "orc_grid.h"

using namespace orc;

namespace {
inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// cv::hal::normL2Sqr_(const float*, const float*, int) [3P, OpenCV 4.5.1 modules/core/src/norm.cpp] as compiled for
// the x86-64 baseline (SSE, 4 lanes, v_muladd = mul + add, four accumulators, blocks of 16 floats):
//   acc[a][l] += t*t for j = 16*blk + 4*a + l;  d = reduce(acc0 + acc1 + acc2 + acc3) with SSE reduce
//   (s0 + s2) + (s1 + s3); scalar tail d += t*t.  For n = 33: two blocks, tail j = 32.
typedef float v4sf __attribute__((vector_size(16)));
inline v4sf ld4(const float* p) { v4sf r; std::memcpy(&r, p, 16); return r; }
inline float l2sqr33(const float* a, const float* b) {
    // written with 4-lane vectors so that it is literally the SSE lane structure (and runs at SSE speed); every lane
    // operation is an IEEE mul followed by an IEEE add (-ffp-contract=off).
    v4sf acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int blk = 0; blk < 2; ++blk) {
        const float *pa = a + 16 * blk, *pb = b + 16 * blk;
        v4sf t0 = ld4(pa) - ld4(pb), t1 = ld4(pa + 4) - ld4(pb + 4), t2 = ld4(pa + 8) - ld4(pb + 8), t3 = ld4(pa + 12) - ld4(pb + 12);
        acc0 = t0 * t0 + acc0; acc1 = t1 * t1 + acc1; acc2 = t2 * t2 + acc2; acc3 = t3 * t3 + acc3;
    }
    v4sf s = ((acc0 + acc1) + acc2) + acc3;
    float d = (s[0] + s[2]) + (s[1] + s[3]);
    float t = a[32] - b[32];
    d += t * t;
    return d;
}
inline bool row_valid(const float* r) { for (int j = 0; j < 33; ++j) if (!std::isfinite(r[j])) return false; return true; }

// one query against the whole train set with the reference's block structure:
//   include/matching.h:594-634 matchBF: for each train block j: cv::batchDistance K=1 keeps the first minimum with a
//   strict '<' on the int bit pattern of sqrt(d2) (NaN never enters) -> lowest index wins inside a block;
//   src/common.cpp:517-529 updateMultivaluedCorrespondence inserts BEFORE an equal distance -> a later block wins ties.
// best match of one query inside one train block [j0, j1): cv::batchDistance K=1 keeps the first minimum with a
// strict '<' on the int bit pattern of sqrt(d2) (NaN never enters) -> lowest index wins inside a block
inline void block_best(const float* q, const float* t33, int j0, int j1, int& bi, float& bd) {
    bi = -1; bd = std::numeric_limits<float>::max();   // batchDistance init: FLT_MAX / -1
    for (int j = j0; j < j1; ++j) {
        float d = std::sqrt(l2sqr33(q, t33 + 33 * (size_t) j));
        if (d < bd) { bd = d; bi = j; }   // NaN compares false; for non-negative floats int-bit compare == float compare
    }
}
// src/common.cpp:517-529 updateMultivaluedCorrespondence with k = 1: inserts BEFORE an equal distance -> a later
// block wins ties
inline void merge_block(int bi, float bd, int& best_idx, float& best_dist) {
    if (bi < 0) return;                   // matches[l] empty / queryIdx == -1
    if (best_idx < 0 || !(best_dist < bd)) { best_idx = bi; best_dist = bd; }
}
}  // namespace

// include/matching.h:594-634 matchBF: the reference loops train blocks INSIDE query blocks and lets OpenCV
// parallelise over the query rows of one knnMatch call (cv::batchDistance: one query row against all rows of the train
// block at a time).  Here a thread takes QB queries through the train block together, so a train row is loaded once per
// QB queries instead of once per query (the query-at-a-time loop streams the 26 MB block from DRAM for every query);
// every (query, train) distance is the same canonical l2sqr33 and every query keeps its own first-minimum / merge rule,
// so results do not depend on QB or on the threading.
extern "C" int orc_match_bf_subset(const float* q33, const int* qsel, int nsel, const float* t33, int mt, int block, int* idx, float* dist) {
    if (block <= 0) return -1;
    constexpr int QB = 16;
    for (int s = 0; s < nsel; ++s) { idx[s] = -1; dist[s] = 0.f; }
    for (int j0 = 0; j0 < mt; j0 += block) {
        int j1 = std::min(mt, j0 + block);
#pragma omp parallel for schedule(dynamic, 1)
        for (int s0 = 0; s0 < nsel; s0 += QB) {
            const int nq = std::min(QB, nsel - s0);
            alignas(64) float qrow[QB][36];
            int bi[QB];
            float bd[QB];
            for (int u = 0; u < nq; ++u) {
                std::memcpy(qrow[u], q33 + 33 * (size_t) (qsel ? qsel[s0 + u] : s0 + u), 33 * sizeof(float));
                bi[u] = -1; bd[u] = std::numeric_limits<float>::max();   // batchDistance init: FLT_MAX / -1
            }
            for (int j = j0; j < j1; ++j) {
                const float* t = t33 + 33 * (size_t) j;
                for (int u = 0; u < nq; ++u) {
                    float d = std::sqrt(l2sqr33(qrow[u], t));
                    if (d < bd[u]) { bd[u] = d; bi[u] = j; }   // as block_best: strict '<', NaN never enters
                }
            }
            for (int u = 0; u < nq; ++u) merge_block(bi[u], bd[u], idx[s0 + u], dist[s0 + u]);
        }
    }
    return 0;
}

extern "C" int orc_match_bf(const float* q33, int mq, const float* t33, int mt, int block, int* idx, float* dist) {
    return orc_match_bf_subset(q33, nullptr, mq, t33, mt, block, idx, dist);
}

namespace {
// FLANN L2_Simple / pcl::L2_Norm_SQR: result += diff * diff, sequentially from dimension 0
inline float l2sqr33_seq(const float* a, const float* b) {
    float s = 0.f;
    for (int i = 0; i < 33; ++i) { float d = a[i] - b[i]; s += d * d; }
    return s;
}
}  // namespace

// include/matching.h:565-592 matchFLANN<FPFH>: pcl::KdTreeFLANN<FPFHSignature33>::nearestKSearch(k = 1) is an EXACT search
//  under L2_Simple, returning the squared distance; the reference takes std::sqrt of it (:587).  The nearest row is
// therefore the argmin of the sequential squared distance; equal distances: lowest index (FLANN's order is unspecified).
// Invalid (non-finite) query rows get no match (:576); non-finite train rows cannot be nearest to anything (their distance is NaN).
extern "C" int orc_match_flann(const float* q33, int mq, const float* t33, int mt, int* idx, float* dist) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < mq; ++i) {
        idx[i] = -1; dist[i] = 0.f;
        const float* q = q33 + 33 * (size_t) i;
        if (!row_valid(q)) continue;
        float best = 0.f;
        int bi = -1;
        for (int j = 0; j < mt; ++j) {
            float d = l2sqr33_seq(q, t33 + 33 * (size_t) j);
            if (d == d && (bi < 0 || d < best)) { best = d; bi = j; }
        }
        if (bi >= 0) { idx[i] = bi; dist[i] = std::sqrt(best); }
    }
    return 0;
}

extern "C" void orc_inverse4(const float* m16, float* out16) {
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m16[4 * c + r]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (piv != col) for (int c = 0; c < 8; ++c) std::swap(a[piv][c], a[col][c]);
        double d = a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            double f = a[r][col];
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out16[4 * c + r] = (float) a[r][4 + c];
}

// include/matching.h:637-678 matchLocal<FPFH> with randomness = 1.
//   transformed query point: pcl::transformPointCloudWithNormals -> Transformer::se3 (x*c0 + (y*c1 + (z*c2 + c3)), as orc_gror.cpp)
//   radiusSearch(p, match_search_radius): strict d2 < r*r, r*r in float (FLT_MAX -> inf: every finite train point), results by
//   ascending (d2, index); only train rows with valid descriptors are offered to KNNResult (:664); dist = pcl::L2_Norm =
//   sqrtf(sequential sum) (:665); KNNResult(1).addPoint keeps the FIRST of equal distances (include/matching.h:69-93) -> the
//   winner is the lexicographic minimum of (descriptor distance, spatial d2, index).
extern "C" int orc_match_local(const float* qpts, int mq, const float* tpts, int mt, const float* q33, const float* t33, const float* G,
                               float radius, int* idx, float* dist) {
    const float r2 = radius * radius;
    Grid g;
    const bool use_grid = std::isfinite(r2) && radius > 0.f;
    if (use_grid) g.build(tpts, mt, radius * 1.001f);
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < mq; ++i) {
        idx[i] = -1; dist[i] = 0.f;
        const float* qf = q33 + 33 * (size_t) i;
        if (!row_valid(qf)) continue;
        const float* P = qpts + 12 * (size_t) i;
        float T[3];
        for (int a = 0; a < 3; ++a) T[a] = G[a] * P[0] + (G[4 + a] * P[1] + (G[8 + a] * P[2] + G[12 + a]));
        float bd = 0.f, bs = 0.f;
        int bi = -1;
        auto offer = [&](int j) {
            const float* Q = tpts + 12 * (size_t) j;
            if (!finite3(Q)) return;
            float s = dist2(T, Q);
            if (!(s < r2)) return;
            const float* tf = t33 + 33 * (size_t) j;
            if (!row_valid(tf)) return;
            float d = std::sqrt(l2sqr33_seq(qf, tf));
            if (bi < 0 || d < bd || (d == bd && (s < bs || (s == bs && j < bi)))) { bd = d; bs = s; bi = j; }
        };
        if (!std::isfinite(T[0]) || !std::isfinite(T[1]) || !std::isfinite(T[2])) continue;   // FLANN: no neighbours of a NaN point
        if (use_grid) g.visit27(T, offer);
        else for (int j = 0; j < mt; ++j) offer(j);
        if (bi >= 0) { idx[i] = bi; dist[i] = bd; }
    }
    return 0;
}

// src/common.cpp:531-547 calculateSmoothedDensities: k-NN of point i (itself included, sorted by (d2, index));
// density = sqrt(d2[k-1]); then the same for nn_indices[1] and take the min.
extern "C" int orc_smoothed_densities(const float* pts, int n, int k, float* out) {
    if (n <= 1 || k < 2) return -1;  // rassert(pcd->size() > 1 && k >= 2)
    Grid g;
    g.build(pts, n, auto_cell(pts, n, 4.f));
    std::vector<float> dk(n);
    std::vector<int> nn1(n);
#pragma omp parallel
    {
        std::vector<Grid::Cand> c(k);
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < n; ++i) {
            int f = g.knn(pts + 12 * (size_t) i, k, c.data());
            dk[i] = f >= k ? std::sqrt(c[k - 1].d2) : std::numeric_limits<float>::quiet_NaN();
            nn1[i] = f >= 2 ? c[1].idx : i;
        }
    }
    for (int i = 0; i < n; ++i) out[i] = std::min(dk[i], dk[nn1[i]]);
    return 0;
}

// src/common.cpp:202-208 calculatePointCloudDensity
extern "C" int orc_cloud_density(const float* pts, int n, float quantile, float* out) {
    std::vector<float> d(n);
    if (orc_smoothed_densities(pts, n, 8, d.data())) return -1;
    int k = std::max(std::min((int) (quantile * (float) d.size() - 1), (int) d.size() - 1), 0);
    std::nth_element(d.begin(), d.begin() + k, d.end());
    *out = d[k];
    return 0;
}

// include/matching.h:395-411 (OneSided), :428-453 (LeftToRight), :492-550 (Cluster) with randomness = 1 and a single
// scale (match_multiscale :264-354 is then the identity on the 1-NN tables).
extern "C" int orc_filter(int matching_id, const float* src, int ns, const float* tgt, int nt,
                          const int* ij_idx, const float* ij_dist, const int* ji_idx, const float* ji_dist,
                          float distance_thr, int cluster_k, lgr_orc_corr* out, int* n_out) {
    std::vector<float> thr_s(ns), thr_t(nt);
    if (orc_smoothed_densities(src, ns, 2, thr_s.data())) return -1;
    if (orc_smoothed_densities(tgt, nt, 2, thr_t.data())) return -1;
    int cnt = 0;
    if (matching_id == ORC_MATCH_ONE_SIDED) {
        for (int i = 0; i < ns; ++i) {
            int j = ij_idx[i];
            if (j < 0) continue;
            float thr = std::min(std::max(thr_s[i], thr_t[j]), distance_thr);
            out[cnt++] = lgr_orc_corr{i, j, ij_dist[i], thr};
        }
    } else if (matching_id == ORC_MATCH_LR) {
        for (int i = 0; i < ns; ++i) {
            int j = ij_idx[i];
            if (j < 0) continue;
            if (ji_idx[j] == i) {
                float thr = std::min(std::max(thr_s[i], thr_t[j]), distance_thr);
                out[cnt++] = lgr_orc_corr{i, j, ji_dist[j], thr};   // distance of the reverse direction (:444)
            }
        }
    } else if (matching_id == ORC_MATCH_CLUSTER) {
        int k = cluster_k;
        std::vector<int> knn_s((size_t) ns * k), knn_t((size_t) nt * k);
        {
            std::vector<float> d2s((size_t) ns * k), d2t((size_t) nt * k);
            orc_knn(src, ns, src, ns, k, knn_s.data(), d2s.data());
            orc_knn(tgt, nt, tgt, nt, k, knn_t.data(), d2t.data());
        }
        // calculateCorrespondenceDistance(i, j, k, mv_ij, tree_src, tree_tgt)  (:524-550)
        auto cdist = [&](int i, int j, const int* knn_a, const int* knn_b, const int* ab_idx) {
            const int* nb = knn_b + (size_t) j * k;
            int consistent = 0, pairs = 0;
            for (int a = 0; a < k; ++a) {
                int in = knn_a[(size_t) i * k + a];
                if (in < 0) continue;
                int mt = ab_idx[in];
                if (mt < 0) continue;          // empty match list
                bool hit = false;
                for (int b = 0; b < k; ++b) if (nb[b] == mt) { hit = true; break; }
                if (hit) consistent++;
                pairs++;
            }
            if (pairs == 0) return 0.f;
            return 1.f - (float) consistent / (float) pairs;
        };
        std::vector<float> di(ns), dj(ns);
#pragma omp parallel for schedule(dynamic, 256)
        for (int i = 0; i < ns; ++i) {
            int j = ij_idx[i];
            if (j < 0) continue;
            di[i] = cdist(i, j, knn_s.data(), knn_t.data(), ij_idx);
            dj[i] = cdist(j, i, knn_t.data(), knn_s.data(), ji_idx);
        }
        for (int i = 0; i < ns; ++i) {
            int j = ij_idx[i];
            if (j < 0) continue;
            if (di[i] < 0.95f && dj[i] < 0.95f) {   // MATCHING_CLUSTER_THRESHOLD include/common.h:52
                float thr = std::min(std::max(thr_s[i], thr_t[j]), distance_thr);
                out[cnt++] = lgr_orc_corr{i, j, std::max(di[i], dj[i]), thr};
            }
        }
    } else {
        return -2;
    }
    *n_out = cnt;
    return 0;
}

// src/correspondence_search.cpp:4-15 + include/matching.h:148-262 with keypoint 'any' (kps = whole cloud,
// src/common.cpp:680-689) and feature_radius set (single scale):
//   log2_radius = floor(log2(r)/log2(scale)); search_radius = powf(scale, log2_radius)        (:172, :230)
//   voxel = sqrtf(M_PI * r*r / feature_nr_points)                                              (:231)
//   downsample (:234) -> normals k (:235) -> [kps normals re-estimate :243-246: no observable effect for FPFH,
//   skipped] -> FPFH (:248) -> 1-NN both ways (:306) -> filter.  The surface is kept in ORC_ORDER_CANONICAL.
// stage_seconds: [0] downsample [1] normals [2] fpfh [3] match [4] filter
// ---- multi-scale matching (feature_radius unset): include/matching.h:163-262 (initialize) and :264-352 (match_multiscale)
namespace {
// the matcher dispatch of match_multiscale (include/matching.h:294-312): guess -> matchLocal (inverse guess for the
// train -> query direction, :296), else bf -> matchBF, else matchFLANN
void match_dispatch(const lgr_orc_params* p, const float* qpts, int mq, const float* tpts, int mt, const float* qf, const float* tf,
                    bool inverse_tn, int* idx, float* dist) {
    if (p->has_guess) {
        float G[16];
        if (inverse_tn) orc_inverse4(p->guess, G); else std::memcpy(G, p->guess, 64);
        orc_match_local(qpts, mq, tpts, mt, qf, tf, G, p->match_search_radius, idx, dist);
    } else if (p->use_bfmatcher) {
        orc_match_bf(qf, mq, tf, mt, p->bf_block_size, idx, dist);
    } else {
        orc_match_flann(qf, mq, tf, mt, idx, dist);
    }
}
struct MsStorage {
    const float* kps = nullptr;       // key-point cloud, 12 floats per point
    int n_kps = 0;
    float iss_radius = 0.f;
    int min_l2 = std::numeric_limits<int>::max(), max_l2 = std::numeric_limits<int>::lowest();
    std::vector<std::vector<int>> kidx_ms;       // per scale: indices into kps
    std::vector<std::vector<float>> feat_ms;     // per scale: rows x 33
};

// :176-208 per-key-point radius level from the 5-NN distance in the full cloud, level pruning; :209-262 per-scale clouds
int ms_initialize(MsStorage& st, const float* pcd, int n, const float* kps, int n_kps, const int* kidx /* or null: identity */,
                  float iss_radius, const lgr_orc_params* p, const float* vp, double* t) {
    st.kps = kps; st.n_kps = n_kps; st.iss_radius = iss_radius;
    const int k = 5;
    if (n < k) return -6;
    std::vector<int> log2_radii(n_kps);
    {
        std::vector<int> nn((size_t) n_kps * k);
        std::vector<float> d2((size_t) n_kps * k);
        orc_knn(kps, n_kps, pcd, n, k, nn.data(), d2.data());   // nearestKSearch(*pcd, kps_indices[i], k): the point itself comes first
        (void) kidx;
        for (int i = 0; i < n_kps; ++i) {
            float density = sqrtf(d2[(size_t) i * k + (k - 1)]);
            float feature_radius = sqrtf((float) p->feature_nr_points * density * density / M_PI);
            log2_radii[i] = (int) std::floor(std::log2(feature_radius) / std::log2(p->scale_factor));
            st.min_l2 = std::min(log2_radii[i], st.min_l2);
            st.max_l2 = std::max(log2_radii[i], st.max_l2);
        }
    }
    std::vector<int> count(st.max_l2 - (st.min_l2 - 1), 0);
    for (int v : log2_radii) count[v - st.min_l2]++;
    int max_nr = *std::max_element(count.begin(), count.end());
    size_t front = 0, back = count.size();
    while (10 * count[front] < max_nr) { ++front; st.min_l2++; }
    while (1000 * count[back - 1] < max_nr) { --back; st.max_l2--; }
    for (int& v : log2_radii) v = std::min(std::max(v, st.min_l2), st.max_l2);
    int nr_scales = st.max_l2 - (st.min_l2 - 1);
    st.kidx_ms.assign(nr_scales, {});
    st.feat_ms.assign(nr_scales, {});
    for (int i = 0; i < n_kps; ++i)
        for (int j = log2_radii[i]; j <= st.max_l2; ++j) st.kidx_ms[j - st.min_l2].push_back(i);
    std::vector<float> prev;
    int n_prev = 0;
    for (int i = 0; i < nr_scales; ++i) {
        float search_radius = powf(p->scale_factor, (float) (st.min_l2 + i));
        float voxel = sqrtf(M_PI * search_radius * search_radius / (float) p->feature_nr_points);
        double ta = now_s();
        const float* in = i == 0 ? pcd : prev.data();
        int n_in = i == 0 ? n : n_prev;
        std::vector<float> ds((size_t) n_in * 12);
        int nd = 0;
        if (orc_downsample(in, n_in, voxel, ORC_ORDER_CANONICAL, ds.data(), &nd)) return -4;
        ds.resize((size_t) nd * 12);
        double tb = now_s();
        orc_normals_knn(ds.data(), nd, nullptr, 0, p->normal_nr_points, vp, p->normals_available);
        double tc = now_s();
        const std::vector<int>& sel = st.kidx_ms[i];
        std::vector<float> sub(sel.size() * 12);
        for (size_t r = 0; r < sel.size(); ++r) memcpy(sub.data() + 12 * r, kps + 12 * (size_t) sel[r], 48);
        st.feat_ms[i].resize(sel.size() * 33);
        orc_fpfh(sub.data(), (int) sel.size(), ds.data(), nd, search_radius, st.feat_ms[i].data(), 0);
        double td = now_s();
        t[0] += tb - ta; t[1] += tc - tb; t[2] += td - tc;
        prev.swap(ds); n_prev = nd;
    }
    return 0;
}

// :264-352: per common level brute-force matches, then one match per query by the proximity vote
void ms_match(const MsStorage& q, const MsStorage& tr, const lgr_orc_params* p, bool inverse_tn, std::vector<int>& out_idx, std::vector<float>& out_dist) {
    std::vector<std::vector<int>> mi(q.n_kps);
    std::vector<std::vector<float>> md(q.n_kps);
    int lo = std::max(q.min_l2, tr.min_l2), hi = std::min(q.max_l2, tr.max_l2);
    for (int level = lo; level <= hi; ++level) {
        const std::vector<int>& qs = q.kidx_ms[level - q.min_l2];
        const std::vector<int>& ts = tr.kidx_ms[level - tr.min_l2];
        const std::vector<float>& qf = q.feat_ms[level - q.min_l2];
        const std::vector<float>& tf = tr.feat_ms[level - tr.min_l2];
        std::vector<int> idx(qs.size());
        std::vector<float> dist(qs.size());
        {   // key-point sub-clouds of this level (kps_multiscale[idx], include/matching.h:243)
            std::vector<float> qp(qs.size() * 12), tp(ts.size() * 12);
            for (size_t r = 0; r < qs.size(); ++r) memcpy(qp.data() + 12 * r, q.kps + 12 * (size_t) qs[r], 48);
            for (size_t r = 0; r < ts.size(); ++r) memcpy(tp.data() + 12 * r, tr.kps + 12 * (size_t) ts[r], 48);
            match_dispatch(p, qp.data(), (int) qs.size(), tp.data(), (int) ts.size(), qf.data(), tf.data(), inverse_tn, idx.data(), dist.data());
        }
        for (size_t i = 0; i < qs.size(); ++i) {
            if (!row_valid(qf.data() + 33 * i) || idx[i] < 0) continue;
            mi[qs[i]].push_back(ts[idx[i]]);
            md[qs[i]].push_back(dist[i]);
        }
    }
    for (int i = 0; i < q.n_kps; ++i) {
        const std::vector<int>& m = mi[i];
        std::vector<float> cnt(m.size(), 0.f);
        for (size_t m1 = 0; m1 < m.size(); ++m1)
            for (size_t m2 = m1; m2 < m.size(); ++m2) {
                const float* a = tr.kps + 12 * (size_t) m[m1];
                const float* b = tr.kps + 12 * (size_t) m[m2];
                float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
                float dist_l2 = std::sqrt((dx * dx + dy * dy) + dz * dz);
                if (dist_l2 < 32 * tr.iss_radius) cnt[m1] += tr.iss_radius / std::max(dist_l2, tr.iss_radius);
            }
        float best_c = 0.f, best_d = 0.f;
        int best = -1;
        for (size_t k = 0; k < m.size(); ++k)
            if (cnt[k] > best_c || (cnt[k] == best_c && md[i][k] < best_d)) { best_c = cnt[k]; best_d = md[i][k]; best = (int) k; }
        out_idx[i] = best >= 0 ? m[best] : -1;
        out_dist[i] = best >= 0 ? md[i][best] : 0.f;
    }
}
}  // namespace

extern "C" int orc_correspondences(const float* src_all, int ns_all, const float* tgt_all, int nt_all, const lgr_orc_params* p,
                                   lgr_orc_corr* out, int* n_out, double* st) {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float search_radius = 0.f, voxel = 0.f;
    if (p->feature_radius > 0.f) {
        int log2_radius = (int) std::floor(std::log2(p->feature_radius) / std::log2(p->scale_factor));
        search_radius = powf(p->scale_factor, (float) log2_radius);
        voxel = sqrtf(M_PI * search_radius * search_radius / (float) p->feature_nr_points);
    }
    std::vector<float> feat[2];
    const float* clouds[2] = {src_all, tgt_all};
    int sizes[2] = {ns_all, nt_all};
    // key points (src/correspondence_search.cpp:8-11): every point, or the ISS detections; kps = pcd[kps_indices]
    // (include/matching.h:167), all later stages work on the key-point clouds and finalize() maps indices back
    std::vector<int> kidx[2];
    std::vector<float> kps[2];
    for (int c = 0; c < 2; ++c) {
        if (p->keypoint_id == ORC_KEYPOINT_ISS) {
            double t0 = now_s();
            kidx[c].resize(sizes[c]);
            int m = 0;
            float r = c == 0 ? p->iss_radius_src : p->iss_radius_tgt;
            if (orc_iss_keypoints(clouds[c], sizes[c], r, 0.975f, 0.975f, 4, kidx[c].data(), &m, nullptr)) return -5;
            kidx[c].resize(m);
            kps[c].resize((size_t) m * 12);
            for (int i = 0; i < m; ++i) memcpy(kps[c].data() + 12 * (size_t) i, clouds[c] + 12 * (size_t) kidx[c][i], 48);
            t[5] += now_s() - t0;
        }
    }
    const bool iss = p->keypoint_id == ORC_KEYPOINT_ISS;
    const float* src = iss ? kps[0].data() : src_all;
    const float* tgt = iss ? kps[1].data() : tgt_all;
    const int ns = iss ? (int) kidx[0].size() : ns_all, nt = iss ? (int) kidx[1].size() : nt_all;
    const float* kclouds[2] = {src, tgt};
    const int ksizes[2] = {ns, nt};
    *n_out = 0;
    if (ns == 0 || nt == 0) { if (st) for (int i = 0; i < 8; ++i) st[i] = t[i]; return 0; }
    std::vector<int> ij(ns), ji(nt);
    std::vector<float> dij(ns), dji(nt);
    const bool need_ji = p->matching_id != ORC_MATCH_ONE_SIDED;
    double t0, t1;
    if (p->feature_radius > 0.f) {
        for (int c = 0; c < 2; ++c) {
            double ta = now_s();
            std::vector<float> ds((size_t) sizes[c] * 12);
            int nd = 0;
            if (orc_downsample(clouds[c], sizes[c], voxel, ORC_ORDER_CANONICAL, ds.data(), &nd)) return -4;
            double tb = now_s();
            const float* vp = c == 0 ? (p->has_vp_src ? p->vp_src : nullptr) : (p->has_vp_tgt ? p->vp_tgt : nullptr);
            orc_normals_knn(ds.data(), nd, nullptr, 0, p->normal_nr_points, vp, p->normals_available);
            double tc = now_s();
            feat[c].resize((size_t) ksizes[c] * 33);
            orc_fpfh(kclouds[c], ksizes[c], ds.data(), nd, search_radius, feat[c].data(), 0);
            double td = now_s();
            t[0] += tb - ta; t[1] += tc - tb; t[2] += td - tc;
        }
        t0 = now_s();
        match_dispatch(p, src, ns, tgt, nt, feat[0].data(), feat[1].data(), false, ij.data(), dij.data());
        if (need_ji) match_dispatch(p, tgt, nt, src, ns, feat[1].data(), feat[0].data(), true, ji.data(), dji.data());
        // NaN query rows have no match (include/matching.h:576,614)
        for (int i = 0; i < ns; ++i) if (!row_valid(feat[0].data() + 33 * (size_t) i)) ij[i] = -1;
        if (need_ji) for (int i = 0; i < nt; ++i) if (!row_valid(feat[1].data() + 33 * (size_t) i)) ji[i] = -1;
        t1 = now_s();
    } else {
        // multi-scale (feature_radius unset): include/matching.h:176-262 + match_multiscale :264-352
        MsStorage st[2];
        for (int c = 0; c < 2; ++c) {
            const float* vp = c == 0 ? (p->has_vp_src ? p->vp_src : nullptr) : (p->has_vp_tgt ? p->vp_tgt : nullptr);
            int rc = ms_initialize(st[c], clouds[c], sizes[c], kclouds[c], ksizes[c], iss ? kidx[c].data() : nullptr,
                                   c == 0 ? p->iss_radius_src : p->iss_radius_tgt, p, vp, t);
            if (rc) return rc;
        }
        t0 = now_s();
        ms_match(st[0], st[1], p, false, ij, dij);
        if (need_ji) ms_match(st[1], st[0], p, true, ji, dji);
        t1 = now_s();
    }
    int rc = orc_filter(p->matching_id, src, ns, tgt, nt, ij.data(), dij.data(), ji.data(), dji.data(),
                        p->distance_thr, p->cluster_k, out, n_out);
    double t2 = now_s();
    t[3] = t1 - t0; t[4] = t2 - t1;
    if (iss && rc == 0)   // finalize(): local key-point indices -> cloud indices
        for (int i = 0; i < *n_out; ++i) { out[i].query = kidx[0][out[i].query]; out[i].match = kidx[1][out[i].match]; }
    if (st) for (int i = 0; i < 8; ++i) st[i] = t[i];
    return rc;
}
