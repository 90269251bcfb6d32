// lgr_compat.hpp -- header-only C++ shim that re-exposes the reference's call surface on top of the C ABI (lgr.h).
//
// Mirrors (same names, argument meaning and error behaviour) of:
//   include/alignment.h:6-19              alignPointClouds / alignRansac / alignGror / alignTeaser
//   include/correspondence_search.h:9-28  CorrespondenceSearch, FeatureBasedCorrespondenceSearch
//   include/sac_prerejective_omp.h:21-56  SampleConsensusPrerejectiveOMP
//   include/downsample.h:32               downsamplePointCloud
//   include/common.h:322-332              estimateFeatures<FPFH>
//   include/matching.h:373-376            matchBF<FPFH>
//   include/transformation.h:6-7          estimateOptimalRigidTransformation
//   include/hypotheses.h:10-12            updateHypotheses
//   src/common.cpp:531-547, 644-655       calculateSmoothedDensities, estimateNormalsPoints
//
// The reference passes pcl::PointCloud<pcl::PointXYZINormal> / pcl::FPFHSignature33 / Eigen::Matrix4f.  Neither PCL
// nor Eigen exists in this image, so the shim is written against three tiny layout-compatible types (lgr::PointN is
// the 48-byte PointXYZINormal, lgr::FPFH the 132-byte signature, lgr::Matrix4f a column-major 4x4).  A maintainer of
// the reference defines LGR_COMPAT_POINT_T / LGR_COMPAT_FPFH_T / LGR_COMPAT_MATRIX4F_T to the real types before
// including this header: every access below is either `.points`, `.size()`, `.data()` or a
// reinterpret of the contiguous storage, which the real types provide with the same layout.
#pragma once
#include <array>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lgr.h"

namespace lgr {

#ifndef LGR_COMPAT_POINT_T
struct alignas(16) PointN {   // pcl::PointXYZINormal
    float x = 0, y = 0, z = 0, _pad0 = 1.f;
    float normal_x = 0, normal_y = 0, normal_z = 0, _pad1 = 0;
    float intensity = 0, curvature = 0, _pad2 = 0, _pad3 = 0;
    PointN() = default;
    PointN(float x_, float y_, float z_, float intensity_ = 0.f, float nx = 0.f, float ny = 0.f, float nz = 0.f)
        : x(x_), y(y_), z(z_), normal_x(nx), normal_y(ny), normal_z(nz), intensity(intensity_) {}
};
#else
using PointN = LGR_COMPAT_POINT_T;
#endif
static_assert(sizeof(PointN) == 48, "PointN must be the 48-byte pcl::PointXYZINormal layout");

#ifndef LGR_COMPAT_FPFH_T
struct FPFH { float histogram[33]; };   // pcl::FPFHSignature33
#else
using FPFH = LGR_COMPAT_FPFH_T;
#endif
static_assert(sizeof(FPFH) == 132, "FPFH must be 33 floats");

#ifndef LGR_COMPAT_MATRIX4F_T
struct Matrix4f {   // column-major like Eigen::Matrix4f
    float m[16];
    static Matrix4f Identity() { Matrix4f r{}; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.f : 0.f; return r; }
    float& operator()(int row, int col) { return m[4 * col + row]; }
    float operator()(int row, int col) const { return m[4 * col + row]; }
    float* data() { return m; }
    const float* data() const { return m; }
};
#else
using Matrix4f = LGR_COMPAT_MATRIX4F_T;
#endif

template <class T> struct Cloud {   // the subset of pcl::PointCloud<T> the path touches
    using Ptr = std::shared_ptr<Cloud<T>>;
    using ConstPtr = std::shared_ptr<const Cloud<T>>;
    std::vector<T> points;
    unsigned width = 0, height = 1;
    bool is_dense = true;
    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
};
using PointNCloud = Cloud<PointN>;
using FPFHCloud = Cloud<FPFH>;

// include/common.h:120-127
struct Correspondence {
    int index_query = 0, index_match = -1;
    float distance = 3.4028235e38f, threshold = 0.f;
    Correspondence() = default;
    Correspondence(int q, int m, float d, float thr) : index_query(q), index_match(m), distance(d), threshold(thr) {}
};
static_assert(sizeof(Correspondence) == sizeof(lgr_corr), "Correspondence must match lgr_corr");
using Correspondences = std::vector<Correspondence>;
using CorrespondencesPtr = std::shared_ptr<Correspondences>;
using CorrespondencesConstPtr = std::shared_ptr<const Correspondences>;

// include/common.h:192-195
struct MultivaluedCorrespondence { std::vector<int> match_indices; std::vector<float> distances; };

// include/common.h:135-163 (string ids kept; translated to the ABI enums in to_abi)
struct AlignmentParameters {
    bool reestimate_frames{true};
    int feature_nr_points{352}, normal_nr_points{30};
    float edge_thr_coef{0.95f};
    float distance_thr{0.f}, iss_radius_src{0.f}, iss_radius_tgt{0.f};
    std::optional<float> feature_radius;
    float scale_factor{2.0f};
    float confidence{0.999f};
    bool use_bfmatcher{true};
    int bf_block_size{10000};
    int ratio_k{2}, cluster_k{40};
    int randomness{1}, n_samples{3};
    std::string alignment_id{"ransac"}, descriptor_id{"shot"}, keypoint_id{"iss"};
    std::string metric_id{"combination"}, matching_id{"cluster"}, lrf_id{"default"};
    std::string weight_id{"constant"}, score_id{"mse"};
    int max_iterations{0};
    bool save_features{false};
    std::string testname;
    std::optional<Matrix4f> ground_truth;
    bool fix_seed = true, normals_available = false;
    float match_search_radius = 0;
    std::optional<Matrix4f> guess;
    std::string dir_path;
    std::optional<std::array<float, 3>> vp_src, vp_tgt;
};

// include/common.h:165-174
struct AlignmentResult {
    PointNCloud::ConstPtr src, tgt;
    Matrix4f transformation;
    CorrespondencesConstPtr correspondences;
    int iterations = 0;
    bool converged = false;
    double time_te = 0.0, time_cs = 0.0;
};

// ---- context: one per host thread, created lazily on device 0 (override with set_device before the first call)
inline int& device_ordinal() { static int d = 0; return d; }
inline void set_device(int d) { device_ordinal() = d; }
inline lgr_ctx* context() {
    static thread_local lgr_ctx* ctx = nullptr;
    // a liblgr_hip.so of another ABI revision would read lgr_params with another layout (include/lgr.h LGR_VERSION)
    if (!ctx && lgr_version() != LGR_VERSION)
        throw std::runtime_error("lgr: liblgr_hip.so has ABI revision " + std::to_string(lgr_version()) + ", this header is revision " + std::to_string(LGR_VERSION));
    if (!ctx && lgr_ctx_create(device_ordinal(), LGR_STREAM_OWN, &ctx) != LGR_OK)
        throw std::runtime_error("lgr: no MI355X device / context creation failed (there is no CPU fallback)");
    return ctx;
}
inline void check(int rc, const char* what) {
    if (rc != LGR_OK) throw std::runtime_error(std::string("lgr: ") + what + ": " + lgr_last_error(context()));
}

// the reference falls back instead of failing on unknown ids (src/alignment.cpp:96-100, src/matching.cpp:60-64,
// src/metric.cpp:296-300): unknown matching -> lr, unknown metric -> correspondences, unknown alignment -> ransac
inline lgr_params to_abi(const AlignmentParameters& p) {
    lgr_params a;
    lgr_default_params(&a);
    a.feature_nr_points = p.feature_nr_points; a.normal_nr_points = p.normal_nr_points;
    a.edge_thr_coef = p.edge_thr_coef; a.distance_thr = p.distance_thr;
    a.feature_radius = p.feature_radius.value_or(0.f);   // unset -> 0 -> multi-scale matching (include/matching.h:176-208)
    a.scale_factor = p.scale_factor; a.confidence = p.confidence; a.bf_block_size = p.bf_block_size;
    a.cluster_k = p.cluster_k; a.randomness = p.randomness; a.n_samples = p.n_samples;
    a.alignment_id = p.alignment_id == "gror" ? LGR_ALIGN_GROR : LGR_ALIGN_RANSAC;
    // detectKeyPoints (src/common.cpp:657-691): "iss" -> ISS, anything else -> every point (with a warning there)
    a.keypoint_id = p.keypoint_id == "iss" ? LGR_KEYPOINT_ISS : LGR_KEYPOINT_ANY;
    a.iss_radius_src = p.iss_radius_src; a.iss_radius_tgt = p.iss_radius_tgt;
    a.matching_id = p.matching_id == "cluster" ? LGR_MATCH_CLUSTER : (p.matching_id == "one_sided" ? LGR_MATCH_ONE_SIDED : LGR_MATCH_LR);
    a.metric_id = p.metric_id == "uniformity" ? LGR_METRIC_UNIFORMITY : p.metric_id == "closest_plane" ? LGR_METRIC_CLOSEST_PLANE
                  : p.metric_id == "combination" ? LGR_METRIC_COMBINATION : LGR_METRIC_CORRESPONDENCES;   // weighted_closest_plane: not built
    a.score_id = p.score_id == "mae" ? LGR_SCORE_MAE : (p.score_id == "mse" ? LGR_SCORE_MSE : (p.score_id == "exp" ? LGR_SCORE_EXP : LGR_SCORE_CONSTANT));
    a.max_iterations = p.max_iterations; a.normals_available = p.normals_available; a.fix_seed = p.fix_seed;
    if (p.vp_src) { a.has_vp_src = 1; std::memcpy(a.vp_src, p.vp_src->data(), 12); }
    if (p.vp_tgt) { a.has_vp_tgt = 1; std::memcpy(a.vp_tgt, p.vp_tgt->data(), 12); }
    a.use_bfmatcher = p.use_bfmatcher ? 1 : 0;
    a.match_search_radius = p.match_search_radius;
    if (p.guess) { a.has_guess = 1; std::memcpy(a.guess, p.guess->data(), 64); }
    return a;
}
inline const float* raw(const PointNCloud& c) { return reinterpret_cast<const float*>(c.points.data()); }

// ---- include/downsample.h:32 (pcd_down may alias pcd_fullsize, src/common.cpp:455-456; reference output order)
inline void downsamplePointCloud(const PointNCloud::ConstPtr& pcd_fullsize, PointNCloud::Ptr& pcd_down, float voxel_size) {
    std::vector<PointN> out(pcd_fullsize->size());
    int n_out = 0;
    check(lgr_downsample(context(), raw(*pcd_fullsize), (int) pcd_fullsize->size(), voxel_size, LGR_ORDER_REFERENCE,
                         reinterpret_cast<float*>(out.data()), &n_out), "downsamplePointCloud");
    out.resize(n_out);
    pcd_down->points = std::move(out);
    pcd_down->width = n_out; pcd_down->height = 1; pcd_down->is_dense = true;
}

// ---- src/common.cpp:644-655
inline void estimateNormalsPoints(int k_points, PointNCloud::Ptr& pcd, const PointNCloud::ConstPtr& surface,
                                  const std::optional<std::array<float, 3>>& vp, bool normals_available) {
    check(lgr_normals_knn(context(), reinterpret_cast<float*>(pcd->points.data()), (int) pcd->size(),
                          surface ? raw(*surface) : nullptr, surface ? (int) surface->size() : 0, k_points,
                          vp ? vp->data() : nullptr, normals_available), "estimateNormalsPoints");
}

// ---- include/common.h:322-332 (only the FPFH specialisation exists on this path; the primary template throws, :318-320)
template <class FeatureT>
inline void estimateFeatures(const PointNCloud::ConstPtr&, const PointNCloud::ConstPtr&, typename Cloud<FeatureT>::Ptr&, float, const AlignmentParameters&) {
    throw std::runtime_error("Feature with proposed reference frame isn't supported!");
}
template <>
inline void estimateFeatures<FPFH>(const PointNCloud::ConstPtr& pcd, const PointNCloud::ConstPtr& surface, FPFHCloud::Ptr& features,
                                   float radius_search, const AlignmentParameters&) {
    features->points.resize(pcd->size());
    check(lgr_fpfh(context(), raw(*pcd), (int) pcd->size(), raw(*surface), (int) surface->size(), radius_search,
                   reinterpret_cast<float*>(features->points.data())), "estimateFeatures<FPFH>");
    features->width = (unsigned) pcd->size();
}

// ---- include/matching.h:373-376 (randomness = 1)
template <class FeatureT>
inline std::vector<MultivaluedCorrespondence> matchBF(const typename Cloud<FeatureT>::ConstPtr& query_features,
                                                      const typename Cloud<FeatureT>::ConstPtr& train_features,
                                                      const AlignmentParameters& parameters) {
    static_assert(sizeof(FeatureT) == 132, "only FPFH is built on this path");
    if (parameters.randomness != 1) throw std::runtime_error("lgr: randomness != 1 is not supported (data/test.yaml:14)");
    int mq = (int) query_features->size(), mt = (int) train_features->size();
    std::vector<int32_t> idx(mq);
    std::vector<float> dist(mq);
    check(lgr_match_bf(context(), reinterpret_cast<const float*>(query_features->points.data()), mq,
                       reinterpret_cast<const float*>(train_features->points.data()), mt, parameters.bf_block_size, idx.data(), dist.data()),
          "matchBF");
    std::vector<MultivaluedCorrespondence> out(mq);
    for (int i = 0; i < mq; ++i)
        if (idx[i] >= 0) { out[i].match_indices.push_back(idx[i]); out[i].distances.push_back(dist[i]); }
    return out;
}

// ---- include/matching.h:367-370 (randomness = 1; exact search, so the rows matchBF finds -- tests/flann_bf_matcher.h:82-83)
template <class FeatureT>
inline std::vector<MultivaluedCorrespondence> matchFLANN(const typename Cloud<FeatureT>::ConstPtr& query_features,
                                                         const typename Cloud<FeatureT>::ConstPtr& train_features,
                                                         const AlignmentParameters& parameters) {
    static_assert(sizeof(FeatureT) == 132, "only FPFH is built on this path");
    if (parameters.randomness != 1) throw std::runtime_error("lgr: randomness != 1 is not supported (data/test.yaml:14)");
    int mq = (int) query_features->size(), mt = (int) train_features->size();
    std::vector<int32_t> idx(mq);
    std::vector<float> dist(mq);
    check(lgr_match_flann(context(), reinterpret_cast<const float*>(query_features->points.data()), mq,
                          reinterpret_cast<const float*>(train_features->points.data()), mt, idx.data(), dist.data()), "matchFLANN");
    std::vector<MultivaluedCorrespondence> out(mq);
    for (int i = 0; i < mq; ++i)
        if (idx[i] >= 0) { out[i].match_indices.push_back(idx[i]); out[i].distances.push_back(dist[i]); }
    return out;
}

// ---- include/matching.h:378-382: the reference takes the train cloud as a pcl::search::KdTree; here it is the cloud itself
template <class FeatureT>
inline std::vector<MultivaluedCorrespondence> matchLocal(const PointNCloud::ConstPtr& query_pcd, const PointNCloud::ConstPtr& train_pcd,
                                                         const typename Cloud<FeatureT>::ConstPtr& query_features,
                                                         const typename Cloud<FeatureT>::ConstPtr& train_features,
                                                         const AlignmentParameters& parameters, const Matrix4f& guess) {
    static_assert(sizeof(FeatureT) == 132, "only FPFH is built on this path");
    if (parameters.randomness != 1) throw std::runtime_error("lgr: randomness != 1 is not supported (data/test.yaml:14)");
    int mq = (int) query_features->size(), mt = (int) train_features->size();
    if ((int) query_pcd->size() != mq || (int) train_pcd->size() != mt) throw std::runtime_error("lgr: matchLocal: clouds and feature clouds differ in size");
    std::vector<int32_t> idx(mq);
    std::vector<float> dist(mq);
    check(lgr_match_local(context(), raw(*query_pcd), mq, raw(*train_pcd), mt, reinterpret_cast<const float*>(query_features->points.data()),
                          reinterpret_cast<const float*>(train_features->points.data()), guess.data(), parameters.match_search_radius,
                          idx.data(), dist.data()), "matchLocal");
    std::vector<MultivaluedCorrespondence> out(mq);
    for (int i = 0; i < mq; ++i)
        if (idx[i] >= 0) { out[i].match_indices.push_back(idx[i]); out[i].distances.push_back(dist[i]); }
    return out;
}

// ---- src/common.cpp:531-547
inline std::vector<float> calculateSmoothedDensities(const PointNCloud::ConstPtr& pcd, int k = 2) {
    if (!(pcd->size() > 1 && k >= 2)) throw std::runtime_error("Assertion 3458240390587502 failed!");   // rassert
    std::vector<float> out(pcd->size());
    check(lgr_smoothed_densities(context(), raw(*pcd), (int) pcd->size(), k, out.data()), "calculateSmoothedDensities");
    return out;
}

// ---- include/transformation.h:6-7
inline void estimateOptimalRigidTransformation(const PointNCloud::ConstPtr& src, const PointNCloud::ConstPtr& tgt,
                                               const Correspondences& inliers, Matrix4f& transformation) {
    check(lgr_refit_svd(context(), raw(*src), raw(*tgt), (int) src->size(), (int) tgt->size(),
                        reinterpret_cast<const lgr_corr*>(inliers.data()), (int) inliers.size(), transformation.data()),
          "estimateOptimalRigidTransformation");
}

// ---- include/hypotheses.h:10-12
inline void updateHypotheses(std::vector<Matrix4f>& transformations, std::vector<float>& metrics, const Matrix4f& new_transformation,
                             float new_metric, const AlignmentParameters& parameters) {
    if (transformations.size() != metrics.size()) throw std::runtime_error("Assertion 45832351834023 failed!");
    int n = (int) metrics.size(), cap = n + 1;
    std::vector<float> buf((size_t) cap * 16), met(cap);
    for (int i = 0; i < n; ++i) { std::memcpy(&buf[16 * (size_t) i], transformations[i].data(), 64); met[i] = metrics[i]; }
    int m = lgr_update_hypotheses(buf.data(), met.data(), n, cap, new_transformation.data(), new_metric, parameters.distance_thr);
    if (m < 0) throw std::runtime_error("lgr: updateHypotheses failed");
    transformations.resize(m); metrics.resize(m);
    for (int i = 0; i < m; ++i) { std::memcpy(transformations[i].data(), &buf[16 * (size_t) i], 64); metrics[i] = met[i]; }
}

// ---- include/correspondence_search.h:9-28
class CorrespondenceSearch {
public:
    virtual CorrespondencesPtr calculateCorrespondences() = 0;
    virtual ~CorrespondenceSearch() = default;
};
class FeatureBasedCorrespondenceSearch : CorrespondenceSearch {
public:
    FeatureBasedCorrespondenceSearch() = delete;
    FeatureBasedCorrespondenceSearch(PointNCloud::ConstPtr src, PointNCloud::ConstPtr tgt, AlignmentParameters parameters)
        : src_(std::move(src)), tgt_(std::move(tgt)), parameters_(std::move(parameters)) {}
    CorrespondencesPtr calculateCorrespondences() override {
        // key points: "iss" or every point (src/common.cpp:657-691; other ids fall back to every point there too)
        if (parameters_.descriptor_id != "fpfh") throw std::runtime_error("lgr: only descriptor 'fpfh' is built on the device path");
        lgr_params a = to_abi(parameters_);
        auto out = std::make_shared<Correspondences>(src_->size());
        int n = 0;
        check(lgr_correspondences(context(), raw(*src_), (int) src_->size(), raw(*tgt_), (int) tgt_->size(), &a,
                                  reinterpret_cast<lgr_corr*>(out->data()), &n), "calculateCorrespondences");
        out->resize(n);
        return out;
    }
protected:
    PointNCloud::ConstPtr src_, tgt_;
    AlignmentParameters parameters_;
};

// ---- include/sac_prerejective_omp.h:21-56
class SampleConsensusPrerejectiveOMP {
public:
    SampleConsensusPrerejectiveOMP() = delete;
    SampleConsensusPrerejectiveOMP(PointNCloud::ConstPtr src, PointNCloud::ConstPtr tgt, CorrespondencesConstPtr correspondences,
                                   AlignmentParameters parameters)
        : src_(std::move(src)), tgt_(std::move(tgt)), correspondences_(std::move(correspondences)), parameters_(std::move(parameters)) {}
    AlignmentResult align() {
        lgr_params a = to_abi(parameters_);
        lgr_result r;
        check(lgr_ransac(context(), raw(*src_), (int) src_->size(), raw(*tgt_), (int) tgt_->size(),
                         reinterpret_cast<const lgr_corr*>(correspondences_->data()), (int) correspondences_->size(), &a, &r, nullptr),
              "SampleConsensusPrerejectiveOMP::align");
        AlignmentResult out;
        out.src = src_; out.tgt = tgt_; out.correspondences = correspondences_;
        std::memcpy(out.transformation.data(), r.transformation, 64);
        out.iterations = r.iterations; out.converged = r.converged != 0; out.time_te = r.time_te;
        return out;
    }
    inline std::string getClassName() const { return "SampleConsensusPrerejectiveOMP"; }
protected:
    PointNCloud::ConstPtr src_, tgt_;
    CorrespondencesConstPtr correspondences_;
    AlignmentParameters parameters_;
};

// ---- include/alignment.h:6-19
inline AlignmentResult alignRansac(const PointNCloud::ConstPtr& src, const PointNCloud::ConstPtr& tgt,
                                   const CorrespondencesPtr& correspondences, const AlignmentParameters& parameters) {
    SampleConsensusPrerejectiveOMP ransac(src, tgt, correspondences, parameters);
    return ransac.align();
}
// src/alignment.cpp:21-35: resolution = distance_thr, K_optimal = 800, iterations = 1, converged = true
inline AlignmentResult alignGror(const PointNCloud::ConstPtr& src, const PointNCloud::ConstPtr& tgt, const CorrespondencesPtr& correspondences,
                                 const AlignmentParameters& parameters) {
    lgr_result r;
    check(lgr_gror(context(), raw(*src), (int) src->size(), raw(*tgt), (int) tgt->size(),
                   reinterpret_cast<const lgr_corr*>(correspondences->data()), (int) correspondences->size(), parameters.distance_thr, 800, &r, nullptr),
          "alignGror");
    AlignmentResult out;
    out.src = src; out.tgt = tgt; out.correspondences = correspondences;
    std::memcpy(out.transformation.data(), r.transformation, 64);
    out.iterations = 1; out.converged = true; out.time_te = r.time_te;
    return out;
}
inline AlignmentResult alignTeaser(const PointNCloud::ConstPtr&, const PointNCloud::ConstPtr&, const CorrespondencesPtr&, const AlignmentParameters&) {
    throw std::runtime_error("Not implemented: support TEASER");   // src/alignment.cpp:40
}
// src/alignment.cpp:72-109 without the two CSV side effects (correspondences csv, transformations.csv)
inline AlignmentResult alignPointClouds(const PointNCloud::ConstPtr& src, const PointNCloud::ConstPtr& tgt, const AlignmentParameters& params) {
    FeatureBasedCorrespondenceSearch corr_search(src, tgt, params);
    CorrespondencesPtr correspondences = corr_search.calculateCorrespondences();
    AlignmentResult result;
    if (params.alignment_id == "gror") result = alignGror(src, tgt, correspondences, params);
    else if (params.alignment_id == "teaser") result = alignTeaser(src, tgt, correspondences, params);
    else result = alignRansac(src, tgt, correspondences, params);   // unknown ids fall back to RANSAC (:96-100)
    return result;
}

}  // namespace lgr
