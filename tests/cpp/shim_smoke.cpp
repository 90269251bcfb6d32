// Compiles against the header-only shim exactly like a reference call site would (tests/point2plane_distance.cpp:81
// style): build two clouds, alignPointClouds, check the transform.  Run by tests/test_host_shim.py on the GPU box.
#include <cmath>
#include <cstdio>
#include <limits>
#include <random>

#include "../../lidar-global-registration_amd/host/lgr_compat.hpp"

using namespace lgr;

int main() {
    std::mt19937 gen(566);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    auto src = std::make_shared<PointNCloud>(), tgt = std::make_shared<PointNCloud>();
    // bumpy surface patch 3 m x 2 m; tgt = src region shifted, then rotated about z by 30 degrees and translated
    auto h = [](float x, float y) { return 0.25f * std::sin(2.1f * x + 0.4f * y) + 0.2f * std::cos(1.3f * y - 0.7f * x) + 0.1f * std::sin(5.f * x * y * 0.2f); };
    const float c = std::cos(0.5235988f), s = std::sin(0.5235988f);
    for (int i = 0; i < 12000; ++i) {
        float x = 3.f * U(gen), y = 2.f * U(gen);
        src->points.emplace_back(x, y, h(x, y), 1.f);
        float x2 = 3.f * U(gen), y2 = 2.f * U(gen), z2 = h(x2, y2);
        tgt->points.emplace_back(c * x2 - s * y2 + 1.f, s * x2 + c * y2 - 2.f, z2 + 0.5f, 1.f);
    }
    AlignmentParameters p;
    p.distance_thr = 0.1f; p.feature_radius = 0.25f; p.bf_block_size = 200000; p.max_iterations = 50000;
    p.keypoint_id = "any"; p.descriptor_id = "fpfh"; p.matching_id = "lr"; p.metric_id = "uniformity";
    p.vp_src = std::array<float, 3>{0.f, 0.f, 10.f}; p.vp_tgt = std::array<float, 3>{1.f, -2.f, 10.5f};
    AlignmentResult r = alignPointClouds(src, tgt, p);
    const Matrix4f& T = r.transformation;
    float err = std::fabs(T(0, 0) - c) + std::fabs(T(1, 0) - s) + std::fabs(T(0, 3) - 1.f) + std::fabs(T(1, 3) + 2.f) + std::fabs(T(2, 3) - 0.5f);
    std::printf("converged=%d iterations=%d correspondences=%zu err=%g\n", (int) r.converged, r.iterations, r.correspondences->size(), err);
    // stand-alone entry points of the surface
    auto down = std::make_shared<PointNCloud>();
    downsamplePointCloud(src, down, 0.05f);
    std::vector<float> dens = calculateSmoothedDensities(src);
    std::printf("downsampled %zu -> %zu, density[0]=%g\n", src->size(), down->size(), dens[0]);
    // the reference's tests/flann_bf_matcher.h:40-97 through the shim: matchBF == matchFLANN == matchLocal(identity, FLT_MAX)
    estimateNormalsPoints(30, down, nullptr, p.vp_src, false);
    auto fsrc = std::make_shared<FPFHCloud>(), ftgt = std::make_shared<FPFHCloud>();
    auto down_t = std::make_shared<PointNCloud>();
    downsamplePointCloud(tgt, down_t, 0.05f);
    estimateNormalsPoints(30, down_t, nullptr, p.vp_tgt, false);
    estimateFeatures<FPFH>(down, down, fsrc, 0.25f, p);
    estimateFeatures<FPFH>(down_t, down_t, ftgt, 0.25f, p);
    AlignmentParameters pl = p;
    pl.guess = Matrix4f::Identity();
    pl.match_search_radius = std::numeric_limits<float>::max();
    auto bf = matchBF<FPFH>(fsrc, ftgt, p), fl = matchFLANN<FPFH>(fsrc, ftgt, p);
    auto lo = matchLocal<FPFH>(down, down_t, fsrc, ftgt, pl, *pl.guess);
    size_t differ = 0, matched = 0;
    // an index may differ only between rows at the same descriptor distance (BF / FLANN keep the lowest index of a tie, Local
    // the spatially nearest: the reference's real scans have no exact ties, this smooth synthetic patch has a few)
    auto close = [](float a, float b) { return std::fabs(a - b) <= 1e-8f + 1e-5f * std::fabs(b); };   // tests/flann_bf_matcher.h:12-14
    for (size_t i = 0; i < bf.size(); ++i) {
        matched += !bf[i].match_indices.empty();
        if (bf[i].match_indices.size() != fl[i].match_indices.size() || bf[i].match_indices.size() != lo[i].match_indices.size()) { ++differ; continue; }
        if (bf[i].match_indices.empty()) continue;
        if (bf[i].match_indices != fl[i].match_indices && !close(bf[i].distances[0], fl[i].distances[0])) ++differ;
        if (bf[i].match_indices != lo[i].match_indices && !close(bf[i].distances[0], lo[i].distances[0])) ++differ;
    }
    std::printf("flann_bf_matcher: %zu queries, %zu matched, %zu differ\n", bf.size(), matched, differ);
    // guided second step: the pose found above as the guess, matchLocal inside alignPointClouds
    AlignmentParameters p2 = p;
    p2.guess = T; p2.match_search_radius = 0.3f; p2.max_iterations = 5000;
    AlignmentResult r2 = alignPointClouds(src, tgt, p2);
    const Matrix4f& T2 = r2.transformation;
    float err2 = std::fabs(T2(0, 0) - c) + std::fabs(T2(1, 0) - s) + std::fabs(T2(0, 3) - 1.f) + std::fabs(T2(1, 3) + 2.f) + std::fabs(T2(2, 3) - 0.5f);
    std::printf("guided: converged=%d err=%g\n", (int) r2.converged, err2);
    // (err: five absolute component errors summed; the global step is a 3-point RANSAC hypothesis refitted on its inliers over a 3 m patch -- 0.15 is
    //  3 cm per component, the guided second step must then be within 1 cm per component)
    return (r.converged && err < 0.15f && down->size() > 100 && down->size() < src->size() && differ == 0 && matched > bf.size() / 2 &&
            r2.converged && err2 < 0.05f) ? 0 : 1;
}
