// orc_preprocess.cpp -- ORACLE (test infrastructure): loader preprocessing, the steps of
// loadPointClouds (src/common.cpp:429-470) between reading the PLY file and the boundary of the hot path:
//   filterDuplicatePoints (:417-427, PointHash / PointEqual include/common.h:202-232): first occurrence of every exact
//       (x, y, z); NaN coordinates never compare equal, so such points are all kept;
//   intensity = 1 (:446-451); voxel = FINE_VOXEL_SIZE_COEFFICIENT (2) * calculatePointCloudDensity (:453-456, quantile 0.8 of
//       the smoothed 8-NN densities, :202-208); downsamplePointCloud in place; estimateNormalsPoints(NORMAL_NR_POINTS = 30).
// Output order: ORC_ORDER_LIBSTDCXX reproduces the reference's iteration order of std::unordered_set / unordered_map
// (this oracle is built against the same libstdc++); ORC_ORDER_CANONICAL keeps input order / voxel-key order.
#include <cmath>
#include <cstring>
#include <functional>
#include <unordered_set>
#include <vector>

#include "../lgr_oracle.h"

namespace {
struct Pt { float v[12]; };
struct PtHash {   // include/common.h:202-210 + include/utils.h:28-32
    size_t operator()(const Pt& p) const {
        size_t seed = 0;
        for (int a = 0; a < 3; ++a) seed ^= std::hash<float>()(p.v[a]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};
struct PtEq {
    bool operator()(const Pt& a, const Pt& b) const { return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2]; }
};
}  // namespace

extern "C" uint64_t orc_point_hash(float x, float y, float z) {   // pinned against oracle/_ref (combineHash<float>, include/utils.h:28-32)
    Pt p{};
    p.v[0] = x; p.v[1] = y; p.v[2] = z;
    return (uint64_t) PtHash()(p);
}

// out holds n points; order as described above
extern "C" int orc_dedupe(const float* pts, int n, int order, float* out, int* n_out) {
    std::unordered_set<Pt, PtHash, PtEq> set;
    set.reserve(n);
    std::vector<int> first;
    first.reserve(n);
    for (int i = 0; i < n; ++i) {
        Pt p;
        memcpy(p.v, pts + 12 * (size_t) i, 48);
        if (set.insert(p).second) first.push_back(i);
    }
    int m = 0;
    if (order == ORC_ORDER_LIBSTDCXX) {
        for (const Pt& p : set) memcpy(out + 12 * (size_t) m++, p.v, 48);
    } else {
        for (int i : first) memcpy(out + 12 * (size_t) m++, pts + 12 * (size_t) i, 48);
    }
    *n_out = m;
    return 0;
}

// out holds n points.  voxel_out (optional): the voxel size used
extern "C" int orc_preprocess(const float* pts, int n, const float* vp, int normals_available, int order, float* out, int* n_out, float* voxel_out) {
    std::vector<float> u((size_t) n * 12);
    int m = 0;
    orc_dedupe(pts, n, order, u.data(), &m);
    for (int i = 0; i < m; ++i) u[12 * (size_t) i + 8] = 1.f;   // intensity carries the weight
    if (m < 2) return -2;                                        // rassert(pcd->size() > 1) in calculateSmoothedDensities
    float density = 0.f;
    if (orc_cloud_density(u.data(), m, 0.8f, &density)) return -3;
    float voxel = 2 * density;
    int nd = 0;
    int rc = orc_downsample(u.data(), m, voxel, order, out, &nd);
    if (rc) return rc;
    orc_normals_knn(out, nd, nullptr, 0, 30, vp, normals_available);
    *n_out = nd;
    if (voxel_out) *voxel_out = voxel;
    return 0;
}
