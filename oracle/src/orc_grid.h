// orc_grid.h -- uniform grid over a point cloud: exact radius / k-NN queries for the ORACLE.
//
// The reference uses pcl::KdTreeFLANN (exact search, results sorted by squared distance, FLANN L2_Simple
// distance ((dx*dx)+dy*dy)+dz*dz in float, radius test strict d2 < r*r).  Tie order among equidistant points is
// FLANN-internal (unspecified); the oracle orders by (d2, index).  See SURVEY.md A.3.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace orc {

static inline bool finite3(const float* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

static inline float dist2(const float* a, const float* b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

struct Grid {
    const float* pts = nullptr;  // stride 12 floats
    int n = 0;
    float origin[3] = {0, 0, 0};
    float h = 1.f;
    int dim[3] = {1, 1, 1};
    std::vector<int> cell_start;  // ncell + 1
    std::vector<int> order;       // point indices sorted by (cell id, index)

    // cell coordinate of a coordinate value along axis a (not clamped). Same float ops as the HIP grid.
    inline int cellc(float v, int a) const { return (int) std::floor((v - origin[a]) / h); }

    void build(const float* p, int n_, float h_) {
        pts = p; n = n_; h = h_;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < n; ++i) {
            const float* q = pts + 12 * (size_t) i;
            if (!finite3(q)) continue;
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], q[a]); mx[a] = std::max(mx[a], q[a]); }
        }
        if (!(mn[0] <= mx[0])) { for (int a = 0; a < 3; ++a) { mn[a] = 0; mx[a] = 0; } }
        for (int a = 0; a < 3; ++a) origin[a] = mn[a];
        for (;;) {
            double cells = 1;
            for (int a = 0; a < 3; ++a) { dim[a] = cellc(mx[a], a) + 1; if (dim[a] < 1) dim[a] = 1; cells *= dim[a]; }
            if (cells <= 128e6) break;
            h *= 2.f;
        }
        size_t ncell = (size_t) dim[0] * dim[1] * dim[2];
        cell_start.assign(ncell + 1, 0);
        std::vector<int> cid(n, -1);
        for (int i = 0; i < n; ++i) {
            const float* q = pts + 12 * (size_t) i;
            if (!finite3(q)) continue;
            int cx = cellc(q[0], 0), cy = cellc(q[1], 1), cz = cellc(q[2], 2);
            cx = std::min(std::max(cx, 0), dim[0] - 1); cy = std::min(std::max(cy, 0), dim[1] - 1); cz = std::min(std::max(cz, 0), dim[2] - 1);
            cid[i] = (cz * dim[1] + cy) * dim[0] + cx;
            cell_start[cid[i] + 1]++;
        }
        for (size_t c = 0; c < ncell; ++c) cell_start[c + 1] += cell_start[c];
        order.resize(cell_start[ncell]);
        std::vector<int> cur(cell_start.begin(), cell_start.end() - 1);
        for (int i = 0; i < n; ++i) if (cid[i] >= 0) order[cur[cid[i]]++] = i;  // ascending index within a cell
    }

    // visit all points of cells [c-1, c+1]^3 around q in canonical order: (cz, cy, cx) lexicographic, then index.
    template <class F> inline void visit27(const float* q, F&& f) const {
        int cx = cellc(q[0], 0), cy = cellc(q[1], 1), cz = cellc(q[2], 2);
        for (int z = std::max(cz - 1, 0); z <= std::min(cz + 1, dim[2] - 1); ++z)
            for (int y = std::max(cy - 1, 0); y <= std::min(cy + 1, dim[1] - 1); ++y)
                for (int x = std::max(cx - 1, 0); x <= std::min(cx + 1, dim[0] - 1); ++x) {
                    size_t c = ((size_t) z * dim[1] + y) * dim[0] + x;
                    for (int s = cell_start[c]; s < cell_start[c + 1]; ++s) f(order[s]);
                }
    }

    struct Cand { float d2; int idx; };
    static inline bool cand_less(const Cand& a, const Cand& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); }

    // exact k nearest by (d2, idx); out sorted ascending; returns count found (<= k)
    int knn(const float* q, int k, Cand* out) const {
        std::vector<Cand> heap;  // max-heap on cand_less
        heap.reserve(k + 1);
        if (!finite3(q)) return 0;
        int c0[3] = {cellc(q[0], 0), cellc(q[1], 1), cellc(q[2], 2)};
        int maxring = 0;
        for (int a = 0; a < 3; ++a) maxring = std::max(maxring, std::max(c0[a], dim[a] - 1 - c0[a]));
        maxring = std::max(maxring, 0);
        for (int s = 0;; ++s) {
            // scan shell at Chebyshev distance s
            for (int z = c0[2] - s; z <= c0[2] + s; ++z) {
                if (z < 0 || z >= dim[2]) continue;
                for (int y = c0[1] - s; y <= c0[1] + s; ++y) {
                    if (y < 0 || y >= dim[1]) continue;
                    bool edge_zy = (z == c0[2] - s || z == c0[2] + s || y == c0[1] - s || y == c0[1] + s);
                    int step = edge_zy ? 1 : 2 * s;
                    if (step == 0) step = 1;
                    for (int x = c0[0] - s; x <= c0[0] + s; x += step) {
                        if (x < 0 || x >= dim[0]) continue;
                        size_t c = ((size_t) z * dim[1] + y) * dim[0] + x;
                        for (int t = cell_start[c]; t < cell_start[c + 1]; ++t) {
                            int i = order[t];
                            Cand cd{dist2(q, pts + 12 * (size_t) i), i};
                            if ((int) heap.size() < k) {
                                heap.push_back(cd);
                                std::push_heap(heap.begin(), heap.end(), cand_less);
                            } else if (cand_less(cd, heap.front())) {
                                std::pop_heap(heap.begin(), heap.end(), cand_less);
                                heap.back() = cd;
                                std::push_heap(heap.begin(), heap.end(), cand_less);
                            }
                        }
                    }
                }
            }
            if (s >= maxring) break;
            if ((int) heap.size() == k) {
                float lim = (float) s * h * 0.999f;
                if (heap.front().d2 <= lim * lim) break;
            }
        }
        std::sort_heap(heap.begin(), heap.end(), cand_less);
        for (size_t i = 0; i < heap.size(); ++i) out[i] = heap[i];
        return (int) heap.size();
    }
};

// pick a grid cell size so that the average occupied cell holds ~target points (2-D manifold heuristic)
static inline float auto_cell(const float* pts, int n, float target) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const float* q = pts + 12 * (size_t) i;
        if (!finite3(q)) continue;
        ++cnt;
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], q[a]); mx[a] = std::max(mx[a], q[a]); }
    }
    if (cnt < 2) return 1.f;
    float e[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    std::sort(e, e + 3);
    float area = std::max(e[2] * e[1], 1e-12f);
    float h = std::sqrt(area * target / (float) cnt);
    float hmin = e[2] / 1024.f;
    return std::max(h, std::max(hmin, 1e-6f));
}

}  // namespace orc
