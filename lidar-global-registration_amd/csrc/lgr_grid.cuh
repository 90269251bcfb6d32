// lgr_grid.cuh -- device-side uniform-grid queries shared by the HIP translation units.
//
// Replaces the pcl::KdTreeFLANN queries of the reference (exact search).  Distances are FLANN L2_Simple:
// d2 = ((dx*dx) + dy*dy) + dz*dz in float (no contraction); neighbours are ordered by (d2, original index), the
// oracle's tie rule.  Radius queries are strict d2 < r*r (FLANN RadiusResultSet).
#pragma once
#include "lgr_internal.h"

__device__ __forceinline__ int lgr_cellc(float v, float o, float h) { return (int) floorf((v - o) / h); }

__device__ __forceinline__ float lgr_dist2(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ bool lgr_finite3(float x, float y, float z) {
    return fabsf(x) <= 3.4028234663852886e38f && fabsf(y) <= 3.4028234663852886e38f && fabsf(z) <= 3.4028234663852886e38f;
}

// Per-thread sorted k-best list kept in LDS: slot-major [k][BLOCK] so lane i always hits bank (i % 32).
template <int BLOCK>
struct KnnList {
    float* d2;   // [kcap][BLOCK]
    int* id;     // [kcap][BLOCK]
    int k, count, t;
    float wreg;   // register copy of the k-th distance once the list is full: most candidates are rejected without touching LDS
    __device__ __forceinline__ void init(float* d2s, int* ids, int k_, int tid) { d2 = d2s; id = ids; k = k_; count = 0; t = tid; wreg = 0.f; }
    __device__ __forceinline__ float worst() const { return d2[(k - 1) * BLOCK + t]; }
    __device__ __forceinline__ void push(float d, int i) {
        int pos;
        if (count < k) pos = count++;
        else {
            if (d > wreg) return;
            float wd = wreg;
            int wi = id[(k - 1) * BLOCK + t];
            if (!(d < wd || (d == wd && i < wi))) return;
            pos = k - 1;
        }
        while (pos > 0) {
            float pd = d2[(pos - 1) * BLOCK + t];
            int pi = id[(pos - 1) * BLOCK + t];
            if (d < pd || (d == pd && i < pi)) {
                d2[pos * BLOCK + t] = pd; id[pos * BLOCK + t] = pi; --pos;
            } else break;
        }
        d2[pos * BLOCK + t] = d; id[pos * BLOCK + t] = i;
        if (count == k) wreg = d2[(k - 1) * BLOCK + t];
    }
    __device__ __forceinline__ float dist(int j) const { return d2[j * BLOCK + t]; }
    __device__ __forceinline__ int index(int j) const { return id[j * BLOCK + t]; }
};

// exact k-NN of (qx,qy,qz) in grid g: ring search with the same termination rule as the oracle
// (after ring s every point closer than s*h has been seen).
template <int BLOCK>
__device__ __forceinline__ void lgr_knn_query(const GridDev& g, float qx, float qy, float qz, KnnList<BLOCK>& L) {
    int c0x = lgr_cellc(qx, g.ox, g.h), c0y = lgr_cellc(qy, g.oy, g.h), c0z = lgr_cellc(qz, g.oz, g.h);
    int maxring = max(max(max(c0x, g.dx - 1 - c0x), max(c0y, g.dy - 1 - c0y)), max(max(c0z, g.dz - 1 - c0z), 0));
    for (int s = 0;; ++s) {
        for (int z = c0z - s; z <= c0z + s; ++z) {
            if (z < 0 || z >= g.dz) continue;
            for (int y = c0y - s; y <= c0y + s; ++y) {
                if (y < 0 || y >= g.dy) continue;
                bool edge = (z == c0z - s) || (z == c0z + s) || (y == c0y - s) || (y == c0y + s);
                const size_t row = ((size_t) z * g.dy + y) * g.dx;
                if (edge || s == 0) {
                    // the whole x run of this (z, y) row belongs to the shell: its cells are contiguous in memory
                    int x0 = max(c0x - s, 0), x1 = min(c0x + s, g.dx - 1);
                    if (x0 > x1) continue;
                    int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
                    for (int t = b; t < e; ++t) {
                        float4 p = g.pxyz[t];
                        L.push(lgr_dist2(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w));
                    }
                } else {
                    for (int x = c0x - s; x <= c0x + s; x += 2 * s) {   // the two end cells of an interior row
                        if (x < 0 || x >= g.dx) continue;
                        int b = g.cell_start[row + x], e = g.cell_start[row + x + 1];
                        for (int t = b; t < e; ++t) {
                            float4 p = g.pxyz[t];
                            L.push(lgr_dist2(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w));
                        }
                    }
                }
            }
        }
        if (s >= maxring) break;
        if (L.count == L.k) {
            float lim = (float) s * g.h * 0.999f;
            if (L.worst() <= lim * lim) break;
        }
    }
}

// visit every point of the 27 cells around q in canonical order: cells (z, y, x) ascending, inside a cell
// ascending original index (the sort that built the grid is stable).  f(sorted_position, float4 xyz_idx)
template <class F>
__device__ __forceinline__ void lgr_visit27(const GridDev& g, float qx, float qy, float qz, F&& f) {
    int cx = lgr_cellc(qx, g.ox, g.h), cy = lgr_cellc(qy, g.oy, g.h), cz = lgr_cellc(qz, g.oz, g.h);
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dz - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dy - 1); ++y) {
            int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
            if (x0 > x1) continue;
            size_t c = ((size_t) z * g.dy + y) * g.dx;
            int b = g.cell_start[c + x0], e = g.cell_start[c + x1 + 1];   // three x-cells are contiguous
            for (int t = b; t < e; ++t) f(t, g.pxyz[t]);
        }
}
