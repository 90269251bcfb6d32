// lgr_grid.cuh -- device-side uniform-grid queries shared by the HIP translation units.
//
// Replaces the pcl::KdTreeFLANN queries of the reference (exact search).  Distances are FLANN L2_Simple:
// d2 = ((dx*dx) + dy*dy) + dz*dz in float (no contraction); neighbours are ordered by (d2, original index), the
// oracle's tie rule.  Radius queries are strict d2 < r*r (FLANN RadiusResultSet).
#pragma once
#include "lgr_internal.h"

// XCD-aware tile order for kernels whose workgroups walk the (cell, Morton)-sorted points: the hardware places workgroup b on XCD
// b % 8 (speed only -- any order gives the same results), so with the identity mapping the eight neighbours b .. b+7 of the sorted order
// pull the same candidate cells into eight different L2s.  lgr_xcd_tile hands XCD x the contiguous tile range [x * per, (x + 1) * per):
// consecutive tiles of the sorted order run back to back on ONE L2.  Launch lgr_xcd_grid(n_tiles) workgroups; a tile >= n_tiles has
// nothing to do.
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int lgr_xcd_grid(int n_tiles) { return ((n_tiles + 7) >> 3) << 3; }
__device__ __forceinline__ int lgr_xcd_tile(int b, int n_tiles) { return (b & 7) * ((n_tiles + 7) >> 3) + (b >> 3); }

__device__ __forceinline__ int lgr_cellc(float v, float o, float h) { return (int) floorf((v - o) / h); }

__device__ __forceinline__ float lgr_dist2(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ bool lgr_finite3(float x, float y, float z) {
    return fabsf(x) <= 3.4028234663852886e38f && fabsf(y) <= 3.4028234663852886e38f && fabsf(z) <= 3.4028234663852886e38f;
}

// Per-thread k-best list kept in LDS, slot-major [k][BLOCK] so lane i always hits bank (i % 32).  While the search runs
// it is a binary MAX-heap under the total order (d2, index): a candidate that beats the root replaces it and sifts down in at
// most log2(k) steps.  (A sorted list with insertion costs up to k shifts per accepted candidate, and a wave pays the longest
// shift of any of its lanes for nearly every candidate: 46 k vector + 57 k scalar instructions per wave in the normals
// kernel.)  finalize() heap-sorts in place, after which dist(j) / index(j) are in ascending (d2, index) order -- the same
// k entries in the same order as a sorted insertion would have produced, because the order is total.
template <int BLOCK>
struct KnnList {
    float* d2;   // [kcap][BLOCK]
    int* id;     // [kcap][BLOCK]
    int k, count, t;
    float wreg;   // register copy of the root's distance once the heap is full: most candidates are rejected without touching LDS
    __device__ __forceinline__ void init(float* d2s, int* ids, int k_, int tid) { d2 = d2s; id = ids; k = k_; count = 0; t = tid; wreg = 0.f; }
    __device__ __forceinline__ static bool less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }
    __device__ __forceinline__ float worst() const { return d2[t]; }   // root (valid when count == k)
    __device__ __forceinline__ void sift_down(int pos, int n, float d, int i) {   // place (d, i) at pos of the heap [0, n)
        for (;;) {
            int c = 2 * pos + 1;
            if (c >= n) break;
            float cd = d2[c * BLOCK + t];
            int ci = id[c * BLOCK + t];
            if (c + 1 < n) {
                float rd = d2[(c + 1) * BLOCK + t];
                int ri = id[(c + 1) * BLOCK + t];
                if (less(cd, ci, rd, ri)) { cd = rd; ci = ri; ++c; }
            }
            if (!less(d, i, cd, ci)) break;
            d2[pos * BLOCK + t] = cd; id[pos * BLOCK + t] = ci;
            pos = c;
        }
        d2[pos * BLOCK + t] = d; id[pos * BLOCK + t] = i;
    }
    __device__ __forceinline__ void push(float d, int i) {
        if (count < k) {
            int pos = count++;   // sift up
            while (pos > 0) {
                int par = (pos - 1) >> 1;
                float pd = d2[par * BLOCK + t];
                int pi = id[par * BLOCK + t];
                if (!less(pd, pi, d, i)) break;
                d2[pos * BLOCK + t] = pd; id[pos * BLOCK + t] = pi;
                pos = par;
            }
            d2[pos * BLOCK + t] = d; id[pos * BLOCK + t] = i;
            if (count == k) wreg = d2[t];
            return;
        }
        if (d > wreg) return;
        if (!less(d, i, wreg, id[t])) return;
        sift_down(0, k, d, i);
        wreg = d2[t];
    }
    __device__ __forceinline__ void finalize() {   // heap sort: ascending (d2, index)
        for (int n = count - 1; n > 0; --n) {
            float ld = d2[n * BLOCK + t];
            int li = id[n * BLOCK + t];
            d2[n * BLOCK + t] = d2[t]; id[n * BLOCK + t] = id[t];
            sift_down(0, n, ld, li);
        }
    }
    __device__ __forceinline__ float dist(int j) const { return d2[j * BLOCK + t]; }
    __device__ __forceinline__ int index(int j) const { return id[j * BLOCK + t]; }
};

// candidates [b, e) of the sorted point array, pushed in ascending position (the order the list's tie rule is defined on).  Four
// loads are issued before the first push: the query kernels run at ~2.5 waves per SIMD (the k-best lists fill the LDS), so a
// load per candidate, each waited for, was the kernels' critical path.
template <int BLOCK>
__device__ __forceinline__ void lgr_knn_scan(const GridDev& g, int b, int e, float qx, float qy, float qz, KnnList<BLOCK>& L) {
    for (int t = b; t < e; t += 4) {
        const int last = e - 1;
        const float4 p0 = g.pxyz[t], p1 = g.pxyz[min(t + 1, last)], p2 = g.pxyz[min(t + 2, last)], p3 = g.pxyz[min(t + 3, last)];
        L.push(lgr_dist2(qx, qy, qz, p0.x, p0.y, p0.z), __float_as_int(p0.w));
        if (t + 1 < e) L.push(lgr_dist2(qx, qy, qz, p1.x, p1.y, p1.z), __float_as_int(p1.w));
        if (t + 2 < e) L.push(lgr_dist2(qx, qy, qz, p2.x, p2.y, p2.z), __float_as_int(p2.w));
        if (t + 3 < e) L.push(lgr_dist2(qx, qy, qz, p3.x, p3.y, p3.z), __float_as_int(p3.w));
    }
}

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
                    lgr_knn_scan(g, b, e, qx, qy, qz, L);
                } else {
                    for (int x = c0x - s; x <= c0x + s; x += 2 * s) {   // the two end cells of an interior row
                        if (x < 0 || x >= g.dx) continue;
                        int b = g.cell_start[row + x], e = g.cell_start[row + x + 1];
                        lgr_knn_scan(g, b, e, qx, qy, qz, L);
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
    L.finalize();
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
            // four loads in flight before the first callback (a load per candidate, each waited for, is a dependent round trip per
            // candidate: the closest-plane metric went from 38 to 16 ms per alignment with this, round 3); same visiting order
            for (int t = b; t < e; t += 4) {
                const int last = e - 1;
                const float4 q0 = g.pxyz[t], q1 = g.pxyz[min(t + 1, last)], q2 = g.pxyz[min(t + 2, last)], q3 = g.pxyz[min(t + 3, last)];
                f(t, q0);
                if (t + 1 < e) f(t + 1, q1);
                if (t + 2 < e) f(t + 2, q2);
                if (t + 3 < e) f(t + 3, q3);
            }
        }
}
