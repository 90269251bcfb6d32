// lgr_knn_wave.cuh -- exact k-NN with ONE WAVE PER QUERY (lane = candidate), round 3.
//
// Same answer as the per-thread heap search of lgr_grid.cuh (lgr_knn_query): the k smallest candidates under the total order
// (d2, original index), d2 = lgr_dist2 in float, in ascending order -- the oracle's rule.  Only the search
// strategy differs, and the strategy cannot change the answer:
//
//   guess a threshold key K = (r2, index) -> collect every grid point with key <= K from the cells the ball of radius
//   sqrt(r2) can touch -> accept when k <= count <= CAP (the k nearest are then the k smallest keys collected: every point
//   that is not collected has a larger key than all of them), otherwise move K and scan again.
//
// A key is the 64-bit integer (bits(d2) << 32) | index: d2 >= +0, so integer order on keys IS the (d2, index) order, keys of
// different points differ, and count(K) grows by one point at a time -- a bisection on K between a value known to give
// fewer than k and one known to give more than CAP always ends (duplicate points, equal distances and all).  The first
// guess of a query is the k-th distance of the previous query of the wave (consecutive queries are neighbours in the grid's
// sorted order) times a factor that aims at ~1.25 k candidates, so nearly every query is done after one scan.
//
// Why: the heap kernel keeps 64 per-thread lists in LDS (k * 512 bytes per wave: 2 waves per SIMD at k = 40), and a wave pays
// a full sift for nearly every candidate because SOME lane accepts it -- ~2200 instructions per query at k = 40.  Here a
// candidate costs one lane a distance and a compare; selection happens once per query: the survivors (<= CAP = 64 KPL) are
// ranked by counting (rank = number of smaller keys, one broadcast LDS read per survivor).
#pragma once
#include "lgr_grid.cuh"

template <int KPL>
struct WaveKnn {
    static constexpr int CAP = 64 * KPL;
    static constexpr int BUF = CAP + 4;   // LDS entries per wave (the last four pad the unrolled rank loop)
    unsigned long long key[KPL];          // this lane's survivors: entries lane + 64 j of the buffer (all ones beyond m)
    int rank[KPL];                        // position of key[j] in ascending order (>= m for padding)
    int m;                                // survivors: k <= m <= CAP, or all grid points when the grid holds fewer than k
};

__device__ __forceinline__ unsigned long long wk_key(float d2, int idx) {
    return ((unsigned long long) __float_as_uint(d2) << 32) | (unsigned) idx;
}
__device__ __forceinline__ float wk_key_d2(unsigned long long key) { return __uint_as_float((unsigned) (key >> 32)); }

// first / last cell of one axis that can hold a point within r of q (r already carries the rounding margin of the
// distance computation).  cell(v) = floor((v - o) / h) is monotonic in v and the grid clamps cells to [0, dim - 1], so
// every point p with |p - q| <= r has its cell inside [lo, hi].
__device__ __forceinline__ void wk_axis(float q, float r, float o, float h, int dim, int& lo, int& hi) {
    float a = q - r, b = q + r;
    a = a - fabsf(a) * 1e-6f;
    b = b + fabsf(b) * 1e-6f;
    lo = (int) fminf(fmaxf(floorf((a - o) / h), 0.f), (float) (dim - 1));
    hi = (int) fminf(fmaxf(floorf((b - o) / h), 0.f), (float) (dim - 1));
}

// one scan: every grid point with key <= kthr goes to buf (the first CAP of them), the return value counts ALL of them
template <int CAP>
__device__ __forceinline__ int wk_scan(const GridDev& g, float qx, float qy, float qz, unsigned long long kthr, unsigned long long* __restrict__ buf) {
    const int lane = threadIdx.x & 63;
    float r = __builtin_sqrtf(wk_key_d2(kthr));
    r = r * 1.00001f + 1e-30f;   // computed d2 <= r2  =>  true |dx| <= sqrt(r2) (1 + 4 ulp)
    if (!(r <= 3.0e38f)) r = __uint_as_float(0x7f800000u);   // the all-ones key ("everything"): the whole grid
    int x0, x1, y0, y1, z0, z1;
    wk_axis(qx, r, g.ox, g.h, g.dx, x0, x1);
    wk_axis(qy, r, g.oy, g.h, g.dy, y0, y1);
    wk_axis(qz, r, g.oz, g.h, g.dz, z0, z1);
    x0 = __builtin_amdgcn_readfirstlane(x0); x1 = __builtin_amdgcn_readfirstlane(x1);
    y0 = __builtin_amdgcn_readfirstlane(y0); y1 = __builtin_amdgcn_readfirstlane(y1);
    z0 = __builtin_amdgcn_readfirstlane(z0); z1 = __builtin_amdgcn_readfirstlane(z1);
    const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;
    int m = 0;
    auto take = [&](bool valid, const float4& p) {
        const unsigned long long key = wk_key(lgr_dist2(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w));
        const bool keep = valid && key <= kthr;
        const unsigned long long mask = __ballot(keep);
        if (mask) {
            const int pos = m + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mask, 0u));
            if (keep && pos < CAP) buf[pos] = key;
            m += __popcll(mask);
        }
    };
    // lane <-> (z, y) row of the box: the x cells of a row are one contiguous run of the sorted point array
    for (int rb = 0; rb < nrows; rb += 64) {
        const int rr = rb + lane;
        int b = 0, e = 0;
        if (rr < nrows) {
            const int z = z0 + rr / ny, y = y0 + rr % ny;
            const size_t row = ((size_t) z * g.dy + y) * g.dx;
            b = g.cell_start[row + x0];
            e = g.cell_start[row + x1 + 1];
        }
        unsigned long long ne = __ballot(e > b);
        while (ne) {
            // four rows per round, their first 64 candidates loaded before any is looked at (one dependent round trip, not four)
            int sb[4], se[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sb[u] = 0; se[u] = 0;
                if (ne) {
                    const int j = __builtin_ctzll(ne);
                    ne &= ne - 1ull;
                    sb[u] = __builtin_amdgcn_readlane(b, j);
                    se[u] = __builtin_amdgcn_readlane(e, j);
                }
            }
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = (sb[u] + lane < se[u]) ? g.pxyz[sb[u] + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (se[u] > sb[u]) take(sb[u] + lane < se[u], p[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                for (int t0 = sb[u] + 64; t0 < se[u]; t0 += 64) {   // the rest of a long row
                    const bool valid = t0 + lane < se[u];
                    const float4 pp = valid ? g.pxyz[t0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                    take(valid, pp);
                }
        }
    }
    return m;
}

// The whole wave answers ONE query (qx, qy, qz wave-uniform and finite, g.n > 0).  r2_guess: in = first threshold to try
// (any positive value; only the number of scans depends on it), out = the guess for the wave's next query.
// buf: this wave's WaveKnn<KPL>::BUF entries of LDS.
template <int KPL>
__device__ __forceinline__ void lgr_wave_knn(const GridDev& g, float qx, float qy, float qz, int k, float& r2_guess,
                                             unsigned long long* __restrict__ buf, WaveKnn<KPL>& W) {
    constexpr int CAP = WaveKnn<KPL>::CAP;
    const int lane = threadIdx.x & 63;
    const unsigned long long ALL = ~0ull;
    unsigned long long lo = 0ull, hi = ALL;      // count(lo) < k (or lo = 0: nothing known), count(hi) > CAP (or hi = ALL: nothing known)
    bool lo_known = false, hi_known = false;
    unsigned long long kthr = g.n <= CAP ? ALL : (((unsigned long long) __float_as_uint(r2_guess) << 32) | 0xffffffffull);
    int m = 0;
    for (int iter = 0;; ++iter) {
        __builtin_amdgcn_wave_barrier();
        m = wk_scan<CAP>(g, qx, qy, qz, kthr, buf);
        if (m <= CAP && (m >= k || kthr == ALL)) break;
        const float r2 = wk_key_d2(kthr);
        unsigned long long next;
        if (m < k) {
            lo = kthr; lo_known = true;
            // the count grows like r2 on a surface: aim at 1.25 k, at least +30 %, at most x 4 per step
            float f = m > 0 ? 1.25f * (float) k / (float) m : 4.f;
            f = fminf(fmaxf(f, 1.3f), 4.f);
            const float nr2 = fmaxf(r2 * f, g.h * g.h * 1e-6f);
            next = nr2 < 3.0e38f ? (((unsigned long long) __float_as_uint(nr2) << 32) | 0xffffffffull) : ALL;
        } else {
            hi = kthr; hi_known = true;
            float f = 1.25f * (float) k / (float) m;
            f = fminf(fmaxf(f, 0.05f), 0.8f);
            float base = r2 < 3.0e38f ? r2 : 3.0e38f;
            next = ((unsigned long long) __float_as_uint(base * f) << 32) | 0xffffffffull;
        }
        // keep the new threshold strictly between what is known; after a few estimates bisect the integer keys (always ends:
        // count() steps by one point per key and count(lo) < k <= CAP < count(hi))
        const bool inside = (!lo_known || next > lo) && (!hi_known || next < hi);
        if (!inside || iter >= 5) {
            if (lo_known && hi_known) next = lo + ((hi - lo) >> 1);
            else if (!inside) next = lo_known ? ALL : 0ull;   // (cannot happen: growing from lo without hi, shrinking from hi without lo)
        }
        kthr = next;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 4) buf[m + lane] = ALL;   // padding for the unrolled rank loop (never smaller than a key)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    W.m = m;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        W.key[j] = (lane + 64 * j < m) ? buf[lane + 64 * j] : ALL;
        W.rank[j] = 0;
    }
    for (int i = 0; i < m; i += 4) {
        const unsigned long long k0 = buf[i], k1 = buf[i + 1], k2 = buf[i + 2], k3 = buf[i + 3];   // wave-uniform addresses: LDS broadcasts
#pragma unroll
        for (int j = 0; j < KPL; ++j)
            W.rank[j] += (int) (k0 < W.key[j]) + (int) (k1 < W.key[j]) + (int) (k2 < W.key[j]) + (int) (k3 < W.key[j]);
    }
    // next guess: the k-th distance found, widened so that the next scan collects about 1.25 k (count ~ r2), but not more than
    // halfway to the buffer's capacity
    const int kk = min(k, m) - 1;
    float dk = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const unsigned long long hit = __ballot(W.rank[j] == kk && W.key[j] != ALL);
        if (hit) dk = __uint_as_float(__builtin_amdgcn_readlane((int) (W.key[j] >> 32), __builtin_ctzll(hit)));
    }
    if (m >= k && dk > 0.f) {
        const float target = fminf(1.25f * (float) k, 0.5f * (float) (k + CAP));
        r2_guess = dk * (target / (float) k);
    }
}
