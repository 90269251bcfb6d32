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

#ifndef WK_GUESS
#define WK_GUESS 1.25f
#endif
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

// Threshold search over a scan functor (scan(kthr) = number of points with key <= kthr among the functor's candidates, the first
// CAP of them in buf).  kmax: the largest threshold the candidate set is COMPLETE for (ALL = every grid point is a candidate).
// Returns false when fewer than k points have a key <= kmax (the caller needs a larger candidate set); otherwise W holds the
// ranked survivors.  r2_guess: in = first threshold to try (any positive value; only the number of scans depends on it),
// out = the guess for the wave's next query.  all_fit: the whole grid fits the buffer (then one scan with ALL does it).
template <int KPL, class Scan>
__device__ __forceinline__ bool wk_search(Scan&& scan, unsigned long long kmax, bool all_fit, float r2_floor, int k, float& r2_guess,
                                          unsigned long long* __restrict__ buf, WaveKnn<KPL>& W) {
    constexpr int CAP = WaveKnn<KPL>::CAP;
    const int lane = threadIdx.x & 63;
    const unsigned long long ALL = ~0ull;
    unsigned long long lo = 0ull, hi = ALL;      // count(lo) < k (or lo = 0: nothing known), count(hi) > CAP (or hi = ALL: nothing known)
    bool lo_known = false, hi_known = false;
    unsigned long long kthr = ((unsigned long long) __float_as_uint(r2_guess) << 32) | 0xffffffffull;
    if (kthr > kmax) kthr = kmax;
    if (all_fit && kmax == ALL) kthr = ALL;
    int m = 0;
    for (int iter = 0;; ++iter) {
        __builtin_amdgcn_wave_barrier();
        m = scan(kthr);
        if (m <= CAP && (m >= k || kthr == ALL)) break;
        if (m < k && kthr == kmax) return false;
        const float r2 = wk_key_d2(kthr);
        unsigned long long next;
        if (m < k) {
            lo = kthr; lo_known = true;
            // the count grows like r2 on a surface: aim at 1.25 k, at least +30 %, at most x 4 per step
            float f = m > 0 ? 1.25f * (float) k / (float) m : 4.f;
            f = fminf(fmaxf(f, 1.3f), 4.f);
            const float nr2 = fmaxf(r2 * f, r2_floor);
            next = nr2 < 3.0e38f ? (((unsigned long long) __float_as_uint(nr2) << 32) | 0xffffffffull) : ALL;
            if (next > kmax) next = kmax;
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
            else if (!inside) next = lo_known ? kmax : 0ull;   // (cannot happen: growing from lo without hi, shrinking from hi without lo)
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
        const float target = fminf(WK_GUESS * (float) k, 0.5f * (float) (k + CAP));
        r2_guess = dk * (target / (float) k);
    }
    return true;
}

// The whole wave answers ONE query (qx, qy, qz wave-uniform and finite, g.n > 0) from its own box of cells.
// buf: this wave's WaveKnn<KPL>::BUF entries of LDS.
template <int KPL>
__device__ __forceinline__ void lgr_wave_knn(const GridDev& g, float qx, float qy, float qz, int k, float& r2_guess,
                                             unsigned long long* __restrict__ buf, WaveKnn<KPL>& W) {
    constexpr int CAP = WaveKnn<KPL>::CAP;
    (void) wk_search<KPL>([&](unsigned long long kthr) { return wk_scan<CAP>(g, qx, qy, qz, kthr, buf); }, ~0ull, g.n <= CAP, g.h * g.h * 1e-6f,
                          k, r2_guess, buf, W);
}

// ---- 64 queries per wave, the queries of one cell sharing one candidate set.
// The queries of a grid cell see nearly the same neighbourhood, and most of a single query's instructions above are set-up (cell
// ranges with their divisions, row addresses, the dependent loads of the cell table and then of the candidates).  So: lane l holds
// query l; the wave takes the queries cell by cell; for a cell it loads the points of the (2 s + 1)^3 cells around it ONCE into
// registers (lane = candidate, up to WK_NBATCH batches of 64; the rows' runs are flattened, so every lane holds a candidate), and each
// query of the cell then runs its threshold search on those registers -- a scan is a distance, a compare and a ballot per batch, and a
// second scan (threshold moved) costs no memory access.  A query may only use thresholds up to its SAFE radius: the distance to
// the nearest face of the box that has cells behind it -- every point closer than that is among the candidates; if k points
// are not found below it (or the box holds more than 64 WK_NBATCH points) the query falls back to lgr_wave_knn with its own box.
constexpr int WK_NBATCH = 8, WK_SMAX = 3;

// emit(l, W): called wave-uniformly once per valid query l (lane index), W = its ranked survivors (W.m = 0 is never passed)
// rowtab: 128 ints of LDS per wave.  valid / px / py / pz: this lane's query (finite when valid).  g.n > 0.
template <int KPL, class Emit>
__device__ __forceinline__ void lgr_wave_knn_tile(const GridDev& g, bool valid, float px, float py, float pz, int k, float& guess,
                                                  unsigned long long* __restrict__ buf, int* __restrict__ rowtab, Emit&& emit) {
    constexpr int CAP = WaveKnn<KPL>::CAP;
    const int lane = threadIdx.x & 63;
    const unsigned long long ALL = ~0ull;
    const float INF = __uint_as_float(0x7f800000u);
    int cx = 0, cy = 0, cz = 0;
    if (valid) {
        cx = min(max(lgr_cellc(px, g.ox, g.h), 0), g.dx - 1);
        cy = min(max(lgr_cellc(py, g.oy, g.h), 0), g.dy - 1);
        cz = min(max(lgr_cellc(pz, g.oz, g.h), 0), g.dz - 1);
    }
    const bool all_fit = g.n <= CAP;
    const float r2_floor = g.h * g.h * 1e-6f;
    unsigned long long rem = __ballot(valid);
    WaveKnn<KPL> W;
    while (rem) {
        const int l0 = __builtin_ctzll(rem);
        const int gx = __builtin_amdgcn_readlane(cx, l0), gy = __builtin_amdgcn_readlane(cy, l0), gz = __builtin_amdgcn_readlane(cz, l0);
        const unsigned long long grp = __ballot(valid && cx == gx && cy == gy && cz == gz) & rem;
        rem &= ~grp;
        // the box: s cells each way, s from the current guess (a query on the cell's border reaches s h beyond it)
        int s = (int) ceilf(__builtin_sqrtf(guess) * 1.05f / g.h);
        s = __builtin_amdgcn_readfirstlane(min(max(s, 1), WK_SMAX));
        const int x0 = max(gx - s, 0), x1 = min(gx + s, g.dx - 1), y0 = max(gy - s, 0), y1 = min(gy + s, g.dy - 1);
        const int z0 = max(gz - s, 0), z1 = min(gz + s, g.dz - 1);
        const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;   // <= 49
        int b = 0, e = 0;
        if (lane < nrows) {
            const int z = z0 + lane / ny, y = y0 + lane % ny;
            const size_t row = ((size_t) z * g.dy + y) * g.dx;
            b = g.cell_start[row + x0];
            e = g.cell_start[row + x1 + 1];
        }
        const int len = e - b;
        int inc = len;   // inclusive scan over the rows
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        const int T = __builtin_amdgcn_readlane(inc, 63);
        const bool fast = T <= 64 * WK_NBATCH;
        float4 p[WK_NBATCH];
        float rs2 = 0.f;
        if (fast) {
            __builtin_amdgcn_wave_barrier();
            rowtab[lane] = lane < nrows ? inc - len : 0x7fffffff;   // first flat position of row `lane`
            rowtab[64 + lane] = b - (inc - len);                     // sorted position = flat position + this
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int bi = 0; bi < WK_NBATCH; ++bi) {
                p[bi] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bi * 64 < T) {
                    const int f = bi * 64 + lane;
                    int r = 0;   // the last row that starts at or before f (empty rows share their successor's start)
#pragma unroll
                    for (int st = 32; st > 0; st >>= 1) if (rowtab[r + st] <= f) r += st;
                    if (f < T) p[bi] = g.pxyz[f + rowtab[64 + r]];
                }
            }
            // this lane's query: how far it can see inside the box.  A point outside the box on the low x side has cell <= x0 - 1, i.e.
            // floor((v - ox) / h) < x0, so v < ox + x0 h (1 + a few ulp): below L = ox + x0 h + margin; likewise above U on the high side.
            // (rounding of v - ox, of the division and of ox + x0 h: < 4.1 ulp of |ox| + dx h; the margin is eight times that)
            const float mgx = 2e-6f * (fabsf(g.ox) + (float) g.dx * g.h), mgy = 2e-6f * (fabsf(g.oy) + (float) g.dy * g.h);
            const float mgz = 2e-6f * (fabsf(g.oz) + (float) g.dz * g.h);
            float rs = INF;
            if (x0 > 0) rs = fminf(rs, px - (g.ox + (float) x0 * g.h + mgx));
            if (x1 < g.dx - 1) rs = fminf(rs, (g.ox + (float) (x1 + 1) * g.h - mgx) - px);
            if (y0 > 0) rs = fminf(rs, py - (g.oy + (float) y0 * g.h + mgy));
            if (y1 < g.dy - 1) rs = fminf(rs, (g.oy + (float) (y1 + 1) * g.h - mgy) - py);
            if (z0 > 0) rs = fminf(rs, pz - (g.oz + (float) z0 * g.h + mgz));
            if (z1 < g.dz - 1) rs = fminf(rs, (g.oz + (float) (z1 + 1) * g.h - mgz) - pz);
            rs = fmaxf(rs, 0.f) * 0.9999f;   // computed d2 <= r2  =>  true |dx| <= sqrt(r2) (1 + 4 ulp)
            rs2 = rs * rs;
        }
        // (the queries that need their own box wait until the cell's candidates are dead: the two paths then share the registers)
        unsigned long long todo = fast ? grp : 0ull, own = fast ? 0ull : grp;
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const float qx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(px), l));
            const float qy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(py), l));
            const float qz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pz), l));
            const float r2max = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rs2), l));
            const unsigned long long kmax = r2max < 3.0e38f ? (((unsigned long long) __float_as_uint(r2max) << 32) | 0xffffffffull) : ALL;
            const bool ok = wk_search<KPL>([&](unsigned long long kthr) {
                int m = 0;
#pragma unroll
                for (int bi = 0; bi < WK_NBATCH; ++bi)
                    if (bi * 64 < T) {
                        const unsigned long long key = wk_key(lgr_dist2(qx, qy, qz, p[bi].x, p[bi].y, p[bi].z), __float_as_int(p[bi].w));
                        const bool keep = bi * 64 + lane < T && key <= kthr;
                        const unsigned long long mask = __ballot(keep);
                        if (mask) {
                            const int pos = m + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mask, 0u));
                            if (keep && pos < CAP) buf[pos] = key;
                            m += __popcll(mask);
                        }
                    }
                return m;
            }, kmax, all_fit, r2_floor, k, guess, buf, W);
            if (ok) emit(l, W);
            else own |= 1ull << l;
        }
        while (own) {
            const int l = __builtin_ctzll(own);
            own &= own - 1ull;
            const float qx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(px), l));
            const float qy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(py), l));
            const float qz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pz), l));
            lgr_wave_knn<KPL>(g, qx, qy, qz, k, guess, buf, W);
            emit(l, W);
        }
    }
}
