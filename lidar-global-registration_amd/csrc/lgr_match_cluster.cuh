// lgr_match_cluster.cuh -- 1. two-level k-means on a sample, assignment and placement of the rows.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
#pragma once
#include "lgr_match_common.cuh"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// 1. clustering (any centres are valid -- they only shape the error bounds and the tile schedule).  Lloyd steps with
//    ORDER-FREE sums: every sample coordinate is turned into a 64-bit integer on a per-call power-of-two grid
//    (2^40 steps up to the largest sample magnitude) and added with integer atomics, so the centres -- hence the tile
//    schedule and the timing -- repeat from run to run whatever order the atomics land in, and a Lloyd step is one
//    kernel (label + accumulate) instead of a labelling kernel plus a deterministic tree reduction per centre.
struct KmAcc { long long sum[33]; long long cnt; };
__device__ __forceinline__ double km_scale(unsigned kmax_bits) {   // 2^(40 - e), 2^e <= largest |sample value| < 2^(e+1)
    return kmax_bits ? ldexp(1.0, 40 - ((int) (kmax_bits >> 23) - 127)) : 1.0;
}
// ---- irregular rows (round 5).  The rotated 30-coordinate operand format (FMT_F16R: six MFMA steps instead of seven, and the only format
// the coarse sweep of the final pass exists for) needs every 11-bin block of every row to have the same sum.  FPFH rows do -- except the
// all-zero rows PCL writes for points whose neighbours carry no weight; 99 + 62 such rows among the 2 M of the planar scene used to cost
// the pair the format (match stage 37 ms instead of 19).  So: the consensus of the three block sums is found by a vote over the k-means
// sample (IRR_CAND candidate rows; the winner needs 99.5 % of the finite sample rows within 1e-5 of its sums, otherwise the lane stays
// off -- rows of arbitrary floats have no consensus), rows off the consensus are IRREGULAR: left out of the sample, the clustering and the
// operand packing (assign_kernel lists them), and matched by an exact side scan in both roles (irregular_scan, lgr_match_rerank.cuh)
// that feeds the same packed atomicMin tables as the exact rerank.  More than IRR_CAP of them on a side: the call is
// rebuilt with the lane off.  Results cannot depend on any of this: every (query, train) pair is still covered by an exact path.
constexpr int IRR_CAND = 16;
constexpr int IRR_CAP = 1024;
struct IrrRef { double s[3]; double tol; int enabled; int best; unsigned n_ok; unsigned agree[IRR_CAND]; };   // zeroed per call
__device__ __forceinline__ void block_sums(const float* v, double& s0, double& s1, double& s2) {
    s0 = 0.0; s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 11; ++k) { s0 += (double) v[k]; s1 += (double) v[11 + k]; s2 += (double) v[22 + k]; }
}
__device__ __forceinline__ bool sums_agree(double a0, double a1, double a2, double r0, double r1, double r2, double tol) {
    return fabs(a0 - r0) <= tol && fabs(a1 - r1) <= tol && fabs(a2 - r2) <= tol;   // (NaN: false)
}
__device__ __forceinline__ double sums_tol(double r0, double r1, double r2) { return 1e-5 * fmax(fmax(fabs(r0), fabs(r1)), fabs(r2)); }
// row of sample s (the same rows km_sample takes): B == nullptr: all 2 * per_side samples from A (evenly spaced), otherwise per_side from each set
__device__ __forceinline__ const float* sample_row(int s, const float* __restrict__ A, int ma, const float* __restrict__ B, int mb, int per_side) {
    const bool fromA = !B || s < per_side;
    const float* X = fromA ? A : B;
    const int m = fromA ? ma : mb;
    if (m <= 0) return nullptr;
    const int t = fromA ? s : s - per_side;
    const long long i = (long long) t * m / (B ? per_side : 2 * per_side);
    return X + (size_t) i * 33;
}
// the winner of the vote (lowest candidate among equals) and whether it carries a consensus; the same for every thread that asks
__device__ __forceinline__ bool irr_decide(const IrrRef* __restrict__ ref, int& best) {
    unsigned top = 0u;
    best = 0;
    for (int c = 0; c < IRR_CAND; ++c) { const unsigned a = ref->agree[c]; if (a > top) { top = a; best = c; } }
    const unsigned n_ok = ref->n_ok;
    return n_ok >= 256u && (unsigned long long) top * 1000ull >= (unsigned long long) n_ok * 995ull;
}
__global__ __launch_bounds__(256) void km_consensus(const float* __restrict__ A, int ma, const float* __restrict__ B, int mb, int per_side, IrrRef* __restrict__ ref) {
    __shared__ double cs[IRR_CAND][3];
    __shared__ int cok[IRR_CAND];
    const int ns = 2 * per_side;
    if (threadIdx.x < IRR_CAND) {
        const float* r = sample_row((int) ((long long) threadIdx.x * ns / IRR_CAND), A, ma, B, mb, per_side);
        float v[33];
        const bool ok = r && row_finite(r, v);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (ok) block_sums(v, s0, s1, s2);
        cs[threadIdx.x][0] = s0; cs[threadIdx.x][1] = s1; cs[threadIdx.x][2] = s2;
        cok[threadIdx.x] = ok ? 1 : 0;
    }
    __syncthreads();
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (s < ns) {
        const float* r = sample_row(s, A, ma, B, mb, per_side);
        float v[33];
        ok = r && row_finite(r, v);
        if (ok) block_sums(v, s0, s1, s2);
    }
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < IRR_CAND; ++c) {
        const bool a = ok && cok[c] && sums_agree(s0, s1, s2, cs[c][0], cs[c][1], cs[c][2], sums_tol(cs[c][0], cs[c][1], cs[c][2]));
        const int n = __popcll(__ballot(a));
        if (lane == 0 && n) atomicAdd(&ref->agree[c], (unsigned) n);
    }
    const int n_ok = __popcll(__ballot(ok));
    if (lane == 0 && n_ok) atomicAdd(&ref->n_ok, (unsigned) n_ok);
}
// irr_ref: the vote of km_consensus (nullptr: lane off).  Sample rows off the consensus are left out like non-finite ones; thread 0 publishes
// the decision (reference sums, tolerance, enabled) for assign_kernel.
__global__ void km_sample(const float* __restrict__ A, int ma, const float* __restrict__ B, int mb, int per_side,
                          float* __restrict__ smp, int* __restrict__ smp_ok, unsigned* __restrict__ kmax /* zeroed: max |v| bits */, IrrRef* __restrict__ irr_ref) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 2 * per_side) return;
    const float* r = sample_row(s, A, ma, B, mb, per_side);
    float v[33];
    bool ok = r && row_finite(r, v);
    if (irr_ref) {
        int best;
        const bool on = irr_decide(irr_ref, best);
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
        if (on) {
            const float* rr = sample_row((int) ((long long) best * (2 * per_side) / IRR_CAND), A, ma, B, mb, per_side);
            float w[33];
            (void) row_finite(rr, w);   // (a winner is a finite row)
            block_sums(w, r0, r1, r2);
            if (ok) {
                double s0, s1, s2;
                block_sums(v, s0, s1, s2);
                ok = sums_agree(s0, s1, s2, r0, r1, r2, sums_tol(r0, r1, r2));
            }
        }
        if (s == 0) { irr_ref->s[0] = r0; irr_ref->s[1] = r1; irr_ref->s[2] = r2; irr_ref->tol = sums_tol(r0, r1, r2); irr_ref->best = best; irr_ref->enabled = on ? 1 : 0; }
    }
    float mx = 0.f;
    for (int k = 0; k < 33; ++k) { smp[(size_t) s * 33 + k] = ok ? v[k] : 0.f; if (ok) mx = fmaxf(mx, fabsf(v[k])); }
    smp_ok[s] = ok ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(kmax, __float_as_uint(mx));
}
__global__ void km_init(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns, float* __restrict__ cen) {
    int c = threadIdx.x;
    if (c >= KCL) return;
    int s = (int) ((long long) c * ns / KCL);
    int tries = 0;
    while (!smp_ok[s] && tries < ns) { s = (s + 1) % ns; ++tries; }
    for (int k = 0; k < 33; ++k) cen[c * 33 + k] = smp_ok[s] ? smp[(size_t) s * 33 + k] : 0.f;
}
__device__ __forceinline__ int nearest_centre(const float* v, const float* __restrict__ cen, float& best) {
    int bi = 0;
    best = __uint_as_float(0x7f800000u);
#pragma unroll 1
    for (int c = 0; c < KCL; ++c) {
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) { float t = v[k] - cen[c * 33 + k]; d = d + t * t; }
        if (d < best) { best = d; bi = c; }
    }
    return bi;
}
// One Lloyd step of the first level: the centres of this step come from the previous step's sums (an empty cluster keeps
// its centre), every sample is labelled and, when acc_out is given, added to its centre's sums.  The last launch
// (acc_out = nullptr) only labels and leaves the final centres in cen_out.
constexpr int KM1_THREADS = 256;
__global__ __launch_bounds__(KM1_THREADS) void km1_step(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns, const unsigned* __restrict__ kmax,
                                                        const float* __restrict__ cen_prev, const KmAcc* __restrict__ acc_prev, KmAcc* __restrict__ acc_out,
                                                        float* __restrict__ cen_out, int* __restrict__ label) {
    __shared__ float cen_s[KCL * 33];
    __shared__ long long acc_s[KCL * 34];
    const double scale = km_scale(*kmax);
    for (int e = threadIdx.x; e < KCL * 33; e += KM1_THREADS) {
        const int c = e / 33, k = e % 33;
        float v = cen_prev[e];
        if (acc_prev && acc_prev[c].cnt > 0) v = (float) (((double) acc_prev[c].sum[k] / scale) / (double) acc_prev[c].cnt);
        cen_s[e] = v;
        if (blockIdx.x == 0) cen_out[e] = v;
    }
    for (int e = threadIdx.x; e < KCL * 34; e += KM1_THREADS) acc_s[e] = 0;
    __syncthreads();
    const int s = blockIdx.x * KM1_THREADS + threadIdx.x;
    if (s < ns) {
        int c = -1;
        if (smp_ok[s]) {
            float v[33], d;
#pragma unroll
            for (int k = 0; k < 33; ++k) v[k] = smp[(size_t) s * 33 + k];
            c = nearest_centre(v, cen_s, d);
            if (acc_out) {
#pragma unroll
                for (int k = 0; k < 33; ++k) atomicAdd((unsigned long long*) &acc_s[c * 34 + k], (unsigned long long) (long long) rint((double) v[k] * scale));
                atomicAdd((unsigned long long*) &acc_s[c * 34 + 33], 1ull);
            }
        }
        label[s] = c;
    }
    if (!acc_out) return;
    __syncthreads();
    long long* out = (long long*) acc_out;
    for (int e = threadIdx.x; e < KCL * 34; e += KM1_THREADS)
        if (acc_s[e] != 0) atomicAdd((unsigned long long*) &out[e], (unsigned long long) acc_s[e]);
}
// second level: `sub` centres inside every cluster.  The samples are first put in cluster order (stable sort of the sample
// indices by their level-1 label, lgr_sort_pairs_u32; coff[p] = first sorted position of cluster p, coff[KCL] = number of
// labelled samples), so a cluster's samples are a contiguous range: seeding is a direct index, and a workgroup of the Lloyd
// step sees one cluster only -- its sub-centres and its integer sums fit in a few KB of LDS.
__global__ void km_label_keys(const int* __restrict__ label, int ns, unsigned* __restrict__ keys, int* __restrict__ vals) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    const int l = label[s];
    keys[s] = l < 0 ? (unsigned) KCL : (unsigned) l;   // unlabelled (non-finite) samples last
    vals[s] = s;
}
__global__ void km_cluster_offsets(const unsigned* __restrict__ keys_sorted, int ns, int* __restrict__ coff /* [KCL + 1] */) {
    const int p = threadIdx.x;
    if (p > KCL) return;
    int lo = 0, hi = ns;   // first position with key >= p
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys_sorted[mid] < (unsigned) p) lo = mid + 1; else hi = mid; }
    coff[p] = lo;
}
// seeds: evenly spaced members of the cluster in sample order -- leaf j starts from the member of rank ceil(j cnt / sub) (the
// first rank r with floor(r sub / cnt) == j); leaves without such a member (cnt < sub) start from the cluster centre
__global__ __launch_bounds__(64) void km2_init(const float* __restrict__ smp, const int* __restrict__ sidx, const int* __restrict__ coff,
                                               const float* __restrict__ cen, int sub, float* __restrict__ cen2) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const int c0 = coff[p], cnt = coff[p + 1] - c0;
    for (int j = lane; j < sub; j += 64) {
        const float* src = cen + p * 33;
        if (cnt > 0) {
            const long long r = ((long long) j * cnt + sub - 1) / sub;
            if (r < cnt && (int) (r * sub / cnt) == j) src = smp + (size_t) sidx[c0 + (int) r] * 33;
        }
        for (int k = 0; k < 33; ++k) cen2[((size_t) p * sub + j) * 33 + k] = src[k];
    }
}
__device__ __forceinline__ int nearest_sub(const float* v, const float* __restrict__ c2 /* [sub][33] of the row's cluster */, int sub, float& best) {
    int bj = 0;
    best = __uint_as_float(0x7f800000u);
#pragma unroll 1
    for (int j = 0; j < sub; ++j) {
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) { float t = v[k] - c2[j * 33 + k]; d = d + t * t; }
        if (d < best) { best = d; bj = j; }
    }
    return bj;
}
constexpr int KM2_THREADS = 256, KM2_LANES = 4, KM2_SAMPLES = KM2_THREADS / KM2_LANES, KM2_PIECES = 48;
// Lloyd step of the second level: a workgroup takes 64 consecutive samples of ONE cluster at a time (grid: x = piece, y = cluster; the pieces
// stride over the cluster), FOUR lanes per sample: each walks a quarter of the cluster's sub-centres (every distance the same k-ordered
// sum as nearest_sub's), the quarters meet through two shuffles as (distance, index) pairs -- the smallest distance, the lowest index among
// equals: nearest_sub's answer bit for bit -- and each lane adds a quarter of the sample's coordinates to the leaf's integer sums in LDS; one
// global atomic per touched sum at the end.  (One lane per sample was a chain of 64 x 33 dependent additions on a device with two waves per
// CU in flight: 57 us per step, six steps per alignment; round 5.)  km2_finalize turns the sums into the new centres (an empty leaf keeps its
// centre) and clears them for the next step.
__global__ __launch_bounds__(KM2_THREADS) void km2_step(const float* __restrict__ smp, const int* __restrict__ sidx, const int* __restrict__ coff,
                                                        const unsigned* __restrict__ kmax, const float* __restrict__ cen2, int sub, KmAcc* __restrict__ acc2) {
    __shared__ float c2s[SUBMAX * 33];
    __shared__ long long acc_s[SUBMAX * 34];
    const int p = blockIdx.y;
    const int c0 = coff[p], s1 = coff[p + 1];
    if (c0 + (int) blockIdx.x * KM2_SAMPLES >= s1) return;
    for (int e = threadIdx.x; e < sub * 33; e += KM2_THREADS) c2s[e] = cen2[(size_t) p * sub * 33 + e];
    for (int e = threadIdx.x; e < sub * 34; e += KM2_THREADS) acc_s[e] = 0;
    __syncthreads();
    const double scale = km_scale(*kmax);
    const int q = threadIdx.x & (KM2_LANES - 1);
    for (int s0 = c0 + blockIdx.x * KM2_SAMPLES; s0 < s1; s0 += gridDim.x * KM2_SAMPLES) {   // (workgroup uniform)
        const int s = s0 + (threadIdx.x >> 2);
        const bool act = s < s1;
        float v[33];
        const float* row = smp + (size_t) sidx[act ? s : s1 - 1] * 33;
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = row[k];
        // this lane's sub-centres: q, q + 4, q + 8, ... (neighbouring lanes read neighbouring LDS rows)
        float best = __uint_as_float(0x7f800000u);
        int bj = 0x7fffffff;
#pragma unroll 1
        for (int j = q; j < sub; j += KM2_LANES) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < 33; ++k) { float t = v[k] - c2s[j * 33 + k]; d = d + t * t; }
            if (d < best) { best = d; bj = j; }
        }
#pragma unroll
        for (int o = 1; o < KM2_LANES; o <<= 1) {
            const float ob = __shfl_xor(best, o);
            const int oj = __shfl_xor(bj, o);
            if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        if (bj == 0x7fffffff) bj = 0;   // (every distance NaN or +inf: nearest_sub answers 0)
        if (act) {
#pragma unroll
            for (int k = 0; k < 33; ++k)
                if ((k & (KM2_LANES - 1)) == q) atomicAdd((unsigned long long*) &acc_s[bj * 34 + k], (unsigned long long) (long long) rint((double) v[k] * scale));
            if (q == 1) atomicAdd((unsigned long long*) &acc_s[bj * 34 + 33], 1ull);
        }
    }
    __syncthreads();
    long long* out = (long long*) (acc2 + (size_t) p * sub);
    for (int e = threadIdx.x; e < sub * 34; e += KM2_THREADS)
        if (acc_s[e] != 0) atomicAdd((unsigned long long*) &out[e], (unsigned long long) acc_s[e]);
}
__global__ __launch_bounds__(64) void km2_finalize(KmAcc* __restrict__ acc2, const unsigned* __restrict__ kmax, float* __restrict__ cen2) {
    const int leaf = blockIdx.x, k = threadIdx.x;
    const long long cnt = acc2[leaf].cnt;
    __syncthreads();
    if (k < 33) {
        if (cnt > 0) cen2[(size_t) leaf * 33 + k] = (float) (((double) acc2[leaf].sum[k] / km_scale(*kmax)) / (double) cnt);
        acc2[leaf].sum[k] = 0;
    } else if (k == 33) acc2[leaf].cnt = 0;
}

// key = (leaf << 22) | (bits(r2) >> 9), leaf = cluster * sub + sub-centre: sort by cluster, leaf, then distance to the
// cluster centre.  Invalid rows: 0xffffffff.  counts[leaf] / counts[MAXLEAF] (invalid) and the squared leaf radii
// rmax[leaf] = max |x - c_leaf|^2 (float bits) are accumulated through LDS.
constexpr int ASSIGN_THREADS = 1024;   // the sub-centres (up to 135 KB of LDS) allow one workgroup per CU: its size is the kernel's occupancy (256 / 512 / 1024
                                        // threads: match stage 16.95 / 16.33 / 16.20 ms at 1M)
// nstat (the operand statistics the f16 packing needs before it can choose its scale; they used to cost a pass of their own over
// both sets): [0] largest finite |x - c|^2 (float bits) over the centres the row will be packed against -- its own cluster's
// (role 0, query side) or all KCL (role 1, train side: one operand copy per column set) --, [1] the largest energy of the three
// coordinates the rotated 30-D format drops (u_j = block sum of x - c over sqrt(11); an upper bound from the row's and the centre's
// block sums, see the loop), [2] set when such a |x - c|^2 overflows float, [3] the number of irregular rows (all of them, listed or not).
__global__ __launch_bounds__(ASSIGN_THREADS) void assign_kernel(const float* __restrict__ X, int m, const float* __restrict__ cen, const float* __restrict__ cen2, int sub,
                                                                unsigned* __restrict__ keys, int* __restrict__ vals, uint8_t* __restrict__ valid,
                                                                int* __restrict__ counts /* [MAXLEAF+1] */, unsigned* __restrict__ rmax /* [MAXLEAF] */,
                                                                int role, unsigned* __restrict__ nstat /* [4] */,
                                                                const IrrRef* __restrict__ irr_ref /* or nullptr: no irregular-row lane */, int* __restrict__ irr_list /* [IRR_CAP] */) {
    // All sub-centres live in LDS (dynamic; up to 16 x 64 x 33 floats = 135 KB): every lane walks the sub-centres of ITS
    // cluster, which from global memory is a per-lane gather of 33 x sub words.  The odd pitch per cluster keeps lanes of
    // different clusters on different banks; lanes of one cluster read the same word (broadcast).
    extern __shared__ float c2s[];
    const int pitch = sub * 33 + 1;
    int* lc = (int*) (c2s + KCL * pitch);
    unsigned* lr = (unsigned*) (lc + MAXLEAF + 1);
    for (int e = threadIdx.x; e < KCL * sub * 33; e += blockDim.x) c2s[(e / (sub * 33)) * pitch + e % (sub * 33)] = cen2[e];
    for (int i = threadIdx.x; i <= MAXLEAF; i += blockDim.x) { lc[i] = 0; if (i < MAXLEAF) lr[i] = 0u; }
    // the three block sums of every centre, in double (the dropped-energy statistic below)
    __shared__ double cs_s[KCL][3];
    if (threadIdx.x < KCL * 3) {
        const int cc = threadIdx.x / 3, j = threadIdx.x % 3;
        double t = 0.0;
        for (int k = 0; k < 11; ++k) t += (double) cen[cc * 33 + 11 * j + k];
        cs_s[cc][j] = t;
    }
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float st_n2 = 0.f, st_drop = 0.f;
    bool st_ovf = false;
    if (i < m) {
        float v[33], r2;
        bool ok = row_finite(X + (size_t) i * 33, v);
        unsigned key = 0xffffffffu;
        // irregular rows (see km_consensus): finite, but off the consensus of the block sums -- listed for the exact side scan, kept out of
        // the clustering statistics and the operands (valid = 2: a train row for the exact scans, not a row of the filter)
        bool irr = false;
        double bx0 = 0.0, bx1 = 0.0, bx2 = 0.0;   // the row's own block sums (exact sums of floats, rounded once)
        if (ok) block_sums(v, bx0, bx1, bx2);
        if (ok && irr_ref && irr_ref->enabled) {
            irr = !sums_agree(bx0, bx1, bx2, irr_ref->s[0], irr_ref->s[1], irr_ref->s[2], irr_ref->tol);
            if (irr) {
                const unsigned pos = atomicAdd(&nstat[3], 1u);
                if (pos < (unsigned) IRR_CAP) irr_list[pos] = i;
            }
        }
        if (ok && !irr) {
            // a finite row whose squared distance to every centre overflows float stays a valid row (its exact distance to
            // a duplicate of itself is 0 in the reference); it lands in leaf 0 of cluster 0 with an infinite radius, and
            // the overflow sends the whole call down the exact dense path (match_impl, force_dense)
            // nearest centre (as nearest_centre) with the packing statistics of every centre on the way
            int c = 0;
            r2 = __uint_as_float(0x7f800000u);
            float own_drop = 0.f;
#pragma unroll 1
            for (int cc = 0; cc < KCL; ++cc) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < 33; ++k) {
                    const float t = v[k] - cen[cc * 33 + k];
                    d = d + t * t;
                }
                // Dropped energy: the packing kernel drops u_j = (sum over block j of t_k) / sqrt 11 with t_k = fl(v_k - c_k).  The block sums of the
                // ROUNDED differences differ from (row's block sum - centre's block sum) by at most 11 roundings of size 2^-24 |t_k|, |t_k| <= sqrt d:
                // the statistic is evaluated from the two block sums with that margin on every |s_j| (round 5: the sums of the t_k themselves, in
                // double, were 528 conversions and double additions per row -- a quarter of this kernel).
                const double mg = 6.6e-7 * (double) sqrtf(d);   // 11 x 2^-24 x 1.006
                const double s0 = fabs(bx0 - cs_s[cc][0]) + mg, s1 = fabs(bx1 - cs_s[cc][1]) + mg, s2 = fabs(bx2 - cs_s[cc][2]) + mg;
                const float dr = (float) (((s0 * s0 + s1 * s1) + s2 * s2) * (1.0001 / 11.0)) * 1.000001f + 1e-20f * d;
                if (d < r2) { r2 = d; c = cc; own_drop = dr; }
                if (role == 1) {
                    if (d < FLT_BIG) st_n2 = fmaxf(st_n2, d); else st_ovf = true;
                    st_drop = fmaxf(st_drop, dr);
                }
            }
            if (role == 0) {
                if (r2 < FLT_BIG) st_n2 = r2; else st_ovf = true;
                st_drop = own_drop;
            }
            float rl2;
            int j = nearest_sub(v, c2s + c * pitch, sub, rl2);
            if (!(r2 < FLT_BIG)) r2 = __uint_as_float(0x7f800000u);
            if (!(rl2 < FLT_BIG)) rl2 = __uint_as_float(0x7f800000u);
            int leaf = c * sub + j;
            key = ((unsigned) leaf << 22) | (__float_as_uint(r2) >> 9);
            atomicAdd(&lc[leaf], 1);
            atomicMax(&lr[leaf], __float_as_uint(rl2));
        }
        if (!ok || irr) atomicAdd(&lc[MAXLEAF], 1);
        keys[i] = key; vals[i] = i; valid[i] = ok ? (irr ? 2 : 1) : 0;
    }
    __syncthreads();
    for (int l = threadIdx.x; l <= MAXLEAF; l += blockDim.x) {
        if (lc[l]) atomicAdd(&counts[l], lc[l]);
        if (l < MAXLEAF && lr[l]) atomicMax(&rmax[l], lr[l]);
    }
    for (int o = 32; o > 0; o >>= 1) { st_n2 = fmaxf(st_n2, __shfl_xor(st_n2, o)); st_drop = fmaxf(st_drop, __shfl_xor(st_drop, o)); }
    const bool any_ovf = __ballot(st_ovf) != 0ull;
    if ((threadIdx.x & 63) == 0) {   // (a plain look first: same-address atomics from every wave would serialise in L2)
        if (st_n2 > 0.f && __float_as_uint(st_n2) > *(volatile unsigned*) &nstat[0]) atomicMax(&nstat[0], __float_as_uint(st_n2));
        if (st_drop > 0.f && __float_as_uint(st_drop) > *(volatile unsigned*) &nstat[1]) atomicMax(&nstat[1], __float_as_uint(st_drop));
        if (any_ovf) nstat[2] = 1u;
    }
}

// sorted position s -> padded position (leaves / clusters start at multiples of their pad units)
__global__ void place_kernel(const unsigned* __restrict__ keys_sorted, const int* __restrict__ vals_sorted, int n_valid,
                             const int* __restrict__ sorted_start /* [leaf] */, const int* __restrict__ pad_start /* [leaf] */,
                             int* __restrict__ perm) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_valid) return;
    int l = (int) (keys_sorted[s] >> 22);
    perm[pad_start[l] + (s - sorted_start[l])] = vals_sorted[s];
}


}  // namespace
