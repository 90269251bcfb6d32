// lgr_match_bounds.cuh -- 3b. exact stage skipping: ball and box lower bounds, pass-0 selection.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
#pragma once
#include "lgr_match_common.cuh"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// 3b. bound-based stage skipping (exact).  Train rows are sorted by leaf (two-level k-means), so a column group g is a
// ball (centre c_g, radius r_g).  For a 256-row block rb,  LB(rb, g) = max(0, min_i |a_i - c_g| - r_g)  is a lower bound
// of every distance in the tile rb x g.  Pass 1 computes the NEAR_T nearest groups of each row block and the NEAR_T
// nearest row blocks of each group; from its minima every row / column gets an upper bound U of its nearest-neighbour
// distance.  Pass 2 computes the remaining tiles with LB <= max U of the block's rows or of the group's columns.  A
// skipped tile holds only pairs with d >= LB > U >= (nearest distance), so it can contain neither the nearest
// neighbour nor a tie of any row or column.  All comparisons carry relative slack far above float rounding.
constexpr float LB_SHRINK = 0.99999f, LB_GROW = 1.00001f;

// LBsq[rb][g]; +inf when the row block has no valid row or the leaf is empty
__global__ __launch_bounds__(256) void lb_kernel(const float* __restrict__ Asorted, const int* __restrict__ permA, const float* __restrict__ cen2,
                                                 const unsigned* __restrict__ r2max, const int* __restrict__ leaf_count, int n_leaves,
                                                 float* __restrict__ LBsq) {
    constexpr int ROW_LD = 34;   // even row pitch: the packed loads below stay 8-byte aligned
    __shared__ __attribute__((aligned(16))) float rows[BLOCK_ROWS * ROW_LD];
    __shared__ int okr[BLOCK_ROWS];
    const int rb = blockIdx.x;
    for (int e = threadIdx.x; e < BLOCK_ROWS * 33; e += 256) rows[(e / 33) * ROW_LD + e % 33] = Asorted[(size_t) rb * BLOCK_ROWS * 33 + e];
    okr[threadIdx.x] = permA[rb * BLOCK_ROWS + threadIdx.x] >= 0;
    __syncthreads();
    typedef float v2f __attribute__((ext_vector_type(2)));
    // |a - c|^2 on packed fp32 math (v_pk_add_f32 / v_pk_fma_f32: two coordinates per instruction, even and odd coordinates
    // in separate accumulators); any summation order is fine here, the bound carries 1e-5 of slack.  Two leaves per thread
    // and row pass: every row read from LDS feeds two centres.
    for (int g0 = threadIdx.x; g0 < n_leaves; g0 += 2 * 256) {
        const int g1 = g0 + 256;
        const bool h1 = g1 < n_leaves;
        v2f c0[16], c1[16];
        float c032, c132;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            c0[k].x = cen2[(size_t) g0 * 33 + 2 * k]; c0[k].y = cen2[(size_t) g0 * 33 + 2 * k + 1];
            c1[k].x = h1 ? cen2[(size_t) g1 * 33 + 2 * k] : 0.f; c1[k].y = h1 ? cen2[(size_t) g1 * 33 + 2 * k + 1] : 0.f;
        }
        c032 = cen2[(size_t) g0 * 33 + 32]; c132 = h1 ? cen2[(size_t) g1 * 33 + 32] : 0.f;
        float dmin0 = __uint_as_float(0x7f800000u), dmin1 = dmin0;
        for (int i = 0; i < BLOCK_ROWS; ++i) {
            if (!okr[i]) continue;
            const float* __restrict__ r = rows + i * ROW_LD;
            v2f d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma clang fp contract(fast)
                v2f a = *reinterpret_cast<const v2f*>(r + 2 * k);
                v2f t0 = a - c0[k], t1 = a - c1[k];
                d0 = t0 * t0 + d0; d1 = t1 * t1 + d1;
            }
            const float r32 = r[32];
            float t0 = r32 - c032, t1 = r32 - c132;
            dmin0 = fminf(dmin0, __builtin_fmaf(t0, t0, d0.x + d0.y));
            dmin1 = fminf(dmin1, __builtin_fmaf(t1, t1, d1.x + d1.y));
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int g = w ? g1 : g0;
            if (g >= n_leaves) break;
            const float dmin = w ? dmin1 : dmin0;
            float out = __uint_as_float(0x7f800000u);
            if (leaf_count[g] > 0 && dmin < FLT_BIG) {
                float lb = sqrtf(dmin) * LB_SHRINK - sqrtf(__uint_as_float(r2max[g])) * LB_GROW;
                lb = lb > 0.f ? lb : 0.f;
                out = lb * lb * LB_SHRINK;
            }
            LBsq[(size_t) rb * n_leaves + g] = out;
        }
    }
}

// The same table from the packed f16 operands (f16 formats): the leaf centres are packed like train rows (one copy per column set,
// pack16_kernel with role 1), a workgroup multiplies the 8 row tiles of its block with the centre tiles of the block's set on the
// matrix cores and keeps the row minimum -- 6 (7) MFMA steps per 32 x 32 (row, centre) tile instead of 33 packed FMAs per pair.
// The filtered value is within eps(x, y) of |a - c|^2 (x = largest |a'| of the block, y = |c'|: the matcher's own proven bound),
// so  dmin - eps  is a valid lower bound of the smallest distance; eps is ~1e-5 (x + y)^2, far below what the bound is used for.
constexpr int LBM_THREADS = 512, LBM_CHUNK = 4;   // 8 waves = the 8 row tiles of a block; centre tiles go through LDS 4 at a time, double buffered
template <int KS>
__global__ __launch_bounds__(LBM_THREADS) void lb_mfma_kernel(const f16x8* __restrict__ Ap, const f16x8* __restrict__ Cp, size_t cset_stride /* fragments */,
                                                              float out_scale, const int* __restrict__ blkcl, const float* __restrict__ nA,
                                                              const float* __restrict__ nC /* [KCL][n_cpad] */, EpsExtra ex,
                                                              const unsigned* __restrict__ r2max, const int* __restrict__ leaf_count, int n_leaves, int n_cpad,
                                                              float* __restrict__ LBsq) {
    // (round 5: the chunks arrive by LDS-DMA, the next one while the current one is multiplied -- staged through registers, every 16-byte piece
    //  of a chunk waited for its own memory round trip, six in a row per chunk: 0.62 ms for 0.09 ms worth of MFMAs, on the matcher's critical chain)
    constexpr int CH_BLOCKS = LBM_CHUNK * KS;   // 1 KB blocks per chunk
    __shared__ __attribute__((aligned(16))) f16x8 cs[2][LBM_CHUNK * KS * 64];
    __shared__ int dmin_s[MAXLEAF + TILE];
    __shared__ float xw[LBM_THREADS / 64];
    const int rb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int p = blkcl[rb];
    constexpr int IINF = 0x7f800000;
    for (int g = tid; g < n_cpad; g += LBM_THREADS) dmin_s[g] = IINF;
    // largest |a'| of the block's valid rows (padding rows carry +inf)
    {
        float n2 = tid < BLOCK_ROWS ? nA[(size_t) rb * BLOCK_ROWS + tid] : 0.f;
        if (!(n2 < FLT_BIG)) n2 = 0.f;
        for (int o = 32; o > 0; o >>= 1) n2 = fmaxf(n2, __shfl_xor(n2, o));
        if (lane == 0) xw[wave] = n2;
    }
    f16x8 a[KS];
    const int row_tile = rb * (BLOCK_ROWS / TILE) + wave;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) a[kk] = Ap[((size_t) row_tile * KS + kk) * 64 + lane];
    const f16x8* Cset = Cp + (size_t) p * cset_stride;
    const int n_ct = n_cpad / TILE, n_ch = (n_ct + LBM_CHUNK - 1) / LBM_CHUNK;
    auto issue = [&](int ch) {   // wave w copies the 1 KB blocks w, w + 8, ... of chunk ch
        const int nblk = min(LBM_CHUNK, n_ct - ch * LBM_CHUNK) * KS;
        const char* src = reinterpret_cast<const char*>(Cset + (size_t) ch * LBM_CHUNK * KS * 64);
        char* dst = reinterpret_cast<char*>(&cs[ch & 1][0]);
        for (int blk = wave; blk < nblk; blk += LBM_THREADS / 64)
            __builtin_amdgcn_global_load_lds((const void*) (src + blk * 1024 + lane * 16), (__attribute__((address_space(3))) void*) (dst + blk * 1024), 16, 0, 0);
    };
    static_assert(CH_BLOCKS * 1024 == (int) sizeof(cs) / 2, "a chunk image is CH_BLOCKS blocks of 1 KB");
    issue(0);
    for (int ch = 0; ch < n_ch; ++ch) {
        const int ct0 = ch * LBM_CHUNK, nt = min(LBM_CHUNK, n_ct - ct0);
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // this wave's pieces of chunk ch (the only DMAs in flight) -- and, the first time, a / nA
        __syncthreads();   // every wave's pieces are in LDS; everybody is done with chunk ch - 1, whose buffer the next DMA takes (first time: dmin_s / xw)
        if (ch + 1 < n_ch) issue(ch + 1);
        const f16x8* cb = &cs[ch & 1][0];
#pragma unroll
        for (int t = 0; t < LBM_CHUNK; ++t) {
            if (t >= nt) break;   // (uniform)
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) acc = mfma_step(a[kk], cb[(t * KS + kk) * 64 + lane], acc);
            // minimum over the wave's 32 rows, on the bit patterns (a negative value -- a distance within eps of zero -- stays
            // negative under the signed-integer order, whichever negative it is; it ends as a bound of 0 below)
            int m = min(__float_as_int(acc[0]), __float_as_int(acc[1]));
#pragma unroll
            for (int g = 2; g < 16; g += 2) m = min(min(m, __float_as_int(acc[g])), __float_as_int(acc[g + 1]));
            auto sw = __builtin_amdgcn_permlane32_swap((unsigned) m, (unsigned) m, false, false);
            m = min(m, (int) (half ? sw[0] : sw[1]));
            if (half == 0) atomicMin(&dmin_s[(ct0 + t) * TILE + lane], m);
        }
    }
    __syncthreads();
    float x2 = xw[0];
#pragma unroll
    for (int w = 1; w < LBM_THREADS / 64; ++w) x2 = fmaxf(x2, xw[w]);
    const float x = sqrtf(x2) * 1.0000002f;
    for (int g = tid; g < n_leaves; g += LBM_THREADS) {
        float out = __uint_as_float(0x7f800000u);
        const float d = __int_as_float(dmin_s[g]) * out_scale;   // NaN never occurs (padding rows: +inf)
        if (leaf_count[g] > 0 && d < FLT_BIG) {
            const float y = sqrtf(nC[(size_t) p * n_cpad + g]) * 1.0000002f;
            const float dm = fmaxf(d - eps_xy(x, y, ex), 0.f);
            float lb = sqrtf(dm) * LB_SHRINK - sqrtf(__uint_as_float(r2max[g])) * LB_GROW;
            lb = lb > 0.f ? lb : 0.f;
            out = lb * lb * LB_SHRINK;
        }
        LBsq[(size_t) rb * n_leaves + g] = out;
    }
}
__global__ void centre_perm_kernel(int n_leaves, int n_cpad, int* __restrict__ perm) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_cpad) perm[g] = g < n_leaves ? g : -1;
}

// ---- bounding-box bounds (section 3b).  A leaf's ball is a poor container in 33 dimensions; its axis-aligned box in a
// fixed orthonormal basis (the principal axes of a sample of both sets) excludes a fifth more tiles at 1M x 1M.  For a row
// block with box [amin, amax] and a leaf with box [blo, bhi] every pair is at least sqrt(sum_k gap_k^2) apart,
// gap_k = max(0, amin_k - bhi_k, blo_k - amax_k).  Any orthonormal V gives a valid bound; float rounding of the rotation
// is taken off every gap (4.1e-6 * largest |x - mu|), the rest is covered like the ball bound's roundings (LB_SHRINK).
constexpr int COV_ROWS = 384, COV_THREADS = 640;   // rows per block; 33 sums + 561 products (a <= b) + the row count = 595 workers
__global__ __launch_bounds__(COV_THREADS) void cov_kernel(const float* __restrict__ smp, const int* __restrict__ smp_ok, int ns,
                                                          float* __restrict__ part /* [blocks][34 * 33 + 1]: sums, products (a <= b), row count */) {
    __shared__ float rows[COV_ROWS * 33];
    __shared__ int okr[COV_ROWS];
    const int r0 = blockIdx.x * COV_ROWS, nr = min(COV_ROWS, ns - r0);
    for (int i = threadIdx.x; i < nr * 33; i += COV_THREADS) rows[i] = smp[(size_t) r0 * 33 + i];
    for (int i = threadIdx.x; i < nr; i += COV_THREADS) okr[i] = smp_ok[r0 + i];
    __syncthreads();
    const int w = threadIdx.x;
    if (w > 594) return;
    float* out = part + (size_t) blockIdx.x * (34 * 33 + 1);
    float acc = 0.f;
    if (w < 33) {
        for (int r = 0; r < nr; ++r) if (okr[r]) acc += rows[r * 33 + w];
        out[w] = acc;
    } else if (w < 594) {
        int p = w - 33, a = 0;
        while (p >= 33 - a) { p -= 33 - a; ++a; }   // pair (a, b = a + p)
        const int b2 = a + p;
        for (int r = 0; r < nr; ++r) if (okr[r]) acc = __builtin_fmaf(rows[r * 33 + a], rows[r * 33 + b2], acc);
        out[33 + a * 33 + b2] = acc;
    } else {
        for (int r = 0; r < nr; ++r) acc += okr[r] ? 1.f : 0.f;
        out[34 * 33] = acc;
    }
}
// block partials -> totals, summed in block order (deterministic basis, hence a deterministic tile schedule)
__global__ void cov_reduce(const float* __restrict__ part, int n_blocks, float* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > 34 * 33) return;
    float acc = 0.f;
    for (int b = 0; b < n_blocks; ++b) acc += part[(size_t) b * (34 * 33 + 1) + e];
    out[e] = acc;
}
// boxes of row segments in the basis (rows of V, y = V (x - mu)); segments: fixed 256-row blocks (starts == nullptr) or
// [starts[s], starts[s + 1]).  box[s][0..32] = min, [33..65] = max (transposed: box[c][s]); rmax2: largest |x - mu|^2 seen.
// X: the rows in their ORIGINAL order, read through perm (position r holds row perm[r]; -1 = padding)
__global__ __launch_bounds__(256, 4) void box_kernel(const float* __restrict__ X, const int* __restrict__ perm, const int* __restrict__ starts, int n_seg,
                                                  const float* __restrict__ V /* [33][33] */, const float* __restrict__ mu, int transposed,
                                                  float* __restrict__ box, unsigned* __restrict__ rmax2) {
    // basis rows padded to 36 floats: a row is nine 16-byte LDS reads (every lane reads the same address: broadcasts); the loops below
    // are fully unrolled so that x / mn / mx stay in registers (a variable index puts them in scratch: 1.4 + 0.6 ms per step at 1M,
    // on the critical path of the pass-0 bounds)
    __shared__ __attribute__((aligned(16))) float Vs[33 * 36];
    __shared__ float mus[33];
    __shared__ float red[4][66];
    for (int i = threadIdx.x; i < 33 * 36; i += 256) { const int k = i / 36, j = i % 36; Vs[i] = j < 33 ? V[k * 33 + j] : 0.f; }
    if (threadIdx.x < 33) mus[threadIdx.x] = mu[threadIdx.x];
    __syncthreads();
    const int seg = blockIdx.x;
    const int b = starts ? starts[seg] : seg * BLOCK_ROWS, e = starts ? starts[seg + 1] : (seg + 1) * BLOCK_ROWS;
    const float inf = __uint_as_float(0x7f800000u);
    float mn[33], mx[33], r2 = 0.f;
#pragma unroll
    for (int k = 0; k < 33; ++k) { mn[k] = inf; mx[k] = -inf; }
    for (int r = b + threadIdx.x; r < e; r += 256) {
        const int o = perm[r];
        if (o < 0) continue;
        asm volatile("" ::: "memory");   // the basis is loop invariant: without this its 300 LDS reads are hoisted out of the row loop into 1200 registers
        float x[33];
        float n2 = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) { x[k] = X[(size_t) o * 33 + k] - mus[k]; n2 = __builtin_fmaf(x[k], x[k], n2); }
        r2 = fmaxf(r2, n2);
#pragma unroll
        for (int k = 0; k < 33; ++k) {
            const float4* vr = reinterpret_cast<const float4*>(&Vs[k * 36]);
            float y = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < 8; ++j4) {
                const float4 v = vr[j4];
                y = __builtin_fmaf(v.x, x[4 * j4], y); y = __builtin_fmaf(v.y, x[4 * j4 + 1], y);
                y = __builtin_fmaf(v.z, x[4 * j4 + 2], y); y = __builtin_fmaf(v.w, x[4 * j4 + 3], y);
            }
            y = __builtin_fmaf(Vs[k * 36 + 32], x[32], y);
            mn[k] = fminf(mn[k], y); mx[k] = fmaxf(mx[k], y);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 33; ++k) {
        float a = mn[k], c = mx[k];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); c = fmaxf(c, __shfl_xor(c, o)); }
        if (lane == 0) { red[wave][k] = a; red[wave][33 + k] = c; }
    }
    for (int o = 32; o > 0; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o));
    if (lane == 0 && r2 > 0.f && __float_as_uint(r2) > *(volatile unsigned*) rmax2) atomicMax(rmax2, __float_as_uint(r2));
    __syncthreads();
    if (threadIdx.x < 66) {
        const int c = threadIdx.x;
        float v = red[0][c];
        for (int w = 1; w < 4; ++w) v = c < 33 ? fminf(v, red[w][c]) : fmaxf(v, red[w][c]);
        box[transposed ? (size_t) c * n_seg + seg : (size_t) seg * 66 + c] = v;
    }
}
// LBsq[rb][leaf] = max(ball bound, box bound)
__global__ __launch_bounds__(256) void box_lb_kernel(const float* __restrict__ boxA /* [n_rb][66] */, const float* __restrict__ boxBt /* [66][n_leaves] */,
                                                     int n_leaves, const unsigned* __restrict__ rmax2, float* __restrict__ LBsq,
                                                     uint2* __restrict__ part /* [n_rb]: zero, finite final bounds of this row block (what lb_stats_kernel counts), or nullptr */) {
    __shared__ float a[66];
    __shared__ unsigned cnt_s[2];
    unsigned n_zero = 0u, n_fin = 0u;
    if (threadIdx.x < 2) cnt_s[threadIdx.x] = 0u;
    const int rb = blockIdx.x;
    if (threadIdx.x < 66) a[threadIdx.x] = boxA[(size_t) rb * 66 + threadIdx.x];
    __syncthreads();
    const float delta = 4.1e-6f * sqrtf(__uint_as_float(*rmax2)) * 1.01f;
#pragma unroll 1   // (with the counters carried across it hipcc unrolled this loop as well: 512 VGPRs, 945 spilled, 0.66 ms instead of 0.07)
    for (int g = threadIdx.x; g < n_leaves; g += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 33; ++k) {
            float gap = fmaxf(a[k] - boxBt[(size_t) (33 + k) * n_leaves + g], boxBt[(size_t) k * n_leaves + g] - a[33 + k]) - delta;
            if (gap > 0.f) s = __builtin_fmaf(gap, gap, s);
        }
        const size_t idx = (size_t) rb * n_leaves + g;
        const float old = LBsq[idx];
        const float lb = s * (LB_SHRINK * LB_SHRINK * LB_SHRINK);
        const bool raise = old < FLT_BIG && lb > old && lb < FLT_BIG;
        if (raise) LBsq[idx] = lb;   // (the box alone: 19.4 % of the tiles, the ball alone 22.2 %, both 17.4 %)
        const float fin = raise ? lb : old;
        n_fin += fin < FLT_BIG ? 1u : 0u;
        n_zero += (fin < FLT_BIG && fin <= 0.f) ? 1u : 0u;
    }
    if (part) {
        // the statistics auto_dense decides on, counted where the final bounds are written (as a pass of its own over the table: 0.1 ms on the
        // critical chain) -- one pair of counts per row block, summed by near_kernel (two global counters instead: 31 000 atomics on two
        // addresses made this kernel 0.77 ms instead of 0.07)
        for (int o = 32; o > 0; o >>= 1) { n_zero += __shfl_xor(n_zero, o); n_fin += __shfl_xor(n_fin, o); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&cnt_s[0], n_zero); atomicAdd(&cnt_s[1], n_fin); }
        __syncthreads();
        if (threadIdx.x == 0) part[rb] = make_uint2(cnt_s[0], cnt_s[1]);
    }
}

// Can the bounds separate anything at all?  A (row block, leaf) pair whose lower bound is ZERO -- the two balls overlap and so do the boxes --
// can never be excluded, whatever upper bounds the passes find.  lb_stats_kernel counts those pairs among the finite entries; when they are
// (nearly) all of them (descriptors without cluster structure: bench.py's `structureless` extreme) pass 0's selection of the nearest leaves,
// the second pass and its bookkeeping only add to a dense computation.  lb_widen() then makes near_kernel
// mark EVERYTHING for pass 0 (near_t = the whole vector): pass 0 becomes the dense pass, the final pass finds nothing left to schedule, and
// the rest of the machinery runs over empty work lists -- decided on the device, no host round trip (a read-back at this point stalls the
// launch queue for longer than the decision is worth).  Results never depend on it (a wider pass 0 is still an exact schedule).
__global__ void lb_stats_kernel(const float* __restrict__ LBsq, size_t n, unsigned long long* __restrict__ out2 /* zero, finite */) {
    unsigned long long z = 0, f = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const float v = LBsq[i];
        const bool fin = v < FLT_BIG;
        f += fin ? 1 : 0;
        z += (fin && v <= 0.f) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) { z += __shfl_xor(z, o); f += __shfl_xor(f, o); }
    if ((threadIdx.x & 63) == 0) { if (z) atomicAdd(&out2[0], z); if (f) atomicAdd(&out2[1], f); }
}
// (the decision itself: every workgroup of near_kernel evaluates it from the same counts -- stats2 from lb_stats_kernel, or the per-row-block
//  pairs of box_lb_kernel, which workgroup 0 also sums into stats2 for the host's report.  Every thread of the workgroup must call it.)
__device__ __forceinline__ bool lb_widen(unsigned long long* __restrict__ stats2, const uint2* __restrict__ part, int n_part, float frac) {
    __shared__ unsigned long long sum_s[2];
    unsigned long long z, f;
    if (part) {
        if (threadIdx.x < 2) sum_s[threadIdx.x] = 0ull;
        __syncthreads();
        unsigned long long pz = 0ull, pf = 0ull;
        for (int i = threadIdx.x; i < n_part; i += blockDim.x) { const uint2 p = part[i]; pz += p.x; pf += p.y; }
        for (int o = 32; o > 0; o >>= 1) { pz += __shfl_xor(pz, o); pf += __shfl_xor(pf, o); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sum_s[0], pz); atomicAdd(&sum_s[1], pf); }
        __syncthreads();
        z = sum_s[0]; f = sum_s[1];
        if (blockIdx.x == 0 && threadIdx.x == 0) { stats2[0] = z; stats2[1] = f; }
        __syncthreads();
    } else { z = stats2[0]; f = stats2[1]; }
    return frac > 0.f && f > 0ull && (double) z >= (double) frac * (double) f;
}

// the near_t smallest finite entries of a strided vector -> need1 = 1; ties go to the lowest index.  One 256-thread block
// per vector: the vector is read once into LDS (dynamic: len words), a bitwise radix select finds the near_t-th smallest
// key (entries are >= 0, so the float bits order like the values), then everything below it and the first ties are marked.
constexpr int NEAR_THREADS = 256;
constexpr int NEAR_LDS_MAX = 36 * 1024;   // entries that fit the dynamic LDS slab (144 KB); longer vectors are re-read from global
template <bool IN_LDS>
__global__ __launch_bounds__(NEAR_THREADS) void near_kernel(int near_t, const float* __restrict__ LBsq, int n_vec, int len, size_t vec_stride, size_t elem_stride,
                                                            uint8_t* __restrict__ need1, size_t need_vec_stride, size_t need_elem_stride,
                                                            unsigned long long* __restrict__ stats2, const uint2* __restrict__ lb_part, int n_part, float widen_frac) {
    extern __shared__ unsigned keys[];
    __shared__ int cnt_s, base_s;
    __shared__ int wave_cnt[NEAR_THREADS / 64];
    const int vec = blockIdx.x, tid = threadIdx.x;
    if (vec >= n_vec) return;
    if (lb_widen(stats2, lb_part, n_part, widen_frac)) {   // (uniform) the bounds separate nothing: every finite entry is "near"
        for (int e = tid; e < len; e += NEAR_THREADS)
            if (LBsq[vec * vec_stride + e * elem_stride] < FLT_BIG) need1[vec * need_vec_stride + e * need_elem_stride] = 1;
        return;
    }
    constexpr unsigned INF = 0x7f800000u;
    auto load = [&](int e) {
        unsigned k = __float_as_uint(LBsq[vec * vec_stride + e * elem_stride]);
        return k > INF ? INF : k;                   // negative values / NaN cannot occur; anything odd counts as "not finite"
    };
    auto key_of = [&](int e) { return IN_LDS ? keys[e] : load(e); };
    int n_fin = 0;
    for (int e = tid; e < len; e += NEAR_THREADS) {
        unsigned k = load(e);
        if (IN_LDS) keys[e] = k;
        n_fin += k < INF ? 1 : 0;
    }
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) n_fin += __shfl_xor(n_fin, o);
    if ((tid & 63) == 0) atomicAdd(&cnt_s, n_fin);
    __syncthreads();
    int k = min(near_t, cnt_s);                     // how many to mark
    __syncthreads();
    if (k == 0) return;
    // Round 5: every entry with a ZERO lower bound as well (the balls and the boxes overlap: no upper bound will ever exclude the pair, and among
    // zeros "the near_t nearest" is an accident of the index order).  Blobs of near-duplicate descriptors give a row block dozens of such leaves;
    // left to the final pass they are swept, kept and finished tile by tile -- eight MFMA steps on the slow path for what pass 0 does in six.
    // (measured at 900 k points over six scene seeds: the worst, 571 -- 85 zero bounds per row block on average -- 38.1 -> 26.1 ms for the match
    // stage, 569: 24.2 -> 19.7, 567: 17.5 -> 16.6, the bench pair's scene 566: 14.4 -> 14.7.)
    for (int e = tid; e < len; e += NEAR_THREADS)
        if (key_of(e) == 0u) need1[vec * need_vec_stride + e * need_elem_stride] = 1;
    // k-th smallest key (1-based) by radix select from the top bit; `k` becomes its rank among the equal keys
    unsigned prefix = 0u;
    for (int bit = 30; bit >= 0; --bit) {           // bit 31 is 0 everywhere
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        const unsigned hi_mask = ~((2u << bit) - 1u);   // bits above `bit`
        int c0 = 0;
        for (int e = tid; e < len; e += NEAR_THREADS) {
            unsigned key = key_of(e);
            c0 += ((key & hi_mask) == prefix && !((key >> bit) & 1u)) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) c0 += __shfl_xor(c0, o);
        if ((tid & 63) == 0 && c0) atomicAdd(&cnt_s, c0);
        __syncthreads();
        const int zeros = cnt_s;
        __syncthreads();
        if (k > zeros) { k -= zeros; prefix |= 1u << bit; }
    }
    // mark the keys below the k-th ...
    for (int e = tid; e < len; e += NEAR_THREADS)
        if (key_of(e) < prefix) need1[vec * need_vec_stride + e * need_elem_stride] = 1;
    // ... and the first k entries equal to it, in index order (rows of NEAR_THREADS consecutive entries)
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int e0 = 0; e0 < len; e0 += NEAR_THREADS) {
        const int e = e0 + tid;
        const bool tie = e < len && key_of(e) == prefix;
        const unsigned long long bal = __ballot(tie);
        if ((tid & 63) == 0) wave_cnt[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = base_s;
        for (int w = 0; w < (tid >> 6); ++w) before += wave_cnt[w];
        before += __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if (tie && before < k) need1[vec * need_vec_stride + e * need_elem_stride] = 1;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < NEAR_THREADS / 64; ++w) t += wave_cnt[w]; base_s += t; }
        __syncthreads();
        if (base_s >= k) break;   // uniform
    }
}


}  // namespace
