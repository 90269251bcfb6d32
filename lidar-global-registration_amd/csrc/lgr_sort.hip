// lgr_sort.hip -- stable LSD radix sort of (key, 32-bit value) pairs for the 0.5-2 M-element sorts of the path (grid builds, voxel
// keys, matcher placement, rerank items): 8-bit digits, three short kernels per digit (per-workgroup digit counts, one-workgroup
// scan of the count table, stable scatter).  rocPRIM serves these sizes with a merge sort -- 21 launches and ten passes over the
// data per sort, 16 sorts per 1M-point pair; here a sort costs one pass pair per USED digit, and callers name the bit ranges
// that can be non-zero (a cell id needs its 20-28 bits, a voxel key three short fields of its 63).
//
// Stability: an element's place inside its digit is (workgroup, round, wave, lane) order = input order; the scatter ranks the
// 64 lanes of a round with match-any ballots (the set of lanes holding the same digit) and the rounds / waves with a table in LDS.
#include <algorithm>

#include "lgr_internal.h"

namespace {

constexpr int RS_THREADS = 256, RS_ITEMS = 8, RS_TILE = RS_THREADS * RS_ITEMS, RS_SLOTS = RS_ITEMS * (RS_THREADS / 64);

template <class K>
__global__ __launch_bounds__(RS_THREADS) void rs_count(const K* __restrict__ keys, size_t n, int shift, unsigned mask, int nblk, unsigned* __restrict__ counts /* [256][nblk] */) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t) blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const size_t i = base + (size_t) r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned) (keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    counts[(size_t) threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit: counts[d][b] -> number of elements of digit d in earlier workgroups; the digit's total goes to tot[d]
__global__ __launch_bounds__(256) void rs_scan(unsigned* __restrict__ counts, int nblk, unsigned* __restrict__ tot) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned carry_s;
    unsigned* row = counts + (size_t) blockIdx.x * nblk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0u;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        const int b = b0 + tid;
        const unsigned c = b < nblk ? row[b] : 0u;
        unsigned incl = c;
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (b < nblk) row[b] = before + incl - c;
        __syncthreads();
        if (tid == 255) carry_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) tot[blockIdx.x] = carry_s;
}

// Stable scatter of one 2048-element tile.  The tile is first put into digit order in LDS, then written out in that order:
// consecutive lanes then write consecutive addresses inside a digit's run (a wave store touches ~8 runs instead of 64 lines).
template <class K>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter(const K* __restrict__ kin, const int* __restrict__ vin, K* __restrict__ kout, int* __restrict__ vout,
                                                         size_t n, int shift, unsigned mask, int nblk, const unsigned* __restrict__ offs /* [256][nblk] */,
                                                         const unsigned* __restrict__ tot /* [256] */) {
    __shared__ unsigned short hist[RS_SLOTS][256];   // [round * waves + wave][digit]: elements of that digit in that wave round
    __shared__ unsigned goff[256];                   // first output position of the tile's elements of digit d
    __shared__ unsigned short tstart[257];           // first tile position of digit d
    __shared__ unsigned wsum[4];
    __shared__ K sk[RS_TILE];
    __shared__ int sv[RS_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < RS_SLOTS * 256; e += RS_THREADS) (&hist[0][0])[e] = 0;
    {   // digit bases: exclusive scan of the digit totals
        const unsigned c = tot[tid];
        unsigned incl = c;
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned before = 0u;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        goff[tid] = before + incl - c + offs[(size_t) tid * nblk + blockIdx.x];
    }
    __syncthreads();
    const size_t base = (size_t) blockIdx.x * RS_TILE;
    const int n_tile = (int) min((size_t) RS_TILE, n - base);
    const unsigned long long lt = (1ull << lane) - 1ull;
    K key[RS_ITEMS];
    int val[RS_ITEMS];
    unsigned short wrank[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int e = r * RS_THREADS + tid;
        const bool valid = e < n_tile;
        key[r] = valid ? kin[base + e] : (K) 0;
        val[r] = valid ? vin[base + e] : 0;
        const unsigned d = (unsigned) (key[r] >> shift) & mask;
        unsigned long long peers = __ballot(valid);   // lanes of this wave round that hold the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        wrank[r] = (unsigned short) __popcll(peers & lt);
        if (valid && (peers & lt) == 0ull) hist[r * (RS_THREADS / 64) + wave][d] = (unsigned short) __popcll(peers);
    }
    __syncthreads();
    {
        unsigned run = 0u;
#pragma unroll 4
        for (int s = 0; s < RS_SLOTS; ++s) { const unsigned c = hist[s][tid]; hist[s][tid] = (unsigned short) run; run += c; }
        // tile position of the first element of every digit: exclusive scan of the tile's digit totals
        unsigned incl = run;
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        __syncthreads();   // (wsum is read above by every wave before it is rewritten here)
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned before = 0u;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        tstart[tid] = (unsigned short) (before + incl - run);
        if (tid == 255) tstart[256] = (unsigned short) (before + incl);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int e = r * RS_THREADS + tid;
        if (e >= n_tile) continue;
        const unsigned d = (unsigned) (key[r] >> shift) & mask;
        const int tp = tstart[d] + hist[r * (RS_THREADS / 64) + wave][d] + wrank[r];
        sk[tp] = key[r];
        sv[tp] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int e = r * RS_THREADS + tid;
        if (e >= n_tile) continue;
        const K k = sk[e];
        const unsigned d = (unsigned) (k >> shift) & mask;
        const size_t pos = (size_t) goff[d] + (unsigned) (e - tstart[d]);
        kout[pos] = k;
        vout[pos] = sv[e];
    }
}

__global__ void rs_copy(const unsigned* __restrict__ a, unsigned* __restrict__ b, size_t nwords) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) b[i] = a[i];
}

template <class K>
int sort_impl(lgr_ctx* ctx, const K* kin, K* kout, const int* vin, int* vout, size_t n, const int* shifts, const int* widths, int np) {
    if (n == 0) return LGR_OK;
    LGR_CHECK(ctx, kin && kout && vin && vout && (const void*) kin != (const void*) kout && vin != vout && n < ((size_t) 1 << 31), LGR_ERR_INVALID_ARG);
    // 8-bit digits over the named bit ranges
    int dsh[16], dw[16], nd = 0;
    for (int p = 0; p < np; ++p)
        for (int b = 0; b < widths[p]; b += 8) {
            LGR_CHECK(ctx, nd < 16, LGR_ERR_INVALID_ARG);
            dsh[nd] = shifts[p] + b; dw[nd] = std::min(8, widths[p] - b); ++nd;
        }
    if (nd == 0) {
        rs_copy<<<cdiv((long long) (n * sizeof(K) / 4), 256), 256, 0, ctx->stream>>>((const unsigned*) kin, (unsigned*) kout, n * sizeof(K) / 4);
        rs_copy<<<cdiv((long long) n, 256), 256, 0, ctx->stream>>>((const unsigned*) vin, (unsigned*) vout, n);
        LGR_HIP(ctx, hipGetLastError());
        return LGR_OK;
    }
    const int nblk = (int) ((n + RS_TILE - 1) / RS_TILE);
    char* tmp;
    const size_t kb = (n * sizeof(K) + 255) & ~(size_t) 255, vb = (n * 4 + 255) & ~(size_t) 255;
    LGR_TRY(lgr_ws_t(ctx, WS_SORT_TMP, kb + vb + (size_t) nblk * 256 * 4 + 1024, &tmp));
    K* tk = (K*) tmp;
    int* tv = (int*) (tmp + kb);
    unsigned* tot = (unsigned*) (tmp + kb + vb);
    unsigned* counts = tot + 256;
    const K* sk = kin;
    const int* sv = vin;
    for (int p = 0; p < nd; ++p) {
        const bool to_out = ((nd - 1 - p) & 1) == 0;   // the last digit lands in the output, the ones before alternate
        K* dk = to_out ? kout : tk;
        int* dv = to_out ? vout : tv;
        const unsigned mask = (1u << dw[p]) - 1u;
        rs_count<K><<<nblk, RS_THREADS, 0, ctx->stream>>>(sk, n, dsh[p], mask, nblk, counts);
        rs_scan<<<256, 256, 0, ctx->stream>>>(counts, nblk, tot);
        rs_scatter<K><<<nblk, RS_THREADS, 0, ctx->stream>>>(sk, sv, dk, dv, n, dsh[p], mask, nblk, counts, tot);
        sk = dk; sv = dv;
    }
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}

}  // namespace

int lgr_sort_pairs_u32(lgr_ctx* ctx, const unsigned* kin, unsigned* kout, const int* vin, int* vout, size_t n, int begin_bit, int end_bit) {
    const int sh = begin_bit, w = std::max(0, end_bit - begin_bit);
    return sort_impl<unsigned>(ctx, kin, kout, vin, vout, n, &sh, &w, 1);
}
int lgr_sort_pairs_u64(lgr_ctx* ctx, const unsigned long long* kin, unsigned long long* kout, const int* vin, int* vout, size_t n,
                       const int* shifts, const int* widths, int n_ranges) {
    return sort_impl<unsigned long long>(ctx, kin, kout, vin, vout, n, shifts, widths, n_ranges);
}

extern "C" int lgr_sort_pairs_u32_dev(lgr_ctx* ctx, const uint32_t* kin, uint32_t* kout, const int32_t* vin, int32_t* vout, size_t n, int begin_bit, int end_bit) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, begin_bit >= 0 && end_bit <= 32 && begin_bit <= end_bit, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    return lgr_sort_pairs_u32(ctx, kin, kout, vin, vout, n, begin_bit, end_bit);
}
extern "C" int lgr_sort_pairs_u64_dev(lgr_ctx* ctx, const uint64_t* kin, uint64_t* kout, const int32_t* vin, int32_t* vout, size_t n,
                                      const int* shifts, const int* widths, int n_ranges) {
    lgr_turn turn__(ctx);   // contexts of one device take turns (lgr_internal.h)
    if (!ctx) return LGR_ERR_INVALID_ARG;
    LGR_CHECK(ctx, n_ranges >= 0 && n_ranges <= 8 && (n_ranges == 0 || (shifts && widths)), LGR_ERR_INVALID_ARG);
    for (int i = 0; i < n_ranges; ++i) LGR_CHECK(ctx, shifts[i] >= 0 && widths[i] >= 0 && shifts[i] + widths[i] <= 64, LGR_ERR_INVALID_ARG);
    LGR_HIP(ctx, hipSetDevice(ctx->device));
    return lgr_sort_pairs_u64(ctx, (const unsigned long long*) kin, (unsigned long long*) kout, vin, vout, n, shifts, widths, n_ranges);
}
