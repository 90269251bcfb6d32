// lgr_match_sweep.cuh -- 3c. the final MFMA pass as two kernels (round 4): match_sweep + match_tiles.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
//
// The final pass of the rotated format tests ~54 M 32 x 32 tiles of the 1M pair with two of the six K steps (coarse rejection,
// CoarseArgs in lgr_match_common.cuh) and finishes under one million of them.  In match_mfma<.., true> both jobs share one kernel,
// and the rare job dictates the shape of the common one: six A fragments, sixteen row minima and the whole six-step chain in
// registers (127 VGPRs, 49 spilled SGPRs, 4 waves per SIMD), a 16 KB column-minimum slab and its flush per item -- for a sweep that
// issues eight MFMAs per wave and stage and is bound by latency (21 % MFMA-busy in round 3).  Here the sweep is a kernel of its own:
//   match_sweep   the same persistent work list, the same LDS-DMA ring of coarse fragments, the same three tests per tile (shell gap,
//                 per-row coarse threshold, per-element re-test against the columns' own bounds) -- but only the two coarse A fragments
//                 and one accumulator pair live in registers, no minima are kept, and a tile that passes is APPENDED to a list (one
//                 atomic per wave and visit).  <= 80 VGPRs: three workgroups per CU instead of two.
//   match_tiles   one wave per listed tile: A and B fragments straight from memory (L2), the six-step chain, row minima folded by the
//                 halving butterfly of match_mfma and column minima folded across the lane halves, both into the tables with
//                 integer atomicMin on the float bits.  Consecutive list entries of a wave usually share the row tile (a wave of the
//                 sweep appends its visit's tiles together): A is reloaded only when it changes.
// The tables receive exactly the minima the fused kernel wrote (the same tiles, the same chain, min is order free), so nothing
// downstream can tell the two apart; lgr_match_options.split_sweep = 0 runs the fused kernel (tests compare both).  The list has a
// fixed capacity; a pass that overflows it (descriptors without structure: nearly every tile passes) is repeated by the fused kernel --
// the tables only ever take minima, so the partial work is harmless.
#pragma once
#include "lgr_match_common.cuh"
#include "lgr_match_mfma.cuh"

namespace {

constexpr int SW_RB = 1;    // row blocks per visit (2: a wave owns two row tiles that share the staged column tile -- half the LDS reads and
                            // barriers per MFMA; measured at 1M: 6.9 ms against 6.3 for one row block at six waves per SIMD: the sweep is bound by
                            // its vector instructions (PMC: 31 per tested tile, half of the SIMD time), not by LDS, so the occupancy wins)
constexpr int SW_OCC = SW_RB == 1 ? 6 : 4;   // waves per SIMD of match_sweep (80 / 128 VGPRs): three / two 512-thread workgroups per CU

// per-column thresholds of the per-element re-test, scaled like the accumulator and rounded UP to bf16 (a larger threshold only keeps
// more): U_col (1 + 1e-5) 1.0001 c_scale -- what match_mfma builds per item in LDS, here once per pass in memory
__global__ void ucol_pack_kernel(const float* __restrict__ u_colv, int mb_pad, float c_scale, unsigned short* __restrict__ out) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= mb_pad) return;
    const float t = ((fmaxf(u_colv[col], 0.f) * 1.00001f) * 1.0001f) * c_scale;
    const unsigned bits = t >= 0.f ? __float_as_uint(t) : 0x7f800000u;   // (NaN, never expected: keep everything)
    out[col] = (unsigned short) (min(bits + 0xffffu, 0x7f800000u) >> 16);
}

// which kernel takes the pass's work list (see launch_mfma): the list's per-XCD offsets go to one of them, zeros to the other
__global__ void pass_select_kernel(const unsigned long long* __restrict__ pass_stages, double half_of_all, const int* __restrict__ xcd_start /* [9] */,
                                   int* __restrict__ xs_sweep, int* __restrict__ xs_plain) {
    const int i = threadIdx.x;
    if (i >= 9) return;
    const bool plain = pass_stages && (double) pass_stages[0] > half_of_all;
    xs_sweep[i] = plain ? 0 : xcd_start[i];
    xs_plain[i] = plain ? xcd_start[i] : 0;
}

__global__ __launch_bounds__(NTHR, SW_OCC) void match_sweep(const f16x8* __restrict__ Ap, const f16x8* __restrict__ Bp, size_t bset_stride /* fragments */,
                                                            float c_scale, const int* __restrict__ blkcl, int ma_pad, int mb_pad, int rg_rows,
                                                            const unsigned* __restrict__ stage_mask, int n_cc, int item_rb, const int2* __restrict__ items,
                                                            const int* __restrict__ xcd_start, int* __restrict__ xcd_ctr, CoarseArgs ca,
                                                            const unsigned short* __restrict__ ucol16 /* [mb_pad] or nullptr */,
                                                            uint2* __restrict__ kept_out, unsigned long long* __restrict__ kept_count /* 64 bits: never wraps below the capacity */, unsigned kept_cap) {
    typedef f16x8 frag;
    constexpr int KS = OpFmt<FMT_F16R>::KS;
    constexpr int STAGE_FRAGS = STAGE_TILES * KS * 64;
    constexpr int CO_FRAGS = STAGE_TILES * 2 * 64, CO_NB = 6, CO_D = 4;   // ring slot (fragments), slots, stages in flight (as match_mfma)
    static_assert(CO_D == 4 && CO_NB == CO_D + 2, "the counted waits below are written for two pairs of stages in flight");
    static_assert(WAVES == 2 * STAGE_TILES, "one coarse piece per wave");
    __shared__ __attribute__((aligned(16))) unsigned char smem[CO_NB * CO_FRAGS * (int) sizeof(frag) + 16];
    int& item_s = *reinterpret_cast<int*>(smem + CO_NB * CO_FRAGS * (int) sizeof(frag));
    unsigned n_tested = 0u, n_rejected = 0u, n_skipped = 0u;   // wave-uniform tile counts
    const int xcd = blockIdx.x % 8;
    const int item0 = xcd_start[xcd], n_items = xcd_start[xcd + 1] - item0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int rg_blocks = rg_rows / BLOCK_ROWS;
    const int n_rb_total = ma_pad / BLOCK_ROWS;
    constexpr int IINF = 0x7f800000;
    const f32x16 nav = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (;;) {
        __syncthreads();
        // (an empty list leaves the counter alone: another kernel may own it.  No early exit on a full list: looking at the list's counter -- the
        //  hottest address of the launch -- once per item cost 1 ms of 5.6; an overflowing pass is repeated by the fused kernel either way)
        if (tid == 0) item_s = n_items > 0 ? atomicAdd(&xcd_ctr[xcd], 1) : 0x7fffffff;
        __syncthreads();
        const int it = item_s;
        if (it >= n_items) break;
        const int2 item = items[item0 + it];
        const int cc = item.x, rb0 = item.y;
        const int n_rb = min(item_rb, n_rb_total - rb0);
        const int col_tile0 = cc * (CHUNK_COLS / TILE);
        const int n_coltiles = min(CHUNK_COLS / TILE, mb_pad / TILE - col_tile0);
        const int n_stages = n_coltiles / STAGE_TILES;
        const unsigned full = n_stages >= 32 ? 0xffffffffu : ((1u << n_stages) - 1u);
        unsigned my_mask = lane < n_rb ? (stage_mask[(size_t) (rb0 + lane) * n_cc + cc] & full) : 0u;

        // A VISIT is a pair of row blocks of the item that are packed against the same centre (or one row block): wave w owns row tile w of
        // both, so ONE staged column tile -- one LDS read of its two coarse fragments -- feeds two MFMA chains.  Per coarse MFMA the kernel
        // reads one B fragment from LDS (1 KB, 8 cycles of the CU's LDS port for 32 cycles of one of its four matrix pipes): at one row tile
        // per wave the LDS port is as loaded as the matrix pipes together, and the eight waves of a workgroup all read the same bytes.  The
        // stages swept are the union of the two blocks' masks; a block whose mask lacks the stage sits it out (wave uniform).
        for (int rbi = 0; rbi < n_rb;) {
            const int rbA = rb0 + rbi;
            const bool pair = SW_RB == 2 && rbi + 1 < n_rb && blkcl[rbA + 1] == blkcl[rbA];
            const unsigned mk[2] = {(unsigned) __builtin_amdgcn_readlane(my_mask, rbi), pair ? (unsigned) __builtin_amdgcn_readlane(my_mask, min(rbi + 1, 63)) : 0u};
            rbi += pair ? 2 : 1;
            const unsigned mask = mk[0] | mk[1];   // uniform over the workgroup
            if (!mask) continue;
            const int set = blkcl[rbA];
            const frag* Bset = Bp + (size_t) set * bset_stride + (size_t) col_tile0 * KS * 64;
            int row_tile[2];
            frag a0[2], a1[2];
            int t_lane[2][2];       // [block][q >> 6]: coarse threshold of tile q = 4 stage + ct in lane q & 63 (column side + error term)
            float e_lane[2][2];     // error term of tile q, same lanes
            float trow[2];          // this lane's row: U_row (1 + 1e-5) 1.0001 c_scale
            unsigned long long skipm[2][2];   // shell test: bit q & 63 of skipm[block][q >> 6] = the wave leaves tile q out
            unsigned kept[2][STAGE_TILES];
            const unsigned skipped_before = n_skipped;
#pragma unroll
            for (int r = 0; r < SW_RB; ++r) {
                row_tile[r] = (rbA + r) * (BLOCK_ROWS / TILE) + wave;
                const int rt = mk[r] ? row_tile[r] : row_tile[0];   // (a block that sits the visit out: any valid address)
                a0[r] = Ap[((size_t) rt * KS + 0) * 64 + lane];
                a1[r] = Ap[((size_t) rt * KS + 1) * 64 + lane];
                t_lane[r][0] = t_lane[r][1] = IINF; e_lane[r][0] = e_lane[r][1] = 0.f; trow[r] = 0.f;
                skipm[r][0] = skipm[r][1] = 0ull;
#pragma unroll
                for (int ct = 0; ct < STAGE_TILES; ++ct) kept[r][ct] = 0u;
            }
            // (a raw barrier: every wave is past the previous visit's LDS reads; the A fragments just requested stay in flight)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
            unsigned to_issue = mask, to_do = mask;
            int issued = 0, done_ = 0;
            auto dma_coarse = [&]() {
                const int s_ = __builtin_ctz(to_issue);
                to_issue &= to_issue - 1u;
                const char* src = reinterpret_cast<const char*>(Bset + (size_t) s_ * STAGE_FRAGS + (size_t) ((wave >> 1) * KS + (wave & 1)) * 64);
                char* dst = reinterpret_cast<char*>(smem) + (issued % CO_NB) * (CO_FRAGS * (int) sizeof(frag)) + wave * 1024;
#ifndef LGR_EXP_SWEEP_NODMA   // (timing ablations, tools/exp_sweep_ablate.sh: wrong results)
                __builtin_amdgcn_global_load_lds((const void*) (src + lane * 16), (__attribute__((address_space(3))) void*) dst, 16, 0, 0);
#else
                (void) src; (void) dst;
#endif
                ++issued;
            };
            while (issued < CO_D && to_issue) dma_coarse();
            // thresholds (ordinary loads behind the first DMAs: one memory round trip per visit); see match_mfma for the derivation
#pragma unroll
            for (int r = 0; r < SW_RB; ++r) {
                if (!mk[r]) continue;   // (wave uniform)
                const int rb = rbA + r;
                const float ur = ca.u_rt[rb * (BLOCK_ROWS / TILE) + wave];
                if (ca.u_row) trow[r] = ((fmaxf(ca.u_row[(size_t) row_tile[r] * TILE + (lane & 31)], 0.f) * 1.00001f) * 1.0001f) * c_scale;
                const float x = ca.xmax[rb / rg_blocks];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    // (a visit tests ~11 of its 128 tile slots on the bench pair: the half of the slots whose stages the mask lacks is not priced)
                    if (!(h2 ? (mk[r] >> 16) : (mk[r] & 0xffffu))) continue;   // wave uniform
                    const int q = lane + 64 * h2;
                    const int gst = min(cc * STAGES_PER_CHUNK + (q >> 2), ca.n_stage_total - 1);
                    const int gct = min(col_tile0 + q, ca.n_ct_total - 1);
                    const float us = ca.u_stage ? __uint_as_float(ca.u_ct[gct]) : 0.f;
                    float xt = x, yt = ca.ymax[(size_t) set * ca.n_stage_total + gst];
                    if (ca.rt_shell) {
                        const float2 sa = ca.rt_shell[rb * (BLOCK_ROWS / TILE) + wave];
                        const float2 sb = ca.ct_shell[(size_t) set * ca.n_ct_total + gct];
                        const float gap = fmaxf(sb.x - sa.y, sa.x - sb.y) - 4e-6f * (sa.y + sb.y);
                        const float U = fmaxf(fmaxf(ur, us), 0.f);
                        const bool skip = gap > 0.f && gap * gap * (0.99999f * 0.99999f * 0.99999f) > U * 1.00001f + 1e-12f;
                        skipm[r][h2] = __ballot(skip);
                        xt = fminf(xt, sa.y); yt = fminf(yt, sb.y);
                    }
                    const float s = xt + yt;
                    const float et = (ca.quad * s) * s + ca.cross * (xt * yt) + ca.lin * s + ca.abs;
                    float t = fmaxf(fmaxf(ca.u_row ? 0.f : ur, us), 0.f) * 1.00001f + et;
                    t = (t * 1.0001f) * c_scale;
                    t_lane[r][h2] = t >= 0.f ? __float_as_int(t) : IINF;   // NaN (never expected): keep everything
                    e_lane[r][h2] = (et * 1.0001f) * c_scale;
                }
            }
            asm volatile("" : : "v"(t_lane[0][0]), "v"(t_lane[0][1]), "v"(e_lane[0][0]), "v"(e_lane[0][1]), "v"(trow[0]) : "memory");
            if (SW_RB == 2) asm volatile("" : : "v"(t_lane[1][0]), "v"(t_lane[1][1]), "v"(e_lane[1][0]), "v"(e_lane[1][1]), "v"(trow[1]) : "memory");
            // one stage of the sweep: four column tiles, two coarse steps each, on the TRANSPOSED product (lane = row of the wave's tile)
            auto compute = [&](int st, int buf) {
                const frag* cs = reinterpret_cast<const frag*>(smem) + buf * CO_FRAGS + lane;   // ring slot: [tile][2 steps][64]
                unsigned need[2];   // tiles of the stage each block still has to test
#pragma unroll
                for (int r = 0; r < SW_RB; ++r) {
                    const unsigned sk4 = (unsigned) ((st < 16 ? skipm[r][0] : skipm[r][1]) >> ((st * STAGE_TILES) & 63)) & 0xfu;
                    const bool on = (mk[r] >> st) & 1u;
                    if (on) n_skipped += (unsigned) __builtin_popcount(sk4);
                    need[r] = on ? (~sk4 & 0xfu) : 0u;
                }
                if (!(need[0] | (SW_RB == 2 ? need[1] : 0u))) return;
#pragma unroll
                for (int ct = 0; ct < STAGE_TILES; ++ct) {
                    if (!(((need[0] | (SW_RB == 2 ? need[1] : 0u)) >> ct) & 1u)) continue;   // (wave uniform)
                    const int q = st * STAGE_TILES + ct;
                    const frag c0 = cs[ct * 128], c1 = cs[ct * 128 + 64];
#pragma unroll
                    for (int r = 0; r < SW_RB; ++r) {
                        if (!((need[r] >> ct) & 1u)) continue;   // (wave uniform)
                        f32x16 acc = mfma_step(c0, a0[r], nav);
                        acc = mfma_step(c1, a1[r], acc);
                        // (a tree of v_min3; measured against the chain of eight dependent ones: no difference, the other waves of the SIMD fill either way)
                        auto ai = [&](int g) { return __float_as_int(acc[g]); };
                        const int m0 = min(min(ai(0), ai(1)), ai(2)), m1 = min(min(ai(3), ai(4)), ai(5)), m2 = min(min(ai(6), ai(7)), ai(8));
                        const int m3 = min(min(ai(9), ai(10)), ai(11)), m4 = min(min(ai(12), ai(13)), ai(14));
                        const int m = min(min(min(m0, m1), m2), min(min(m3, m4), ai(15)));
                        const int t_sel = st < 16 ? t_lane[r][0] : t_lane[r][1];
                        const float e_sel = st < 16 ? e_lane[r][0] : e_lane[r][1];
                        const int tc = __builtin_amdgcn_readlane(t_sel, q & 63);
                        const float te = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e_sel), q & 63));
                        const int thr = max(__float_as_int(trow[r] + te), tc);   // (both >= +0: integer order is float order; IINF = keep everything)
                        if (__ballot(m <= thr) == 0ull) continue;
                        // The tile goes on when some element is at or below max(U_row, U_col) (1 + 1e-5) + error term.  Row side: m <= trow + te
                        // for some lane (one compare: m is the lane's minimum).  Column side, only when no lane passes that: some element with
                        // acc - u_col <= te, i.e. min_c (acc_c - u_c) <= te -- sixteen subtractions and eight v_min3 (the thresholds carry a
                        // factor 1.0001 for exactly this kind of float evaluation; 97 vector instructions as sixteen max / add / compare / or).
                        bool kp = __ballot(m <= __float_as_int(trow[r] + te)) != 0ull;
                        if (!kp && ucol16) {   // acc[g] = column (g & 3) + 8 (g >> 2) + 4 half of the tile
                            const unsigned short* uc = ucol16 + ((size_t) (col_tile0 + q) << 5) + 4 * half;
                            float dmin = 3.4028234663852886e38f;
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                const uint2 w = *reinterpret_cast<const uint2*>(uc + 8 * gq);   // four bf16 thresholds
                                const float u0 = __uint_as_float(w.x << 16), u1 = __uint_as_float(w.x & 0xffff0000u);
                                const float u2 = __uint_as_float(w.y << 16), u3 = __uint_as_float(w.y & 0xffff0000u);
                                dmin = fminf(fminf(dmin, acc[4 * gq] - u0), acc[4 * gq + 1] - u1);
                                dmin = fminf(fminf(dmin, acc[4 * gq + 2] - u2), acc[4 * gq + 3] - u3);
                            }
                            kp = __ballot(dmin <= te) != 0ull;
                        } else if (!kp && !ucol16) kp = true;   // (no per-column bounds: the first test decides)
                        if (kp) kept[r][ct] |= 1u << st;
                    }
                }
            };
            while (to_do) {
                const int st = __builtin_ctz(to_do);
                to_do &= to_do - 1u;
                int st2 = -1;
                if (to_do) { st2 = __builtin_ctz(to_do); to_do &= to_do - 1u; }
                const int n_now = st2 >= 0 ? 2 : 1;
                const int newer = issued - done_ - n_now;   // DMAs issued after this pair's
                if (newer >= 2) asm volatile("s_waitcnt vmcnt(2)" : : : "memory");
                else if (newer == 1) asm volatile("s_waitcnt vmcnt(1)" : : : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
                __builtin_amdgcn_s_barrier();
                if (to_issue) dma_coarse();   // into the two slots the previous pair used
                if (to_issue) dma_coarse();
#ifndef LGR_EXP_SWEEP_NOCOMPUTE
                compute(st, done_ % CO_NB);
                if (st2 >= 0) compute(st2, (done_ + 1) % CO_NB);
#endif
                done_ += n_now;
            }
            unsigned nk = 0u;
#pragma unroll
            for (int r = 0; r < SW_RB; ++r)
#pragma unroll
                for (int ct = 0; ct < STAGE_TILES; ++ct) nk += (unsigned) __builtin_popcount(kept[r][ct]);
            {
                const unsigned swept = (unsigned) (STAGE_TILES * (__builtin_popcount(mk[0]) + (SW_RB == 2 ? __builtin_popcount(mk[1]) : 0))), sk_now = n_skipped - skipped_before;
                n_tested += swept - sk_now;
                n_rejected += swept - sk_now - nk;
            }
            if (nk) {   // append this wave's tiles: one atomic per wave and visit
                unsigned long long base = 0ull;
                if (lane == 0) base = atomicAdd(kept_count, (unsigned long long) nk);
                base = ((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (base >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) base);
                unsigned off = 0u;
#pragma unroll
                for (int r = 0; r < SW_RB; ++r)
#pragma unroll
                    for (int ct = 0; ct < STAGE_TILES; ++ct) {
                        const unsigned m = kept[r][ct];
                        if (lane < 32 && ((m >> lane) & 1u)) {
                            const unsigned long long pos = base + off + (unsigned) __builtin_popcount(m & ((1u << lane) - 1u));
                            if (pos < (unsigned long long) kept_cap) kept_out[pos] = make_uint2((unsigned) row_tile[r], (unsigned) (col_tile0 + lane * STAGE_TILES + ct));
                        }
                        off += (unsigned) __builtin_popcount(m);
                    }
            }
        }
    }
    if (ca.cnt && lane == 0) {
        if (n_tested) atomicAdd(&ca.cnt[0], (unsigned long long) n_tested);
        if (n_rejected) atomicAdd(&ca.cnt[1], (unsigned long long) n_rejected);
        if (n_skipped) atomicAdd(&ca.cnt[2], (unsigned long long) n_skipped);
    }
}

// Which (row block, leaf) pairs of the final pass does the list touch?  (Round 5.)  The sweep writes no table; match_tiles writes the minima of
// the listed tiles only -- 0.7 M of the 51 M tested on the bench pair.  Initialising the table entries of EVERY scheduled pair ahead of the pass
// (2 GB of +inf, 0.7 ms between the passes) and scanning them all again in the rerank is work for entries that stay +inf: only the pairs that
// hold a listed tile are marked (with their schedule byte), initialised behind the sweep and reported as computed.  A scheduled pair without
// a listed tile has had every tile rejected -- none of its elements can be a nearest neighbour or a tie of its row or column -- and "not
// computed" says the same as +inf.  When the plain kernel takes the pass (pass_select_kernel) or the list overflows (the fused kernel repeats
// the pass), every scheduled pair is marked.
__global__ void touched_kernel(const uint2* __restrict__ kept, const unsigned long long* __restrict__ kept_count, unsigned kept_cap, const int* __restrict__ xs_plain /* [9] */,
                               const uint8_t* __restrict__ sched, const int* __restrict__ tile_leaf, int n_leaves, size_t n_pairs, uint8_t* __restrict__ touched /* zeroed */) {
    const size_t i0 = (size_t) blockIdx.x * blockDim.x + threadIdx.x, step = (size_t) gridDim.x * blockDim.x;
    if (xs_plain[8] > 0 || *kept_count > (unsigned long long) kept_cap) {
        for (size_t i = i0; i < n_pairs; i += step) touched[i] = sched[i];
        return;
    }
    const size_t n = (size_t) *kept_count;
    for (size_t e = i0; e < n; e += step) {
        const uint2 ent = kept[e];
        const size_t idx = (size_t) (ent.x / (BLOCK_ROWS / TILE)) * n_leaves + tile_leaf[ent.y];
        touched[idx] = sched[idx];   // (every writer of a byte writes the same value)
    }
}

// the listed tiles in full: one wave per tile, TL_RUN consecutive list entries per wave visit
constexpr int TL_WAVES = 4, TL_RUN = 4;
template <bool COLDIR>
__global__ __launch_bounds__(64 * TL_WAVES) void match_tiles(const f16x8* __restrict__ Ap, const f16x8* __restrict__ Bp, size_t bset_stride, float out_scale,
                                                             const int* __restrict__ blkcl, int ma_pad, int mb_pad, int rg_rows, const int* __restrict__ tile_group,
                                                             int* __restrict__ rowmin, int* __restrict__ colmin, const uint2* __restrict__ kept,
                                                             const unsigned long long* __restrict__ kept_count, unsigned kept_cap) {
    typedef f16x8 frag;
    constexpr int KS = OpFmt<FMT_F16R>::KS;
    constexpr int IINF = 0x7f800000;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    if (*kept_count > (unsigned long long) kept_cap) return;   // overflow: the fused kernel repeats the whole pass
    const unsigned n = (unsigned) *kept_count;
    const unsigned n_waves = gridDim.x * TL_WAVES, w_id = blockIdx.x * TL_WAVES + (threadIdx.x >> 6);
    const int rg_blocks = rg_rows / BLOCK_ROWS;
    const f32x16 nav = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    frag a[KS];
    unsigned cur_rt = 0xffffffffu;
    for (unsigned e0 = w_id * TL_RUN; e0 < n; e0 += n_waves * TL_RUN) {
        for (unsigned e = e0; e < min(n, e0 + TL_RUN); ++e) {
            const uint2 ent = kept[e];
            const unsigned rt = ent.x, ct = ent.y;   // wave uniform
            const int rb = (int) (rt / (BLOCK_ROWS / TILE));
            if (rt != cur_rt) {
                cur_rt = rt;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) a[kk] = Ap[((size_t) rt * KS + kk) * 64 + lane];
            }
            const frag* bsrc = Bp + (size_t) blkcl[rb] * bset_stride + (size_t) ct * KS * 64 + lane;
            frag b[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) b[kk] = bsrc[kk * 64];
            f32x16 acc = mfma_step(a[0], b[0], nav);
#pragma unroll
            for (int kk = 1; kk < KS; ++kk) acc = mfma_step(a[kk], b[kk], acc);
            // column minima: the 16 registers of a lane are 16 rows of one column; the two lane halves hold the other 16 rows of the same column
            if (COLDIR) {
                int cm = min(__float_as_int(acc[0]), __float_as_int(acc[1]));
#pragma unroll
                for (int g = 2; g < 16; g += 2) cm = min(min(cm, __float_as_int(acc[g])), __float_as_int(acc[g + 1]));
                cm = min(cm, __shfl_xor(cm, 32));
                if (lane < 32 && cm != IINF) {
                    const int v = __float_as_int(__int_as_float(cm) * out_scale);
                    atomicMin(&colmin[(size_t) (rb / rg_blocks) * mb_pad + (size_t) ct * TILE + lane], v);
                }
            }
            // row minima: the halving butterfly of match_mfma's flush_rows (16 registers x 32 lanes -> one value per lane)
            {
                int rmin[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) rmin[g] = __float_as_int(acc[g]);
                int w8[8], w4[4], w2[2], w1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    auto sw = __builtin_amdgcn_permlane16_swap((unsigned) rmin[j], (unsigned) rmin[8 + j], false, false);
                    w8[j] = min((int) sw[0], (int) sw[1]);
                }
                {
                    const bool up = (lane & 8) != 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int keep = up ? w8[4 + j] : w8[j], send = up ? w8[j] : w8[4 + j];
                        w4[j] = min(keep, __builtin_amdgcn_update_dpp(0, send, 0x128 /* row_ror:8 */, 0xf, 0xf, true));
                    }
                }
                {
                    const bool up = (lane & 4) != 0;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        int keep = up ? w4[2 + j] : w4[j], send = up ? w4[j] : w4[2 + j];
                        const int t = __builtin_amdgcn_update_dpp(0, send, 0x141 /* row_half_mirror */, 0xf, 0xf, true);
                        w2[j] = min(keep, __builtin_amdgcn_update_dpp(0, t, 0x1b /* quad_perm [3,2,1,0] */, 0xf, 0xf, true));
                    }
                }
                {
                    const bool up = (lane & 2) != 0;
                    int keep = up ? w2[1] : w2[0], send = up ? w2[0] : w2[1];
                    w1 = min(keep, __builtin_amdgcn_update_dpp(0, send, 0x4e /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
                }
                w1 = min(w1, __builtin_amdgcn_update_dpp(0, w1, 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
                const int g = (lane >> 1) & 15;
                w1 = __float_as_int(__int_as_float(w1) * out_scale);
                if ((lane & 1) == 0 && w1 != IINF)
                    atomicMin(&rowmin[(size_t) tile_group[ct] * ma_pad + (size_t) rt * TILE + (g & 3) + 8 * (g >> 2) + 4 * half], w1);
            }
        }
    }
}

}  // namespace
