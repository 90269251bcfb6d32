// lgr_match_mfma.cuh -- 3. the MFMA filter kernel and its work list.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
#pragma once
#include "lgr_match_common.cuh"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// 3. MFMA filter kernel.  One work item = one row group (item_rb row blocks of 256) x one 4096-column chunk (32 stages of 128).
//    Wave w of row block rb owns row tile rb*8 + w; all waves share the column stage staged in LDS.
//    The column operand set is chosen per row block: Bp + blkcl[rb] * bset_stride.
//    stage_mask[rb][chunk] (optional) selects the stages to compute: bound-based skipping, section 3b.
//    Row minima are flushed per column group (tile_group[tile], a leaf of the train side) with an integer atomicMin
//    on the float bits; column minima per row group likewise (no-return atomics; one owner per entry).  Both tables must
//    be initialised to +inf bits, so several masked passes accumulate into the same tables.
//    CO = true (rotated format, launches that have upper bounds: CoarseArgs): per row block a coarse sweep over the active
//    stages (two of the six K steps per tile, tested against the stage threshold, survivors recorded in bit masks; coarse
//    fragments only, staged through a ring of six LDS slots, two pairs of stages ahead, two stages per barrier) and then the recorded tiles in full,
//    outside the per-stage barriers, on B fragments read straight from memory.
#ifdef EXP_PROF
__device__ unsigned long long g_prof[16];
#define PROF_T(var) unsigned long long var = wall_clock64()
#define PROF_ADD(slot, a, b) do { if (tid == 0) atomicAdd(&g_prof[slot], (b) - (a)); } while (0)
#define PROF_CNT(slot) do { if (tid == 0) atomicAdd(&g_prof[slot], 1ull); } while (0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, a, b)
#define PROF_CNT(slot)
#endif
__device__ __forceinline__ f32x16 mfma_step(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_step(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <bool COLDIR, int FMT, bool CO /* coarse rejection (rotated format, CoarseArgs set) */>
#ifndef LGR_MM_OCC_CO
#define LGR_MM_OCC_CO LGR_MM_OCC
#endif
__global__ __launch_bounds__(NTHR, CO ? LGR_MM_OCC_CO : LGR_MM_OCC) void match_mfma(const typename OpFmt<FMT>::frag* __restrict__ Ap, const typename OpFmt<FMT>::frag* __restrict__ Bp,
                                                     size_t bset_stride /* fragments */, float c_scale /* 2^2s, F16 only */, float out_scale /* 2^-2s */,
                                                     const int* __restrict__ blkcl, const float* __restrict__ nA, int ma_pad, int mb_pad,
                                                     int rg_rows, const int* __restrict__ tile_group, const unsigned* __restrict__ stage_mask,
                                                     int* __restrict__ rowmin /* [n_groups][ma_pad] */,
                                                     int* __restrict__ colmin /* [ma_pad/rg_rows][mb_pad] */,
                                                     int n_cc, int item_rb, const int2* __restrict__ items, const int* __restrict__ xcd_start,
                                                     int* __restrict__ xcd_ctr, CoarseArgs ca) {
    // column stage double buffered in LDS: the next stage is prefetched into registers while the current one is
    // consumed and written to the other buffer afterwards -> one barrier per stage, global latency hidden
    constexpr bool F16 = FMT != FMT_F32;
    static_assert(!CO || FMT == FMT_F16R, "the rotated K order puts a coarse d2~ into the first two steps (pack16_kernel)");
    constexpr bool COARSE = CO, use_coarse = CO;
    unsigned n_tested = 0u, n_rejected = 0u, n_skipped = 0u;   // wave-uniform tile counts
    typedef typename OpFmt<FMT>::frag frag;
    constexpr int KS = OpFmt<FMT>::KS;
    constexpr int STAGE_FRAGS = STAGE_TILES * KS * 64;
    constexpr int STAGE_VEC4 = STAGE_FRAGS * (int) sizeof(frag) / 16;   // 16-byte pieces per stage
    // ONE __shared__ object: with a second one beside the LDS-DMA staging array hipcc waits vmcnt(0) before the first
    // ds_read of every stage, i.e. for the DMA of the NEXT stage it has just issued (cdna_hip_programming.md, projection GEMM
    // item 4a) -- the staging then never overlaps the stage's own MFMA chains
    constexpr int CO_FRAGS = STAGE_TILES * 2 * 64, CO_NB = 6, CO_D = 4;   // coarse kernel: ring slot (fragments), slots, stages in flight
    static_assert(CO_D == 4 && CO_NB == CO_D + 2, "the counted waits below are written for two pairs of stages in flight");
    constexpr int BS_BYTES = CO ? CO_NB * CO_FRAGS * (int) sizeof(frag) : 2 * STAGE_FRAGS * (int) sizeof(frag);
    // (coarse launch: + the chunk's per-column thresholds as bf16, rounded up -- 8 KB; 72.6 KB per workgroup, two per CU)
    constexpr int UCOL_BYTES = CO ? CHUNK_COLS * 2 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[BS_BYTES + CHUNK_COLS * 4 + (CHUNK_COLS / TILE) * 4 + 16 + UCOL_BYTES];
    frag (*Bs)[STAGE_FRAGS] = reinterpret_cast<frag (*)[STAGE_FRAGS]>(smem);
    int* const cmin_s = reinterpret_cast<int*>(smem + BS_BYTES);
    int* const tg_s = cmin_s + CHUNK_COLS;
    int& item_s = tg_s[CHUNK_COLS / TILE];
    unsigned short* const ucol_s = reinterpret_cast<unsigned short*>(smem + BS_BYTES + CHUNK_COLS * 4 + (CHUNK_COLS / TILE) * 4 + 16);

    // Persistent workgroups over a compacted work list.  An item is (column chunk, item_rb row blocks) with at least
    // one stage to compute.  Hardware places workgroup i on XCD i % 8; the list is partitioned per XCD (XCD x owns the
    // chunks x, x + 8, ...; items ordered by chunk, then rows), and the workgroups of an XCD pull items in order from
    // a shared counter: a chunk's B operand stays in one L2 while its items run, chunks of different cost interleave
    // across the XCDs, and nobody idles behind a static partition.  (Speed only: any item order gives the same tables.)
    const int xcd = blockIdx.x % 8;
    const int item0 = xcd_start[xcd], n_items = xcd_start[xcd + 1] - item0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int rg_blocks = rg_rows / BLOCK_ROWS;
    const int n_rb_total = ma_pad / BLOCK_ROWS;
    constexpr int IINF = 0x7f800000;   // +inf as bits
    static_assert(STAGE_VEC4 % 64 == 0, "a stage is a whole number of 1 KB DMA pieces");
    if (COLDIR) {
        for (int i = tid; i < CHUNK_COLS; i += NTHR) cmin_s[i] = IINF;   // every flush leaves the array at +inf again
    }
    const unsigned cmin_lane_base = (unsigned) (uintptr_t) (__attribute__((address_space(3))) int*) &cmin_s[lane & 31];
    int cur_cc = -1, col_tile0 = 0, n_coltiles = 0;
    unsigned full = 0u;
    unsigned tend[STAGE_TILES] = {0u, 0u, 0u, 0u};
    PROF_T(t_wg0);
    PROF_CNT(8);

  for (;;) {
    __syncthreads();   // all waves are done with the previous item (tg_s, item_s, cmin_s)
    if (tid == 0) item_s = n_items > 0 ? atomicAdd(&xcd_ctr[xcd], 1) : 0x7fffffff;   // (an empty list leaves the counter alone: another kernel may own it)
    __syncthreads();
    const int it = item_s;
    if (it >= n_items) break;
    const int2 item = items[item0 + it];
    const int cc = item.x, rb0 = item.y;
    const int n_rb = min(item_rb, n_rb_total - rb0);
    if (cc != cur_cc) {
        cur_cc = cc;
        col_tile0 = cc * (CHUNK_COLS / TILE);
        n_coltiles = min(CHUNK_COLS / TILE, mb_pad / TILE - col_tile0);
        const int n_stages = n_coltiles / STAGE_TILES;
        full = n_stages >= 32 ? 0xffffffffu : ((1u << n_stages) - 1u);
        // column group (train leaf) of every 32-column tile of this chunk; tend[ct] bit st = tile ct of stage st is
        // the last tile of its group (uniform registers: nothing is loaded between the MFMA chains)
        if (tid < CHUNK_COLS / TILE) tg_s[tid] = tid < n_coltiles ? tile_group[col_tile0 + tid] : -1;
        if (CO && ca.u_colv) {
            // per-column row of the thresholds, scaled like the accumulator: U_col (1 + 1e-5) 1.0001 c_scale, as bf16 rounded UP (a larger
            // threshold only keeps more); the chunk's columns are the same for all the item's row blocks
            for (int i = tid; i < CHUNK_COLS; i += NTHR) {
                const int col = col_tile0 * TILE + i;
                float t = 0.f;
                if (col < mb_pad) t = ((fmaxf(ca.u_colv[col], 0.f) * 1.00001f) * 1.0001f) * c_scale;
                const unsigned bits = t >= 0.f ? __float_as_uint(t) : 0x7f800000u;   // (NaN, never expected: keep everything)
                ucol_s[i] = (unsigned short) (min(bits + 0xffffu, 0x7f800000u) >> 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < STAGE_TILES; ++ct) {
            int t = (lane & 31) * STAGE_TILES + ct;
            bool e = t < n_coltiles && (t == n_coltiles - 1 || tg_s[t + 1] != tg_s[t]);
            tend[ct] = __builtin_amdgcn_readfirstlane((unsigned) (__ballot(e) & 0xffffffffull));
        }
    }
    // the stage masks of the item's row blocks, fetched once (lane rbi holds the mask of row block rb0 + rbi)
    unsigned my_mask = full;
    if (stage_mask) my_mask = lane < n_rb ? (stage_mask[(size_t) (rb0 + lane) * n_cc + cc] & full) : 0u;
    bool col_dirty = false;

    for (int rbi = 0; rbi < n_rb; ++rbi) {
        const int rb = rb0 + rbi;
        unsigned mask = stage_mask ? __builtin_amdgcn_readlane(my_mask, rbi) : full;   // uniform over the workgroup
        if (mask) {
            PROF_T(t_v0);
            PROF_CNT(9);
            col_dirty = true;
            const int row_tile = rb * (BLOCK_ROWS / TILE) + wave * RW;
            const frag* Bset = Bp + (size_t) blkcl[rb] * bset_stride + (size_t) col_tile0 * KS * 64;
            // A fragments (coalesced 256-B loads) and the |a'|^2 of the 16 rows each lane's accumulators cover
            frag a[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a[kk] = Ap[((size_t) row_tile * KS + kk) * 64 + lane];
            f32x16 nav = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (!F16) {   // f32 operands: |a'|^2 through the accumulator input (the f16 format carries it in K slots)
#pragma unroll
                for (int g = 0; g < 16; ++g) nav[g] = nA[row_tile * TILE + (g & 3) + 8 * (g >> 2) + 4 * half];
            }
            // coarse-rejection thresholds of the 128 column tiles of this (row block, chunk) for this wave's 32 rows: tile q = 4 stage + ct
            // sits in lane q & 63 of t_lane[q >> 6] (bit pattern of a float >= 0, scaled like the accumulator; +inf = keep everything)
            int t_lane[2] = {IINF, IINF};
            // Round 3: the sweep's product is TRANSPOSED (columns x rows), so a lane holds one ROW of the wave's tile (row lane & 31, 16 of
            // the tile's 32 columns) and compares its smallest coarse value with a threshold made of ITS OWN row's upper bound and the
            // tile's column bound: thr = max(U_row, U_cols(tile)) (1 + 1e-5) + error term.  With the tile's largest row bound for every row
            // (round 2) one loose row of the 32 kept a tile alive for all of them: 15 % of the tested tiles went on.
            float e_lane[2] = {0.f, 0.f};   // error term of tile q (scaled like the accumulator), same lanes as t_lane
            float trow = 0.f;               // this lane's row: U_row (1 + 1e-5) 1.0001 c_scale
            unsigned long long skipm[2] = {0ull, 0ull};   // shell test: bit q & 63 of skipm[q >> 6] = this wave leaves tile q out (wave uniform)
            unsigned kept[STAGE_TILES] = {0u, 0u, 0u, 0u};   // coarse sweep: bit st of kept[ct] = tile ct of stage st goes on (wave uniform)
            int rmin[16];   // float bit patterns, see the epilogue note
#pragma unroll
            for (int g = 0; g < 16; ++g) rmin[g] = IINF;

            // Column stages go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers; each wave
            // instruction copies one contiguous 1 KB piece, the stage image has the same order in memory and in LDS).
            auto stage_dma = [&](int stage, int to_buf) {
                const char* src = reinterpret_cast<const char*>(Bset + (size_t) stage * STAGE_FRAGS);
                char* dst = reinterpret_cast<char*>(Bs[to_buf]);
#pragma unroll
                for (int piece = wave; piece < STAGE_VEC4 / 64; piece += WAVES)
                    __builtin_amdgcn_global_load_lds((const void*) (src + piece * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*) (dst + piece * 1024), 16, 0, 0);
            };
            // first active stage of this row block (barrier first: every wave is past the previous row block's LDS reads)
            int st = __builtin_ctz(mask);
            const unsigned sweep_mask = mask, skipped_before = n_skipped;
            // (a raw barrier: the A fragments just requested stay in flight while the first stage is requested)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
            if (!COARSE) {
                stage_dma(st, 0);
                __syncthreads();   // waits for the A fragments and the DMA (vmcnt(0)) and makes the stage visible
            }
            PROF_T(t_v1);
            PROF_ADD(0, t_v0, t_v1);
            // Stage loop: the DMA of the next active stage into the other buffer is issued before the current stage is
            // consumed; the barrier at the end of the stage waits for it.
            // On gfx950 the f32 MFMA runs on the FP32 lanes the VALU uses (equal peak rate; no co-execution was
            // measured: removing the epilogue saved exactly its VALU time), so the epilogue is kept minimal:
            //  * |a'|^2 enters through the accumulator input of the first MFMA step (f32) or through spare K slots
            //    (f16): d2~ = S + |a'|^2 costs nothing;
            //  * minima are taken on the bit patterns with v_min_i32 / v_min3_i32 (one instruction per slot, no
            //    canonicalising v_max pair as a float min of raw MFMA output needs).  Signed-int order equals float
            //    order except among negative values, where it keeps the one closest to zero; d2~ < 0 only within
            //    the proven error eps of a true distance >= 0, so the filtered minimum stays within eps;
            //  * a VALU lane swap instead of an LDS shuffle folds the two lane halves of the column chain.
            // The B fragment of the next tile is fetched from LDS before the epilogue runs.
            // Epilogue of one finished 32x32 tile: tile ct of stage st; nxt = the stage computed after st.
            auto row_min1 = [&](const f32x16& acc) {
#pragma unroll
                for (int g = 0; g < 16; ++g) rmin[g] = min(rmin[g], __float_as_int(acc[g]));
            };
            auto col_min = [&](const f32x16& acc, int st, int ct) {
                if (COLDIR) {
                    int cm = min(__float_as_int(acc[0]), __float_as_int(acc[1]));
#pragma unroll
                    for (int g = 2; g < 16; g += 2) cm = min(min(cm, __float_as_int(acc[g])), __float_as_int(acc[g + 1]));
                    // The two lane halves (rows 4 * half + ...) hold partial minima of the same 32 columns.  Round 3: they are NOT folded in
                    // the VALU any more (v_permlane32_swap + select + min = 3 vector instructions per tile on the unit that bounds this
                    // kernel): both halves issue the LDS atomic on the same word and the LDS unit, which has slack, merges them.
                    // (inline asm: behind a compiler-visible LDS atomic hipcc waits vmcnt(0), i.e. for the LDS-DMA of the
                    // next stage, in the middle of the current one; the s_waitcnt lgkmcnt(0) before the column flush's
                    // barrier retires these.)  The address is a per-lane base + a compile-time tile offset + a per-stage scalar.
#ifdef LGR_MM_FOLD   // A/B switch (tools/exp_env.sh): the round-2 epilogue folded the halves with v_permlane32_swap before the atomic
                    {
                        auto sw = __builtin_amdgcn_permlane32_swap((unsigned) cm, (unsigned) cm, false, false);
                        cm = min(cm, (int) (half ? sw[0] : sw[1]));
                    }
#endif
                    const unsigned lds_addr = cmin_lane_base + (unsigned) (st * STAGE_TILES * TILE * 4);   // (st is wave uniform: one add per stage)
                    static_assert(STAGE_TILES == 4 && TILE == 32, "the tile offsets below are literal");
                    switch (ct) {   // ct is a literal after unrolling: the tile's 128-byte offset rides in the instruction
                        case 0: asm volatile("ds_min_i32 %0, %1" : : "v"(lds_addr), "v"(cm)); break;
                        case 1: asm volatile("ds_min_i32 %0, %1 offset:128" : : "v"(lds_addr), "v"(cm)); break;
                        case 2: asm volatile("ds_min_i32 %0, %1 offset:256" : : "v"(lds_addr), "v"(cm)); break;
                        default: asm volatile("ds_min_i32 %0, %1 offset:384" : : "v"(lds_addr), "v"(cm)); break;
                    }
                }
            };
            auto tile_ends_group = [&](int st, int ct) {
                const unsigned te = ct == 0 ? tend[0] : ct == 1 ? tend[1] : ct == 2 ? tend[2] : tend[3];
                return ((te >> st) & 1u) != 0u;
            };
            auto flush_rows = [&](int grp) {
                {
                    PROF_CNT(10);
                    // Halving butterfly over the 32 lanes of each half wave: at every step a lane keeps half of
                    // its registers and receives the partner's copy of them, so 16 registers x 32 lanes reduce to
                    // one value per lane with 16 + 8 + 4 + 2 + 1 exchanges instead of 16 x 5; lane bits 4..1 then
                    // select the register (= row) the lane ends up holding, and one atomic instruction with 16
                    // active lanes per half wave writes all rows.  All exchanges stay in the VALU (no LDS round trip):
                    // lane ^ 16 is v_permlane16_swap -- swapping the odd 16-lane rows of rmin[j] with the even rows of
                    // rmin[8 + j] hands every lane exactly the partner copy of the register it keeps, in both directions
                    // at once -- and lane ^ 8 / 4 / 2 / 1 are DPP operands of the v_min (row_ror:8; row_half_mirror then
                    // quad_perm [3,2,1,0] = ^7 ^3; quad_perm [2,3,0,1]; quad_perm [1,0,3,2]).
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
                    // register index held by this lane: bit 3 <- lane bit 4, bit 2 <- bit 3, bit 1 <- bit 2, bit 0 <- bit 1
                    const int g = (lane >> 1) & 15;
                    if (F16) w1 = __float_as_int(__int_as_float(w1) * out_scale);   // back to d2~ (monotonic)
                    if ((lane & 1) == 0 && w1 != IINF)
                        atomicMin(&rowmin[(size_t) grp * ma_pad + row_tile * TILE + (g & 3) + 8 * (g >> 2) + 4 * half], w1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) rmin[r] = IINF;
                }
            };
            auto maybe_flush = [&](int st, int ct, int nxt) {
                // flush the row minima when the column group (train leaf) ends, or before skipped stages
                if (tile_ends_group(st, ct) || (ct == STAGE_TILES - 1 && nxt != st + 1)) flush_rows(tg_s[st * STAGE_TILES + ct]);
            };
            // Epilogue of one finished 32x32 tile: tile ct of stage st; nxt = the stage computed after st.
            auto epilogue = [&](const f32x16& acc, int st, int ct, int nxt) {
                row_min1(acc);
                col_min(acc, st, ct);
                maybe_flush(st, ct, nxt);
            };
            // Schedules tried and measured on a dense 400k x 400k probe (35.9 ms as is; MFMA chains alone 24.9 ms: the
            // chip holds ~1.4 GHz under this f16 MFMA load): deferring a tile's epilogue behind the next tile's MFMA
            // chain (software pipeline, with and without register double buffering of the B fragments) -1..-2 % at
            // 4 waves/SIMD with spills, +8 % at 2 waves/SIMD; no stage DMA -14 %; no barrier 0 %; no column minima -5 %;
            // two column tiles per epilogue (one v_min3 per accumulator pair for the row minima, 8 fewer vector
            // instructions per tile): +13 % at 4 waves/SIMD (spills), -1.4 % at 2 waves/SIMD.
            auto compute = [&](int st, int buf, int nxt) {
                frag b[KS];
                if (COARSE && use_coarse) {
                    // Coarse sweep: the first two steps of a tile give a coarse d2~ (see CoarseArgs); tiles with an element
                    // under the stage threshold are only RECORDED here (kept[ct] bit st) and finished after the sweep,
                    // outside the per-stage barriers -- every wave does the same work per stage, so nobody waits at the
                    // barrier for a wave that happens to hold the few full tiles.
                    const int t_sel = st < 16 ? t_lane[0] : t_lane[1];   // (st is wave uniform)
                    const float e_sel = st < 16 ? e_lane[0] : e_lane[1];
                    // threshold of this lane's row against tile q: max(row part + error term, column part + error term), as a bit pattern
                    auto lane_thr = [&](int q) -> int {
                        const int tc = __builtin_amdgcn_readlane(t_sel, q & 63);
                        const float te = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e_sel), q & 63));
                        return max(__float_as_int(trow + te), tc);   // (both >= +0: integer order is float order; IINF = keep everything)
                    };
                    // A tile the lane test keeps is tested again PER ELEMENT when the columns' own bounds are at hand: element (row lane & 31,
                    // column c) goes on only if its coarse value is at most max(U_row, U_col(c)) (1 + 1e-5) + error term.  (The first test uses
                    // the largest column bound of the tile for all 32 columns: one loose column kept a tile alive for every row -- 5.8 M of
                    // 54 M tested tiles went on, 0.7 M would with no column side at all.)  Registers of the transposed product: acc[g] =
                    // column (g & 3) + 8 (g >> 2) + 4 half of the tile.
                    auto keep_tile = [&](const f32x16& acc, int q) -> bool {
                        if (!ca.u_colv) return true;
                        const float te = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e_sel), q & 63));
                        const unsigned short* uc = ucol_s + (q << 5) + 4 * half;
                        bool kp = false;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const uint2 w = *reinterpret_cast<const uint2*>(uc + 8 * gq);   // four bf16 thresholds
                            const float c0 = __uint_as_float(w.x << 16), c1 = __uint_as_float(w.x & 0xffff0000u);
                            const float c2 = __uint_as_float(w.y << 16), c3 = __uint_as_float(w.y & 0xffff0000u);
                            kp = kp || acc[4 * gq] <= fmaxf(trow, c0) + te || acc[4 * gq + 1] <= fmaxf(trow, c1) + te
                                    || acc[4 * gq + 2] <= fmaxf(trow, c2) + te || acc[4 * gq + 3] <= fmaxf(trow, c3) + te;
                        }
                        return __ballot(kp) != 0ull;
                    };
                    const frag* cs = reinterpret_cast<const frag*>(smem) + buf * CO_FRAGS + lane;   // ring slot `buf`: [tile][2 steps][64]
                    const unsigned sk4 = (unsigned) ((st < 16 ? skipm[0] : skipm[1]) >> ((st * STAGE_TILES) & 63)) & 0xfu;   // tiles the shell test leaves out
                    if (sk4) {
                        n_skipped += (unsigned) __builtin_popcount(sk4);
                        if (sk4 == 0xfu) return;
#pragma unroll
                        for (int ct = 0; ct < STAGE_TILES; ++ct) {
                            if ((sk4 >> ct) & 1u) continue;
                            const frag c0 = cs[ct * 128], c1 = cs[ct * 128 + 64];
                            f32x16 acc = mfma_step(c0, a[0], nav);   // transposed: lane = row
                            acc = mfma_step(c1, a[1], acc);
                            int m = min(__float_as_int(acc[0]), __float_as_int(acc[1]));
#pragma unroll
                            for (int g = 2; g < 16; g += 2) m = min(min(m, __float_as_int(acc[g])), __float_as_int(acc[g + 1]));
                            if (__ballot(m <= lane_thr(st * STAGE_TILES + ct)) != 0ull && keep_tile(acc, st * STAGE_TILES + ct)) kept[ct] |= 1u << st;
                        }
                        return;
                    }
                    b[0] = cs[0];
                    b[1] = cs[64];
#pragma unroll
                    for (int ct = 0; ct < STAGE_TILES; ++ct) {
                        f32x16 acc = mfma_step(b[0], a[0], nav);   // transposed: lane = row
                        acc = mfma_step(b[1], a[1], acc);
                        if (ct + 1 < STAGE_TILES) {
                            b[0] = cs[(ct + 1) * 128];
                            b[1] = cs[(ct + 1) * 128 + 64];
                        }
                        // smallest coarse d2~ of the lane's 16 elements against the stage threshold (bit patterns: the
                        // signed-int order errs only among negative values, which are below any threshold anyway)
                        int m = min(__float_as_int(acc[0]), __float_as_int(acc[1]));
#pragma unroll
                        for (int g = 2; g < 16; g += 2) m = min(min(m, __float_as_int(acc[g])), __float_as_int(acc[g + 1]));
                        if (__ballot(m <= lane_thr(st * STAGE_TILES + ct)) != 0ull && keep_tile(acc, st * STAGE_TILES + ct)) kept[ct] |= 1u << st;   // (the tested / rejected counts are taken from the masks after the sweep)
                    }
                    return;
                }
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = Bs[buf][kk * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < STAGE_TILES; ++ct) {
                    f32x16 acc = mfma_step(a[0], b[0], nav);
#pragma unroll
                    for (int kk = 1; kk < KS; ++kk) acc = mfma_step(a[kk], b[kk], acc);
                    if (ct + 1 < STAGE_TILES) {
#pragma unroll
                        for (int kk = 0; kk < KS; ++kk) b[kk] = Bs[buf][((ct + 1) * KS + kk) * 64 + lane];
                    }
                    epilogue(acc, st, ct, nxt);
                }
            };
            if (COARSE) {
                // Coarse sweep over the active stages.  Only the two coarse fragments of each tile are staged (8 x 1 KB pieces
                // per stage, one LDS-DMA instruction per wave) into a ring of CO_NB slots, CO_D stages ahead: a stage of the
                // sweep is too short (eight MFMAs per wave) to cover the latency of a DMA issued one stage earlier.  Per
                // PAIR of stages: counted vmcnt (this wave's pieces of the pair have landed; newer ones stay in flight), raw barrier
                // (everybody's pieces have landed, everybody is done with the previous pair), issue the DMAs of the pair after
                // the next into the slots the previous pair used, compute both.  (__syncthreads() would drain the DMA queue.)
                static_assert(WAVES == 2 * STAGE_TILES, "one coarse piece per wave");
                unsigned to_issue = mask, to_do = mask;
                int issued = 0, done_ = 0;
                auto dma_coarse = [&]() {
                    const int s_ = __builtin_ctz(to_issue);
                    to_issue &= to_issue - 1u;
                    const char* src = reinterpret_cast<const char*>(Bset + (size_t) s_ * STAGE_FRAGS + (size_t) ((wave >> 1) * KS + (wave & 1)) * 64);
                    char* dst = reinterpret_cast<char*>(smem) + (issued % CO_NB) * (CO_FRAGS * (int) sizeof(frag)) + wave * 1024;
                    __builtin_amdgcn_global_load_lds((const void*) (src + lane * 16), (__attribute__((address_space(3))) void*) dst, 16, 0, 0);
                    ++issued;
                };
                while (issued < CO_D && to_issue) dma_coarse();
                // The thresholds come from ordinary loads, issued HERE, behind the first DMAs: hipcc waits for them with vmcnt(0), i.e. for
                // the A fragments, the DMAs and these loads together -- one memory round trip per visit.  (Round 2 had them in registers
                // before the first DMA, behind the visit's full barrier: two round trips in a row, 15 % of the launch by the in-kernel timers.)
                if (use_coarse) {
                    const float ur = ca.u_rt[rb * (BLOCK_ROWS / TILE) + wave];   // this wave's 32 rows
                    if (ca.u_row) trow = ((fmaxf(ca.u_row[(size_t) row_tile * TILE + (lane & 31)], 0.f) * 1.00001f) * 1.0001f) * c_scale;
                    const float x = ca.xmax[rb / rg_blocks];
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int q = lane + 64 * r;
                        const int gst = min(cc * STAGES_PER_CHUNK + (q >> 2), ca.n_stage_total - 1);
                        const int gct = min(col_tile0 + q, ca.n_ct_total - 1);
                        const float us = ca.u_stage ? __uint_as_float(ca.u_ct[gct]) : 0.f;
                        float xt = x, yt = ca.ymax[(size_t) blkcl[rb] * ca.n_stage_total + gst];
                        // Shell test (round 3): rows and columns are sorted by their distance to the cluster centre inside their leaves, so
                        // this wave's 32 rows and the 32 columns of tile q are thin radial shells about the centre both are packed against;
                        // |a - b| >= | |a - c| - |b - c| |, so when the gap of the two shells exceeds the upper bounds of all its rows and
                        // columns the tile holds no nearest neighbour and no tie: not even its coarse steps are issued.  The shells' outer
                        // radii are also the tile's own max |a'| and max |b'|: the error term of the coarse threshold is stated in those
                        // (it was the row group's and the stage's maxima: (x + y)^2 up to 2.5 times larger, and one tile in six went on).
                        if (ca.rt_shell) {
                            const float2 sa = ca.rt_shell[rb * (BLOCK_ROWS / TILE) + wave];
                            const float2 sb = ca.ct_shell[(size_t) blkcl[rb] * ca.n_ct_total + gct];
                            const float gap = fmaxf(sb.x - sa.y, sa.x - sb.y) - 4e-6f * (sa.y + sb.y);   // (norms: 33-term float sums, 2e-6 on a radius)
                            const float U = fmaxf(fmaxf(ur, us), 0.f);
                            const bool skip = gap > 0.f && gap * gap * (0.99999f * 0.99999f * 0.99999f) > U * 1.00001f + 1e-12f;
                            skipm[r] = __ballot(skip);
                            xt = fminf(xt, sa.y); yt = fminf(yt, sb.y);
                        }
                        const float s = xt + yt;
                        const float et = (ca.quad * s) * s + ca.cross * (xt * yt) + ca.lin * s + ca.abs;
                        // column side (and, without per-row bounds, the tile's largest row bound) + error term; the row side is added per lane
                        float t = fmaxf(fmaxf(ca.u_row ? 0.f : ur, us), 0.f) * 1.00001f + et;
                        t = (t * 1.0001f) * c_scale;
                        t_lane[r] = t >= 0.f ? __float_as_int(t) : IINF;   // NaN (never expected): keep everything
                        e_lane[r] = (et * 1.0001f) * c_scale;
                    }
                }
                asm volatile("" : : "v"(t_lane[0]), "v"(t_lane[1]), "v"(e_lane[0]), "v"(e_lane[1]), "v"(trow) : "memory");
                // Two stages per barrier (round 3): a stage of the sweep is eight MFMAs per wave, and at one raw barrier per stage the waves
                // spent as long waiting for the slowest of the eight as computing (a stage visit took 1.9 k cycles against ~1 k of issue).
                while (to_do) {
                    st = __builtin_ctz(to_do);
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
                    compute(st, done_ % CO_NB, -1);
                    if (st2 >= 0) compute(st2, (done_ + 1) % CO_NB, -1);
                    done_ += n_now;
                }
            }
            mask &= mask - 1u;   // st is taken
            int buf = 0;
            while (!COARSE) {
                int nxt = -1;
                if (mask) { nxt = __builtin_ctz(mask); mask &= mask - 1u; }
                if (nxt >= 0) stage_dma(nxt, buf ^ 1);
                compute(st, buf, nxt);
                if (nxt < 0) break;
                PROF_T(t_s0);
                __syncthreads();   // DMA landed (vmcnt(0)) and visible; all waves done with the buffer refilled next
                PROF_T(t_s1);
                PROF_ADD(2, t_s0, t_s1);
                buf ^= 1;
                st = nxt;
            }
            if (COARSE) {
                // (n_skipped so far counts every visit of this wave; the difference to its value before the sweep is this visit's)
                {
                    const unsigned swept = (unsigned) (STAGE_TILES * __builtin_popcount(sweep_mask)), sk_now = n_skipped - skipped_before;
                    n_tested += swept - sk_now;
                    n_rejected += swept - sk_now - (unsigned) (__builtin_popcount(kept[0]) + __builtin_popcount(kept[1]) + __builtin_popcount(kept[2]) + __builtin_popcount(kept[3]));
                }
                // the recorded tiles, in column order: the whole chain on B fragments read straight from memory (the stage
                // image has the same order there as in LDS), the usual epilogue, row minima flushed when the group changes
                PROF_T(t_k0);
                unsigned any = kept[0] | kept[1] | kept[2] | kept[3];
                int prev_grp = -1;
                while (any) {
                    const int ks_ = __builtin_ctz(any);
                    any &= any - 1u;
#pragma unroll
                    for (int ct = 0; ct < STAGE_TILES; ++ct) {
                        if (!((kept[ct] >> ks_) & 1u)) continue;
                        const int grp = __builtin_amdgcn_readfirstlane(tg_s[ks_ * STAGE_TILES + ct]);
                        if (grp != prev_grp && prev_grp >= 0) flush_rows(prev_grp);
                        prev_grp = grp;
                        const frag* bsrc = Bset + (size_t) ks_ * STAGE_FRAGS + (size_t) ct * KS * 64 + lane;
                        frag b[KS];
#pragma unroll
                        for (int kk = 0; kk < KS; ++kk) b[kk] = bsrc[kk * 64];
                        f32x16 acc = mfma_step(a[0], b[0], nav);
#pragma unroll
                        for (int kk = 1; kk < KS; ++kk) acc = mfma_step(a[kk], b[kk], acc);
                        row_min1(acc);
                        col_min(acc, ks_, ct);
                    }
                }
                if (prev_grp >= 0) flush_rows(prev_grp);
                PROF_T(t_k1);
                PROF_ADD(5, t_k0, t_k1);
            }
            PROF_T(t_v2);
            PROF_ADD(1, t_v1, t_v2);
        }
    }
    // column minima of this item -> table (the item covers exactly one row group: single owner, plain read-modify-write)
    if (COLDIR && col_dirty) {
        PROF_T(t_c0);
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");   // the ds_min_i32 of col_min
        __syncthreads();
        int rg = rb0 / rg_blocks;
        int ncols = n_coltiles * TILE;
        int* dst = colmin + (size_t) rg * mb_pad + col_tile0 * TILE;
        constexpr int NCM = CHUNK_COLS / NTHR;   // 8 columns per thread: all loads in flight before the merge
        // (atomics without return: nobody waits for the table's old values -- a read-modify-write cost one exposed memory round trip per item)
#pragma unroll
        for (int j = 0; j < NCM; ++j) {
            int i = tid + NTHR * j;
            const int cur = i < ncols ? cmin_s[i] : IINF;
            if (cur != IINF) {
                const int v = F16 ? __float_as_int(__int_as_float(cur) * out_scale) : cur;
                atomicMin(&dst[i], v);
                cmin_s[i] = IINF;
            }
        }
        PROF_T(t_c1);
        PROF_ADD(3, t_c0, t_c1);
    }
  }
    if (use_coarse && ca.cnt && lane == 0) {
        if (n_tested) atomicAdd(&ca.cnt[0], (unsigned long long) n_tested);
        if (n_rejected) atomicAdd(&ca.cnt[1], (unsigned long long) n_rejected);
        if (n_skipped) atomicAdd(&ca.cnt[2], (unsigned long long) n_skipped);
    }
    PROF_T(t_wg1);
    PROF_ADD(4, t_wg0, t_wg1);
}

// work list of match_mfma: flag every (XCD-major chunk, item row) that has something to compute, scan, emit
__global__ void items_flag_kernel(const unsigned* __restrict__ mask, int n_rb, int n_cc, int item_rb, int n_ir, int ccx, int* __restrict__ flags) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 8 * ccx * n_ir) return;
    int xcd = j / (ccx * n_ir), rem = j % (ccx * n_ir);
    int cc = (rem / n_ir) * 8 + xcd, ir = rem % n_ir;
    int f = 0;
    if (cc < n_cc) {
        if (!mask) f = 1;
        else
            for (int r = ir * item_rb; r < min(n_rb, (ir + 1) * item_rb); ++r) f |= mask[(size_t) r * n_cc + cc] != 0u ? 1 : 0;
    }
    flags[j] = f;
}
__global__ void items_emit_kernel(const int* __restrict__ flags, const int* __restrict__ pos, int item_rb, int n_ir, int ccx,
                                  int2* __restrict__ items, int* __restrict__ xcd_start /* [9] */) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_xcd = ccx * n_ir;
    if (j >= 8 * per_xcd) return;
    int xcd = j / per_xcd, rem = j % per_xcd;
    if (flags[j]) items[pos[j]] = make_int2((rem / n_ir) * 8 + xcd, (rem % n_ir) * item_rb);
    if (rem == 0) xcd_start[xcd] = pos[j];
    if (j == 8 * per_xcd - 1) xcd_start[8] = pos[j] + flags[j];
}


}  // namespace
