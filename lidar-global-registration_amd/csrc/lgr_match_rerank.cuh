// lgr_match_rerank.cuh -- 4. exact canonical distance, error bounds, table scans, schedule, table init, self-check, exact re-rank.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
#pragma once
#include "lgr_match_common.cuh"
#include "lgr_match_cluster.cuh"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// exact canonical distance: cv::hal::normL2Sqr_ (OpenCV 4.5.1, SSE baseline: 4 lanes x 4 accumulators over blocks
// of 16 floats, mul then add, reduce ((acc0+acc1)+acc2)+acc3 then (s0+s2)+(s1+s3), scalar tail) followed by sqrt.
// Must stay op-for-op identical to oracle/src/orc_matching.cpp:l2sqr33 (compiled with -ffp-contract=off).
__device__ __forceinline__ float exact_l2(const float* __restrict__ a, const float* __restrict__ b) {
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[i][l] = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                int j = 16 * blk + 4 * i + l;
                float t = a[j] - b[j];
                acc[i][l] = t * t + acc[i][l];
            }
    float s[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) s[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
    float d = (s[0] + s[2]) + (s[1] + s[3]);
    float t = a[32] - b[32];
    d = d + t * t;
    return __builtin_sqrtf(d);   // IEEE-correct sequence (NOT __fsqrt_rn, which is the 1-ulp v_sqrt_f32 on gfx950)
}

// tie rank of train index j: highest bf block first, lowest index inside a block first (smaller rank wins)
__device__ __forceinline__ unsigned tie_rank(int j, int block, int nblocks) {
    int blk = j / block;
    return (unsigned) ((nblocks - 1 - blk) * (long long) block + (j - blk * block));
}

// ---------------------------------------------------------------------------------------------------------------
// 4a. candidate groups per query.  table[g][q_pad] holds, for padded query position i and train group g, the
// filtered minimum v of d2~ = S + |a'|^2.  Proven bound of |filtered - true| for
// every pair of (query i, group g)  (DESIGN.md "matcher margin"): centring (2 roundings) + fma chain of 34 products
// + norm rounding + the column-direction add:  eps = 4 g40 (x + y)^2, g40 = 40u/(1-40u), u = 2^-24, where x, y are
// |q - c| and the group's max |t - c| for the centre c the pair was computed with.
//   upper = v + eps, lower = v - eps;  UB = min_g upper;  group g is a candidate iff lower_g <= UB + slack, with
//   slack = 1e-5 * d2(UB) (two rows whose true d2 differ by less may tie or swap in the canonical float distance)
//         + float rounding of the comparison.
struct RerankCounters { unsigned n_items; unsigned n_dense; unsigned pad0; unsigned pad1; };

// extra terms of the f16-split operand path (0 on the f32 path): eps += lin * (x + y) + abs

template <bool ROWDIR>
__device__ __forceinline__ float group_eps(int i, int g, float xq, const float* __restrict__ nT_sets, const float* __restrict__ gmax,
                                           int n_groups, int p_of_query, const int* __restrict__ cl_of_group, int t_pad, EpsExtra ex) {
    // ROWDIR: query = row i of cluster p (xq = |a'|), train group g of columns: y = gmaxB[p][g]
    // COLDIR: query = column i, train group g = row group of cluster p(g): x = gmaxA[g], y = |b - c_p(g)| (per set)
    // evaluated in float, inflated by 1e-5 (the five roundings below are worth 3e-7): an upper bound of the proven eps
    float x, y;
    if (ROWDIR) { x = xq; y = gmax[(size_t) p_of_query * n_groups + g]; }
    else {
        int p = cl_of_group[g];
        x = gmax[g];
        y = sqrtf(nT_sets[(size_t) p * t_pad + i]) * 1.0000002f;
    }
    return eps_xy(x, y, ex);
}

// Table scan of one query: calls f(group, value) for every computed, finite entry table[g][i].  Four loads are in flight
// before the first value is used (the loop body is short; one dependent global load per iteration was the whole cost).
// `own` (columns): the computed-flags of the query's own leaf.  A block of 256 columns can span two leaves, so the
// block's list is a superset; entries that were never computed are never initialised (init_tables_sparse_kernel) and must not
// be read.  nullptr: the list is exact (rows: one row block per workgroup) or everything was computed.
template <class F>
__device__ __forceinline__ void scan_groups(const float* __restrict__ table, size_t q_pad, int i, int n_list, int n_groups,
                                            const int* __restrict__ list_s, const uint8_t* __restrict__ own, F&& f) {
    const int n_it = n_list < 0 ? n_groups : n_list;
    const float inf = __uint_as_float(0x7f800000u);
    int k = 0;
    constexpr int U = 8;   // table loads in flight per thread (4: rerank_count 0.60 + 0.79 ms at 1M, col_u 0.29)
    for (; k + U <= n_it; k += U) {
        int g[U];
        float v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            g[j] = n_list < 0 ? k + j : list_s[k + j];
            v[j] = (!own || own[g[j]]) ? table[(size_t) g[j] * q_pad + i] : inf;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) if (v[j] < FLT_BIG) f(g[j], v[j]);
    }
    for (; k < n_it; ++k) {
        const int g = n_list < 0 ? k : list_s[k];
        const float v = (!own || own[g]) ? table[(size_t) g * q_pad + i] : inf;
        if (v < FLT_BIG) f(g, v);
    }
}

// Which table entries were computed at all (skipping leaves most of them at +inf): byte matrices derived from the
// done | scheduled tiles, so the table scans below read only the entries that can be finite.
//   rows:  comp_r[row block][column group]      cols:  comp_c[leaf][row group]
struct CompView { const uint8_t* m; int stride; const int* row_of_tile; };   // m == nullptr: everything was computed
__device__ __forceinline__ const uint8_t* comp_row(const CompView& c, int i, int block_row) {
    if (!c.m) return nullptr;
    int r = c.row_of_tile ? c.row_of_tile[i / TILE] : block_row;
    return c.m + (size_t) r * c.stride;
}
// compact list (dynamic LDS) of the groups computed for any query of this block; returns its length, or -1 when
// nothing was skipped (iterate all groups).  Every thread of the block must call it.
__device__ __forceinline__ int comp_list(const CompView& c, int i0, int n_i, int n_groups, int* list_s, int span = 0, bool* several = nullptr) {
    __shared__ int cnt_s;
    if (several) *several = false;
    if (!c.m) return -1;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    const int t0 = i0 / TILE, t1 = (min(i0 + (span ? span : (int) blockDim.x), n_i) - 1) / TILE;
    if (several && c.row_of_tile) *several = c.row_of_tile[t0] != c.row_of_tile[t1];   // tiles are sorted by leaf
    for (int g = threadIdx.x; g < n_groups; g += blockDim.x) {
        uint8_t f = 0;
        if (c.row_of_tile) {
            int prev = -1;
            for (int t = t0; t <= t1; ++t) { int r = c.row_of_tile[t]; if (r != prev) { f |= c.m[(size_t) r * c.stride + g]; prev = r; } }
        } else f = c.m[(size_t) (i0 / BLOCK_ROWS) * c.stride + g];
        if (f) list_s[atomicAdd(&cnt_s, 1)] = g;
    }
    __syncthreads();
    return cnt_s;
}
// tile states (section 3b).  sched: 1 = the whole (row block, leaf) tile, 2 = column-partial (final pass), SCHED_NOT_P0 = without the stages
// pass 0 already took.  done: 1 = computed, DONE_SHELL = pass 0 computed the stages of overlapping shells only (mask_kernel).
constexpr uint8_t SCHED_NOT_P0 = 4, DONE_SHELL = 8;
__global__ void comp_rows_kernel(const uint8_t* __restrict__ done, const uint8_t* __restrict__ sched, const int* __restrict__ group_leaf,
                                 int n_rb, int n_leaves, int n_groups, uint8_t* __restrict__ comp_r) {
    size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t) n_rb * n_groups) return;
    int rb = (int) (idx / n_groups), g = (int) (idx % n_groups);
    size_t t = (size_t) rb * n_leaves + group_leaf[g];
    // bit 1 (column-partial, see sched_kernel) leaves the row entries incomplete: not for the row scans; DONE_SHELL (pass 0 took the stages
    // of overlapping shells only) leaves minima over real pairs in them: good upper bounds, and everything they miss lies above them
    comp_r[idx] = (((done[t] | sched[t]) & 1) || (done[t] & DONE_SHELL)) ? 1 : 0;
}
__global__ void comp_cols_kernel(const uint8_t* __restrict__ done, const uint8_t* __restrict__ sched, int n_rb, int n_leaves, int n_rg,
                                 int rg_blocks, uint8_t* __restrict__ comp_c) {
    size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t) n_leaves * n_rg) return;
    int l = (int) (idx / n_rg), rg = (int) (idx % n_rg);
    uint8_t v = 0;
    for (int rb = rg * rg_blocks; rb < min(n_rb, (rg + 1) * rg_blocks); ++rb) v |= done[(size_t) rb * n_leaves + l] | sched[(size_t) rb * n_leaves + l];
    comp_c[idx] = v;
}

// upper bounds after a masked pass (section 3b): largest over the row block / the leaf of  min_g (filtered + eps)
__global__ __launch_bounds__(BLOCK_ROWS) void row_u_kernel(const float* __restrict__ table, int n_groups, int q_pad, const int* __restrict__ permQ,
                                                            const float* __restrict__ nQ, const int* __restrict__ blkclQ,
                                                            const float* __restrict__ gmax, EpsExtra ex, CompView comp, float* __restrict__ u_rb,
                                                            float* __restrict__ u_rt /* [row tiles]: the same maximum per 32-row tile */,
                                                            float* __restrict__ u_row /* [rows]: every row's own bound (-1: padding), or nullptr */) {
    extern __shared__ int list_s[];
    const int i = blockIdx.x * BLOCK_ROWS + threadIdx.x;
    const int n_list = comp_list(comp, blockIdx.x * BLOCK_ROWS, q_pad, n_groups, list_s);
    float ub = -1.f;   // padding rows need nothing
    if (i < q_pad && permQ[i] >= 0) {
        int p = blkclQ[blockIdx.x];
        float xq = sqrtf(nQ[i]) * 1.0000002f;
        ub = __uint_as_float(0x7f800000u);
        scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, nullptr, [&](int g, float v) {
            float e = group_eps<true>(i, g, xq, nullptr, gmax, n_groups, p, nullptr, q_pad, ex);
            ub = fminf(ub, v + e);
        });
    }
    if (u_row && i < q_pad) u_row[i] = ub;
    for (int o = 16; o > 0; o >>= 1) ub = fmaxf(ub, __shfl_xor(ub, o));
    if ((threadIdx.x & 31) == 0 && i < q_pad) u_rt[i / TILE] = ub;
    ub = fmaxf(ub, __shfl_xor(ub, 32));
    __shared__ float sh[BLOCK_ROWS / 64];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ub;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLOCK_ROWS / 64; ++w) ub = fmaxf(ub, sh[w]);
        u_rb[blockIdx.x] = ub;
    }
}
__global__ void col_u_kernel(const float* __restrict__ table, int n_rg, int t_pad, const int* __restrict__ permT,
                             const float* __restrict__ nT_sets, const float* __restrict__ gmaxA, const int* __restrict__ cl_of_rg,
                             const int* __restrict__ tile_group, EpsExtra ex, CompView comp, unsigned* __restrict__ u_leaf /* float bits, >= 0 */,
                             unsigned* __restrict__ u_stage /* [stages] or nullptr: the same maximum per 128-column stage */,
                             unsigned* __restrict__ u_ct /* [column tiles] or nullptr: per 32-column tile (plain store: one writer) */,
                             float* __restrict__ u_colv /* [columns] or nullptr: every column's own bound (0: padding) */) {
    extern __shared__ int list_s[];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    bool several;
    const int n_list = comp_list(comp, blockIdx.x * blockDim.x, t_pad, n_rg, list_s, 0, &several);
    float ub = 0.f;   // padding columns need nothing
    if (j < t_pad && permT[j] >= 0) {
        ub = __uint_as_float(0x7f800000u);
        scan_groups(table, (size_t) t_pad, j, n_list, n_rg, list_s, several ? comp_row(comp, j, 0) : nullptr, [&](int g, float v) {
            float e = group_eps<false>(j, g, 0.f, nT_sets, gmaxA, n_rg, 0, cl_of_rg, t_pad, ex);
            ub = fminf(ub, v + e);
        });
        ub = ub > 0.f ? ub : 0.f;
    }
    if (u_colv && j < t_pad) u_colv[j] = ub;
    // the 32 columns of a tile share a leaf: one atomic per tile (t_pad is a multiple of the block size, so whole waves get here)
    for (int o = 16; o > 0; o >>= 1) ub = fmaxf(ub, __shfl_xor(ub, o));
    if (u_ct && (threadIdx.x & 31) == 0 && j < t_pad) u_ct[j / TILE] = ub > 0.f ? __float_as_uint(ub) : 0u;
    if ((threadIdx.x & 31) == 0 && j < t_pad && ub > 0.f) {
        atomicMax(&u_leaf[tile_group[j / TILE]], __float_as_uint(ub));
        if (u_stage) atomicMax(&u_stage[j / STAGE_COLS], __float_as_uint(ub));
    }
}
// +inf for the table entries a masked pass is about to compute for the first time (the tables hold 10 GB at 1M x 1M and
// only a fifth of them is ever computed or read: no blanket fill).  Row table: (group of a newly scheduled leaf, the 256
// rows of the block).  Column table: the columns of the leaf in the block's row group, written by the lowest newly
// scheduled block of the group unless an earlier pass already computed that (row group, leaf).  Entries that only a
// boundary stage touches (a stage is computed when any leaf it overlaps is scheduled) may hold anything: nothing reads
// them until their own (block, leaf) is scheduled, and that initialises them here.
// From the schedule's side (round 5): a thread per (row block, leaf) pair finds out what its pair needs (most need nothing and are done), then the
// wave's lanes together write the entries of each pair that needs some -- 16 k workgroups that mostly return at once.  (Rounds 1-4: a workgroup per
// (row block, slice of 64 leaves) that scanned its slice, synchronised and walked it again: 63 k workgroups, 0.36 + 0.18 ms per pair at 1M against
// 2 x 0.16.)
__global__ __launch_bounds__(256) void init_tables_sparse_kernel(const uint8_t* __restrict__ sched, const uint8_t* __restrict__ done, int n_rb, int n_leaves,
                                                                 const int* __restrict__ leaf_g0 /* [n_leaves + 1] */, const int* __restrict__ group_start,
                                                                 int rg_blocks, int* __restrict__ rowmin, size_t ma_pad, int* __restrict__ colmin, size_t mb_pad) {
    static_assert(BLOCK_ROWS == 256, "a row block is one 16-byte store per lane");
    const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = idx < (size_t) n_rb * n_leaves;
    const int rb = in ? (int) (idx / n_leaves) : 0, l = in ? (int) (idx % n_leaves) : 0;
    int f = 0;
    if (in) {
        const uint8_t sv = sched[idx];
        f = ((sv & 1) && !(sv & SCHED_NOT_P0)) ? 1 : 0;   // rows of the block: whole-leaf tiles only; not again when pass 0 has minima in them
        if (sv && colmin) {
            const int rb_lo = (rb / rg_blocks) * rg_blocks, rb_hi = min(n_rb, rb_lo + rg_blocks);
            bool first = true;
            for (int r = rb_lo; r < rb_hi; ++r) {
                if (done[(size_t) r * n_leaves + l]) first = false;
                if (r < rb && sched[(size_t) r * n_leaves + l]) first = false;
            }
            if (first) f |= 2;
        }
    }
    unsigned long long m = __ballot(f != 0);
    const int lane = threadIdx.x & 63;
    constexpr int IINF = 0x7f800000;
    const int4 inf4 = make_int4(IINF, IINF, IINF, IINF);
    while (m) {
        const int src = __ffsll((long long) m) - 1;
        m &= m - 1ull;
        const int prb = __shfl(rb, src), pl = __shfl(l, src), pf = __shfl(f, src);
        const int g0 = leaf_g0[pl], g1 = leaf_g0[pl + 1];
        if (pf & 1)
            for (int g = g0; g < g1; ++g) reinterpret_cast<int4*>(rowmin + (size_t) g * ma_pad + (size_t) prb * BLOCK_ROWS)[lane] = inf4;
        if ((pf & 2) && g0 < g1) {
            const int rg = prb / rg_blocks;
            for (int col = group_start[g0] + lane; col < group_start[g1]; col += 64) colmin[(size_t) rg * mb_pad + col] = IINF;
        }
    }
}

// tile scheduling of one pass (section 3b).  sched_kernel: tiles of the previous pass become done; a tile not yet
// done is scheduled when  LBsq <= beta_sq * U  of its row block or (both directions) of its leaf.  mask_kernel turns the
// scheduled (row block, leaf) tiles into stage masks: a stage is computed when any leaf it overlaps is scheduled.
struct MaskStats { unsigned long long stages[8]; };
// Tile states: bit 0 = the whole (row block, leaf) tile is scheduled; bit 1 (col_partial, final pass only) = only the
// block's ROWS do not need the leaf, some of its COLUMNS may: mask_kernel then takes just the 128-column stages whose own
// columns ask for it (LB^2 <= the stage's largest U^2) instead of the whole leaf because of its worst column.
__device__ __forceinline__ bool col_stage_needed(float lbsq, unsigned ustage_bits) { return lbsq <= __uint_as_float(ustage_bits) * LB_GROW + 1e-12f; }
// prev_shell: the pass just finished was pass 0 with the shell selection -- its tiles are DONE_SHELL, not done, and a later pass that
// needs such a tile takes the rest of it (SCHED_NOT_P0)
__global__ void sched_kernel(int both, float beta_sq, const float* __restrict__ LBsq, const float* __restrict__ u_rb,
                             const unsigned* __restrict__ u_leaf, int n_rb, int n_leaves, int col_partial, int prev_shell,
                             uint8_t* __restrict__ done, uint8_t* __restrict__ sched) {
    const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t) n_rb * n_leaves) return;
    const int rb = (int) (idx / n_leaves), g = (int) (idx % n_leaves);
    uint8_t d = done[idx];
    if (sched[idx]) d |= prev_shell ? DONE_SHELL : 1;
    done[idx] = d;
    uint8_t s = 0;
    if (!(d & 1)) {
        float lb = LBsq[idx], urb = u_rb[rb];
        bool need = urb >= 0.f && lb <= beta_sq * (urb * LB_GROW + 1e-12f);
        s = need ? 1 : 0;
        if (both && !need) { float ug = __uint_as_float(u_leaf[g]); if (lb <= beta_sq * (ug * LB_GROW + 1e-12f)) s = col_partial ? 2 : 1; }
        if (s && (d & DONE_SHELL)) s |= SCHED_NOT_P0;
    }
    sched[idx] = s;
}
// one lane per (row block, chunk, stage): the 32 stages of a (row block, chunk) pair are the 32 lanes of a half wave, the mask is
// their ballot (one thread per pair walking its 32 stages was a chain of 128 dependent loads: 0.30 + 0.37 ms per pair at 1M)
// Shell bound (round 3, passes with upper bounds): rows and columns are sorted inside their leaves by their distance to the cluster
// centre (assign_kernel's key), so a row block and a column stage are thin radial shells [r0, r1] about the centre c of the row block's
// cluster -- the centre both are packed against, whose squared distances are the norms nA / nB[set].  By the reverse triangle inequality
// every pair has |a - b| >= | |a - c| - |b - c| | >= gap(shells): a stage of a scheduled leaf whose shell gap exceeds the upper bounds of
// the block's rows AND of the stage's columns holds no nearest neighbour and no tie of either, and is dropped.  (The radii of a cluster's
// rows spread over several times a typical nearest-neighbour distance: tools/exp_radial_bound.py.)
struct ShellArgs {
    const float2* rshA;           // [row blocks] shell (min, max |a - c|) of the block's valid rows about its own centre (min rounded down, max up)
    const float2* sshB;           // [KCL][column stages] shell of the stage's columns about every centre
    const int* blkcl;             // [row blocks]
    const float* u_rb;            // [row blocks] largest U^2 of the block's rows (< 0: none); nullptr in pass 0 (no bounds yet)
    int cols;                     // 0: row direction only; 1: u_stage holds the stages' largest column U^2
};
__global__ __launch_bounds__(256) void mask_kernel(int pass, const uint8_t* __restrict__ sched, const int* __restrict__ tile_group,
                                                   int n_rb, int n_cc, int n_leaves, int n_stage_total, const float* __restrict__ LBsq, const unsigned* __restrict__ u_stage,
                                                   ShellArgs sh, unsigned* __restrict__ mask, unsigned* __restrict__ mask_acc /* [n_rb][n_cc]: stages of all passes so far */,
                                                   MaskStats* __restrict__ stats) {
    static_assert(STAGES_PER_CHUNK == 32, "a half wave per (row block, chunk) pair");
    const long long n_pairs = (long long) n_rb * n_cc;
    const int s = threadIdx.x & 31;
    unsigned long long count = 0ull, fresh = 0ull;   // stages of this pass; those among them that no earlier pass computed
    // (the two half waves of a wave hold consecutive pairs.  The loop's trip count is the WAVE's: the upper half wave of the last round may
    // have no pair left and then runs the body predicated off -- the cross-half shuffle below never reads a lane that has left the loop)
    for (long long idx0 = (((long long) blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 1; idx0 < n_pairs; idx0 += ((long long) gridDim.x * blockDim.x) >> 5) {
        const long long idx = idx0 + ((threadIdx.x >> 5) & 1);
        const bool valid = idx < n_pairs;
        bool on = false;
        if (valid) {
            const int rb = (int) (idx / n_cc), cc = (int) (idx % n_cc);
            const int gst = cc * STAGES_PER_CHUNK + s;
            if (gst < n_stage_total) {
                // shell gap of (row block, stage) about the centre of the block's cluster (the norms are 33-term float sums: 2e-6 relative on
                // a radius; the shells are widened by 4e-6 of their radii); > 0: no pair of the two is closer than that
                float gap = 0.f;
                if (sh.rshA) {
                    const float2 sa = sh.rshA[rb], sb = sh.sshB[(size_t) sh.blkcl[rb] * n_stage_total + gst];
                    gap = fmaxf(sb.x - sa.y, sa.x - sb.y) - 4e-6f * (sa.y + sb.y);
                }
                const bool overlap = !(gap > 0.f);   // what pass 0 takes of a scheduled leaf when the shells are known
                int gprev = -1;
#pragma unroll
                for (int ct = 0; ct < STAGE_TILES; ++ct) {
                    const int g = tile_group[gst * STAGE_TILES + ct];
                    if (g == gprev) continue;
                    gprev = g;
                    const uint8_t sv = sched[(size_t) rb * n_leaves + g];
                    bool want = (sv & 1) || ((sv & 2) && col_stage_needed(LBsq[(size_t) rb * n_leaves + g], u_stage[gst]));
                    if (sh.rshA && !sh.u_rb) want = want && overlap;     // pass 0: a heuristic selection, it only looks for good upper bounds
                    if (sv & SCHED_NOT_P0) want = want && !overlap;       // ... and a later pass does not repeat it
                    on = on || want;
                }
                if (on && sh.u_rb && gap > 0.f && gap < FLT_BIG) {
                    const float lb = gap * gap * (LB_SHRINK * LB_SHRINK * LB_SHRINK);
                    const float urb = sh.u_rb[rb];
                    const bool rows_need = urb >= 0.f && lb <= urb * LB_GROW + 1e-12f;
                    const bool cols_need = sh.cols != 0 && col_stage_needed(lb, u_stage[gst]);
                    on = rows_need || cols_need;
                }
            }
        }
        const unsigned long long bal = __ballot(on);
        int fresh_here = 0;
        if (s == 0 && valid) {
            const unsigned m = (unsigned) (bal >> (threadIdx.x & 32));
            mask[idx] = m;
            // a stage that straddles two leaves is computed whole by every pass that schedules one of them: count it ONCE for the
            // executed fraction (lgr_match_last_work); the per-pass sums are the work that was really issued (lgr_match_last_issued)
            const unsigned before = mask_acc[idx];
            mask_acc[idx] = before | m;
            fresh_here = __popc(m & ~before);
        }
        fresh += (unsigned long long) (fresh_here + __shfl_xor(fresh_here, 32));
        count += (unsigned long long) __popcll(bal);
    }
    // one atomic per workgroup (a wave each was half a million same-address atomics)
    __shared__ unsigned long long cnt_s[4], fresh_s[4];
    if ((threadIdx.x & 63) == 0) { cnt_s[threadIdx.x >> 6] = count; fresh_s[threadIdx.x >> 6] = fresh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long c = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3], f = fresh_s[0] + fresh_s[1] + fresh_s[2] + fresh_s[3];
        if (c) atomicAdd(&stats->stages[pass], c);
        if (f) atomicAdd(&stats->stages[7], f);   // (passes use slots 0 .. 6 at most: LGR_PRUNE_BETAS)
    }
}

// The same masks from the SCHEDULE's side (round 5).  mask_kernel above evaluates every (row block, chunk, stage) -- 31 M lanes at 1M rows, each with
// its shell pair and up to four schedule bytes -- although 4 % (pass 0) or 15 % (final pass) of the (row block, leaf) pairs are scheduled at all.
// Here a thread takes one (row block, leaf) pair, leaves at once when it is not scheduled, and otherwise walks the leaf's own stages (a leaf is a
// contiguous run of tiles: leaf_stage_range_kernel) with exactly mask_kernel's per-leaf test and per-stage shell refinement, OR-ing the stages it
// wants into the (zeroed) mask words.  A stage shared by two leaves is the OR of their tests in both kernels, and the refinement depends on
// (row block, stage) only, so the masks are identical bit for bit (lgr_match_options.self_check compares them).  mask_stats_kernel then does
// mask_kernel's bookkeeping over the finished words.
__global__ void leaf_stage_range_kernel(const int* __restrict__ tile_leaf, int n_tiles, int n_leaves, int* __restrict__ first_stage /* 0x7f7f7f7f */, int* __restrict__ last_stage /* -1 */) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int g = tile_leaf[t];
    if (g < 0 || g >= n_leaves) return;
    // (a leaf is ONE run of tiles: its first and its last tile write, nobody else -- 31 k atomics on 2 k words took 44 us in front of pass 0)
    if (t == 0 || tile_leaf[t - 1] != g) first_stage[g] = t / STAGE_TILES;
    if (t == n_tiles - 1 || tile_leaf[t + 1] != g) last_stage[g] = t / STAGE_TILES;
}
__global__ __launch_bounds__(256) void mask_sparse_kernel(const uint8_t* __restrict__ sched, const int* __restrict__ first_stage, const int* __restrict__ last_stage,
                                                          int n_rb, int n_cc, int n_leaves, int n_stage_total, const float* __restrict__ LBsq, const unsigned* __restrict__ u_stage,
                                                          ShellArgs sh, unsigned* __restrict__ mask /* zeroed */) {
    const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t) n_rb * n_leaves) return;
    const uint8_t sv = sched[idx];
    if (!(sv & 3)) return;
    const int rb = (int) (idx / n_leaves), g = (int) (idx % n_leaves);
    const int s0 = first_stage[g], s1 = min(last_stage[g], n_stage_total - 1);
    float2 sa = make_float2(0.f, 0.f);
    const float2* sshB = nullptr;
    if (sh.rshA) { sa = sh.rshA[rb]; sshB = sh.sshB + (size_t) sh.blkcl[rb] * n_stage_total; }
    const float lbsq = LBsq[idx];
    const float urb = sh.u_rb ? sh.u_rb[rb] : -1.f;
    for (int gst = s0; gst <= s1; ++gst) {
        float gap = 0.f;
        if (sh.rshA) { const float2 sb = sshB[gst]; gap = fmaxf(sb.x - sa.y, sa.x - sb.y) - 4e-6f * (sa.y + sb.y); }
        const bool overlap = !(gap > 0.f);
        bool on = (sv & 1) || ((sv & 2) && col_stage_needed(lbsq, u_stage[gst]));
        if (sh.rshA && !sh.u_rb) on = on && overlap;
        if (sv & SCHED_NOT_P0) on = on && !overlap;
        if (on && sh.u_rb && gap > 0.f && gap < FLT_BIG) {
            const float lb = gap * gap * (LB_SHRINK * LB_SHRINK * LB_SHRINK);
            const bool rows_need = urb >= 0.f && lb <= urb * LB_GROW + 1e-12f;
            const bool cols_need = sh.cols != 0 && col_stage_needed(lb, u_stage[gst]);
            on = rows_need || cols_need;
        }
        if (on) atomicOr(&mask[(size_t) rb * n_cc + (gst / STAGES_PER_CHUNK)], 1u << (gst % STAGES_PER_CHUNK));
    }
}
__global__ void mask_compare_kernel(const unsigned* __restrict__ a, const unsigned* __restrict__ b, long long n, unsigned* __restrict__ n_diff) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(n_diff, 1u);
}
__global__ __launch_bounds__(256) void mask_stats_kernel(int pass, const unsigned* __restrict__ mask, unsigned* __restrict__ mask_acc, long long n_pairs, MaskStats* __restrict__ stats) {
    unsigned long long count = 0ull, fresh = 0ull;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (long long) gridDim.x * blockDim.x) {
        const unsigned m = mask[i];
        if (!m) continue;
        const unsigned before = mask_acc[i];
        mask_acc[i] = before | m;
        count += (unsigned long long) __popc(m);
        fresh += (unsigned long long) __popc(m & ~before);
    }
    for (int o = 32; o > 0; o >>= 1) { count += __shfl_xor(count, o); fresh += __shfl_xor(fresh, o); }
    __shared__ unsigned long long cnt_s[4], fresh_s[4];
    if ((threadIdx.x & 63) == 0) { cnt_s[threadIdx.x >> 6] = count; fresh_s[threadIdx.x >> 6] = fresh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long c = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3], f = fresh_s[0] + fresh_s[1] + fresh_s[2] + fresh_s[3];
        if (c) atomicAdd(&stats->stages[pass], c);
        if (f) atomicAdd(&stats->stages[7], f);
    }
}

// Self-check of the filter bound (lgr_match_options.self_check; tests, up to 1M x 1M): for sampled queries and every computed group,
// |filtered minimum - exact minimum of the squared distance (double)| / eps, maximised through atomicMax on the float
// bits.  eps is a proven bound, so the ratio must stay <= 1; tests assert it on both operand formats.
template <bool ROWDIR>
__global__ void check_kernel(const float* __restrict__ table, int n_groups, int q_pad, int group_size, const int* __restrict__ starts,
                             const float* __restrict__ Qsorted, const int* __restrict__ permQ, const float* __restrict__ Tsorted,
                             const int* __restrict__ permT, int t_pad, const float* __restrict__ nQ, const int* __restrict__ blkclQ,
                             const float* __restrict__ nQ_sets, const float* __restrict__ gmax, const int* __restrict__ cl_of_group,
                             EpsExtra ex, CompView comp, int stride, const uint8_t* __restrict__ done, const uint8_t* __restrict__ sched,
                             int n_leaves, const float* __restrict__ LBsq, const unsigned* __restrict__ u_stage,
                             const float* __restrict__ uq_rows /* coarse rejection: u_row (per row), or nullptr */,
                             const float* __restrict__ uq_cols /* coarse rejection: u_colv (per column), or nullptr */, unsigned* __restrict__ worst) {
    extern __shared__ int list_s[];
    const int i = blockIdx.x * stride;   // sampled padded query position
    // the computed groups of the row block (rows) / of the leaf (columns) this query lives in
    const int n_list = comp_list(comp, ROWDIR ? (i / BLOCK_ROWS) * BLOCK_ROWS : (i / TILE) * TILE, q_pad, n_groups, list_s, ROWDIR ? BLOCK_ROWS : TILE);
    // columns: a row group is computed row block by row block; the table holds the minimum over the computed ones only
    const int my_leaf = (!ROWDIR && done && i < q_pad) ? comp.row_of_tile[i / TILE] : -1;
    if (i >= q_pad || permQ[i] < 0) return;
    const int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    const float xq = ROWDIR ? sqrtf(nQ[i]) * 1.0000002f : 0.f;
    float q[33];
    for (int k = 0; k < 33; ++k) q[k] = Qsorted[(size_t) i * 33 + k];
    for (int kk = 0; kk < (n_list < 0 ? n_groups : n_list); ++kk) {
        const int g = n_list < 0 ? kk : list_s[kk];
        const float v = table[(size_t) g * q_pad + i];
        const int j0 = starts ? starts[g] : g * group_size, j1 = starts ? starts[g + 1] : min(t_pad, j0 + group_size);
        // best: exact minimum over the rows the entry is GUARANTEED to cover; best_all: over every row that MAY have reached it.
        // Columns: a row block covers the column for sure when (row block, the column's own leaf) was computed for the
        // column's stage; a stage that straddles two leaves is computed whole, so row blocks scheduled only for the OTHER leaf
        // reach the column too -- their minima land in the entry when it is already initialised and are lost otherwise
        // (extra information either way: the rerank only needs the guaranteed rows).
        double best = 1e300, best_all = 1e300;
        for (int j = j0 + (int) threadIdx.x; j < j1; j += blockDim.x) {
            if (permT[j] < 0) continue;
            bool sure = true;
            if (my_leaf >= 0) {
                const int gst = i / STAGE_COLS;
                bool on = false;
                sure = false;
                int gprev = -1;
                for (int ct = 0; ct < STAGE_TILES; ++ct) {
                    const int gl = comp.row_of_tile[gst * STAGE_TILES + ct];
                    if (gl == gprev || gl < 0) continue;
                    gprev = gl;
                    const size_t t = (size_t) (j / BLOCK_ROWS) * n_leaves + gl;
                    const uint8_t sv = done[t] | sched[t];
                    const bool c = (sv & 1) || ((sv & 2) && u_stage && col_stage_needed(LBsq[t], u_stage[gst]));   // mask_kernel's rule
                    on = on || c || (sv & DONE_SHELL);   // (pass 0's shell selection may have reached the column: not guaranteed)
                    if (gl == my_leaf) sure = c;
                }
                if (!on) continue;
            }
            double d = 0;
            for (int k = 0; k < 33; ++k) { double t = (double) q[k] - (double) Tsorted[(size_t) j * 33 + k]; d += t * t; }
            best_all = d < best_all ? d : best_all;
            if (sure) best = d < best ? d : best;
        }
        for (int o = 32; o > 0; o >>= 1) {
            double other = __shfl_xor(best, o); best = other < best ? other : best;
            other = __shfl_xor(best_all, o); best_all = other < best_all ? other : best_all;
        }
        __shared__ double sh[4], sh_all[4];
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = best; sh_all[threadIdx.x >> 6] = best_all; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int) (blockDim.x >> 6); ++w) { best = sh[w] < best ? sh[w] : best; best_all = sh_all[w] < best_all ? sh_all[w] : best_all; }
            if (best_all < 1e299) {   // rows reached the entry
                float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
                // upper side (entry <= guaranteed minimum + eps, and finite): required unless the coarse rejection may have
                // left the minimum out -- it abandons tiles whose elements all lie above the U^2 of their rows and columns, so
                // an entry may exceed the exact minimum (or stay +inf) when that minimum is above the query's own U^2 (u_rb of
                // its row block / u_stage of its column stage).  Lower side (entry >= minimum over all rows - eps): always.
                bool upper = best < 1e299;
                if (ROWDIR && uq_rows) upper = upper && best <= (double) uq_rows[i];
                if (!ROWDIR && uq_cols) upper = upper && best <= (double) uq_cols[i];
                float ratio = 0.f;
                if (v < FLT_BIG) {
                    ratio = (float) (fmax(best_all - (double) v, 0.0) / (double) e);
                    if (upper) ratio = fmaxf(ratio, (float) (fmax((double) v - best, 0.0) / (double) e));
                } else if (upper) ratio = 1e30f;
                atomicMax(worst, __float_as_uint(ratio));
            }
        }
    }
}

constexpr int CAND_KEEP = 4;   // smallest lower bounds kept per query by rerank_count (up to CAND_KEEP - 1 candidates without a rescan)
template <bool ROWDIR>
__global__ void rerank_count(const float* __restrict__ table, int n_groups, int q_pad, const int* __restrict__ permQ,
                             const float* __restrict__ nQ /* ROWDIR: |a'|^2 per padded row */, const int* __restrict__ blkclQ,
                             const float* __restrict__ nQ_sets /* COLDIR: |b - c_p|^2 [KCL][q_pad] */, const float* __restrict__ gmax,
                             const int* __restrict__ cl_of_group, int dense_limit /* < 0: every query takes the dense path */, EpsExtra ex, CompView comp,
                             float* __restrict__ thr_out, int* __restrict__ counts, int* __restrict__ cand /* [q_pad][CAND_KEEP] */,
                             unsigned* __restrict__ dense, RerankCounters* __restrict__ cnt) {
    extern __shared__ int list_s[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool several;
    const int n_list = comp_list(comp, blockIdx.x * blockDim.x, q_pad, n_groups, list_s, 0, &several);
    if (i >= q_pad) return;
    counts[i] = 0;
    int o = permQ[i];
    if (o < 0) return;
    if (dense_limit < 0) {   // the filter is not usable for this call (a centred norm overflows float): exact brute force
        unsigned pos = atomicAdd(&cnt->n_dense, 1u);
        dense[pos] = (unsigned) o;
        return;
    }
    int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    float nq = ROWDIR ? nQ[i] : 0.f;
    float xq = ROWDIR ? sqrtf(nq) * 1.0000002f : 0.f;
    // one scan: the smallest upper bound, and the CAND_KEEP smallest lower bounds with their groups.  Candidates are the
    // groups whose lower bound does not exceed thr (derived from the smallest upper bound); there is about one per query,
    // so they are almost always among the kept ones and neither a second scan here nor one in rerank_emit is needed.
    float ub = __uint_as_float(0x7f800000u);
    float lo[CAND_KEEP];
    int lg[CAND_KEEP];
#pragma unroll
    for (int j = 0; j < CAND_KEEP; ++j) { lo[j] = __uint_as_float(0x7f800000u); lg[j] = -1; }
    const uint8_t* own = (ROWDIR || !several) ? nullptr : comp_row(comp, i, 0);   // the list is exact unless the block spans two leaves
    scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, own, [&](int g, float v) {
        float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
        ub = fminf(ub, v + e);
        float l = v - e;
        int gi = g;
        if (l < lo[CAND_KEEP - 1]) {
#pragma unroll
            for (int j = 0; j < CAND_KEEP; ++j)
                if (l < lo[j]) { float tl = lo[j]; int tg = lg[j]; lo[j] = l; lg[j] = gi; l = tl; gi = tg; }
        }
    });
    if (!(ub < FLT_BIG)) return;     // no valid train row at all
    double d2 = fmax((double) ub, 0.0);   // both tables hold d2~ = S + |a'|^2
    float thr = (float) ((double) ub + 1e-5 * d2 + 8.0 * 5.9604644775390625e-8 * fabs((double) ub) + 1e-30);
    if (thr < ub) thr = ub;
    int nc = 0;
    if (lo[CAND_KEEP - 1] <= thr) {
        // the kept list may be incomplete: count by a second scan, rerank_emit rescans too (cand[0] = -1)
        scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, own, [&](int g, float v) {
            float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
            nc += (v - e <= thr) ? 1 : 0;
        });
        cand[(size_t) i * CAND_KEEP] = -1;
    } else {
#pragma unroll
        for (int j = 0; j < CAND_KEEP - 1; ++j)
            if (lo[j] <= thr) { cand[(size_t) i * CAND_KEEP + nc] = lg[j]; ++nc; }
    }
    if (nc > dense_limit) {
        unsigned pos = atomicAdd(&cnt->n_dense, 1u);
        dense[pos] = (unsigned) o;
        return;
    }
    thr_out[i] = thr;
    counts[i] = nc;
}

template <bool ROWDIR>
__global__ void rerank_emit(const float* __restrict__ table, int n_groups, int q_pad, const float* __restrict__ nQ,
                            const int* __restrict__ blkclQ, const float* __restrict__ nQ_sets, const float* __restrict__ gmax,
                            const int* __restrict__ cl_of_group, EpsExtra ex, CompView comp, const float* __restrict__ thr_in,
                            const int* __restrict__ counts, const int* __restrict__ cand, const int* __restrict__ offs,
                            unsigned* __restrict__ item_q, unsigned* __restrict__ item_g) {
    extern __shared__ int list_s[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool several;
    const int n_list = comp_list(comp, blockIdx.x * blockDim.x, q_pad, n_groups, list_s, 0, &several);
    if (i >= q_pad || counts[i] == 0) return;
    int p = ROWDIR ? blkclQ[i / BLOCK_ROWS] : 0;
    float xq = ROWDIR ? sqrtf(nQ[i]) * 1.0000002f : 0.f;
    float thr = thr_in[i];
    int pos = offs[i];
    if (cand[(size_t) i * CAND_KEEP] >= 0) {   // the candidates rerank_count kept
        for (int j = 0; j < counts[i]; ++j) { item_q[pos] = (unsigned) i; item_g[pos] = (unsigned) cand[(size_t) i * CAND_KEEP + j]; ++pos; }
        return;
    }
    scan_groups(table, (size_t) q_pad, i, n_list, n_groups, list_s, (ROWDIR || !several) ? nullptr : comp_row(comp, i, 0), [&](int g, float v) {
        float e = group_eps<ROWDIR>(i, g, xq, nQ_sets, gmax, n_groups, p, cl_of_group, q_pad, ex);
        if (v - e <= thr) { item_q[pos] = (unsigned) i; item_g[pos] = (unsigned) g; ++pos; }
    });
}

// 4b. exact distances of the (query position, train group) items, sorted by group: a workgroup takes 256 consecutive
// items (almost always one group) and every thread scans the group's train rows for its own query with the canonical
// distance.  The train row address is wave uniform (made explicit with readfirstlane), so the rows arrive through the
// scalar cache as SGPR operands of the VALU ops: no vector loads, no LDS in the inner loop.
constexpr int RQ_THREADS = 256;
__global__ __launch_bounds__(RQ_THREADS) void rerank_grouped(const float* __restrict__ Q, const int* __restrict__ permQ,
                                                             const float* __restrict__ Tsorted, const int* __restrict__ permT, int t_pad,
                                                             int group_size, const int* __restrict__ starts /* variable groups, or nullptr */,
                                                             int block, int nblocks, const unsigned* __restrict__ item_g,
                                                             const unsigned* __restrict__ item_q, unsigned n_items,
                                                             unsigned long long* __restrict__ best) {
    __shared__ unsigned next_g;
    const int tid = threadIdx.x;
    const unsigned idx = blockIdx.x * RQ_THREADS + tid;
    const bool act = idx < n_items;
    const unsigned g = act ? item_g[idx] : 0xffffffffu;
    const int qo = act ? permQ[item_q[idx]] : -1;
    float q[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) q[k] = act ? Q[(size_t) qo * 33 + k] : 0.f;
    unsigned long long bk = ~0ull;
    unsigned cur = item_g[blockIdx.x * RQ_THREADS];   // items are sorted: the first one has the smallest group
    while (cur != 0xffffffffu) {
        const int j0 = __builtin_amdgcn_readfirstlane(starts ? starts[cur] : (int) cur * group_size);
        const int j1 = __builtin_amdgcn_readfirstlane(starts ? starts[cur + 1] : min(t_pad, j0 + group_size));
        // waves without an item of this group skip it (wave-uniform branch)
        if (__ballot(act && g == cur) != 0ull) {
            for (int j = j0; j < j1; ++j) {
                const int to = __builtin_amdgcn_readfirstlane(permT[j]);
                if (to < 0) continue;                // padding
                const float* __restrict__ tp = Tsorted + (size_t) j * 33;   // wave-uniform address -> scalar loads
                float t[33];
#pragma unroll
                for (int k = 0; k < 33; ++k) t[k] = tp[k];
                float d = exact_l2(q, t);
                if (act && g == cur && d < FLT_BIG) {   // batchDistance keeps only d < FLT_MAX
                    unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(to, block, nblocks);
                    bk = key < bk ? key : bk;
                }
            }
        }
        __syncthreads();
        if (tid == 0) next_g = 0xffffffffu;
        __syncthreads();
        if (act && g > cur) atomicMin(&next_g, g);
        __syncthreads();
        cur = next_g;
    }
    if (act && bk != ~0ull) atomicMin(&best[qo], bk);
}

// 4b'. MFMA re-filter of the items (f16 operand formats).  rerank_grouped computes the exact distance to every train row of
// an item's group (~660 per query at 1M) to find one neighbour.  The packed operands can say which of those rows matter: a
// wave takes 32 consecutive items, gathers the queries' operand fragments into one MFMA tile and multiplies it with the
// group's train tiles; an element (query, train row) goes on to the exact distance only when  filtered - eps <= thr  -- the
// very criterion that made the GROUP a candidate, applied per element (|filtered - d2| <= eps holds per element), so the
// nearest neighbour and every tie pass it.  Output: (query position, train position) pairs for rerank_pairs.
// ROWDIR: queries are rows (gathered A fragments), trains the group's column tiles from the B set of the queries' cluster;
// the accumulator holds 16 queries per lane.  COLDIR: queries are columns (gathered B fragments of the row group's set),
// trains the row tiles of the group; the accumulator holds one query per lane.  Items are sorted by group and, inside a
// group, by query position (clusters are contiguous), so a batch of 32 items is a few runs of equal (group, cluster).
struct RefilterArgs {
    const f16x8* Ap; const f16x8* Bp; size_t bset_stride; float out_scale; int ks;   // ks = 6 (rotated) or 7; 0 = re-filter off
    const int* blkclA;    // cluster of every 256-row block of A (the set a row / a row group is computed against)
    int pair_cap;         // >= 0: upper limit of the pair buffer (env LGR_MATCH_PAIR_CAP, tests)
};
constexpr int RF_THREADS = 256;
#ifndef RF_OCC
#define RF_OCC 4   // waves per SIMD of rerank_refilter (round 5: the allocator took 154 VGPRs and two; the kernel waits on barriers and tile loads 64 % of its wave cycles)
#endif
template <bool ROWDIR, int KS>
__global__ __launch_bounds__(RF_THREADS) __attribute__((amdgpu_waves_per_eu(RF_OCC, RF_OCC))) void rerank_refilter(RefilterArgs ra, const unsigned* __restrict__ item_g, const unsigned* __restrict__ item_q, unsigned n_items,
                                                              int n_groups, int q_pad, int t_pad, int group_size, const int* __restrict__ starts,
                                                              const float* __restrict__ nQ, const int* __restrict__ blkclQ, const float* __restrict__ nQ_sets,
                                                              const float* __restrict__ gmax, const int* __restrict__ cl_of_group, EpsExtra ex,
                                                              const float* __restrict__ thr_in, unsigned cap, unsigned* __restrict__ n_pairs,
                                                              unsigned* __restrict__ pair_q, unsigned* __restrict__ pair_t) {
    __shared__ float thr_s[RF_THREADS / 64][32], eps_s[RF_THREADS / 64][32];
    __shared__ int q_s[RF_THREADS / 64][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, c = lane & 31;
    // passing pairs are collected per wave in LDS and leave with one atomic per flush (an atomic per pair on the one global
    // counter serialises in L2: 12 ms for 3 M pairs)
    constexpr int RF_BUF = 256;
    __shared__ unsigned buf_q[RF_THREADS / 64][RF_BUF], buf_t[RF_THREADS / 64][RF_BUF];
    int n_buf = 0;   // wave uniform
    auto flush = [&]() {
        if (n_buf == 0) return;
        unsigned pos0 = 0;
        if (lane == 0) pos0 = atomicAdd(n_pairs, (unsigned) n_buf);
        pos0 = (unsigned) __builtin_amdgcn_readfirstlane((int) pos0);
        for (int e = lane; e < n_buf; e += 64)
            if (pos0 + e < cap) { pair_q[pos0 + e] = buf_q[wave][e]; pair_t[pos0 + e] = buf_t[wave][e]; }
        n_buf = 0;
    };
    // The four waves of a workgroup hold 4 x 32 consecutive items, which mostly belong to one or two (group, cluster) runs: the
    // run is chosen for the whole workgroup and its train tiles go through LDS once for all four waves (every wave fetching
    // its own copy made the kernel a re-reader of operand tiles: 5-6 GB per launch at 1M).  A wave without items in the run
    // only joins the barriers.
    __shared__ __attribute__((aligned(16))) f16x8 tile_s[KS * 64];
    __shared__ unsigned long long runkey_s[RF_THREADS / 64];
    // XCD-aware block order (round 5): the items are sorted by group, a group's ~700 items are five or six consecutive blocks -- which the
    // hardware spreads over as many XCDs, each L2 fetching the group's 192 KB of fragments for itself (2.2 GB of HBM-side traffic per launch at
    // 3.2 TB/s: the kernel's bound).  XCD x takes the contiguous block range [x per, (x + 1) per): a group's blocks run back to back on one L2.
    const int n_blk = (int) ((n_items + 32u * (RF_THREADS / 64) - 1u) / (32u * (RF_THREADS / 64)));
    const int blk = (int) (blockIdx.x & 7u) * ((n_blk + 7) >> 3) + (int) (blockIdx.x >> 3);
    if (blk >= n_blk) return;
    const unsigned base = ((unsigned) blk * (RF_THREADS / 64) + wave) * 32u;
    // lane c (both halves) owns item base + c
    const bool have = base + c < n_items;
    const unsigned my_g = have ? item_g[base + c] : 0xffffffffu;
    const int my_q = have ? (int) item_q[base + c] : 0;
    const int my_p = !have ? -1 : ROWDIR ? blkclQ[my_q / BLOCK_ROWS] : cl_of_group[my_g];
    float my_e = 0.f, my_thr = 0.f;
    if (have) {
        const float xq = ROWDIR ? sqrtf(nQ[my_q]) * 1.0000002f : 0.f;
        my_e = group_eps<ROWDIR>(my_q, (int) my_g, xq, nQ_sets, gmax, n_groups, my_p, cl_of_group, q_pad, ex);
        my_thr = thr_in[my_q];
    }
    const unsigned long long my_key = have ? (((unsigned long long) my_g << 8) | (unsigned) my_p) : ~0ull;
    bool todo = have;
    constexpr int PIECES = KS * 64;   // 16-byte pieces of a train tile
    for (;;) {
        // the run of this round: all pending items of the workgroup with the smallest pending (group, cluster)
        {
            const unsigned long long pend = __ballot(todo);
            unsigned long long first_key = ~0ull;
            if (pend) {   // items are sorted: the first pending lane holds the wave's smallest key
                const int first = __ffsll((long long) pend) - 1;
                const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) my_key, first);
                const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (my_key >> 32), first);
                first_key = ((unsigned long long) hi << 32) | lo;
            }
            __syncthreads();   // the previous round's readers of runkey_s / tile_s are done
            if (lane == 0) runkey_s[wave] = first_key;
            __syncthreads();
        }
        unsigned long long rk = runkey_s[0];
#pragma unroll
        for (int w = 1; w < RF_THREADS / 64; ++w) rk = runkey_s[w] < rk ? runkey_s[w] : rk;
        if (rk == ~0ull) break;
        const unsigned g = (unsigned) (rk >> 8);
        const int p = (int) (rk & 0xffull);
        const bool act = todo && my_key == rk;
        todo = todo && !act;
        const bool wave_on = __ballot(act) != 0ull;
        f16x8 qf[KS];
        // Pre-test of a tile (round 5): about one pair per query passes in the whole launch, so nearly every 32 x 32 tile ends with nothing to emit --
        // but the exact test below is sixteen multiply / subtract / compare / ballot / branch groups per tile, the bulk of the kernel's vector
        // instructions.  bound = (thr + eps) / out_scale, inflated by 1e-6 (its own roundings and the exact test's are worth 3e-7): whatever the
        // exact test passes has acc <= bound, so a tile whose smallest acc - bound is positive in every lane is left at once.
        float bnd_r[16];
        float bnd_lane = 0.f;
        if (wave_on) {
            if (ROWDIR) {   // thresholds by accumulator row: inactive rows can never pass
                if (half == 0) { thr_s[wave][c] = act ? my_thr : -__uint_as_float(0x7f800000u); eps_s[wave][c] = act ? my_e : 0.f; q_s[wave][c] = my_q; }
            }
            // the queries' fragments: lane (c, half) reads the 8 halves of its query for every K step
            const f16x8* src = ROWDIR ? ra.Ap : ra.Bp + (size_t) p * ra.bset_stride;
            const size_t o = ((size_t) (my_q >> 5) * KS) * 64 + (my_q & 31) + 32 * half;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) qf[kk] = src[o + (size_t) kk * 64];
            auto inflate = [&](float t) {   // (thr + eps) / out_scale, rounded up; -inf stays -inf, NaN keeps the tile
                const float b = t / ra.out_scale;
                return b >= 0.f ? b * 1.000001f + 1e-30f : (b < 0.f ? b * 0.999999f + 1e-30f : __uint_as_float(0x7f800000u));
            };
            if (ROWDIR) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    bnd_r[r] = inflate(thr_s[wave][row] + eps_s[wave][row]);
                }
            } else bnd_lane = act ? inflate(my_thr + my_e) : -__uint_as_float(0x7f800000u);
        }
        const int j0 = starts ? starts[g] : (int) g * group_size, j1 = starts ? starts[g + 1] : min(t_pad, j0 + group_size);
        const f16x8* tsrc = ROWDIR ? ra.Bp + (size_t) p * ra.bset_stride : ra.Ap;
        // train tiles: fetched by the whole workgroup (the next one into registers while the current one is consumed), one LDS copy
        f16x8 nx0, nx1;
        auto fetch = [&](int t0) {
            const f16x8* tb = tsrc + (size_t) (t0 >> 5) * KS * 64;
            nx0 = tb[threadIdx.x];
            if ((int) threadIdx.x + RF_THREADS < PIECES) nx1 = tb[threadIdx.x + RF_THREADS];
        };
        if (j0 < j1) fetch(j0);
        for (int t0 = j0; t0 < j1; t0 += TILE) {
            __syncthreads();   // the previous tile has been read by every wave
            tile_s[threadIdx.x] = nx0;
            if ((int) threadIdx.x + RF_THREADS < PIECES) tile_s[threadIdx.x + RF_THREADS] = nx1;
            __syncthreads();
            if (t0 + TILE < j1) fetch(t0 + TILE);
            if (!wave_on) continue;
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const f16x8 tf = tile_s[kk * 64 + lane];
                acc = ROWDIR ? mfma_step(qf[kk], tf, acc) : mfma_step(tf, qf[kk], acc);
            }
            // ROWDIR: acc[r] = (query row (r&3) + 8 (r>>2) + 4 half, train column c); COLDIR: (train row ..., query column c)
            {
                float dmin;
                if (ROWDIR) {
                    dmin = acc[0] - bnd_r[0];
#pragma unroll
                    for (int r = 1; r < 16; r += 3) dmin = fminf(fminf(dmin, acc[r] - bnd_r[r]), fminf(acc[r + 1] - bnd_r[r + 1], acc[r + 2] - bnd_r[r + 2]));
                } else {
                    dmin = acc[0];
#pragma unroll
                    for (int r = 1; r < 16; r += 3) dmin = fminf(fminf(dmin, acc[r]), fminf(acc[r + 1], acc[r + 2]));
                    dmin = dmin - bnd_lane;
                }
                if (__ballot(!(dmin > 0.f)) == 0ull) continue;   // (NaN keeps the tile)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r] * ra.out_scale;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool pass = ROWDIR ? (v - eps_s[wave][row] <= thr_s[wave][row]) : (act && v - my_e <= my_thr);
                const unsigned long long bal = __ballot(pass);
                if (bal != 0ull) {   // wave uniform
                    if (pass) {
                        const int slot = n_buf + __popcll(bal & ((1ull << lane) - 1ull));
                        buf_q[wave][slot] = ROWDIR ? (unsigned) q_s[wave][row] : (unsigned) my_q;
                        buf_t[wave][slot] = ROWDIR ? (unsigned) (t0 + c) : (unsigned) (t0 + row);
                    }
                    n_buf += __popcll(bal);
                    if (n_buf > RF_BUF - 64) flush();
                }
            }
        }
    }
    flush();
}
// exact distances of the re-filtered pairs (a thread per pair; same key as rerank_grouped)
__global__ void rerank_pairs(const float* __restrict__ Q, const int* __restrict__ permQ, const float* __restrict__ Tsorted, const int* __restrict__ permT,
                             int block, int nblocks, const unsigned* __restrict__ pair_q, const unsigned* __restrict__ pair_t, unsigned n_pairs,
                             unsigned long long* __restrict__ best) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const int qo = permQ[pair_q[i]];
    const int j = (int) pair_t[i], to = permT[j];
    if (qo < 0 || to < 0) return;
    float q[33], t[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) { q[k] = Q[(size_t) qo * 33 + k]; t[k] = Tsorted[(size_t) j * 33 + k]; }
    const float d = exact_l2(q, t);
    if (d < FLT_BIG) atomicMin(&best[qo], ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(to, block, nblocks));
}

// 4c. dense fallback (degenerate data: more than half of all groups qualify, e.g. huge sets of identical rows):
// plain exact brute force over the original train rows, parallel over (256 dense queries) x (column chunk).
constexpr int DENSE_CHUNK = 8192;
__global__ __launch_bounds__(256) void rerank_dense(const float* __restrict__ Q, const float* __restrict__ T,
                                                    const uint8_t* __restrict__ validT, int nt, int block, int nblocks,
                                                    const unsigned* __restrict__ dense, unsigned n_dense,
                                                    unsigned long long* __restrict__ best) {
    __shared__ float Ts[64 * 33];
    __shared__ uint8_t vTs[64];
    int c0 = blockIdx.y * DENSE_CHUNK, c1 = min(nt, c0 + DENSE_CHUNK);
    for (unsigned base = blockIdx.x * 256; base < n_dense; base += gridDim.x * 256) {
        unsigned di = base + threadIdx.x;
        bool act = di < n_dense;
        unsigned qi = act ? dense[di] : 0;
        float q[33];
#pragma unroll
        for (int k = 0; k < 33; ++k) q[k] = act ? Q[(size_t) qi * 33 + k] : 0.f;
        unsigned long long bk = ~0ull;
        for (int j0 = c0; j0 < c1; j0 += 64) {
            __syncthreads();
            int nj = min(64, c1 - j0);
            for (int i = threadIdx.x; i < nj * 33; i += 256) Ts[i] = T[(size_t) j0 * 33 + i];
            if (threadIdx.x < nj) vTs[threadIdx.x] = validT[j0 + threadIdx.x];
            __syncthreads();
            if (act) {
                for (int jj = 0; jj < nj; ++jj) {
                    if (!vTs[jj]) continue;
                    float d = exact_l2(q, Ts + jj * 33);
                    if (!(d < FLT_BIG)) continue;
                    unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(j0 + jj, block, nblocks);
                    bk = key < bk ? key : bk;
                }
            }
        }
        if (act && bk != ~0ull) atomicMin(&best[qi], bk);
        __syncthreads();
    }
}

// 4d. irregular rows (lgr_match_cluster.cuh, km_consensus): they are in no operand set and no table, so every pair they are part of is
// computed here -- the listed rows of one set (a few dozen of a million; at most IRR_CAP) against every finite row of the other, exact
// distances folded into the same packed keys as every other exact path, for BOTH roles at once: thread j holds row x_j of the full set and
// keeps the best listed row for it (best_x: the listed rows as train rows), and the wave folds its 64 distances to listed row y_i into one
// atomicMin on best_y[y_i] (the listed rows as queries).  (rerank_dense, the path of queries that scan everything, runs a thread per QUERY:
// 161 queries of the planar scene kept two waves per workgroup busy for 5.8 ms; here the million rows are the parallel dimension: 0.5 ms.)
__global__ __launch_bounds__(256) void irregular_scan(const float* __restrict__ X, const uint8_t* __restrict__ validX, int nx, const float* __restrict__ Y,
                                                      const int* __restrict__ irr_list, int n_irr, int block, int nblocks_x, int nblocks_y,
                                                      unsigned long long* __restrict__ best_x /* [nx] keys over Y's indices, or nullptr */,
                                                      unsigned long long* __restrict__ best_y /* [ny] keys over X's indices, or nullptr */) {
    __shared__ float Ys[64 * 33];
    __shared__ int Yj[64];
    const int xj = blockIdx.x * 256 + threadIdx.x;
    const bool act = xj < nx && validX[xj] != 0;
    float x[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) x[k] = act ? X[(size_t) xj * 33 + k] : 0.f;
    const unsigned rank_x = act ? tie_rank(xj, block, nblocks_x) : 0u;
    unsigned long long bk = ~0ull;
    for (int j0 = 0; j0 < n_irr; j0 += 64) {
        __syncthreads();
        const int nj = min(64, n_irr - j0);
        if (threadIdx.x < nj) Yj[threadIdx.x] = irr_list[j0 + threadIdx.x];
        __syncthreads();
        for (int i = threadIdx.x; i < nj * 33; i += 256) Ys[i] = Y[(size_t) Yj[i / 33] * 33 + i % 33];
        __syncthreads();
        for (int jj = 0; jj < nj; ++jj) {
            const float d = exact_l2(x, Ys + jj * 33);
            const bool fin = act && d < FLT_BIG;
            if (fin && best_x) {
                const unsigned long long key = ((unsigned long long) __float_as_uint(d) << 32) | tie_rank(Yj[jj], block, nblocks_y);
                bk = key < bk ? key : bk;
            }
            if (best_y) {
                unsigned long long ky = fin ? (((unsigned long long) __float_as_uint(d) << 32) | rank_x) : ~0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned long long other = __shfl_xor(ky, o);
                    ky = other < ky ? other : ky;
                }
                if ((threadIdx.x & 63) == 0 && ky != ~0ull) atomicMin(&best_y[Yj[jj]], ky);
            }
        }
    }
    if (act && best_x && bk != ~0ull) atomicMin(&best_x[xj], bk);
}

__global__ void fill_u64(unsigned long long* __restrict__ p, int n, unsigned long long v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void rerank_finalize(const unsigned long long* __restrict__ best, int nq, int block, int nblocks,
                                int32_t* __restrict__ idx, float* __restrict__ dist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    unsigned long long k = best[i];
    if (k == ~0ull) { idx[i] = -1; dist[i] = 0.f; return; }
    unsigned rank = (unsigned) (k & 0xffffffffu);
    int qb = rank / block, r = rank % block;
    int blk = nblocks - 1 - qb;
    idx[i] = blk * block + r;
    dist[i] = __uint_as_float((unsigned) (k >> 32));
}

int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// one side (A or B) after clustering
struct Side {
    int m = 0, n_valid = 0, n_pad = 0;
    int* perm = nullptr;          // [n_pad] padded position -> original row or -1
    uint8_t* valid = nullptr;     // [m]
    int* blkcl = nullptr;         // [n_pad / 256] cluster of each 256-row block (device)
    int* leaf_start = nullptr;    // [n_leaves + 1] padded start of every leaf (device); leaf l covers [start[l], start[l+1])
    int* leaf_count = nullptr;    // [MAXLEAF + 1] valid rows per leaf (device; [MAXLEAF] = invalid rows)
    unsigned* r2max = nullptr;    // [MAXLEAF] squared leaf radius bits (device)
    float nstat_n2 = 0.f, nstat_drop = 0.f;   // assign_kernel's packing statistics of this side (largest |x - c|^2, dropped energy)
    bool nstat_ovf = false;
    int n_irr = 0;                // irregular rows (finite, off the block-sum consensus: km_consensus); > IRR_CAP: the list is incomplete, the caller rebuilds without the lane
    int* irr_list = nullptr;      // [min(n_irr, IRR_CAP)] original row indices (device)
    std::vector<int> h_blkcl;     // host copies
    std::vector<int> h_leaf_start;
};

// assign + sort + place one side.  Leaves start at multiples of leaf_unit, clusters at multiples of cluster_unit
// (a multiple of 256 and of leaf_unit); padding positions carry perm = -1.
int build_side(lgr_ctx* ctx, const float* d_x, int m, const float* cen, const float* cen2, int sub, int leaf_unit, int cluster_unit,
               int ws_keys, int ws_perm, int role, const IrrRef* irr_ref, Side* s) {
    s->m = m;
    const int n_leaves = KCL * sub;
    unsigned *keys, *keys2;
    int *vals, *vals2;
    char* kbuf;
    size_t body = (((size_t) m * 17 + 255) & ~(size_t) 255);
    LGR_TRY(lgr_ws_t(ctx, ws_keys, body + 16384 + (size_t) IRR_CAP * 4, &kbuf));
    keys = (unsigned*) kbuf; keys2 = keys + m; vals = (int*) (keys2 + m); vals2 = vals + m;
    s->valid = (uint8_t*) (vals2 + m);
    int* counts = (int*) (kbuf + body);               // [MAXLEAF + 1]
    unsigned* rmax = (unsigned*) (kbuf + body + 8192);   // [MAXLEAF]
    s->leaf_count = counts; s->r2max = rmax;
    LGR_HIP(ctx, hipMemsetAsync(counts, 0, 16384, ctx->stream));
    const size_t assign_lds = ((size_t) KCL * (sub * 33 + 1) + 2 * MAXLEAF + 1) * 4;
    if (assign_lds > 64 * 1024) LGR_HIP(ctx, hipFuncSetAttribute((const void*) assign_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) assign_lds));
    unsigned* nstat = (unsigned*) (kbuf + body + 12288);   // [4], inside the 16 KB cleared above
    s->irr_list = (int*) (kbuf + body + 16384);
    assign_kernel<<<cdiv(m, ASSIGN_THREADS), ASSIGN_THREADS, assign_lds, ctx->stream>>>(d_x, m, cen, cen2, sub, keys, vals, s->valid, counts, rmax, role, nstat, irr_ref, s->irr_list);
    LGR_TRY(lgr_sort_pairs_u32(ctx, keys, keys2, vals, vals2, (size_t) m, 0, 32));
    int* h;
    LGR_TRY(lgr_pinned(ctx, 8192, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, counts, (MAXLEAF + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + MAXLEAF + 8, nstat, 16, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(&s->nstat_n2, h + MAXLEAF + 8, 4);
    memcpy(&s->nstat_drop, h + MAXLEAF + 9, 4);
    s->nstat_ovf = h[MAXLEAF + 10] != 0;
    s->n_irr = h[MAXLEAF + 11];
    std::vector<int> starts(2 * (size_t) MAXLEAF + 2, 0);   // [0..MAXLEAF): sorted start, [MAXLEAF..2*MAXLEAF]: padded start
    int acc = 0, pacc = 0;
    s->h_blkcl.clear();
    s->h_leaf_start.assign(n_leaves + 1, 0);
    for (int c = 0; c < KCL; ++c) {
        int cluster_begin = pacc;
        for (int j = 0; j < sub; ++j) {
            int l = c * sub + j;
            starts[l] = acc; starts[MAXLEAF + l] = pacc;
            s->h_leaf_start[l] = pacc;
            acc += h[l];
            pacc += pad_to(h[l], leaf_unit);
        }
        pacc = pad_to(pacc, cluster_unit);
        for (int b = 0; b < (pacc - cluster_begin) / BLOCK_ROWS; ++b) s->h_blkcl.push_back(c);
    }
    s->h_leaf_start[n_leaves] = pacc;
    starts[MAXLEAF + n_leaves] = pacc;
    s->n_valid = acc; s->n_pad = pacc;
    if (s->n_pad == 0) return LGR_OK;
    int* pbuf;
    LGR_TRY(lgr_ws_t(ctx, ws_perm, (size_t) s->n_pad + s->h_blkcl.size() + starts.size() + 64, &pbuf));
    s->perm = pbuf; s->blkcl = pbuf + s->n_pad;
    int* d_starts = s->blkcl + s->h_blkcl.size();
    s->leaf_start = d_starts + MAXLEAF;
    LGR_HIP(ctx, hipMemsetAsync(s->perm, 0xff, (size_t) s->n_pad * 4, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(s->blkcl, s->h_blkcl.data(), s->h_blkcl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host staging buffers go out of scope
    if (s->n_valid) place_kernel<<<cdiv(s->n_valid, 256), 256, 0, ctx->stream>>>(keys2, vals2, s->n_valid, d_starts, d_starts + MAXLEAF, s->perm);
    return LGR_OK;
}

template <bool ROWDIR>
int run_rerank(lgr_ctx* ctx, EpsExtra ex, CompView comp, const float* table, int n_groups, int group_size, const int* starts, const float* Q, const Side& qs,
               const float* nQ, const float* nQ_sets, const float* gmax, const int* cl_of_group,
               const float* T, const float* Tsorted, const Side& ts, int block, unsigned long long* best, int32_t* d_idx, float* d_dist,
               unsigned* stat_items, unsigned* stat_dense, bool force_dense, RefilterArgs ra, unsigned* stat_pairs) {
    const int q_pad = qs.n_pad;
    unsigned* dense;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_DENSE, (size_t) q_pad * (4 + CAND_KEEP) + 64, &dense));
    float* thr = (float*) (dense + q_pad);
    int* counts = (int*) (dense + 2 * (size_t) q_pad);
    int* offs = (int*) (dense + 3 * (size_t) q_pad);
    int* cand = (int*) (dense + 4 * (size_t) q_pad);
    char* misc;
    LGR_TRY(lgr_ws_t(ctx, WS_MATCH_MISC, 4096, &misc));
    RerankCounters* cnt = (RerankCounters*) (misc + 64);
    LGR_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(RerankCounters), ctx->stream));
    int nblocks = (ts.m + block - 1) / block;
    int dense_limit = force_dense ? -1 : std::max(64, n_groups / 2);
    rerank_count<ROWDIR><<<cdiv(q_pad, 256), 256, (size_t) (n_groups + 8) * 4, ctx->stream>>>(table, n_groups, q_pad, qs.perm, nQ, qs.blkcl, nQ_sets, gmax,
                                                                   cl_of_group, dense_limit, ex, comp, thr, counts, cand, dense, cnt);
    size_t tb = 0;
    LGR_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, counts, offs, 0, (size_t) q_pad, rocprim::plus<int>(), ctx->stream));
    void* tmp;
    LGR_TRY(lgr_ws(ctx, WS_GRID_TMP, tb, &tmp));
    LGR_HIP(ctx, rocprim::exclusive_scan(tmp, tb, counts, offs, 0, (size_t) q_pad, rocprim::plus<int>(), ctx->stream));
    int* h;
    LGR_TRY(lgr_pinned(ctx, 64, (void**) &h));
    LGR_HIP(ctx, hipMemcpyAsync(h, offs + (q_pad - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 1, counts + (q_pad - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipMemcpyAsync(h + 2, cnt, sizeof(RerankCounters), hipMemcpyDeviceToHost, ctx->stream));
    LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned n_items = (unsigned) (h[0] + h[1]);
    unsigned n_dense = ((RerankCounters*) (h + 2))->n_dense;
    *stat_items = n_items; *stat_dense = n_dense;
    if (n_items) {
        unsigned* ib;
        LGR_TRY(lgr_ws_t(ctx, WS_MATCH_ITEMS, (size_t) 4 * n_items + 64, &ib));
        unsigned *item_q = ib, *item_g = ib + n_items, *item_q2 = ib + 2 * (size_t) n_items, *item_g2 = ib + 3 * (size_t) n_items;
        rerank_emit<ROWDIR><<<cdiv(q_pad, 256), 256, (size_t) (n_groups + 8) * 4, ctx->stream>>>(table, n_groups, q_pad, nQ, qs.blkcl, nQ_sets, gmax, cl_of_group,
                                                                      ex, comp, thr, counts, cand, offs, item_q, item_g);
        int bits = 1;
        while ((1 << bits) < n_groups) ++bits;
        LGR_TRY(lgr_sort_pairs_u32(ctx, item_g, item_g2, (const int*) item_q, (int*) item_q2, (size_t) n_items, 0, bits));
        bool refiltered = false;
        *stat_pairs = 0;
        if (ra.ks) {
            // MFMA re-filter of the items -> (query, train row) pairs -> exact distances of the pairs only.  The pair buffer
            // holds 8 per item (1-3 are typical); if it ever overflows, the group scan below does the whole job instead.
            unsigned cap = 8u * n_items + 1024u;
            if (ra.pair_cap >= 0) cap = std::min(cap, (unsigned) ra.pair_cap);   // tests: force the overflow fallback
            unsigned* pb;
            LGR_TRY(lgr_ws_t(ctx, WS_MATCH_PAIRS, (size_t) 2 * cap + 64, &pb));
            unsigned *n_pairs = pb, *pair_q = pb + 16, *pair_t = pb + 16 + cap;
            LGR_HIP(ctx, hipMemsetAsync(n_pairs, 0, 4, ctx->stream));
            const int rf_grid = ((cdiv(n_items, 32 * (RF_THREADS / 64)) + 7) >> 3) << 3;   // (a multiple of 8: the kernel's XCD-aware block order)
#define LGR_RF_ARGS ra, item_g2, item_q2, n_items, n_groups, q_pad, ts.n_pad, group_size, starts, nQ, qs.blkcl, nQ_sets, gmax, cl_of_group, ex, thr, cap, n_pairs, pair_q, pair_t
            if (ra.ks == 6) rerank_refilter<ROWDIR, 6><<<rf_grid, RF_THREADS, 0, ctx->stream>>>(LGR_RF_ARGS);
            else rerank_refilter<ROWDIR, 7><<<rf_grid, RF_THREADS, 0, ctx->stream>>>(LGR_RF_ARGS);
#undef LGR_RF_ARGS
            LGR_HIP(ctx, hipMemcpyAsync(h + 8, n_pairs, 4, hipMemcpyDeviceToHost, ctx->stream));
            LGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const unsigned np = (unsigned) h[8];
            *stat_pairs = np;
            if (np <= cap) {
                if (np) rerank_pairs<<<cdiv(np, 256), 256, 0, ctx->stream>>>(Q, qs.perm, Tsorted, ts.perm, block, nblocks, pair_q, pair_t, np, best);
                refiltered = true;
            }
        }
        if (!refiltered)
            rerank_grouped<<<cdiv(n_items, RQ_THREADS), RQ_THREADS, 0, ctx->stream>>>(Q, qs.perm, Tsorted, ts.perm, ts.n_pad, group_size, starts, block, nblocks,
                                                                                     item_g2, item_q2, n_items, best);
    }
    if (n_dense) {
        dim3 g(std::min(cdiv(n_dense, 256), 64), cdiv(ts.m, DENSE_CHUNK));
        rerank_dense<<<g, 256, 0, ctx->stream>>>(Q, T, ts.valid, ts.m, block, nblocks, dense, n_dense, best);
    }
    rerank_finalize<<<cdiv(qs.m, 256), 256, 0, ctx->stream>>>(best, qs.m, block, nblocks, d_idx, d_dist);
    LGR_HIP(ctx, hipGetLastError());
    return LGR_OK;
}


}  // namespace
