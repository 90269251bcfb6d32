// lgr_match_common.cuh -- constants, operand formats and small helpers of the matcher (included by lgr_match.hip).
#pragma once
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include "lgr_internal.h"

namespace {


constexpr int KK = 17;              // K = 34 -> 17 MFMA steps of k = 2
constexpr int TILE = 32;
constexpr int RW = 1;               // row tiles per wave
constexpr int WAVES = 8;
constexpr int NTHR = WAVES * 64;    // threads per workgroup of the MFMA kernel
constexpr int BLOCK_ROWS = TILE * RW * WAVES;   // 256
constexpr int STAGE_TILES = 4;
constexpr int STAGE_COLS = STAGE_TILES * TILE;  // 128
constexpr int CHUNK_COLS = 4096;
constexpr int PAD = 256;
constexpr int KCL = 16;             // k-means centres (operand centring)
constexpr int SUBMAX = 64;          // second-level centres per cluster ("leaves": sort order + skip bounds)
constexpr int MAXLEAF = KCL * SUBMAX;
#ifndef LGR_KM_SAMPLE
#define LGR_KM_SAMPLE 16384
#endif
constexpr int KM_SAMPLE = LGR_KM_SAMPLE;    // sample rows per side
#ifndef LGR_KM_ITERS
#define LGR_KM_ITERS 12   // Lloyd iterations, first / second level (round 1: 6 / 4 -> 10 / 8 -> 16 / 8: 81.7 -> 80.2 -> 79.5 ms per 1M pair, tighter leaves; with the exact integer sums of round 2 the 1M sample reaches its fixed point by 10 / 6 -- same tile fraction as 16 / 8 -- and every step is ~30-60 us of exposed launch latency)
#endif
constexpr int KM_ITERS = LGR_KM_ITERS;
#ifndef LGR_KM2_ITERS
#define LGR_KM2_ITERS 6
#endif
constexpr int KM2_ITERS = LGR_KM2_ITERS;
#ifndef LGR_MM_OCC
#define LGR_MM_OCC 4          // waves per SIMD of match_mfma (2: 256 VGPRs, one workgroup per CU; 4: 128 VGPRs, two)
#endif
#ifndef LGR_NEAR_T
#define LGR_NEAR_T 40
#endif
constexpr int NEAR_T = LGR_NEAR_T;          // pass 0 visits the NEAR_T nearest leaves of a row block / row blocks of a leaf (measured at 1M: 16 / 24 / 32 / 48 / 64 -> 75.5 / 73.1 / 72.2 / 73.7 / 76.4 ms per pair in round 1; with the per-tile coarse thresholds of round 2 the final pass is cheaper per tile: 16 / 20 / 24 / 28 / 32 -> 40.4 / 39.8 / 39.4 / 39.45 / 39.8-40.1; round 3, pass 0 takes only the stages of overlapping shells of those leaves and is three times cheaper per leaf: 28 / 36 / 48 / 64 -> 30.7-31.2 / 29.9-30.5 / 30.0-30.6 / 30.6)
#ifndef LGR_AUTO_DENSE_FRAC
#define LGR_AUTO_DENSE_FRAC 0.9f   // lgr_match_options.auto_dense: pass 0 takes everything when this fraction of the (row block, leaf) lower bounds is zero
#endif
#ifndef LGR_PRUNE_BETAS
#define LGR_PRUNE_BETAS 1.0f   // ONE final pass.  Intermediate thresholds (0.5f, 1.0f / 0.7f, 1.0f: a sweeping pass between pass 0 and the final one) were measured
                               // again in round 4 with the sweep kernel: 26.6-27.4 ms per pair against 25.4 at pass-0 widths 24-40 (a pass boundary costs 1.2 ms).  More than one
                               // entry is NOT supported any more: the repair of an overflowing tile list (match_impl) covers the last pass only (static_assert there).
#endif
#ifndef LGR_GROUP_COLS
#define LGR_GROUP_COLS 1024
#endif
constexpr int GROUP_COLS = LGR_GROUP_COLS;     // largest column group of the row-minimum table (leaves are cut into such pieces)
constexpr int STAGES_PER_CHUNK = CHUNK_COLS / STAGE_COLS;   // 32 -> one 32-bit stage mask per (row block, chunk)
constexpr float FLT_BIG = 3.4028234663852886e38f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MFMA operand formats.  F32: v_mfma_f32_32x32x2_f32, K = 34 (33 dims + norm slot) -> 17 steps, fragment = 1 float.
// F16: v_mfma_f32_32x32x16_f16 (16x the f32 rate) on two-term f16 splits of the scaled f32 operands,
//      x * 2^s = h1 + h2 (+ residual <= 2^-22 |x|):  a.b ~ a1.b1 + a1.b2 + a2.b1  -> concatenated K = 3 * 33 + 6 norm slots
//      = 105, padded to 112 = 7 steps, fragment = 8 halves (lane l: row l & 31, k = 16 * step + 8 * (l >> 5) + j).
// F16R: the same on 30 coordinates.  Every 11-bin block of an FPFH row sums to 100, so differences of rows have no component
//      along the block's all-ones direction; in a Helmert basis of the block that direction is one coordinate, the other
//      10 carry the whole distance.  K = 3 * 30 + 6 = 96 = 6 steps (-1/7 of the MFMA work, LDS reads and operand bytes).
//      Used only when the dropped coordinates are (numerically) constant over both sets; their largest measured energy
//      enters the error bound, so any input stays exact (match_impl, "rot").
enum { FMT_F32 = 0, FMT_F16 = 1, FMT_F16R = 2 };
template <int FMT> struct OpFmt;
template <> struct OpFmt<FMT_F32> { typedef float frag; static constexpr int KS = 17; };
template <> struct OpFmt<FMT_F16> { typedef f16x8 frag; static constexpr int KS = 7; };
template <> struct OpFmt<FMT_F16R> { typedef f16x8 frag; static constexpr int KS = 6; };
struct F16Scale { float s_mul; float inv_s2; float a_norm[3]; };   // 2^s, 2^-2s, the three a-side norm-slot constants

// Coarse rejection (rotated format, passes that have upper bounds; DESIGN.md 3b "coarse rejection").  After the first two
// MFMA steps of a 32 x 32 tile the accumulator holds  c = |a'|^2_lead + |b'|^2_lead - 2 a1.b1  (scaled by 2^2s): the
// filtered d2~ without the split cross terms a1.b2 + a2.b1 and without the lower norm terms, so
//     |c 2^-2s - d2| <= eps(x, y) + delta(x, y),   delta = 2^-10 x y + 2^-11 (x^2 + y^2) (1 + 2^-9) + lin (x + y) + abs2
// with x = max |a'| of the row group, y = max |b'| of the column stage (|h2| <= 2^-11 |h| per element, or the f16 flush
// limit: the linear / absolute terms).  A tile is abandoned when every c exceeds
//     T = max(U_rows, U_cols) (1 + 1e-5) + eps + delta
// -- U_rows = largest U^2 of the tile's 32 rows (u_rt), U_cols = largest U^2 of the tile's 32 columns (u_ct): no element of
// the tile can be the nearest neighbour (or tie with it) of its row or of its column, which is the same statement the
// bound-based skipping makes about a whole (row block, leaf) tile.
struct CoarseArgs {
    const float* u_rb;          // [row blocks] or nullptr: no coarse rejection in this launch
    const float* u_rt;          // [row tiles]: largest U^2 of the 32 rows of a tile (the row side of a wave's threshold)
    const float* u_row;         // [rows] U^2 of every row (< 0: padding), or nullptr: the sweep then tests with the tile's u_rt for all rows
    const unsigned* u_stage;    // [column stages] float bits, or nullptr (row direction only)
    const unsigned* u_ct;       // [column tiles] float bits: largest U^2 of the 32 columns of a tile (with u_stage)
    const float* u_colv;        // [columns] U^2 of every column (0: padding), or nullptr: tiles the first test keeps are not re-tested per element
    int n_ct_total;
    const float* xmax;          // [row groups] max |a'| (gmaxA)
    const float* ymax;          // [KCL][column stages] max |b'| per set (group_max_kernel over 128-column windows)
    int n_stage_total;
    float quad, cross, lin, abs;   // T = max(U) (1 + 1e-5) + quad (x + y)^2 + cross x y + lin (x + y) + abs
    unsigned long long* cnt;    // [3]: tiles tested, tiles abandoned, tiles skipped by the shell test before any MFMA (or nullptr)
    // radial shells [min |x - c|, max |x - c|] of the 32-row / 32-column tiles (shell test per wave and tile; nullptr: off)
    const float2* rt_shell;     // [row tiles] about the row block's own centre
    const float2* ct_shell;     // [KCL][column tiles] about every centre
};

// proven bound of |filtered - d2| for a pair whose operands have |a'| <= x and |b'| <= y: evaluated in float,
// inflated by 1e-5 (its five roundings are worth 3e-7)
struct EpsExtra { float lin, abs, quad; };   // quad: multiplier of the 4 g40 (x + y)^2 term (1 on the f32 path)
__device__ __forceinline__ float eps_xy(float x, float y, EpsExtra ex) {
    const float c_quad = 9.5367477e-6f * ex.quad;   // 4 g40 = 4 * 40 u / (1 - 40 u) = 9.53677e-6, rounded up
    const float s = x + y;
    return ((c_quad * s) * s + ex.lin * s + ex.abs) * 1.00001f + 1e-30f;
}

__device__ __forceinline__ unsigned f2key(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}
__device__ __forceinline__ bool row_finite(const float* __restrict__ r, float* v) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 33; ++k) { v[k] = r[k]; ok = ok && (fabsf(v[k]) <= FLT_BIG); }
    return ok;
}


}  // namespace
