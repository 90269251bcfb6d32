// lgr_match_pack.cuh -- 2. MFMA operand packing (f32, f16 split, rotated f16 split), norms.
// Part of the brute-force FPFH matcher; see the header of lgr_match.hip and DESIGN.md section 3.
#pragma once
#include "lgr_match_common.cuh"

namespace {

// 2. pack.  P layout: [tile][kk][half][i] floats (tile = 32 rows): MFMA lane l of step kk reads P[(tile*KK+kk)*64 + l].
// role 0 (rows): centre = the row's own cluster (blkcl[pos / 256]), operand [-2 x', 1], nrm = |x'|^2
// role 1 (cols): blockIdx.y = cluster set p, centre c_p for every column, operand [x', |x'|^2]
// padding positions (perm < 0): rows [0.., 1] / cols [0.., +inf], nrm = +inf
__global__ void pack_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, int role,
                            const float* __restrict__ cen, const int* __restrict__ blkcl,
                            float* __restrict__ P, float* __restrict__ nrm, unsigned* __restrict__ ovf) {
    int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n_pad) return;
    int set = role == 1 ? blockIdx.y : 0;
    int o = perm[pos];
    int c = role == 1 ? set : blkcl[pos / BLOCK_ROWS];
    float v[33];
    float n2 = 0.f;
    if (o >= 0) {
#pragma unroll
        for (int k = 0; k < 33; ++k) { v[k] = X[(size_t) o * 33 + k] - cen[c * 33 + k]; n2 = n2 + v[k] * v[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = 0.f;
        n2 = __uint_as_float(0x7f800000u);
    }
    nrm[(size_t) set * n_pad + pos] = n2;
    if (o >= 0 && !(n2 < FLT_BIG)) *ovf = 1u;   // |x'|^2 overflows float: the filter cannot represent this row
    int tile = pos >> 5, r = pos & 31;
    float* base = P + ((size_t) set * (n_pad / TILE) + tile) * KK * 64 + r;
#pragma unroll
    for (int k = 0; k < 34; ++k) {
        float val;
        if (k < 33) val = role == 0 ? -2.0f * v[k] : v[k];
        else val = role == 0 ? 1.0f : n2;
        base[(k >> 1) * 64 + (k & 1) * 32] = val;
    }
}

// f16-split operands (OpFmt<true>).  Same roles / sets / nrm output as pack_kernel; P holds f16x8 fragments:
// fragment (tile, step, lane) at ((set * tiles + tile) * 7 + step) * 64 + lane, lane = row | (khalf << 5).
// Concatenated K index c: [0,33) a1.b1, [33,66) a1.b2, [66,99) a2.b1, 99..104 norm slots, rest 0.
// rows: h = split(-2 x' 2^s);  cols: h = split(x' 2^s);  a norm enters as the three-term f16 expansion of
// N = |x'|^2 2^2s against the constants A1..A3 on the other side (N = A1 B1 + A2 B2 + A3 B3 up to 2^-33 N or the f16
// flush limit).
// Helmert coordinates of one 11-bin block: y_k = (x_0 + .. + x_{k-1} - k x_k) / sqrt(k (k + 1)), k = 1..10 (orthonormal, all
// orthogonal to (1,..,1)); *u = (x_0 + .. + x_10) / sqrt(11) is the dropped coordinate.
__device__ __forceinline__ void helmert11(const float* __restrict__ x, float* __restrict__ y, float* u) {
    const float rs[10] = {0.70710678118654752f, 0.40824829046386302f, 0.28867513459481288f, 0.22360679774997897f, 0.18257418583505537f,
                          0.15430334996209191f, 0.13363062095621219f, 0.11785113019775793f, 0.10540925533894598f, 0.09534625892455924f};
    float pre = x[0];
#pragma unroll
    for (int k = 1; k <= 10; ++k) {
        y[k - 1] = (pre - (float) k * x[k]) * rs[k - 1];
        pre = pre + x[k];
    }
    *u = pre * 0.30151134457776363f;
}

template <bool ROT>
__global__ __launch_bounds__(256) void pack16_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, int role,
                              const float* __restrict__ cen, const int* __restrict__ blkcl, F16Scale sc,
                              _Float16* __restrict__ P, float* __restrict__ nrm, float2* __restrict__ shell /* [sets][n_pad / 32] or nullptr */) {
    int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = pos < n_pad;
    if (!in_range) pos = n_pad - 1;           // (nothing is stored for these lanes)
    // the row is read once (a 132-byte gather) and packed for every set: 16 column sets, one per centre, or the row set
    const int n_sets = role == 1 ? KCL : 1;
    const int o = perm[pos];
    float x0[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) x0[k] = o >= 0 ? X[(size_t) o * 33 + k] : 0.f;
#pragma unroll 1
    for (int set = 0; set < n_sets; ++set) {
    const int c = role == 1 ? set : blkcl[pos / BLOCK_ROWS];
    float v[33];
    float n2 = 0.f;
    if (o >= 0) {
#pragma unroll
        for (int k = 0; k < 33; ++k) { v[k] = x0[k] - cen[c * 33 + k]; n2 = n2 + v[k] * v[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 33; ++k) v[k] = 0.f;
        n2 = __uint_as_float(0x7f800000u);
    }
    if (in_range) nrm[(size_t) set * n_pad + pos] = n2;   // |x'|^2 in all 33 coordinates: the magnitude the error bounds are stated in
    if (shell) {
        // radial shell of the 32-position tile about this centre: (min, max) of |x'| over its finite rows (min rounded down, max up; an
        // empty tile: (+inf, 0)) -- the 32 rows of a tile are the 32 lanes of a half wave (shell bound, lgr_match_rerank.cuh / match_mfma)
        const bool fin = in_range && n2 < FLT_BIG;
        float mn = fin ? n2 : __uint_as_float(0x7f800000u), mx = fin ? n2 : 0.f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
        if (in_range && (pos & 31) == 0)
            shell[(size_t) set * (n_pad / TILE) + (pos >> 5)] = make_float2(mn < FLT_BIG ? sqrtf(mn) * 0.9999998f : mn, sqrtf(mx) * 1.0000002f);
    }
    float y[30], u0, u1, u2;
    if (ROT) { helmert11(v, y, &u0); helmert11(v + 11, y + 10, &u1); helmert11(v + 22, y + 20, &u2); }
    if (!in_range) continue;
    constexpr int nd = ROT ? 30 : 33, ks = ROT ? OpFmt<FMT_F16R>::KS : OpFmt<FMT_F16>::KS;
    int tile = pos >> 5, r = pos & 31;
    // the row's K = 16 ks halves are assembled in registers (all indices are compile-time constants) and leave as
    // 2 ks 16-byte pieces: piece (step, khalf) of row r sits at fragment (step * 64 + khalf * 32 + r)
    _Float16 hv[ks * 16];
    auto put = [&](int cidx, _Float16 h) { hv[cidx] = h; };
    const float mul = role == 0 ? -2.0f * sc.s_mul : sc.s_mul;
    float n2m = n2;                           // the norm the MFMA chain must see: of the operand coordinates
    if (ROT && o >= 0) {
        n2m = 0.f;
#pragma unroll
        for (int k = 0; k < 30; ++k) n2m = n2m + y[k] * y[k];
    }
    // K order.  Plain format: [0,nd) a1.b1, [nd,2nd) a1.b2, [2nd,3nd) a2.b1, then the six norm slots.  Rotated format: the
    // first two MFMA steps (K slots 0..31) hold a1.b1 and the leading norm terms -- a coarse d2~ the kernel tests before
    // it spends the other four steps on a tile (match_mfma, "coarse rejection") -- then [32,62) a1.b2, [62,92) a2.b1 and
    // the four remaining norm terms.
    constexpr int o12 = ROT ? 32 : nd, o21 = ROT ? 62 : 2 * nd;
#pragma unroll
    for (int k = 0; k < nd; ++k) {
        float x = (ROT ? y[k] : v[k]) * mul;  // exact (power of two)
        _Float16 h1 = (_Float16) x;           // round to nearest
        _Float16 h2 = (_Float16) (x - (float) h1);
        if (role == 0) { put(k, h1); put(o12 + k, h1); put(o21 + k, h2); }
        else { put(k, h1); put(o12 + k, h2); put(o21 + k, h1); }
    }
    // norm slots: 3 nd .. 3 nd + 2 carry |b'|^2 (expansion on the column side, constants on the row side), the next three
    // |a'|^2 the other way round, so d2~ 2^2s = |b'|^2 - 2 a'.b' + |a'|^2 comes out of the MFMA chain with C = 0
    const bool rows = role == 0;
    const _Float16 c0 = (_Float16) sc.a_norm[0], c1 = (_Float16) sc.a_norm[1], c2 = (_Float16) sc.a_norm[2];
    _Float16 b1, b2, b3;
    if (o >= 0) {
        float N = n2m * (sc.s_mul * sc.s_mul);
        b1 = (_Float16) (N / sc.a_norm[0]);
        float r1 = __builtin_fmaf(-sc.a_norm[0], (float) b1, N);
        b2 = (_Float16) (r1 / sc.a_norm[1]);
        float r2 = __builtin_fmaf(-sc.a_norm[1], (float) b2, r1);
        b3 = (_Float16) (r2 / sc.a_norm[2]);
    } else {
        b1 = (_Float16) __uint_as_float(0x7f800000u); b2 = (_Float16) 0.f; b3 = (_Float16) 0.f;   // padding: +inf
    }
    // columns: [expansion | constants], rows: [constants | expansion]
    if (ROT) {
        put(30, rows ? c0 : b1); put(31, rows ? b1 : c0);
        put(92, rows ? c1 : b2); put(93, rows ? c2 : b3); put(94, rows ? b2 : c1); put(95, rows ? b3 : c2);
    } else {
        put(3 * nd + 0, rows ? c0 : b1); put(3 * nd + 1, rows ? c1 : b2); put(3 * nd + 2, rows ? c2 : b3);
        put(3 * nd + 3, rows ? b1 : c0); put(3 * nd + 4, rows ? b2 : c1); put(3 * nd + 5, rows ? b3 : c2);
#pragma unroll
        for (int cidx = 3 * nd + 6; cidx < ks * 16; ++cidx) put(cidx, (_Float16) 0.f);
    }
    f16x8* base = reinterpret_cast<f16x8*>(P) + ((size_t) set * (n_pad / TILE) + tile) * ks * 64;
#pragma unroll
    for (int piece = 0; piece < 2 * ks; ++piece) {
        f16x8 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = hv[piece * 8 + j];
        base[(piece >> 1) * 64 + ((piece & 1) << 5) + r] = w;
    }
    }   // sets
}

// original rows in padded (cluster-sorted) order, contiguous, for the exact rerank (padding rows are never read)
__global__ void gather_rows_kernel(const float* __restrict__ X, const int* __restrict__ perm, int n_pad, float* __restrict__ out) {
    size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t) n_pad * 33) return;
    int pos = (int) (e / 33), k = (int) (e % 33);
    int o = perm[pos];
    out[e] = o >= 0 ? X[(size_t) o * 33 + k] : 0.f;
}

// per-group maxima of sqrt(norm) (finite entries only): out[set][g]; groups are [starts[g], starts[g+1]) or, with
// starts == nullptr, fixed windows of `group` positions
__global__ void group_max_kernel(const float* __restrict__ nrm, int n_pad, int group, const int* __restrict__ starts, float* __restrict__ out) {
    int g = blockIdx.x, set = blockIdx.y, n_groups = gridDim.x;
    float m = 0.f;
    int p0 = starts ? starts[g] : g * group, p1 = starts ? starts[g + 1] : min(n_pad, (g + 1) * group);
    for (int pos = p0 + threadIdx.x; pos < p1; pos += blockDim.x) {
        float v = nrm[(size_t) set * n_pad + pos];
        if (v < FLT_BIG) m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int) (blockDim.x >> 6); ++w) m = fmaxf(m, sh[w]);
        out[(size_t) set * n_groups + g] = sqrtf(m) * 1.0000002f;   // rounded up
    }
}


// shells of windows of `group` consecutive tiles from the tiles' shells (pack16_kernel): out[set][g] = (min of the minima, max of the maxima);
// out_max (optional) = the maxima alone (the stage maxima of the coarse thresholds)
__global__ void shell_reduce_kernel(const float2* __restrict__ tile_shell, int n_sets, int n_tiles, int group, float2* __restrict__ out, float* __restrict__ out_max) {
    const int n_groups = (n_tiles + group - 1) / group;
    const long long id = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long) n_sets * n_groups) return;
    const int set = (int) (id / n_groups), g = (int) (id % n_groups);
    float mn = __uint_as_float(0x7f800000u), mx = 0.f;
    for (int t = g * group; t < min(n_tiles, (g + 1) * group); ++t) {
        const float2 v = tile_shell[(size_t) set * n_tiles + t];
        mn = fminf(mn, v.x); mx = fmaxf(mx, v.y);
    }
    out[id] = make_float2(mn, mx);
    if (out_max) out_max[id] = mx;
}

}  // namespace
