// What exactly does v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 compute along k?  (gfx950; run on the GPU box)
// Compares the hardware result for random operands with candidate float formulas evaluated on the device with explicit fmaf.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const float* A /*[16][4]*/, const float* B /*[4][16]*/, const float* C /*[16][16]*/, float* D) {
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 c;
    for (int q = 0; q < 4; ++q) c[q] = C[(4 * (l >> 4) + q) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) D[(4 * (l >> 4) + q) * 16 + (l & 15)] = c[q];
}
__global__ void k32(const float* A /*[32][2]*/, const float* B /*[2][32]*/, const float* C /*[32][32]*/, float* D) {
    int l = threadIdx.x;
    float a = A[(l & 31) * 2 + (l >> 5)], b = B[(l >> 5) * 32 + (l & 31)];
    f32x16 c;
    for (int q = 0; q < 16; ++q) c[q] = C[((q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[q];
}

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main() {
    const int trials = 2000;
    long cnt[8] = {0}, total = 0;
    const char* names[8] = {"fma chain k ascending", "fma chain k descending", "exact sum rounded once (double)", "pairwise: fma(a0,b0,c) + ... tree (k0,k1)+(k2,k3)",
                            "mul-add unfused ascending", "fma chain, products k1,k0,k3,k2", "c + (exact dot rounded)", "fma(a3,b3, fma(a2,b2, fma(a1,b1, a0*b0 + c)))"};
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 4); hipMalloc(&dB, 64 * 4); hipMalloc(&dC, 1024 * 4); hipMalloc(&dD, 1024 * 4);
    srand(1);
    auto rnd = [](int mode) { float v = (float) rand() / RAND_MAX; if (mode == 0) return v * 100.f; return (v - 0.5f) * ldexpf(1.f, rand() % 24 - 12); };
    for (int shape = 0; shape < 2; ++shape) {
        memset(cnt, 0, sizeof cnt); total = 0;
        const int M = shape ? 32 : 16, K = shape ? 2 : 4;
        for (int t = 0; t < trials; ++t) {
            float A[64], B[64], C[1024], D[1024];
            for (int i = 0; i < 64; ++i) { A[i] = rnd(t & 1); B[i] = rnd(t & 1); }
            for (int i = 0; i < M * M; ++i) C[i] = rnd(t & 1);
            hipMemcpy(dA, A, 256, hipMemcpyHostToDevice); hipMemcpy(dB, B, 256, hipMemcpyHostToDevice); hipMemcpy(dC, C, M * M * 4, hipMemcpyHostToDevice);
            if (shape) k32<<<1, 64>>>(dA, dB, dC, dD); else k16<<<1, 64>>>(dA, dB, dC, dD);
            hipMemcpy(D, dD, M * M * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j) {
                    float a[4], b[4], c = C[i * M + j];
                    for (int k = 0; k < K; ++k) { a[k] = A[i * K + k]; b[k] = B[k * M + j]; }
                    float r[8];
                    float s = c; for (int k = 0; k < K; ++k) s = fmaf(a[k], b[k], s); r[0] = s;
                    s = c; for (int k = K - 1; k >= 0; --k) s = fmaf(a[k], b[k], s); r[1] = s;
                    double ds = c; for (int k = 0; k < K; ++k) ds += (double) a[k] * b[k]; r[2] = (float) ds;
                    if (K == 4) { float p = fmaf(a[1], b[1], a[0] * b[0]), q = fmaf(a[3], b[3], a[2] * b[2]); r[3] = (p + q) + c; } else r[3] = fmaf(a[1], b[1], a[0] * b[0]) + c;
                    s = c; for (int k = 0; k < K; ++k) { float pr = a[k] * b[k]; s = s + pr; } r[4] = s;
                    s = c; if (K == 4) { int o[4] = {1, 0, 3, 2}; for (int k = 0; k < 4; ++k) s = fmaf(a[o[k]], b[o[k]], s); } else { s = fmaf(a[1], b[1], s); s = fmaf(a[0], b[0], s); } r[5] = s;
                    double dd = 0; for (int k = 0; k < K; ++k) dd += (double) a[k] * b[k]; r[6] = c + (float) dd;
                    s = a[0] * b[0] + c; for (int k = 1; k < K; ++k) s = fmaf(a[k], b[k], s); r[7] = s;
                    for (int f = 0; f < 8; ++f) if (bits(r[f]) == bits(D[i * M + j])) cnt[f]++;
                    total++;
                }
        }
        printf("%s: %ld results\n", shape ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", total);
        for (int f = 0; f < 8; ++f) printf("   %-60s matches %6.2f %%\n", names[f], 100.0 * cnt[f] / total);
    }
    return 0;
}
