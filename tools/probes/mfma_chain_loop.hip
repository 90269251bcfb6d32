// Does a LOOP of v_mfma_f32_16x16x4_f32 with masked (zero) weights and three interleaved accumulators equal the per-bin fmaf chain
// over the non-zero weights?  (the structure of fpfh_mfma_kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void loopk(const float* W /*[G][16][4]*/, const float* H /*[G][4][48]*/, int G, float* D /*[16][48]*/) {
    int l = threadIdx.x, i = l & 15, k = l >> 4;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    for (int g = 0; g < G; ++g) {
        float w = W[(g * 16 + i) * 4 + k];
        if (__ballot(w != 0.f) == 0ull) continue;
        const float* h = H + (size_t) (g * 4 + k) * 48 + i;
        float b0 = h[0], b1 = h[16], b2 = h[32];
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b0, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b1, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b2, a2, 0, 0, 0);
    }
    for (int q = 0; q < 4; ++q) { D[(4 * k + q) * 48 + i] = a0[q]; D[(4 * k + q) * 48 + 16 + i] = a1[q]; D[(4 * k + q) * 48 + 32 + i] = a2[q]; }
}
int main() {
    const int G = 300;
    std::vector<float> W(G * 64), H(G * 4 * 48), D(16 * 48);
    srand(3);
    for (auto& w : W) w = (rand() % 3 == 0) ? 16.f + 4000.f * rand() / RAND_MAX : 0.f;
    for (auto& h : H) h = (rand() % 4 == 0) ? 0.f : 30.f * rand() / RAND_MAX;
    float *dW, *dH, *dD;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dH, H.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice);
    loopk<<<1, 64>>>(dW, dH, G, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 48; ++j) {
            float s = 0.f;
            for (int g = 0; g < G; ++g)
                for (int k = 0; k < 4; ++k) { float w = W[(g * 16 + i) * 4 + k]; if (w != 0.f) s = fmaf(w, H[(g * 4 + k) * 48 + j], s); }
            if (memcmp(&s, &D[i * 48 + j], 4)) { if (bad < 5) printf("i %d j %d cpu %.9g gpu %.9g\n", i, j, s, D[i * 48 + j]); ++bad; }
        }
    printf("mismatches %d of %d\n", bad, 16 * 48);
    return 0;
}
