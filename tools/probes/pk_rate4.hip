// Fourth round: v_cndmask_b32_e32 against the distance (in VALU instructions) to the v_cmp that wrote VCC, and a second read of the same VCC.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int REP = 4096;
template <int K, int READS>
__global__ void probe(float* out, float a, float b) {
    float s[16], t[16], u[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = a + i + threadIdx.x; t[i] = b + i; }
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = a * i;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(t[i]), "v"(a) : "vcc");
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(a));
#pragma unroll
            for (int k = 0; k < READS; ++k) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(s[(i + k) & 15]) : "v"(t[i]), "v"(a) : );
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += s[i] + t[i] + u[i & 7];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}
template <int K, int READS>
static void run(float* d, double ghz) {
    const int wps = 4;
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    probe<K, READS><<<256 * wps, 256>>>(d, 1.0f, 0.5f);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    probe<K, READS><<<256 * wps, 256>>>(d, 1.0f, 0.5f);
    (void) hipEventRecord(e1);
    (void) hipEventSynchronize(e1);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9 / ((double) REP * 16 * wps);
    printf("v_cmp + %d v_mul + %d v_cndmask_e32: %6.2f cycles per group -> %.2f per cndmask (v_cmp 4.2, v_mul 2.35 taken off)\n", K, READS, cyc, (cyc - 4.2 - 2.35 * K) / READS);
}
int main() {
    float* d;
    (void) hipMalloc(&d, 4096);
    const double ghz = 2.4;
    run<0, 1>(d, ghz); run<1, 1>(d, ghz); run<2, 1>(d, ghz); run<3, 1>(d, ghz); run<4, 1>(d, ghz); run<6, 1>(d, ghz); run<8, 1>(d, ghz); run<12, 1>(d, ghz); run<16, 1>(d, ghz);
    run<0, 2>(d, ghz); run<0, 3>(d, ghz); run<0, 4>(d, ghz); run<2, 2>(d, ghz);
    return 0;
}
