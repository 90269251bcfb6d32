// (s_and_b64 writes SCC: the clobber is declared -- without it the loop branch, whose compare hipcc hoists to the top of the body, never falls through.)
// Third round: when is v_cndmask_b32_e32 (VCC) slow on gfx950?  (pk_rate2: 23 cycles with a VCC written once by s_mov, 2 beside a v_cmp.)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int REP = 4096;
template <int KIND>
__global__ void probe(float* out, float a, float b) {
    float s[16], t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = a + i + threadIdx.x; t[i] = b + i; }
    unsigned long long m = 0x5555555555555555ull, m2 = 0x3333333333333333ull;
    asm volatile("s_mov_b64 vcc, %0" : : "s"(m) : "vcc");
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) { asm volatile("s_and_b64 vcc, %0, %1" : : "s"(m), "s"(m2) : "vcc", "scc"); asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(s[i]) : "v"(t[i]), "v"(a) : ); }
                if (KIND == 1) { if ((i & 3) == 0) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(t[i]), "v"(a) : "vcc"); asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(s[i]) : "v"(t[i]), "v"(a) : ); }
                if (KIND == 2) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(s[i]) : "v"(t[i]), "v"(a) : );
                if (KIND == 3) { asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(t[i]), "v"(a)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(s[i]) : "v"(t[i]), "v"(a), "s"(m2)); }
                if (KIND == 4) { asm volatile("s_and_b64 %0, %1, %2" : "=s"(m2) : "s"(m), "s"(m2) : "scc"); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(s[i]) : "v"(t[i]), "v"(a), "s"(m2)); }
                if (KIND == 5) { asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(t[i]), "v"(a) : "vcc"); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(t[i]) : "v"(a)); asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(s[i]) : "v"(t[i]), "v"(a) : ); }
                if (KIND == 6) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(a) : );   // dst = src0 (pk_rate)
                if (KIND == 7) asm volatile("v_addc_co_u32_e32 %0, vcc, %1, %2, vcc" : "=v"(s[i]) : "v"(t[i]), "v"(a) : "vcc");
            }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += s[i] + t[i];
    if (acc == 12345.678f) out[threadIdx.x] = acc + (float) m2;
}
template <int KIND>
static void run(const char* name, int per, float* d, double ghz) {
    for (int wps : {1, 4}) {
        hipEvent_t e0, e1;
        (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
        probe<KIND><<<256 * wps, 256>>>(d, 1.0f, 0.5f);
        (void) hipDeviceSynchronize();
        (void) hipEventRecord(e0);
        probe<KIND><<<256 * wps, 256>>>(d, 1.0f, 0.5f);
        (void) hipEventRecord(e1);
        (void) hipEventSynchronize(e1);
        float ms = 0;
        (void) hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %d waves/SIMD: %8.3f ms  -> %.2f cycles per GROUP and SIMD (%d instructions per group)\n", name, wps, ms, ms * 1e-3 * ghz * 1e9 / ((double) REP * 64 * wps), per);
    }
}
int main() {
    float* d;
    (void) hipMalloc(&d, 4096);
    const double ghz = 2.4;
    run<0>("s_and_b64 vcc + v_cndmask_e32 vcc", 2, d, ghz);
    run<1>("v_cmp vcc every 4th + v_cndmask_e32 vcc", 1, d, ghz);
    run<2>("v_cndmask_e64 ..., vcc (stale vcc)", 1, d, ghz);
    run<3>("v_cmp_e64 sgpr + v_cndmask_e64 sgpr", 2, d, ghz);
    run<4>("s_and_b64 sgpr + v_cndmask_e64 sgpr", 2, d, ghz);
    run<5>("v_cmp vcc + v_mul + v_cndmask_e32 vcc", 3, d, ghz);
    run<6>("v_cndmask_e32 dst=src0, stale vcc", 1, d, ghz);
    run<7>("v_addc_co_u32 vcc, ..., vcc", 1, d, ghz);
    return 0;
}
