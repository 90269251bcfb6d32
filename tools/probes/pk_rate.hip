// Issue rate of packed f32 instructions against their scalar equivalents on gfx950, one to eight waves per SIMD, no MFMA nearby.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/pk_rate.hip -o build/pk_rate && build/pk_rate
// Every kernel runs REP x 64 independent instructions per wave (16 accumulators, 4 rounds); cycles per instruction and SIMD =
// elapsed * clock / (instructions issued on one SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int REP = 4096;

template <int KIND>
__global__ void probe(float* out, float a, float b) {
    float s[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = a + i + threadIdx.x; p[i] = v2f{a + i, b + threadIdx.x}; }
    const v2f pa{a, a}, pb{b, b};
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
                if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
                if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
                if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(a));
                if (KIND == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(s[i]), "v"(a) : "vcc");
                if (KIND == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
                if (KIND == 9) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
                if (KIND == 10) asm volatile("v_add_f64 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
                if (KIND == 11) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(p[i]) : "v"(s[i]));
                if (KIND == 12) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
                if (KIND == 13) asm volatile("v_floor_f32 %0, %0" : "+v"(s[i]));
                if (KIND == 14) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
                if (KIND == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(a));
            }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += s[i] + p[i].x + p[i].y;
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int KIND>
static void run(const char* name, float* d, double ghz) {
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 256;                 // one wave per SIMD per workgroup
        const int blocks = 256 * wps;            // wps workgroups per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<KIND><<<blocks, threads>>>(d, 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<KIND><<<blocks, threads>>>(d, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double) REP * 64 * wps;   // instructions one SIMD issued
        printf("%-16s %d waves/SIMD: %8.3f ms  -> %.2f cycles per instruction and SIMD (at %.2f GHz)\n", name, wps, ms, ms * 1e-3 * ghz * 1e9 / per_simd, ghz);
    }
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const double ghz = 2.4;
    run<0>("v_fma_f32", d, ghz); run<4>("v_mul_f32", d, ghz); run<1>("v_pk_fma_f32", d, ghz); run<2>("v_pk_mul_f32", d, ghz); run<3>("v_pk_add_f32", d, ghz);
    run<5>("v_rcp_f32", d, ghz); run<12>("v_sqrt_f32", d, ghz); run<6>("v_cndmask_b32", d, ghz); run<7>("v_cmp_lt_f32", d, ghz); run<8>("v_mul_lo_u32", d, ghz);
    run<9>("v_fma_f64", d, ghz); run<10>("v_add_f64", d, ghz); run<11>("v_cvt_f64_f32", d, ghz); run<13>("v_floor_f32", d, ghz); run<14>("v_max3_f32", d, ghz); run<15>("v_mov_b32", d, ghz);
    return 0;
}
