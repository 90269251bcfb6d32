// Second round of issue-rate probes on gfx950 (see pk_rate.hip): v_cndmask forms, integer ops, and MFMA beside VALU from OTHER waves of the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int REP = 4096;

template <int KIND>
__global__ void probe(float* out, float a, float b) {
    float s[16], t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = a + i + threadIdx.x; t[i] = b + i; }
    unsigned long long m = threadIdx.x & 1 ? 0x5555555555555555ull : 0x3333333333333333ull;
    m = __builtin_amdgcn_readfirstlane((int) m) | ((unsigned long long) __builtin_amdgcn_readfirstlane((int) (m >> 32)) << 32);
    asm volatile("s_mov_b64 vcc, %0" : : "s"(m) : "vcc");
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(s[i]) : "v"(t[i]), "v"(a) : );
                if (KIND == 1) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(s[i]) : "v"(t[i]), "v"(a), "s"(m));
                if (KIND == 2) { asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(t[i]), "v"(a) : "vcc"); asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(s[i]) : "v"(t[i]), "v"(a) : ); }
                if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
                if (KIND == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
                if (KIND == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
                if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(s[i]));
                if (KIND == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
                if (KIND == 8) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(s[i]) : "v"(a));
                if (KIND == 9) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(s[i]));
                if (KIND == 10) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
                if (KIND == 11) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(s[i]) : "s"(a));
                if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
                if (KIND == 13) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(s[i]) : "v"(t[i]), "s"(m));
                if (KIND == 14) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
                if (KIND == 15) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(s[i]));
            }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += s[i];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

// waves with (blockIdx.x & 1) == role 0 issue MFMAs only, the others VALU only (two workgroups of 256 threads per CU -> each SIMD holds one of each)
template <int MODE>   // 0: both, 1: MFMA waves only (others exit), 2: VALU waves only, 3: one wave does both interleaved (6 VALU per MFMA)
__global__ void mix(float* out, float a, float b) {
    const bool mfma_role = (blockIdx.x & 1) == 0;
    f32x16 acc0 = {0}, acc1 = {0};
    h8 A, B;
#pragma unroll
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16) (a + i); B[i] = (_Float16) (b + i); }
    float s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = a + i + threadIdx.x;
    if (MODE == 3) {
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[4 * q + (i & 3)]) : "v"(a));
            }
        }
    } else if (mfma_role) {
        if (MODE == 2) return;
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc1, 0, 0, 0);
            }
        }
    } else {
        if (MODE == 1) return;
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += s[i] + acc0[i] + acc1[i];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int KIND>
static void run(const char* name, float* d, double ghz) {
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
        const double per_simd = (double) REP * 64 * wps * (KIND == 2 ? 2 : 1);
        printf("%-28s %d waves/SIMD: %8.3f ms  -> %.2f cycles per instruction and SIMD (at %.2f GHz)\n", name, wps, ms, ms * 1e-3 * ghz * 1e9 / per_simd, ghz);
    }
}
template <int MODE>
static void run_mix(const char* name, float* d) {
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    mix<MODE><<<512, 256>>>(d, 1.0f, 0.5f);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    mix<MODE><<<512, 256>>>(d, 1.0f, 0.5f);
    (void) hipEventRecord(e1);
    (void) hipEventSynchronize(e1);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %8.3f ms\n", name, ms);
}

int main() {
    float* d;
    (void) hipMalloc(&d, 4096);
    const double ghz = 2.4;
    run<0>("v_cndmask_e32 vcc (s_mov)", d, ghz); run<1>("v_cndmask_e64 sgpr pair", d, ghz); run<13>("v_cndmask_e64 0, v, sgpr", d, ghz); run<2>("v_cmp + v_cndmask (pair)", d, ghz);
    run<3>("v_add_f32", d, ghz); run<4>("v_max_f32", d, ghz); run<5>("v_and_b32", d, ghz); run<6>("v_lshlrev_b32", d, ghz); run<7>("v_add_u32", d, ghz);
    run<8>("v_lshl_add_u32", d, ghz); run<9>("v_cvt_f32_i32", d, ghz); run<10>("v_min3_f32", d, ghz); run<11>("v_sub_f32 v, s, v", d, ghz); run<12>("v_fmac_f32", d, ghz);
    run<14>("v_mad_u32_u24", d, ghz); run<15>("v_bfe_u32", d, ghz);
    run_mix<1>("MFMA waves alone (4 x 4096 x 32x32x16 f16 per wave, 1 wave/SIMD)", d);
    run_mix<2>("VALU waves alone (32 x 4096 v_mul_f32 per wave, 1 wave/SIMD)", d);
    run_mix<0>("both on the same SIMDs", d);
    run_mix<3>("2 waves/SIMD each: MFMA + 6 v_mul_f32 interleaved (16384 MFMA per wave)", d);
    return 0;
}
