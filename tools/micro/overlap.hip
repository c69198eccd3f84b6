// Micro-probe: which VALU instructions overlap with v_mfma_f32_32x32x16_f16 on gfx950 -- from ANOTHER wave of the same SIMD
// (block = 8 waves: waves 0-3 run role A, waves 4-7 role B, one of each per SIMD) and INSIDE one wave (1 MFMA : 5 fillers).
// Everything is volatile inline asm so neither SLP packing nor the scheduler can move it.
// build: hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip ; run: ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

enum { IDLE, MFMA, FMA, PKFMA, EXP, MAX3, CVT, MUL, IL_FMA, IL_EXP, IL_PK, IL_MAX3, IL_CVT };

#define MF(c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
template <int OP> __device__ __forceinline__ void filler(float &x, f2v &p, float k, float s) {
    if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(s));
    if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(k));
    if constexpr (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
    if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if constexpr (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(s));
    if constexpr (OP == CVT) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(k));
}

template <int ROLE> __device__ __forceinline__ float work(int iters, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    f16v c0 = {}, c1 = {};
    float v[8];
    f2v p[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed * i; p[i] = f2v{seed, seed * i}; }
    const float k = 1.0001f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (ROLE == MFMA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { MF(c0); MF(c1); }
        } else if constexpr (ROLE >= FMA && ROLE <= MUL) {
#pragma unroll
            for (int j = 0; j < 40; ++j) filler<ROLE>(v[j & 7], p[j & 7], k, seed);
        } else if constexpr (ROLE >= IL_FMA) {
            constexpr int OP = ROLE == IL_FMA ? FMA : ROLE == IL_EXP ? EXP : ROLE == IL_PK ? PKFMA : ROLE == IL_MAX3 ? MAX3 : CVT;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j & 1) MF(c1); else MF(c0);
#pragma unroll
                for (int q = 0; q < 5; ++q) filler<OP>(v[(j * 5 + q) & 7], p[(j * 5 + q) & 7], k, seed);
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) r += v[i] + p[i][0] + p[i][1];
    return r;
}

template <int RA, int RB> __global__ void __launch_bounds__(512) probe(float *out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    float r;
    if (wave < 4) r = work<RA>(iters, seed);
    else r = work<RB>(iters, seed);
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int RA, int RB> static void run(const char *name, float *out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    probe<RA, RB><<<256, 512>>>(out, iters, 0.5f);
    (void)hipEventRecord(e0);
    probe<RA, RB><<<256, 512>>>(out, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.1f us  %6.0f ns/iter\n", name, ms * 1e3, ms * 1e6 / iters);
}

int main() {
    float *out; (void)hipMalloc(&out, 4096);
    printf("per iteration: role MFMA = 8 MFMAs, VALU roles = 40 instructions, IL = 8 x (1 MFMA + 5 fillers)\n");
    run<MFMA, IDLE>("MFMA | idle", out);
    run<MFMA, MFMA>("MFMA | MFMA", out);
    run<FMA, IDLE>("v_fma_f32 | idle", out);
    run<FMA, FMA>("v_fma_f32 | v_fma_f32", out);
    run<MUL, MUL>("v_mul_f32 | v_mul_f32", out);
    run<PKFMA, PKFMA>("v_pk_fma_f32 | v_pk_fma_f32", out);
    run<EXP, EXP>("v_exp_f32 | v_exp_f32", out);
    run<MAX3, MAX3>("v_max3_f32 | v_max3_f32", out);
    run<CVT, CVT>("v_cvt_pkrtz | v_cvt_pkrtz", out);
    run<MFMA, FMA>("MFMA | v_fma_f32", out);
    run<MFMA, MUL>("MFMA | v_mul_f32", out);
    run<MFMA, PKFMA>("MFMA | v_pk_fma_f32", out);
    run<MFMA, EXP>("MFMA | v_exp_f32", out);
    run<MFMA, MAX3>("MFMA | v_max3_f32", out);
    run<MFMA, CVT>("MFMA | v_cvt_pkrtz", out);
    run<IL_FMA, IDLE>("in-wave 1 MFMA : 5 v_fma_f32 | idle", out);
    run<IL_EXP, IDLE>("in-wave 1 MFMA : 5 v_exp_f32 | idle", out);
    run<IL_PK, IDLE>("in-wave 1 MFMA : 5 v_pk_fma_f32 | idle", out);
    run<IL_MAX3, IDLE>("in-wave 1 MFMA : 5 v_max3_f32 | idle", out);
    run<IL_CVT, IDLE>("in-wave 1 MFMA : 5 v_cvt_pkrtz | idle", out);
    run<IL_FMA, IL_FMA>("in-wave 1 MFMA : 5 v_fma_f32 | same", out);
    return 0;
}
