// Shared device/host helpers for the gfx950 kernels of libsfast_hip.so.
// wave = 64 lanes everywhere; no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sfast_hip.h"

namespace sfast {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char *fmt, ...);
void set_kernel_name(const char *fmt, ...);
int check_launch(const char *what);

#define SFAST_REQUIRE(cond, code, ...)  \
    do {                                \
        if (!(cond)) {                  \
            sfast::set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

// SFAST_BATCH_INVARIANT=1 (read once in sfast_hip_init; round 6): launch heuristics that size a grid from the
// BATCH -- statistics splits of the GroupNorm, the row blocks of the split-K reduce, the 32- vs 64-row attention kernel -- are evaluated
// for a reference batch of this many samples instead, so a sample's arithmetic does not depend on how many samples ride with it.
// 0 = off (every heuristic sees the real batch).
extern int g_batch_ref;

// ---- element types ------------------------------------------------------------------------------
using f16 = _Float16;
using bf16 = __bf16;

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Elem;
template <> struct Elem<f16> {
    using vec8 = f16x8;
    using vec4 = f16x4;
    using vec2 = f16x2;
    static __device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
    static __device__ __forceinline__ f16 from_f32(float v) { return (f16)v; }
};
template <> struct Elem<bf16> {
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    using vec2 = bf16x2;
    static __device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
    static __device__ __forceinline__ bf16 from_f32(float v) { return (bf16)v; }
};
template <> struct Elem<float> {
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};

// 16-byte chunk <-> 8 floats
template <typename T> __device__ __forceinline__ void unpack8(const u32x4 &raw, float (&f)[8]) {
    typename Elem<T>::vec8 v = __builtin_bit_cast(typename Elem<T>::vec8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    typename Elem<T>::vec8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = Elem<T>::from_f32(f[i]);
    return __builtin_bit_cast(u32x4, v);
}
template <typename T> __device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
    typename Elem<T>::vec4 v;
    v[0] = Elem<T>::from_f32(a);
    v[1] = Elem<T>::from_f32(b);
    v[2] = Elem<T>::from_f32(c);
    v[3] = Elem<T>::from_f32(d);
    return __builtin_bit_cast(u32x2, v);
}
template <typename T> __device__ __forceinline__ void unpack4(const u32x2 &raw, float (&f)[4]) {
    typename Elem<T>::vec4 v = __builtin_bit_cast(typename Elem<T>::vec4, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
}

// 16 bytes of zeros in device memory: staging loads of out-of-range chunks are redirected here so
// they stay unconditional (no exec-masked branch, no select on the loaded registers).
static __device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

// ---- activations (fp32) -------------------------------------------------------------------------
__device__ __forceinline__ float act_relu(float v) { return fmaxf(v, 0.0f); }
__device__ __forceinline__ float act_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float act_silu(float v) { return v * act_sigmoid(v); }
// Normal CDF through erfc (Abramowitz & Stegun 7.1.26, |abs err| <= 1.5e-7 -- below fp32 erff's own last-ulp
// noise once multiplied by v and rounded to f16/bf16): 2 transcendentals + 8 FMAs instead of libm erff's
// ~50-instruction two-branch polynomial. The GEGLU epilogue evaluates this 10M times per SD1.5 FF layer; with erff
// it was VALU-bound for ~40 % of the kernel (per-workgroup phase trace, profiles/r01_igemm_phase_trace.log).
// Negative arguments use erfc directly, so there is no 1 + erf(x) cancellation in the tail.
// Round 6: the same formula with the constants folded (z = |v| / sqrt(2) never formed: 0.3275911 / sqrt(2) = 0.2316419, the 0.5 in the
// coefficients, exp(-z^2) = exp2(-0.72134752 v^2)) and the sign select replaced by max(v, 0) - |v| h (v >= 0: v - v h, v < 0: v h):
// 11 full-rate VALU operations + 2 transcendentals per value instead of 16 + 2 -- the GEGLU epilogue is ~2.5 us of VALU work per
// tile round of the 256-row kernels (profiles/r06_pp_ksweep_run28.log: 5 us of fixed cost per round).
__device__ __forceinline__ float act_gelu_erf(float v) {
    const float a = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, a, 1.0f));
    float p = fmaf(0.5307027145f, t, -0.7265760135f);
    p = fmaf(p, t, 0.7107068705f);
    p = fmaf(p, t, -0.142248368f);
    p = fmaf(p, t, 0.127414796f);
    const float h = p * t * __builtin_amdgcn_exp2f(v * v * -0.72134752044448170368f);  // 0.5 * erfc(|v| / sqrt(2))
    return fmaf(-a, h, fmaxf(v, 0.0f));
}
// tanh from one exp + one rcp; |x| < 0.1 uses the odd series (the rational form cancels there)
__device__ __forceinline__ float act_tanh(float v) {
    const float x = fabsf(v);
    const float e = __builtin_amdgcn_exp2f(x * -2.88539008177792681472f);  // exp(-2|x|)
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    const float x2 = x * x;
    const float small = x * fmaf(x2, fmaf(x2, 0.13333333f, -0.33333333f), 1.0f);
    return copysignf(x < 0.1f ? small : big, v);
}
__device__ __forceinline__ float act_gelu_tanh(float v) {
    const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
    return 0.5f * v * (1.0f + act_tanh(k0 * (v + k1 * v * v * v)));
}
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
    case SFAST_ACT_RELU: return fmaxf(v, 0.0f);
    case SFAST_ACT_GELU: return act_gelu_erf(v);
    case SFAST_ACT_GELU_TANH: return act_gelu_tanh(v);
    case SFAST_ACT_SILU: return act_silu(v);
    case SFAST_ACT_SIGMOID: return act_sigmoid(v);
    case SFAST_ACT_TANH: return act_tanh(v);
    default: return v;
    }
}

// ---- wave-level reductions (64 lanes) ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
static inline bool aligned8(const void *p) { return (((uintptr_t)p) & 7) == 0; }

static inline size_t dtype_bytes(int dtype) { return dtype == SFAST_F32 ? 4 : 2; }

}  // namespace sfast
