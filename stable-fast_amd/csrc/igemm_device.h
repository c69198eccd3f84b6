// Device helpers shared by the MFMA implicit-GEMM kernels (igemm.hip, igemm_glds.hip):
// MFMA wrappers, the LDS tile swizzle and the fused epilogues.
#pragma once
#include "igemm.h"

namespace sfast {

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// byte offset of 16-B chunk `chunk` (0..7) of tile row `row` (128 B per row), XOR-swizzled
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// final epilogue for 4 consecutive columns n..n+3 of row m (fp32 in, T out)
template <typename T>
__device__ __forceinline__ void epilogue4(const IgemmArgs &a, int m, int n, float (&v)[4]) {
    if (a.bias) {
        float b[4];
        unpack4<T>(*reinterpret_cast<const u32x2 *>((const T *)a.bias + n), b);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += b[i];
    }
    if (a.rowbias) {
        float b[4];
        const int bi = m / a.rows_per_batch;
        unpack4<T>(*reinterpret_cast<const u32x2 *>((const T *)a.rowbias + (int64_t)bi * a.ld_rowbias + n), b);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += b[i];
    }
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.res) {
        unpack4<T>(*reinterpret_cast<const u32x2 *>((const T *)a.res + (int64_t)m * a.ldr + n), r);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] *= a.alpha;
    }
    if (a.res_before_act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += r[i];
    }
    if (a.act != SFAST_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], a.act);
    }
    if (!a.res_before_act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += r[i];
    }
    *reinterpret_cast<u32x2 *>((T *)a.out + (int64_t)m * a.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
}

template <typename T>
__device__ __forceinline__ void epilogue4_geglu(const IgemmArgs &a, int m, int n, float (&h)[4], float (&g)[4]) {
    if (a.bias) {
        float bh[4], bg[4];
        unpack4<T>(*reinterpret_cast<const u32x2 *>((const T *)a.bias + n), bh);
        unpack4<T>(*reinterpret_cast<const u32x2 *>((const T *)a.bias + a.N + n), bg);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] += bh[i];
            g[i] += bg[i];
        }
    }
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = h[i] * act_gelu_erf(g[i]);
    *reinterpret_cast<u32x2 *>((T *)a.out + (int64_t)m * a.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
}


// ---- whole-tile epilogue ---------------------------------------------------------------------------------
// acc[fn][fm] holds D[n][m] fragments (32x32 C/D layout). Every bias / row-bias / residual vector of the
// tile is requested unconditionally in ONE batch (absent operands and out-of-range groups are redirected
// to the device zero block: no branch between the loads, no select on their results), then fp32 epilogue
// math and 8-byte stores. In-place residuals (out == res) stay correct: a group's residual is read by the
// same lane that later writes it, and no other workgroup touches it.
// The operand vectors are fetched by epilogue_prefetch() BEFORE the K loop (their latency hides under the
// whole main loop instead of being paid once more at the end of every workgroup -- the short-K GEMMs of
// the transformer blocks are chains of exposed memory round trips otherwise) and consumed by
// epilogue_finish(). Split-K launches skip both (the reduce kernel applies the epilogue).
template <int FH, int FM> struct EpiOperands {
    u32x2 vb[FH][4];       // bias (GEGLU: bias of the h half)
    u32x2 vb2[FM][FH][4];  // row-bias (GEGLU: bias of the g half, index [0])
    u32x2 vr[FM][FH][4];   // residual
};

template <typename T, int FN, int FM, bool GEGLU>
__device__ __forceinline__ void epilogue_prefetch(const IgemmArgs &a, EpiOperands<(GEGLU ? FN / 2 : FN), FM> &e, int mbase, int nbase,
                                                  int l31, int hi) {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    const g2_ptr zero = (g2_ptr)(const void *)g_zero16;
    constexpr int FH = GEGLU ? FN / 2 : FN;
    if (a.splits > 1) return;
#pragma unroll
    for (int fh = 0; fh < FH; ++fh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nbase + fh * 32 + 8 * g + 4 * hi;
            const bool nok = n < a.N;
            e.vb[fh][g] = *((nok && a.bias) ? (g2_ptr)(const void *)((const T *)a.bias + n) : zero);
            if (GEGLU) e.vb2[0][fh][g] = *((nok && a.bias) ? (g2_ptr)(const void *)((const T *)a.bias + a.N + n) : zero);
        }
    if (GEGLU) return;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mbase + fm * 32 + l31;
        const bool mok = m < a.M;
        const int bi = a.rowbias ? m / a.rows_per_batch : 0;
#pragma unroll
        for (int fh = 0; fh < FH; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                const bool ok = mok && n < a.N;
                e.vb2[fm][fh][g] = *((ok && a.rowbias) ? (g2_ptr)(const void *)((const T *)a.rowbias + (int64_t)bi * a.ld_rowbias + n) : zero);
                e.vr[fm][fh][g] = *((ok && a.res) ? (g2_ptr)(const void *)((const T *)a.res + (int64_t)m * a.ldr + n) : zero);
            }
    }
}

template <typename T, int FN, int FM, bool GEGLU>
__device__ __forceinline__ void epilogue_finish(const IgemmArgs &a, f32x16 (&acc)[FN][FM], const EpiOperands<(GEGLU ? FN / 2 : FN), FM> &e,
                                                int mbase, int nbase, int l31, int hi, int split_idx) {
    const bool partial = a.splits > 1;
    constexpr int FH = GEGLU ? FN / 2 : FN;  // output fragments along n
    constexpr int o = GEGLU ? FN / 2 : 0;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mbase + fm * 32 + l31;
        const bool mok = m < a.M;
        if (partial) {
            if (!mok) continue;
            const int64_t np = GEGLU ? 2 * (int64_t)a.N : (int64_t)a.N;
            float *p = a.partial + ((int64_t)split_idx * a.M + m) * np;
#pragma unroll
            for (int fh = 0; fh < FH; ++fh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                    if (n >= a.N) continue;
                    *reinterpret_cast<f32x4 *>(p + n) =
                        f32x4{acc[fh][fm][4 * g], acc[fh][fm][4 * g + 1], acc[fh][fm][4 * g + 2], acc[fh][fm][4 * g + 3]};
                    if (GEGLU) {
                        *reinterpret_cast<f32x4 *>(p + a.N + n) = f32x4{acc[fh + o][fm][4 * g], acc[fh + o][fm][4 * g + 1],
                                                                         acc[fh + o][fm][4 * g + 2], acc[fh + o][fm][4 * g + 3]};
                    }
                }
            continue;
        }
#pragma unroll
        for (int fh = 0; fh < FH; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                float v[4], b0[4], b1[4];
                unpack4<T>(e.vb[fh][g], b0);
                unpack4<T>(e.vb2[GEGLU ? 0 : fm][fh][g], b1);
                if (GEGLU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float h = acc[fh][fm][4 * g + i] + b0[i];
                        const float gt = acc[fh + o][fm][4 * g + i] + b1[i];
                        v[i] = h * act_gelu_erf(gt);
                    }
                } else {
                    float r[4];
                    unpack4<T>(e.vr[fm][fh][g], r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = acc[fh][fm][4 * g + i] + b0[i] + b1[i];
                        const float rr = r[i] * a.alpha;
                        if (a.res_before_act) t += rr;
                        if (a.act != SFAST_ACT_NONE) t = apply_act(t, a.act);
                        if (!a.res_before_act) t += rr;
                        v[i] = t;
                    }
                }
                if (mok && n < a.N) *reinterpret_cast<u32x2 *>((T *)a.out + (int64_t)m * a.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
    }
}

// Late form for the 5-fragment tiles: operands are fetched fragment by fragment right before use (one batch of
// 12 vectors per 32x32 fragment), which keeps the epilogue inside the register budget of two waves per SIMD.
template <typename T, int FN, int FM>
__device__ __forceinline__ void epilogue_late(const IgemmArgs &a, f32x16 (&acc)[FN][FM], int mbase, int nbase, int l31, int hi,
                                              int split_idx) {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    const g2_ptr zero = (g2_ptr)(const void *)g_zero16;
    if (a.splits > 1) {
        EpiOperands<FN, FM> *none = nullptr;
        (void)none;
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = mbase + fm * 32 + l31;
            if (m >= a.M) continue;
            float *p = a.partial + ((int64_t)split_idx * a.M + m) * (int64_t)a.N;
#pragma unroll
            for (int fh = 0; fh < FN; ++fh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                    if (n >= a.N) continue;
                    *reinterpret_cast<f32x4 *>(p + n) =
                        f32x4{acc[fh][fm][4 * g], acc[fh][fm][4 * g + 1], acc[fh][fm][4 * g + 2], acc[fh][fm][4 * g + 3]};
                }
        }
        return;
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mbase + fm * 32 + l31;
        const bool mok = m < a.M;
        const int bi = a.rowbias ? m / a.rows_per_batch : 0;
#pragma unroll
        for (int fh = 0; fh < FN; ++fh) {
            u32x2 vb[4], vb2[4], vr[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                const bool ok = mok && n < a.N;
                vb[g] = *((ok && a.bias) ? (g2_ptr)(const void *)((const T *)a.bias + n) : zero);
                vb2[g] = *((ok && a.rowbias) ? (g2_ptr)(const void *)((const T *)a.rowbias + (int64_t)bi * a.ld_rowbias + n) : zero);
                vr[g] = *((ok && a.res) ? (g2_ptr)(const void *)((const T *)a.res + (int64_t)m * a.ldr + n) : zero);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                float v[4], b0[4], b1[4], r[4];
                unpack4<T>(vb[g], b0);
                unpack4<T>(vb2[g], b1);
                unpack4<T>(vr[g], r);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = acc[fh][fm][4 * g + i] + b0[i] + b1[i];
                    const float rr = r[i] * a.alpha;
                    if (a.res_before_act) t += rr;
                    if (a.act != SFAST_ACT_NONE) t = apply_act(t, a.act);
                    if (!a.res_before_act) t += rr;
                    v[i] = t;
                }
                if (mok && n < a.N) *reinterpret_cast<u32x2 *>((T *)a.out + (int64_t)m * a.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

}  // namespace sfast
