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
// acc[fn][fm] holds D[n][m] fragments (32x32 C/D layout). For each 32-row activation fragment the
// epilogue runs in two phases so the memory system sees ONE batch of loads instead of a chain of
// load -> wait -> store groups: (1) every bias / row-bias / residual vector of the fragment column is
// requested unconditionally (absent operands and out-of-range groups are redirected to the device zero
// block, so there is no branch between the loads and no select on their results); (2) fp32 epilogue
// math and the 8-byte stores. In-place residuals (out == res) stay correct: a group's residual is read
// in phase 1 and only that group's lanes write it in phase 2.
template <typename T, int FN, int FM, bool GEGLU>
__device__ __forceinline__ void epilogue_tile(const IgemmArgs &a, f32x16 (&acc)[FN][FM], int mbase, int nbase, int l31, int hi,
                                              int split_idx) {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    const g2_ptr zero = (g2_ptr)(const void *)g_zero16;
    const bool partial = a.splits > 1;
    constexpr int FH = GEGLU ? FN / 2 : FN;  // output fragments along n
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
                        constexpr int o = GEGLU ? FN / 2 : 0;
                        *reinterpret_cast<f32x4 *>(p + a.N + n) = f32x4{acc[fh + o][fm][4 * g], acc[fh + o][fm][4 * g + 1],
                                                                         acc[fh + o][fm][4 * g + 2], acc[fh + o][fm][4 * g + 3]};
                    }
                }
            continue;
        }
        // ---- phase 1: batched operand loads ----------------------------------------------------------
        u32x2 vb[FH][4], vb2[FH][4], vr[FH][4];
        const int bi = a.rowbias ? m / a.rows_per_batch : 0;
#pragma unroll
        for (int fh = 0; fh < FH; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                const bool ok = mok && n < a.N;
                const g2_ptr pb = (ok && a.bias) ? (g2_ptr)(const void *)((const T *)a.bias + n) : zero;
                vb[fh][g] = *pb;
                if (GEGLU) {
                    const g2_ptr pg = (ok && a.bias) ? (g2_ptr)(const void *)((const T *)a.bias + a.N + n) : zero;
                    vb2[fh][g] = *pg;
                } else {
                    const g2_ptr prb = (ok && a.rowbias) ? (g2_ptr)(const void *)((const T *)a.rowbias + (int64_t)bi * a.ld_rowbias + n) : zero;
                    vb2[fh][g] = *prb;
                    const g2_ptr pr = (ok && a.res) ? (g2_ptr)(const void *)((const T *)a.res + (int64_t)m * a.ldr + n) : zero;
                    vr[fh][g] = *pr;
                }
            }
        // ---- phase 2: fp32 math + stores -----------------------------------------------------------------
#pragma unroll
        for (int fh = 0; fh < FH; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                float v[4], b0[4], b1[4];
                unpack4<T>(vb[fh][g], b0);
                unpack4<T>(vb2[fh][g], b1);
                if (GEGLU) {
                    constexpr int o = GEGLU ? FN / 2 : 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float h = acc[fh][fm][4 * g + i] + b0[i];
                        const float gt = acc[fh + o][fm][4 * g + i] + b1[i];
                        v[i] = h * act_gelu_erf(gt);
                    }
                } else {
                    float r[4];
                    unpack4<T>(vr[fh][g], r);
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

}  // namespace sfast
