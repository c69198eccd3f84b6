// Device helpers shared by the MFMA implicit-GEMM kernels (igemm.hip, igemm_glds.hip):
// MFMA wrappers, the LDS tile swizzle and the fused epilogues.
#pragma once
#include "igemm.h"

namespace sfast {

// Profiling only (sfast_hip_set_trace): thread 0 of every workgroup stamps the 100 MHz wall clock into
// slot `slot` of its 16-slot record. One uniform scalar branch when tracing is off.
__device__ __forceinline__ void trace_mark(const IgemmArgs &a, int slot) {
    if (a.trace != nullptr) {
        if (threadIdx.x == 0) {
            unsigned long long *rec = a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
            rec[slot] = wall_clock64();
            if (slot == 0) rec[8] = clock64();  // shader-clock counter: (rec[9] - rec[8]) / wall time = effective clock
        }
    }
}
__device__ __forceinline__ void trace_finish(const IgemmArgs &a) {
    if (a.trace != nullptr) {
        trace_mark(a, 5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trace_mark(a, 6);
        if (threadIdx.x == 0) {
            unsigned long long *rec = a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
            rec[9] = clock64();
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
            rec[7] = ((unsigned long long)xcc << 32) | hw;
        }
    }
}
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

// Waves per SIMD the kernel is compiled for (second __launch_bounds__ argument): what the workgroup's LDS footprint
// lets a CU hold, capped at 3 -- the register allocator then stays inside 512 / that many registers instead of
// trading occupancy for scheduling freedom in the epilogue.
constexpr int igemm_min_waves(int threads, int lds_bytes) {
    const int w = (163840 / lds_bytes) * (threads / 256);
    return w < 1 ? 1 : (w > 3 ? 3 : w);
}

// Requests the whole argument block in one burst of scalar loads behind ONE wait. Left alone, the compiler fetches
// kernarg fields lazily, group by group, each group behind its own `s_waitcnt lgkmcnt(0)`: six serialized scalar
// round trips (~1 us) at the top of every workgroup of a 6-13 us kernel.
__device__ __forceinline__ void touch_args(const IgemmArgs &a) {
    asm volatile("" ::"s"(a.x), "s"(a.x2), "s"(a.w[0]), "s"(a.w[1]), "s"(a.w[2]), "s"(a.w[3]), "s"(a.bias), "s"(a.rowbias), "s"(a.res),
                 "s"(a.out), "s"(a.partial), "s"(a.M), "s"(a.N), "s"(a.K), "s"(a.ldx), "s"(a.ldw), "s"(a.ldo), "s"(a.ldr),
                 "s"(a.ld_rowbias), "s"(a.rows_per_seg), "s"(a.rows_per_batch), "s"(a.act), "s"(a.res_before_act), "s"(a.alpha));
    asm volatile("" ::"s"(a.tiles_m), "s"(a.tiles_n), "s"(a.ktiles), "s"(a.ktiles_per_split), "s"(a.splits), "s"(a.trace));
    asm volatile("" ::"s"(a.xmap), "s"(a.x_lxn), "s"(a.x_lxm), "s"(a.x_tn), "s"(a.x_tm), "s"(a.x_sp), "s"(a.x_per), "s"(a.x_order));
}
__device__ __forceinline__ void touch_conv_args(const IgemmArgs &a) {
    asm volatile("" ::"s"(a.H), "s"(a.W), "s"(a.C1), "s"(a.C2), "s"(a.Ho), "s"(a.Wo), "s"(a.KH), "s"(a.KW), "s"(a.stride_h), "s"(a.stride_w),
                 "s"(a.pad_h), "s"(a.pad_w), "s"(a.dil_h), "s"(a.dil_w), "s"(a.ups));
}

// floor(n / d) for 0 <= n <= 2^22 and d >= 1 with a precomputed rd = rcp((float)d): float(n) is exact and the
// quotient estimate is off by at most one (n/d * 2^-22.4 < 1), fixed by one compare each way. ~8 VALU ops instead of
// the ~40 of the integer sequence; the per-row pixel decode at the top of a conv workgroup is pure latency at one
// workgroup per CU (measured 4 us of a 30 us conv, profiles/r01_igemm_phase_trace.log).
__device__ __forceinline__ int fdiv22(int n, int d, float rd) {
    int q = (int)((float)n * rd);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}
struct PixelDecoder {
    int hw, wo;
    float r_hw, r_wo;
    bool fast;
    __device__ __forceinline__ PixelDecoder(const IgemmArgs &a)
        : hw(a.Ho * a.Wo), wo(a.Wo), r_hw(__builtin_amdgcn_rcpf((float)(a.Ho * a.Wo))), r_wo(__builtin_amdgcn_rcpf((float)a.Wo)),
          fast(a.M <= (1 << 22)) {}
    __device__ __forceinline__ void operator()(int m, int &b, int &ho, int &wo_) const {
        if (fast) {
            b = fdiv22(m, hw, r_hw);
            const int rem = m - b * hw;
            ho = fdiv22(rem, wo, r_wo);
            wo_ = rem - ho * wo;
        } else {
            b = m / hw;
            const int rem = m - b * hw;
            ho = rem / wo;
            wo_ = rem - ho * wo;
        }
    }
};
// ---- which tile / K-split a workgroup computes ------------------------------------------------------------------
// The hardware places block b on XCD b % 8 and each XCD has its own L2, so what an XCD's blocks have in common decides how often
// an operand crosses the fabric: a weight element is fetched once per XCD that owns a tile in its column box and K-split, an
// activation element once per XCD that owns a tile in its row box and K-split.
//  * xmap = 1 (host: choose_xcd_map picks the factorisation of 8 into K-split x row x column boxes that minimises
//    activation_bytes * column_boxes + weight_bytes * row_boxes; K-split boxes replicate nothing): inside a box the order is
//    tile_n fastest, then tile_m, then split.
//  * xmap = 0 (tile counts that do not divide): each XCD a contiguous run of row-major tiles, splits on blockIdx.y.
struct BlockTile {
    int tile_m, tile_n, split;
};
__device__ __forceinline__ BlockTile decode_block(const IgemmArgs &a) {
    BlockTile t;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, k = bid >> 3;
#ifdef SFAST_PROBES
    if (a.xmap == 2) {
        const int lid = xcd * a.x_per + k;
        if (lid >= a.tiles_m * a.tiles_n * a.splits) {  // surplus block of the last run(s): nothing to do (tile_m < 0, every kernel returns)
            t.tile_m = t.tile_n = t.split = -1;
            return t;
        }
        const int d1 = a.x_order ? a.tiles_n : a.tiles_m, d2 = a.x_order ? a.tiles_m : a.tiles_n;
        const int q = fdiv22(lid, d1, __builtin_amdgcn_rcpf((float)d1));
        const int s = fdiv22(q, d2, __builtin_amdgcn_rcpf((float)d2));
        const int i1 = lid - q * d1, i2 = q - s * d2;
        t.tile_m = a.x_order ? i2 : i1;
        t.tile_n = a.x_order ? i1 : i2;
        t.split = s;
    } else
#endif
    if (a.xmap) {
        const int jn = xcd & ((1 << a.x_lxn) - 1);
        const int im = (xcd >> a.x_lxn) & ((1 << a.x_lxm) - 1);
        const int is = xcd >> (a.x_lxn + a.x_lxm);
        const int q = fdiv22(k, a.x_tn, __builtin_amdgcn_rcpf((float)a.x_tn));  // grids stay far below 2^22 blocks
        const int s = fdiv22(q, a.x_tm, __builtin_amdgcn_rcpf((float)a.x_tm));
        t.tile_n = jn * a.x_tn + (k - q * a.x_tn);
        t.tile_m = im * a.x_tm + (q - s * a.x_tm);
        t.split = is * a.x_sp + s;
    } else {
        const int nblk = a.tiles_m * a.tiles_n;
        const int q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        t.tile_m = lid / a.tiles_n;
        t.tile_n = lid - t.tile_m * a.tiles_n;
        t.split = blockIdx.y;
    }
    // everything above is wave-uniform; the float reciprocal runs on the VALU, so pin the results back into SGPRs
    t.tile_m = __builtin_amdgcn_readfirstlane(t.tile_m);
    t.tile_n = __builtin_amdgcn_readfirstlane(t.tile_n);
    t.split = __builtin_amdgcn_readfirstlane(t.split);
    return t;
}

// row -> batch index for the per-batch row bias (time-embedding projection): same trick
struct BatchOfRow {
    int d;
    float rd;
    bool fast;
    __device__ __forceinline__ BatchOfRow(const IgemmArgs &a)
        : d(a.rows_per_batch > 0 ? a.rows_per_batch : 1), rd(__builtin_amdgcn_rcpf((float)(a.rows_per_batch > 0 ? a.rows_per_batch : 1))),
          fast(a.M <= (1 << 22)) {}
    __device__ __forceinline__ int operator()(int m) const { return fast ? fdiv22(m, d, rd) : m / d; }
};

// Activations of the MFMA GEMM epilogue run as one ROLLED 16-element loop per 32x32 fragment (uniform dynamic index
// into the fragment's registers). A per-element `switch (a.act)` inlined 64x made every kernel ~100 KB of code --
// more than the 64 KB instruction cache, with the executed path scattered across it -- and the UNet path itself
// only ever runs the no-activation form here (SiLU is fused into GroupNorm, GELU into the GEGLU kernel).
template <int FN, int FM, int FH> __device__ __forceinline__ void apply_act_tile(f32x16 (&acc)[FN][FM], int act) {
#pragma unroll
    for (int fh = 0; fh < FH; ++fh)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
#pragma unroll 1
            for (int r = 0; r < 16; ++r) acc[fh][fm][r] = apply_act(acc[fh][fm][r], act);
        }
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
// Epilogue operand addressing. One row base per operand and per 32-row fragment, with absence folded in (a missing
// operand or an out-of-range row points at the device zero block); a load is then base + (in range ? n : 0) halves:
// a predicate AND, a select and one 64-bit add. The first version re-derived `present && in range ? ptr : zero`
// per load -- ~30 instructions each, 1800 for a 128x160 tile, most of a 2 us workgroup prologue.
template <typename T> struct EpiRow {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    typedef const T __attribute__((address_space(1))) * gT_ptr;
    gT_ptr base;
    bool have;
    __device__ __forceinline__ EpiRow(const void *p, int64_t elem_off, bool row_ok) {
        have = row_ok & (p != nullptr);
        base = have ? (gT_ptr)p + elem_off : (gT_ptr)(const void *)g_zero16;
    }
    __device__ __forceinline__ u32x2 load(int n, bool nok) const { return *(g2_ptr)(base + ((have & nok) ? n : 0)); }
};

template <int FH, int FM> struct EpiOperands {
    u32x2 vb[FH][4];       // bias (GEGLU: bias of the h half)
    u32x2 vb2[FM][FH][4];  // row-bias (GEGLU: bias of the g half, index [0])
    u32x2 vr[FM][FH][4];   // residual
};

template <typename T, int FN, int FM, bool GEGLU>
__device__ __forceinline__ void epilogue_prefetch(const IgemmArgs &a, EpiOperands<(GEGLU ? FN / 2 : FN), FM> &e, int mbase, int nbase,
                                                  int l31, int hi, bool whole = false) {
    // whole: this workgroup holds the sum over all K-splits (splitk_join) -- the epilogue runs although a.splits > 1
    constexpr int FH = GEGLU ? FN / 2 : FN;
    if (a.splits > 1 && !whole) return;
    const EpiRow<T> bias(a.bias, 0, true), bias_g(a.bias, a.N, true);
#pragma unroll
    for (int fh = 0; fh < FH; ++fh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nbase + fh * 32 + 8 * g + 4 * hi;
            const bool nok = n < a.N;
            e.vb[fh][g] = bias.load(n, nok);
            if (GEGLU) e.vb2[0][fh][g] = bias_g.load(n, nok);
        }
    if (GEGLU) return;
    const BatchOfRow batch_of(a);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mbase + fm * 32 + l31;
        const bool mok = m < a.M;
        const int bi = a.rowbias ? batch_of(m) : 0;
        const EpiRow<T> rowbias(a.rowbias, (int64_t)bi * a.ld_rowbias, mok), res(a.res, (int64_t)m * a.ldr, mok);
#pragma unroll
        for (int fh = 0; fh < FH; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                const bool nok = n < a.N;
                e.vb2[fm][fh][g] = rowbias.load(n, nok);
                e.vr[fm][fh][g] = res.load(n, nok);
            }
    }
}

// fp32 slab store of one workgroup's partial tile (split-K); the reduce kernel applies the epilogue
template <int FN, int FM, bool GEGLU>
__device__ __forceinline__ void store_partial(const IgemmArgs &a, f32x16 (&acc)[FN][FM], int mbase, int nbase, int l31, int hi, int split_idx) {
    constexpr int FH = GEGLU ? FN / 2 : FN;
    constexpr int o = GEGLU ? FN / 2 : 0;
    const int64_t np = GEGLU ? 2 * (int64_t)a.N : (int64_t)a.N;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mbase + fm * 32 + l31;
        if (m >= a.M) continue;
        float *p = a.partial + ((int64_t)split_idx * a.M + m) * np;
#pragma unroll
        for (int fh = 0; fh < FH; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                if (n >= a.N) continue;
                *reinterpret_cast<f32x4 *>(p + n) = f32x4{acc[fh][fm][4 * g], acc[fh][fm][4 * g + 1], acc[fh][fm][4 * g + 2], acc[fh][fm][4 * g + 3]};
                if (GEGLU) {
                    *reinterpret_cast<f32x4 *>(p + a.N + n) =
                        f32x4{acc[fh + o][fm][4 * g], acc[fh + o][fm][4 * g + 1], acc[fh + o][fm][4 * g + 2], acc[fh + o][fm][4 * g + 3]};
                }
            }
    }
}

// ---- where a finished group of 4 output values goes ------------------------------------------------------------
// STAGED = false: straight to global memory (8 bytes per lane, one row per lane: 32 row segments per store instruction).
// STAGED = true : into an LDS image of the output tile (row stride BNO*2 + 8 bytes: the 16 lanes of a ds_write_b64 group hit
//                 16 different 8-byte bank pairs); flush_staged_tile() then writes whole 16-byte chunks of contiguous row
//                 segments and, on request, reduces the tile to GroupNorm partial statistics on the way out.
struct StageCtx {
    char *lds;
    int m0, n0, ldr;
};
template <typename T, bool STAGED>
__device__ __forceinline__ void put4(const IgemmArgs &a, const StageCtx &sc, int m, int n, bool ok, u32x2 v) {
    if constexpr (STAGED) {
        *reinterpret_cast<u32x2 *>(sc.lds + (m - sc.m0) * sc.ldr + (n - sc.n0) * 2) = v;
    } else {
        if (ok) *reinterpret_cast<u32x2 *>((T *)a.out + (int64_t)m * a.ldo + n) = v;
    }
}

// Pass 1 over one 32x32 fragment: acc = out_scale*acc + bias + row-bias (+ residual when it belongs before the activation, or
// when there is no activation and the order is moot). Without an activation the fragment is finished and stored here.
template <typename T, bool STAGED>
__device__ __forceinline__ void fragment_pass1(const IgemmArgs &a, const StageCtx &sc, f32x16 &acc, const u32x2 (&vb)[4],
                                               const u32x2 (&vb2)[4], const u32x2 (&vr)[4], bool mok, int m, int nfrag, int hi) {
    const bool res_now = a.res_before_act || a.act == SFAST_ACT_NONE;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float b0[4], b1[4], r[4];
        unpack4<T>(vb[g], b0);
        unpack4<T>(vb2[g], b1);
        unpack4<T>(vr[g], r);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * g + i] = fmaf(acc[4 * g + i], a.out_scale, b0[i]) + b1[i] + (res_now ? r[i] * a.alpha : -0.0f);
    }
    if (a.act == SFAST_ACT_NONE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nfrag + 8 * g + 4 * hi;
            put4<T, STAGED>(a, sc, m, n, mok && n < a.N, pack4<T>(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]));
        }
    }
}
// Pass 2 (only with an activation): add the residual that belongs after it, store.
template <typename T, bool STAGED>
__device__ __forceinline__ void fragment_pass2(const IgemmArgs &a, const StageCtx &sc, const f32x16 &acc, const u32x2 (&vr)[4], bool mok, int m,
                                               int nfrag, int hi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float r[4];
        unpack4<T>(vr[g], r);
        const int n = nfrag + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[4 * g + i] + (a.res_before_act ? -0.0f : r[i] * a.alpha);
        put4<T, STAGED>(a, sc, m, n, mok && n < a.N, pack4<T>(v[0], v[1], v[2], v[3]));
    }
}

// Activation tail (only when a.act != NONE): one switch over the tile, then the residual that belongs AFTER the
// activation is fetched again, one fragment ahead, and the tile is stored. Re-fetching instead of keeping the
// prefetched residual live across the switch keeps the register peak of this rare path out of the kernel's budget.
template <typename T, int FN, int FM, bool STAGED>
__device__ __forceinline__ void epilogue_act_tail(const IgemmArgs &a, const StageCtx &sc, f32x16 (&acc)[FN][FM], int mbase, int nbase, int l31,
                                                  int hi) {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    const g2_ptr zero = (g2_ptr)(const void *)g_zero16;
    apply_act_tile<FN, FM, FN>(acc, a.act);
    const bool post = a.res != nullptr && !a.res_before_act;
    u32x2 vr[2][4];
    auto fetch = [&](int f, int buf) {
        const int fm = f / FN, fh = f % FN;
        const int m = mbase + fm * 32 + l31;
        const EpiRow<T> res(a.res, (int64_t)m * a.ldr, post && m < a.M);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nbase + fh * 32 + 8 * g + 4 * hi;
            vr[buf][g] = res.load(n, n < a.N);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int f = 0; f < FN * FM; ++f) {
        const int fm = f / FN, fh = f % FN;
        if (f + 1 < FN * FM) fetch(f + 1, (f + 1) & 1);
        const int m = mbase + fm * 32 + l31;
        fragment_pass2<T, STAGED>(a, sc, acc[fh][fm], vr[f & 1], m < a.M, m, nbase + fh * 32, hi);
    }
}

template <typename T, int FN, int FM, bool GEGLU, bool STAGED>
__device__ __forceinline__ void epilogue_finish(const IgemmArgs &a, const StageCtx &sc, f32x16 (&acc)[FN][FM],
                                                const EpiOperands<(GEGLU ? FN / 2 : FN), FM> &e, int mbase, int nbase, int l31, int hi,
                                                int split_idx, bool whole) {
    if (a.splits > 1 && !whole) {
        store_partial<FN, FM, GEGLU>(a, acc, mbase, nbase, l31, hi, split_idx);
        return;
    }
    constexpr int FH = GEGLU ? FN / 2 : FN;  // output fragments along n
    constexpr int o = GEGLU ? FN / 2 : 0;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mbase + fm * 32 + l31;
        const bool mok = m < a.M;
#pragma unroll
        for (int fh = 0; fh < FH; ++fh) {
            if (GEGLU) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                    float v[4], b0[4], b1[4];
                    unpack4<T>(e.vb[fh][g], b0);
                    unpack4<T>(e.vb2[0][fh][g], b1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float h = acc[fh][fm][4 * g + i] + b0[i];
                        const float gt = acc[fh + o][fm][4 * g + i] + b1[i];
                        v[i] = h * act_gelu_erf(gt);
                    }
                    put4<T, STAGED>(a, sc, m, n, mok && n < a.N, pack4<T>(v[0], v[1], v[2], v[3]));
                }
            } else {
                fragment_pass1<T, STAGED>(a, sc, acc[fh][fm], e.vb[fh], e.vb2[fm][fh], e.vr[fm][fh], mok, m, nbase + fh * 32, hi);
            }
        }
    }
    if constexpr (!GEGLU) {
        if (a.act != SFAST_ACT_NONE) epilogue_act_tail<T, FN, FM, STAGED>(a, sc, acc, mbase, nbase, l31, hi);
    }
}

// Late form for the 5-fragment tiles at two workgroups per CU: operands are fetched fragment by fragment, one
// fragment AHEAD of the one being finished (12 vectors each), which keeps the epilogue inside the register budget
// of two waves per SIMD while still overlapping every operand round trip but the first with math and stores.
// (Program order = load(f+1), store(f): legal for in-place residuals because fragments never overlap.)
// AHEAD = false (the 12-wave form of igemm_pp.h: 168 registers, 96 of them accumulators): one operand set, fetched and consumed per fragment
template <typename T, int FN, int FM, bool STAGED, bool AHEAD = true>
__device__ __forceinline__ void epilogue_late(const IgemmArgs &a, const StageCtx &sc, f32x16 (&acc)[FN][FM], int mbase, int nbase, int l31,
                                              int hi, int split_idx, bool whole) {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    const g2_ptr zero = (g2_ptr)(const void *)g_zero16;
    if (a.splits > 1 && !whole) {
        store_partial<FN, FM, false>(a, acc, mbase, nbase, l31, hi, split_idx);
        return;
    }
    const BatchOfRow batch_of(a);
    u32x2 vb[AHEAD ? 2 : 1][4], vb2[AHEAD ? 2 : 1][4], vr[AHEAD ? 2 : 1][4];
    const EpiRow<T> bias(a.bias, 0, true);
    auto fetch = [&](int f, int buf) {
        const int fm = f / FN, fh = f % FN;
        const int m = mbase + fm * 32 + l31;
        const bool mok = m < a.M;
        const int bi = a.rowbias ? batch_of(m) : 0;
        const EpiRow<T> rowbias(a.rowbias, (int64_t)bi * a.ld_rowbias, mok), res(a.res, (int64_t)m * a.ldr, mok);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nbase + fh * 32 + 8 * g + 4 * hi;
            const bool nok = n < a.N;
            vb[buf][g] = bias.load(n, nok);
            vb2[buf][g] = rowbias.load(n, nok);
            vr[buf][g] = res.load(n, nok);
        }
    };
    if constexpr (AHEAD) fetch(0, 0);
#pragma unroll
    for (int f = 0; f < FN * FM; ++f) {
        const int fm = f / FN, fh = f % FN;
        if constexpr (AHEAD) {
            if (f + 1 < FN * FM) fetch(f + 1, (f + 1) & 1);
        } else {
            fetch(f, 0);
        }
        const int m = mbase + fm * 32 + l31;
        constexpr int B1 = AHEAD ? 1 : 0;
        fragment_pass1<T, STAGED>(a, sc, acc[fh][fm], vb[f & B1], vb2[f & B1], vr[f & B1], m < a.M, m, nbase + fh * 32, hi);
    }
    if (a.act != SFAST_ACT_NONE) epilogue_act_tail<T, FN, FM, STAGED>(a, sc, acc, mbase, nbase, l31, hi);
}

// By-operand-type form (round 6, the 256-row tiles of igemm_pp.h): instead of walking the fragments with all three operand kinds in
// flight per fragment, every kind is requested for the WHOLE wave tile at once -- bias (FN * 4 vectors, shared by the pixel fragments),
// then the row bias, then the residual (FN * FM * 4 vectors = 48 registers for a six-fragment wave, beside 96 accumulator registers) --
// and an ABSENT kind is skipped (a wave-uniform branch) instead of being fetched from the zero block. Memory round trips per workgroup:
// one for the bias + FM per further operand kind that exists (typically three) instead of one per fragment (six, serialised: ~12 us of epilogue
// behind a 52 us K loop in the 12-wave form, profiles/r06_pp_loop_probe_run19.log). Same arithmetic sequence per element as
// fragment_pass1: v = fma(acc, out_scale, bias); v += row bias; v += alpha * residual.
template <typename T, int FN, int FM, bool STAGED>
__device__ __forceinline__ void epilogue_by_type(const IgemmArgs &a, const StageCtx &sc, f32x16 (&acc)[FN][FM], int mbase, int nbase, int l31,
                                                 int hi, int split_idx, bool whole) {
    if (a.splits > 1 && !whole) {
        store_partial<FN, FM, false>(a, acc, mbase, nbase, l31, hi, split_idx);
        return;
    }
    {
        u32x2 vb[FN][4];
        const EpiRow<T> bias(a.bias, 0, true);
#pragma unroll
        for (int fh = 0; fh < FN; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                vb[fh][g] = bias.load(n, n < a.N);
            }
#pragma unroll
        for (int fh = 0; fh < FN; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float b0[4];
                unpack4<T>(vb[fh][g], b0);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[fh][fm][4 * g + i] = fmaf(acc[fh][fm][4 * g + i], a.out_scale, b0[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const bool res_now = a.res_before_act || a.act == SFAST_ACT_NONE;
    // one pixel fragment row (FN fragments, FN * 4 vectors) per round trip: all FM rows at once is 48 data + 48 address registers for a
    // six-fragment wave -- past the 168-register budget of the 12-wave kernels (measured: 150 spilled registers)
    auto add_rows = [&](const void *base_ptr, bool per_batch, float scale) __attribute__((always_inline)) {
        const BatchOfRow batch_of(a);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            u32x2 v[FN][4];
            const int m = mbase + fm * 32 + l31;
            const int64_t row_off = per_batch ? (int64_t)batch_of(m) * a.ld_rowbias : (int64_t)m * a.ldr;
            const EpiRow<T> row(base_ptr, row_off, m < a.M);
#pragma unroll
            for (int fh = 0; fh < FN; ++fh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                    v[fh][g] = row.load(n, n < a.N);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int fh = 0; fh < FN; ++fh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float r[4];
                    unpack4<T>(v[fh][g], r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[fh][fm][4 * g + i] += r[i] * scale;
                    __builtin_amdgcn_sched_barrier(0);  // one vector at a time: unpacking all of them first costs 4 registers per vector
                }
        }
    };
    if (a.rowbias != nullptr) add_rows(a.rowbias, true, 1.0f);
    if (a.res != nullptr && res_now) add_rows(a.res, false, a.alpha);
    if (a.act != SFAST_ACT_NONE) {
        epilogue_act_tail<T, FN, FM, STAGED>(a, sc, acc, mbase, nbase, l31, hi);
        return;
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mbase + fm * 32 + l31;
#pragma unroll
        for (int fh = 0; fh < FN; ++fh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + fh * 32 + 8 * g + 4 * hi;
                put4<T, STAGED>(a, sc, m, n, m < a.M && n < a.N,
                                pack4<T>(acc[fh][fm][4 * g], acc[fh][fm][4 * g + 1], acc[fh][fm][4 * g + 2], acc[fh][fm][4 * g + 3]));
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

// ---- staged tile -> global memory (+ GroupNorm partial statistics) ------------------------------------------------
// Thread t of the NTC consumer threads owns chunk column t % (BNO/8) (8 output channels = 16 bytes) of rows t / (BNO/8) + k*RPP:
// 16 consecutive lanes write 256 contiguous bytes of one output row. On request the same pass reduces the tile to the
// statistics the consuming GroupNorm needs, so that it becomes ONE normalisation pass instead of a statistics pass plus a
// normalisation pass (the reference's Triton GroupNorm is that two-kernel form, triton/ops/group_norm.py:111-165 + :272-349):
//   a statistics UNIT is `gn_unit` consecutive channels (a divisor of every consumer's channels-per-group); a tile emits one
//   {mean, M2} record per unit SLOT it overlaps (a unit cut by the tile's column range yields one record in each tile),
//   sums are shifted by the slot's first element in tile row 0 (exactly additive within the tile), order of summation fixed.
// The consumer merges records of different tiles with Chan's parallel-variance formula (norm.hip: gn_nhwc_apply_pre_kernel).
template <typename T, int BM, int BNO, int NTC>
__device__ __forceinline__ void flush_staged_tile(const IgemmArgs &a, char *lds, int m0, int n0, int ctid) {
    constexpr int CPR = BNO / 8, LDR = BNO * 2 + 8, RPP = NTC / CPR, NTF = RPP * CPR;
    constexpr int SCR = ((BM * LDR + 15) / 16) * 16;  // float4 per flush thread: {s1, s2} of the chunk's two unit parts
    __syncthreads();                                  // every fragment of the tile is in LDS
    const bool stats = a.gn_stats != nullptr;
    const int c = ctid % CPR, rp = ctid / CPR;
    const int n = n0 + c * 8;
    const int unit = stats ? a.gn_unit : 8;
    const int U0 = n / unit;                                  // unit of the chunk's first channel
    const int nb = min(8, (U0 + 1) * unit - n);               // channels of the chunk inside U0; the rest belong to U0 + 1
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
    if (ctid < NTF && n < a.N) {
        float sha = 0.f, shb = 0.f;
        if (stats) {
            const int ca = max(n0, U0 * unit) - n0, cb = (U0 + 1) * unit - n0;  // first in-tile channel of either unit slot
            sha = (float)*reinterpret_cast<const T *>(lds + ca * 2);
            shb = nb < 8 ? (float)*reinterpret_cast<const T *>(lds + cb * 2) : 0.f;
        }
        const char *src = lds + rp * LDR + c * 16;
        T *dst = (T *)a.out + (int64_t)(m0 + rp) * a.ldo + n;
        const int64_t dstep = (int64_t)RPP * a.ldo;
#pragma unroll 4
        for (int r = rp; r < BM; r += RPP) {
            if (m0 + r < a.M) {
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(src), hi2 = *reinterpret_cast<const u32x2 *>(src + 8);
                const u32x4 v = {lo[0], lo[1], hi2[0], hi2[1]};
                *reinterpret_cast<u32x4 *>(dst) = v;
                if (stats) {
                    float f[8];
                    unpack8<T>(v, f);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i < nb) {
                            const float d = f[i] - sha;
                            s1a += d;
                            s2a = fmaf(d, d, s2a);
                        } else {
                            const float d = f[i] - shb;
                            s1b += d;
                            s2b = fmaf(d, d, s2b);
                        }
                    }
                }
            }
            src += RPP * LDR;
            dst += dstep;
        }
    }
    if (!stats) return;
    float *red = reinterpret_cast<float *>(lds + SCR);
    if (ctid < NTF) *reinterpret_cast<f32x4 *>(red + ctid * 4) = f32x4{s1a, s2a, s1b, s2b};
    __syncthreads();
    // slot j of this tile <-> unit n0/unit + j. Eight lanes per slot: lane l sums the row phases l, l+8, ... of the slot's chunk
    // columns, then a fixed three-step butterfly -- the order of summation is the same in every launch (bitwise reproducible), and
    // no thread walks the whole scratch array alone (a single-thread walk was ~2 us of serial LDS round trips per workgroup).
    constexpr int LPS = 8;
    static_assert(NTC % 64 == 0, "slot lanes must not straddle waves");
    const int ufirst = n0 / unit;
    const int nend = min(n0 + BNO, a.N);
    const int j = ctid / LPS, l = ctid % LPS;
    {
        const int U = ufirst + j;
        const int lo = max(n0, U * unit), hi_ = min(nend, (U + 1) * unit);
        const bool live = j < a.gn_slots && hi_ > lo;
        float s1 = 0.f, s2 = 0.f;
        if (live) {
            for (int cc = (lo - n0) >> 3; cc <= (hi_ - 1 - n0) >> 3; ++cc) {
                const int part = ((n0 + cc * 8) / unit == U) ? 0 : 2;
#pragma unroll 4
                for (int r = l; r < RPP; r += LPS) {
                    const float *q = red + (r * CPR + cc) * 4 + part;
                    s1 += q[0];
                    s2 += q[1];
                }
            }
        }
#pragma unroll
        for (int off = 1; off < LPS; off <<= 1) {
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        if (j < a.gn_slots && l == 0) {
            float mean = 0.f, m2 = 0.f;
            if (hi_ > lo) {
                const float sh = (float)*reinterpret_cast<const T *>(lds + (lo - n0) * 2);
                const float cnt = (float)(hi_ - lo) * (float)min(BM, a.M - m0);
                mean = sh + s1 / cnt;
                m2 = fmaxf(s2 - s1 * s1 / cnt, 0.f);
            }
            const int tile_m = m0 / BM, tile_n = n0 / BNO;
            float *o = a.gn_stats + (((int64_t)tile_m * a.tiles_n + tile_n) * a.gn_slots + j) * 2;
            o[0] = mean;
            o[1] = m2;
        }
    }
}

// One entry for the three kernel files: split-K slab store, or the fused epilogue with direct (STAGED = false) or staged stores.
// STAGED is a KERNEL template parameter (separate instantiations, chosen on the host from IgemmArgs::stage_out): with both store
// paths in one kernel the 128-wide tiles ran out of their 256 registers and spilled.
// ---- split-K without a second kernel ----------------------------------------------------------------------------------------
// Every workgroup of a split tile leaves its fp32 partial accumulators in the workspace IN FRAGMENT ORDER (thread-linear 16-byte
// groups: the slab layout is private to this function, every access a fully coalesced 1 KiB per wave), publishes them (agent-scope
// fence) and draws a ticket from the tile's counter; the workgroup that draws the last ticket re-reads ALL the slabs in split order
// 0 .. splits-1 (so the sum does not depend on which workgroup happened to finish last: bitwise reproducible, and the same order as
// splitk_reduce_kernel) and goes on into the ordinary epilogue -- bias / residual / activation / staged stores / GroupNorm
// statistics -- as if it had computed the whole K range. atomicInc wraps the counter back to 0 with the last ticket: the counters
// are zero again when the launch ends (sfast_hip.h, SFAST_EXT_WS_TICKETS).
// Measured reason (tools/ws_loop_probe.py, profiles/r03_ws_loop_probe_run7.log): with the reduce as a second launch the part of a
// split conv that is NOT its K loop costs 20-24 us against 10 us for an unsplit one.
// Agent-scope accesses of the slabs: relaxed ATOMIC 8-byte stores / loads at agent scope compile to sc1 accesses -- the stores write
// through this XCD's L2, the loads do not hit in the L1 or in a stale L2 line -- and are ordinary compiler-visible memory operations
// (the compiler tracks their vmcnt; a first attempt with inline-asm dwordx4 loads carried not-yet-landed registers across the loop
// back edge). With them no fence is needed around the ticket: a release / acquire fence at agent scope is buffer_wbl2 / buffer_inv,
// a write-back resp. invalidation of the whole 4 MiB L2 per workgroup -- the fenced version made a joined launch 20-30 us SLOWER than
// the two-launch form (profiles/r03_splitk_join_fenced_run8.json.log: 157 it/s against 183).
typedef unsigned long long slab_word;  // two floats
__device__ __forceinline__ void store_agent(slab_word *p, float x, float y) {
    __hip_atomic_store(p, (slab_word)__float_as_uint(x) | ((slab_word)__float_as_uint(y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ slab_word load_agent(const slab_word *p) {
    return __hip_atomic_load(const_cast<slab_word *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int FN, int FM, int NTC>
__device__ __forceinline__ bool splitk_join(const IgemmArgs &a, f32x16 (&acc)[FN][FM], char *smem, int tile_id, int split_idx, int ctid) {
    constexpr int Q = FN * FM * 8;              // 8-byte words per thread (accumulator registers r, r + 1 of a fragment)
    constexpr int64_t SLAB = (int64_t)Q * NTC;  // words per (tile, split); word q of thread t at q * NTC + t: 512 contiguous bytes per wave
    slab_word *const tile0 = reinterpret_cast<slab_word *>(a.partial) + (int64_t)tile_id * a.splits * SLAB + ctid;
    slab_word *const mine = tile0 + split_idx * SLAB;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 8; ++r) store_agent(mine + ((fn * FM + fm) * 8 + r) * NTC, acc[fn][fm][2 * r], acc[fn][fm][2 * r + 1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's part of the slab has been written through
    __syncthreads();                                   // ... every thread's; and nobody reads or lands anything in the LDS ring any more
    if (ctid == 0)
        *reinterpret_cast<volatile unsigned *>(smem) = __builtin_amdgcn_atomic_inc32(a.tickets + tile_id, (unsigned)(a.splits - 1), __ATOMIC_RELAXED, "agent");
    __syncthreads();
    const unsigned ticket = *reinterpret_cast<volatile unsigned *>(smem);
    if (ticket != (unsigned)(a.splits - 1)) return false;
    // the last ticket: sum all slabs in split order 0 .. S-1 (its own included: read back, so that the order never depends on who is last)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;
    for (int sp = 0; sp < a.splits; ++sp) {
        const slab_word *src = tile0 + sp * SLAB;
        slab_word v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = load_agent(src + q * NTC);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            acc[(q / 8) / FM][(q / 8) % FM][2 * (q % 8)] += __uint_as_float((unsigned)v[q]);
            acc[(q / 8) / FM][(q / 8) % FM][2 * (q % 8) + 1] += __uint_as_float((unsigned)(v[q] >> 32));
        }
    }
    __syncthreads();  // the ticket word sits in the LDS the staged epilogue is about to use
    return true;
}

// JOIN = false compiles the in-kernel split-K join out (the single-stream LDS-DMA kernels of igemm_glds.hip sit at the SGPR limit;
// igemm_run never hands them tickets)
// Round 4: the join is EVIDENCE code (measured slower than the reduce launch; its ticket protocol rests on
// sc1 cache behaviour rather than on release / acquire) -- it is compiled only into the probe build (-DSFAST_PROBES, build.py
// --probes -> libsfast_hip_probes.so); the product library ignores SFAST_EXT_WS_TICKETS and always runs the reduce launch.
// which late epilogue run_epilogue uses unless a kernel says otherwise: 2 = by operand type (round 6), 0 = one fragment ahead (rounds 1 - 5;
// `SFAST_EXTRA_CFLAGS=-DSFAST_LATE_FORM_DEFAULT=0 python stable-fast_amd/build.py` rebuilds that arm for an A/B)
#ifndef SFAST_LATE_FORM_DEFAULT
#define SFAST_LATE_FORM_DEFAULT 2
#endif
#ifdef SFAST_PROBES
constexpr bool kJoinDefault = true;
#else
constexpr bool kJoinDefault = false;
#endif
template <typename T, int BM, int BNO, int FN, int FM, bool GEGLU, bool EPI_EARLY, int NTC, bool STAGED, bool JOIN = kJoinDefault, int LATE_FORM = SFAST_LATE_FORM_DEFAULT>
__device__ __forceinline__ void run_epilogue(const IgemmArgs &a, f32x16 (&acc)[FN][FM],
                                             EpiOperands<(EPI_EARLY ? (GEGLU ? FN / 2 : FN) : 1), (EPI_EARLY ? FM : 1)> &epi, char *smem,
                                             int m0, int n0, int mbase, int nbase, int l31, int hi, int ctid, int split_idx) {
    bool whole = false;
    if constexpr (JOIN) if (a.splits > 1 && a.tickets != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA request of this wave is still on its way into the ring
        const int tile_id = (m0 / BM) * a.tiles_n + n0 / BNO;
        if (!splitk_join<FN, FM, NTC>(a, acc, smem, tile_id, split_idx, ctid)) return;
        whole = true;  // from here on: the workgroup that owns the finished tile
        if constexpr (EPI_EARLY) epilogue_prefetch<T, FN, FM, GEGLU>(a, epi, mbase, nbase, l31, hi, true);
    }
    StageCtx sc{smem, m0, n0, BNO * 2 + 8};
    if constexpr (STAGED) {
        __syncthreads();  // LDS ring is free: every wave is past its last fragment read and its last LDS-DMA has landed
        if constexpr (EPI_EARLY)
            epilogue_finish<T, FN, FM, GEGLU, true>(a, sc, acc, epi, mbase, nbase, l31, hi, split_idx, whole);
        else
            if constexpr (LATE_FORM == 2)
                epilogue_by_type<T, FN, FM, true>(a, sc, acc, mbase, nbase, l31, hi, split_idx, whole);
            else
                epilogue_late<T, FN, FM, true, LATE_FORM == 0>(a, sc, acc, mbase, nbase, l31, hi, split_idx, whole);
        flush_staged_tile<T, BM, BNO, NTC>(a, smem, m0, n0, ctid);
    } else {
        if constexpr (EPI_EARLY)
            epilogue_finish<T, FN, FM, GEGLU, false>(a, sc, acc, epi, mbase, nbase, l31, hi, split_idx, whole);
        else
            if constexpr (LATE_FORM == 2)
                epilogue_by_type<T, FN, FM, false>(a, sc, acc, mbase, nbase, l31, hi, split_idx, whole);
            else
                epilogue_late<T, FN, FM, false, LATE_FORM == 0>(a, sc, acc, mbase, nbase, l31, hi, split_idx, whole);
    }
}

}  // namespace sfast
