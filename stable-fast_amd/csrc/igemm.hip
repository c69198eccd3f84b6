// MFMA implicit-GEMM for gfx950: every Linear / 1x1 conv / 3x3 conv / Linear+GEGLU of the UNet.
//
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] )         X: activations, W: weights [N][K]
//
// Replaces (reference, src/sfast/csrc/operators/):
//   cublas/CUDABlas.cc:720-915 gemm_and_bias (+ cublas_gemm.cpp:798-948 linear / linear_add),
//   cudnn/cudnn_convolution_impl.cc:890-987 fused conv+bias+add+act,
//   cutlass/cutlass_dual_linear_kernel.cu:238-245 DualGemm GEGLU.
//
// CDNA4 design (not a translation of the CUTLASS 128x128x32 / warp 64x32 tiling):
//   * v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate. The WEIGHT fragment is the MFMA A operand
//     and the ACTIVATION fragment the B operand, i.e. the wave computes D[n][m]. In the 32x32 C/D
//     layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) a lane then owns 4
//     CONSECUTIVE output columns n for one output row m -> bias / residual / store are 8-byte
//     vector accesses with no LDS transpose.
//   * both operands are K-contiguous in memory (activations NHWC, weights [Cout][kh][kw][Cin] =
//     the channels_last physical layout of the live nn.Parameter), so a fragment is ONE 16-byte
//     read. K-tile = 64 halves = 128 B per tile row; LDS rows are XOR-swizzled in 16-B chunks
//     (chunk ^= (row>>1)&7) which makes both the ds_write_b128 staging and the ds_read_b128
//     fragment reads bank-conflict free for the instruction lane groups of gfx950.
//   * conv = the same kernel with an implicit im2col gather: a K-chunk k -> (tap, channel),
//     a tile row m -> (b, ho, wo); out-of-image taps are zero-filled in registers. Nearest-2x
//     upsample and the channel concat of the up-blocks are folded into the gather, so neither is
//     ever materialised.
//   * register-staged double buffering (global_load_dwordx4 of tile t+1 issued before the MFMAs
//     of tile t, ds_write after), one barrier per K-tile.
//   * GEGLU: the W tile interleaves 32 hidden rows with 32 gate rows per wave so h and g of the
//     same output element land in the same lane/register slot -> in-register h * gelu(g); the
//     [M, 2N] intermediate never exists.
//   * split-K (fp32 slabs + a reduce/epilogue kernel) for the weight-streaming 16x16 / 8x8 levels
//     where M <= 512 gives too few tiles for 256 CUs.
#include <algorithm>
#include <vector>

#include "igemm.h"
#include "igemm_device.h"

namespace sfast {


// MODE 0: linear (row m -> x + m*ldx). MODE 1: conv (implicit im2col, NHWC).
template <typename T, int BM, int BN, int WM, int WN, int MODE, bool GEGLU, bool STAGED = false, bool W8 = false>
__device__ __forceinline__ void igemm_body(const IgemmArgs &a) {
    using vec8 = typename Elem<T>::vec8;
    constexpr int NT = WM * WN * 64;
    constexpr int FM = BM / (WM * 32);  // 32-row activation fragments per wave
    constexpr int FN = BN / (WN * 32);  // 32-row weight fragments per wave
    constexpr int XCH = BM * 8 / NT;    // 16-B chunks staged per thread per K-tile
    constexpr int WCH = BN * 8 / NT;
    constexpr int RPP = NT / 8;         // tile rows covered per staging pass
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int BNO = GEGLU ? BN / 2 : BN;  // output columns per tile
    constexpr int WNB = FN * 32;              // weight rows per wave-n group
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile/wave mismatch");
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "staging mismatch");
    static_assert(!GEGLU || (FN % 2 == 0), "GEGLU needs paired fragments");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    touch_args(a);
    if (MODE == 1) touch_conv_args(a);
    trace_mark(a, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, hi = lane >> 5;

    const BlockTile bt = decode_block(a);  // XCD-aware (tile, K-split) of this workgroup
    if (bt.tile_m < 0) return;  // surplus block of an xmap = 2 grid (wave-uniform, before any barrier)
    const int tile_n = bt.tile_n, tile_m = bt.tile_m;
    const int m0 = tile_m * BM, n0 = tile_n * BNO;
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);

    const int kc = tid & 7;
    const int rbase = tid >> 3;

    // ---- per-thread staging metadata -----------------------------------------------------------
    const T *xrow[XCH];          // MODE 0
    int xbhw[XCH], xh[XCH], xw[XCH];  // MODE 1
    const PixelDecoder decode(a);
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int m = m0 + rbase + i * RPP;
        if (MODE == 0) {
            xrow[i] = (m < a.M) ? (const T *)a.x + (int64_t)m * a.ldx : nullptr;
        } else {
            if (m < a.M) {
                int b, ho, wo;
                decode(m, b, ho, wo);
                xbhw[i] = b * a.H * a.W;
                xh[i] = ho * a.stride_h - a.pad_h;
                xw[i] = wo * a.stride_w - a.pad_w;
            } else {
                xbhw[i] = 0;
                xh[i] = -(1 << 28);
                xw[i] = -(1 << 28);
            }
        }
    }
    const T *wrow[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int j = rbase + i * RPP;
        if (GEGLU) {
            const int grp = j / WNB, within = j % WNB;
            const int half = within / (WNB / 2), i2 = within % (WNB / 2);
            const int ncol = n0 + grp * (WNB / 2) + i2;
            wrow[i] = (ncol < a.N) ? (const T *)(half ? a.w[1] : a.w[0]) + (int64_t)ncol * a.ldw : nullptr;  // a.w[1] = first gate row (api_gemm_conv.hip)
        } else {
            const int n = n0 + j;
            if (n < a.N) {
                const int rs = a.rows_per_seg;
                const int seg = (n >= rs) + (n - rs >= rs) + (n - rs - rs >= rs);
                const void *base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : seg == 2 ? a.w[2] : a.w[3];
                if constexpr (W8)  // int8 rows: ldw counts bytes; kept as a T pointer of half the element offset (only its address is used)
                    wrow[i] = (const T *)((const char *)base + (int64_t)(n - seg * a.rows_per_seg) * a.ldw);
                else
                    wrow[i] = (const T *)base + (int64_t)(n - seg * a.rows_per_seg) * a.ldw;
            } else {
                wrow[i] = nullptr;
            }
        }
    }

    u32x4 xreg[XCH], wreg[WCH];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // All staging loads are UNCONDITIONAL: an invalid chunk (image border tap, row >= M, k >= K) reads a
    // known-valid address and is zeroed with a select afterwards. A predicated load
    // (`ok ? *p : 0`) makes hipcc wrap every load in its own exec-masked branch with a full
    // s_waitcnt at the join -- eight serialised memory round trips per K-tile.
    // The zero fill itself comes from memory too (a 16-byte device-resident zero block), so nothing
    // consumes the loaded registers before store_tile: the loads stay in flight across the MFMAs.
    // Both candidates are cast to the GLOBAL address space: a generic pointer select would lower to
    // flat_load, which also ticks lgkmcnt and would make every ds_read wait drain the prefetch.
    auto ldg16 = [&](const T *p, const T *, bool ok) -> u32x4 {
        typedef const u32x4 __attribute__((address_space(1))) * gvec_ptr;
        const gvec_ptr q = ok ? (gvec_ptr)(const void *)p : (gvec_ptr)(const void *)g_zero16;
        return *q;
    };
    auto load_tile = [&](int kt) {
        const int k = kt * 64 + kc * 8;
        const bool kvalid = k < a.K;
        const int ks = kvalid ? k : 0;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const bool ok = kvalid & (xrow[i] != nullptr);
                xreg[i] = ldg16(xrow[i] + ks, (const T *)a.x, ok);
            }
        } else {
            const int cin = a.C1 + a.C2;
            const int tap = ks / cin;
            const int c = ks - tap * cin;
            const int r = tap / a.KW, s = tap - r * a.KW;
            const bool first = c < a.C1;
            const T *base = first ? (const T *)a.x : (const T *)a.x2;
            const int pitch = first ? a.C1 : a.C2;
            const int cc = first ? c : c - a.C1;
            const int dh = r * a.dil_h, dw = s * a.dil_w;
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                int hi_ = xh[i] + dh, wi_ = xw[i] + dw;
                bool ok;
                if (a.ups) {
                    ok = (unsigned)hi_ < (unsigned)(2 * a.H) && (unsigned)wi_ < (unsigned)(2 * a.W);
                    hi_ >>= 1;
                    wi_ >>= 1;
                } else {
                    ok = (unsigned)hi_ < (unsigned)a.H && (unsigned)wi_ < (unsigned)a.W;
                }
                ok = ok && kvalid;
                const int64_t off = ((int64_t)(xbhw[i] + hi_ * a.W + wi_)) * pitch + cc;
                xreg[i] = ldg16(base + off, base, ok);
            }
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const bool ok = kvalid & (wrow[i] != nullptr);
            if constexpr (W8) {
                // 8 int8 weights (8 bytes) of this chunk; widened to T when the tile is written to LDS (store_tile)
                typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
                const g2_ptr q = ok ? (g2_ptr)(const void *)((const char *)wrow[i] + ks) : (g2_ptr)(const void *)g_zero16;
                const u32x2 v = *q;
                wreg[i] = u32x4{v[0], v[1], 0u, 0u};
            } else {
                wreg[i] = ldg16(wrow[i] + ks, (const T *)a.w[0], ok);
            }
        }
    };
    auto store_tile = [&](int stage) {
        char *xs = smem + stage * STAGE;
        char *ws = xs + BM * 128;
#pragma unroll
        for (int i = 0; i < XCH; ++i)
            *reinterpret_cast<u32x4 *>(xs + lds_off(rbase + i * RPP, kc)) = xreg[i];
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            if constexpr (W8) {
                vec8 w;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int b8 = (int)(int8_t)((wreg[i][e >> 2] >> (8 * (e & 3))) & 0xffu);
                    w[e] = Elem<T>::from_f32((float)b8);  // exact: |int8| <= 128 is representable in f16 and bf16
                }
                *reinterpret_cast<u32x4 *>(ws + lds_off(rbase + i * RPP, kc)) = __builtin_bit_cast(u32x4, w);
            } else {
                *reinterpret_cast<u32x4 *>(ws + lds_off(rbase + i * RPP, kc)) = wreg[i];
            }
        }
    };

    // epilogue operands (bias / row-bias / residual) are requested now and consumed after the K loop when the
    // tile shape leaves registers for them (the 5-fragment tiles would drop to one wave per SIMD)
    constexpr bool EPI_EARLY = GEGLU || FN * FM <= 4;
    EpiOperands<(EPI_EARLY ? (GEGLU ? FN / 2 : FN) : 1), (EPI_EARLY ? FM : 1)> epi;
    if constexpr (EPI_EARLY) epilogue_prefetch<T, FN, FM, GEGLU>(a, epi, m0 + wm * (FM * 32), n0 + wn * (GEGLU ? WNB / 2 : WNB), l31, hi);

    f32x16 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;

    // fragment reads run one 16-wide K step AHEAD of the MFMAs that consume them (two register sets): at one wave per
    // SIMD nothing else hides the ds_read latency
    auto compute = [&](int stage) {
        const char *xs = smem + stage * STAGE;
        const char *ws = xs + BM * 128;
        vec8 af[2][FN], bf[2][FM];
        auto read_frags = [&](int ks, int set) {
            const int chunk = ks * 2 + hi;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                af[set][fn] = *reinterpret_cast<const vec8 *>(ws + lds_off(wn * WNB + fn * 32 + l31, chunk));
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
                bf[set][fm] = *reinterpret_cast<const vec8 *>(xs + lds_off(wm * (FM * 32) + fm * 32 + l31, chunk));
        };
        read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) read_frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = mfma32(af[ks & 1][fn], bf[ks & 1][fm], acc[fn][fm]);
        }
    };

    // ---- main loop -------------------------------------------------------------------------------
    trace_mark(a, 1);
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        trace_mark(a, 2);
        store_tile(0);
        __syncthreads();
        trace_mark(a, 3);
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int cur = (kt - kt_begin) & 1;
            const bool more = kt + 1 < kt_end;
            if (more) load_tile(kt + 1);
            compute(cur);
            if (more) store_tile(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: batched operand loads, fp32 math, 8-byte stores (igemm_device.h) -------------------------
    trace_mark(a, 4);
    run_epilogue<T, BM, BNO, FN, FM, GEGLU, EPI_EARLY, NT, STAGED>(a, acc, epi, smem, m0, n0, m0 + wm * (FM * 32), n0 + wn * (GEGLU ? WNB / 2 : WNB), l31, hi,
                                                             tid, bt.split);
    trace_finish(a);
}

template <typename T, int BM, int BN, int WM, int WN, int MODE, bool GEGLU, bool STAGED = false>
__global__ void __launch_bounds__(WM *WN * 64, igemm_min_waves(WM *WN * 64, 2 * (BM + BN) * 128)) igemm_kernel(const IgemmArgs a) {
    igemm_body<T, BM, BN, WM, WN, MODE, GEGLU, STAGED>(a);
}

// Weight-only int8 linear: out = dq_scale * (x . Wq^T) + bias with Wq int8 [N][K] -- the reference's cutlass_qlinear_dynamic
// (csrc/operators/cutlass/cutlass_qlinear_dynamic_kernel.cu:259-294: mixed f16 x s8 tensor-op GEMM, alpha = weight.q_scale()).
// Here the int8 rows are widened to T on their way into LDS (the MFMA then runs the ordinary f16 / bf16 tile), the scale is the
// epilogue's out_scale: half the weight bytes cross HBM, the arithmetic is exact in the weights.
template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM *WN * 64, igemm_min_waves(WM *WN * 64, 2 * (BM + BN) * 128)) igemm_w8_kernel(const IgemmArgs a) {
    igemm_body<T, BM, BN, WM, WN, 0, false, false, true>(a);
}

// Grouped launch: blockIdx.z selects one of up to SFAST_MAX_GEMM_GROUPS independent problems of identical shape that share the
// activation operand -- the cross-attention K/V projections of every transformer block read the same text context
// (libs/xformers + diffusers Attention.to_k / to_v; one cublas_lowp_linear each in the reference). Weight and output
// pointers of the group travel in the kernel-argument block; everything else is the plain kernel.
struct IgemmGroupTab {
    const void *x[SFAST_MAX_GEMM_GROUPS];
    const void *w0[SFAST_MAX_GEMM_GROUPS];
    const void *w1[SFAST_MAX_GEMM_GROUPS];
    const void *bias[SFAST_MAX_GEMM_GROUPS];
    void *out[SFAST_MAX_GEMM_GROUPS];
};
template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM *WN * 64, igemm_min_waves(WM *WN * 64, 2 * (BM + BN) * 128))
    igemm_grouped_kernel(const IgemmArgs a, const IgemmGroupTab g) {
    IgemmArgs b = a;
    const int z = blockIdx.z;
    b.x = g.x[z];
    b.w[0] = g.w0[z];
    b.w[1] = g.w1[z];
    b.w[2] = g.w1[z];
    b.w[3] = g.w1[z];
    b.bias = g.bias[z];
    b.out = g.out[z];
    igemm_body<T, BM, BN, WM, WN, 0, false>(b);
}

// split-K reduce + epilogue: one thread per 4 consecutive output columns.
template <typename T, bool GEGLU>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const IgemmArgs a) {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    const g2_ptr zero = (g2_ptr)(const void *)g_zero16;
    const int n4 = a.N / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.M * n4;
    if (idx >= total) return;
    int m;
    if (total <= (1 << 22)) {
        m = fdiv22((int)idx, n4, __builtin_amdgcn_rcpf((float)n4));
    } else {
        m = (int)(idx / n4);
    }
    const int n = ((int)(idx - (int64_t)m * n4)) * 4;
    const int64_t NP = GEGLU ? 2 * (int64_t)a.N : (int64_t)a.N;
    // epilogue operands first: their round trip overlaps the slab reads instead of following them
    u32x2 vb = *(a.bias ? (g2_ptr)(const void *)((const T *)a.bias + n) : zero);
    u32x2 vb2, vr;
    if (GEGLU) {
        vb2 = *(a.bias ? (g2_ptr)(const void *)((const T *)a.bias + a.N + n) : zero);
        vr = u32x2{0u, 0u};
    } else {
        const BatchOfRow batch_of(a);
        const int bi = a.rowbias ? batch_of(m) : 0;
        vb2 = *(a.rowbias ? (g2_ptr)(const void *)((const T *)a.rowbias + (int64_t)bi * a.ld_rowbias + n) : zero);
        vr = *(a.res ? (g2_ptr)(const void *)((const T *)a.res + (int64_t)m * a.ldr + n) : zero);
    }
    float v[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
    const float *p0 = a.partial + (int64_t)m * NP + n;
    const int64_t zstride = (int64_t)a.M * NP;
    int z = 0;
    for (; z + 3 < a.splits; z += 4) {  // four slab loads in flight, summation order z = 0, 1, 2, ...
        f32x4 t[4], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = *reinterpret_cast<const f32x4 *>(p0 + (z + k) * zstride);
            if (GEGLU) u[k] = *reinterpret_cast<const f32x4 *>(p0 + (z + k) * zstride + a.N);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] += t[k][i];
                if (GEGLU) g[i] += u[k][i];
            }
    }
    for (; z < a.splits; ++z) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(p0 + z * zstride);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += t[i];
        if (GEGLU) {
            const f32x4 u = *reinterpret_cast<const f32x4 *>(p0 + z * zstride + a.N);
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] += u[i];
        }
    }
    float b0[4], b1[4], r[4], o[4];
    unpack4<T>(vb, b0);
    unpack4<T>(vb2, b1);
    unpack4<T>(vr, r);
    if (GEGLU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (v[i] + b0[i]) * act_gelu_erf(g[i] + b1[i]);
    } else {
        const bool res_now = a.res_before_act || a.act == SFAST_ACT_NONE;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaf(v[i], a.out_scale, b0[i]) + b1[i] + (res_now ? r[i] * a.alpha : -0.0f);
        if (a.act != SFAST_ACT_NONE) {
            f32x4 ov = {o[0], o[1], o[2], o[3]};
#pragma unroll 1
            for (int i = 0; i < 4; ++i) ov[i] = apply_act(ov[i], a.act);  // rolled: one copy of the activation switch
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = ov[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += res_now ? -0.0f : r[i] * a.alpha;
        }
    }
    *reinterpret_cast<u32x2 *>((T *)a.out + (int64_t)m * a.ldo + n) = pack4<T>(o[0], o[1], o[2], o[3]);
}

#ifdef SFAST_PROBES  // measured SLOWER than reduce + separate GroupNorm in the SD1.5 step (profiles/r04_reduce_gn_ab_run{7,8,9}.log): probe build only
// Split-K reduce + epilogue + the GroupNorm(+SiLU) that CONSUMES the output, in one launch (round 4): workgroup (g, b) owns group g of
// sample b -- every pixel, the group's N / G channels -- sums the K-split slabs in order 0 .. S-1, runs the conv / GEMM epilogue (bias,
// row bias, residual, activation), stores the f16 / bf16 output as the plain reduce does, and then normalises exactly those stored
// values: two exact passes over registers (mean, then squared deviations; block reductions in a fixed order -> bitwise reproducible),
// affine, optional SiLU, second store. At the 16x16 / 8x8 levels this removes the separate single-pass GroupNorm launch that
// followed every split-K conv (4.5 - 6 us each on <= 2.6 MB, profiles/r03_kernels_per_op_run14.json): same arithmetic contract as
// sfast_hip_group_norm on the stored tensor (fp32 statistics, biased variance, rstd = rsqrt(var + eps)).
// Thread t owns the float4 column quads t, t + 512, ... of the (pixels x channels-of-the-group) block: ITEMS <= 8.
template <typename T, int ITEMS>
__global__ void __launch_bounds__(512) splitk_reduce_gn_kernel(const IgemmArgs a) {
    typedef const u32x2 __attribute__((address_space(1))) * g2_ptr;
    const g2_ptr zero = (g2_ptr)(const void *)g_zero16;
    __shared__ float red[2][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, b = blockIdx.y;
    const int HW = a.rows_per_batch, cpg = a.N / a.gn_groups, q4 = cpg / 4;
    const int nitems = HW * q4;
    const float rq4 = __builtin_amdgcn_rcpf((float)q4);
    const int64_t zstride = (int64_t)a.M * a.N;
    const bool res_now = a.res_before_act || a.act == SFAST_ACT_NONE;
    // phase 1: the slab loads. K-split index OUTERMOST, four splits x all items per round (up to 32 independent 16-byte loads in flight
    // per thread), every element still summed in the order z = 0, 1, 2, ... History (profiles/r04_reduce_gn_ab_run{7,8}.log): item by
    // item with the store in between, then item by item without it, both measured ~21 us for a 16x16 conv's reduce -- the compiler
    // does not interleave the runtime-length split loops of different items, so each item paid its own round trips.
    float v[ITEMS][4];
    const float *p0[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int i = tid + it * 512;
        const bool on = i < nitems;
        const int p = on ? fdiv22(i, q4, rq4) : 0, cq = on ? i - p * q4 : 0;
        p0[it] = a.partial + (int64_t)(b * HW + p) * a.N + g * cpg + cq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[it][e] = 0.f;
    }
    int z = 0;
    for (; z + 3 < a.splits; z += 4) {
        f32x4 t[ITEMS][4];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[it][k] = *reinterpret_cast<const f32x4 *>(p0[it] + (z + k) * zstride);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[it][e] += t[it][k][e];
    }
    for (; z < a.splits; ++z) {
        f32x4 t[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) t[it] = *reinterpret_cast<const f32x4 *>(p0[it] + z * zstride);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[it][e] += t[it][e];
    }
    // phase 2: epilogue, store, and the rounded values the statistics are taken over
    float s1 = 0.f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int i = tid + it * 512;
        const bool on = i < nitems;
        const int p = on ? fdiv22(i, q4, rq4) : 0, cq = on ? i - p * q4 : 0;
        const int m = b * HW + p, n = g * cpg + cq * 4;
        const u32x2 vb = *(a.bias ? (g2_ptr)(const void *)((const T *)a.bias + n) : zero);
        const u32x2 vb2 = *(a.rowbias ? (g2_ptr)(const void *)((const T *)a.rowbias + (int64_t)b * a.ld_rowbias + n) : zero);
        const u32x2 vr = *(a.res ? (g2_ptr)(const void *)((const T *)a.res + (int64_t)m * a.ldr + n) : zero);
        float b0[4], b1[4], r[4], o[4];
        unpack4<T>(vb, b0);
        unpack4<T>(vb2, b1);
        unpack4<T>(vr, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(v[it][e], a.out_scale, b0[e]) + b1[e] + (res_now ? r[e] * a.alpha : -0.0f);
        if (a.act != SFAST_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = apply_act(o[e], a.act) + (res_now ? -0.0f : r[e] * a.alpha);
        }
        const u32x2 packed = pack4<T>(o[0], o[1], o[2], o[3]);
        if (on) *reinterpret_cast<u32x2 *>((T *)a.out + (int64_t)m * a.ldo + n) = packed;
        unpack4<T>(packed, v[it]);  // the statistics see the STORED (rounded) values, as a separate GroupNorm launch would
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[it][e] = on ? v[it][e] : 0.f;
            s1 += v[it][e];
        }
    }
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    s1 = wave_sum(s1);
    if (lane == 0) red[0][wave] = s1;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot += red[0][w];
    const float mean = tot * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const bool on = tid + it * 512 < nitems;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = on ? v[it][e] - mean : 0.f;
            s2 = fmaf(d, d, s2);
        }
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wave] = s2;
    __syncthreads();
    float tot2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot2 += red[1][w];
    const float rstd = rsqrtf(tot2 * inv_n + a.gn_eps);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int i = tid + it * 512;
        if (i < nitems) {
            const int p = fdiv22(i, q4, rq4), cq = i - p * q4;
            const int m = b * HW + p, n = g * cpg + cq * 4;
            float ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.gn_gamma) unpack4<T>(*reinterpret_cast<const u32x2 *>((const T *)a.gn_gamma + n), ga);
            if (a.gn_beta) unpack4<T>(*reinterpret_cast<const u32x2 *>((const T *)a.gn_beta + n), be);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = (v[it][e] - mean) * rstd * ga[e] + be[e];
                o[e] = a.gn_act == SFAST_ACT_SILU ? act_silu(y) : y;
            }
            *reinterpret_cast<u32x2 *>((T *)a.gn_out + (int64_t)m * a.N + n) = pack4<T>(o[0], o[1], o[2], o[3]);
        }
    }
}

#endif  // SFAST_PROBES

// can the split-K reduce of an [M, N] problem also run the GroupNorm of its output? (host-side rule shared by igemm_run and the API query)
bool igemm_reduce_gn_ok(int M, int N, int rows_per_batch, int groups) {
    if (groups <= 0 || N % groups || rows_per_batch <= 0 || M % rows_per_batch) return false;
    const int cpg = N / groups;
    return cpg % 4 == 0 && N % 4 == 0 && (int64_t)rows_per_batch * (cpg / 4) <= 512 * 8 && M / rows_per_batch <= 65535;
}

// Split-K reduce + epilogue + GroupNorm partial statistics: a workgroup owns R = TY*RT complete output rows, thread (c, ty) the
// 8-channel chunk column c of rows ty, ty+TY, ... -- the row block is one statistics tile (tile_n = 1, bno = N) in the record layout
// of flush_staged_tile (igemm_device.h). Shifts: the slot's first element in the block's row 0, shared through LDS.
template <typename T, int RT>
__global__ void __launch_bounds__(512) splitk_reduce_rows_kernel(const IgemmArgs a, int TY) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    typedef const u32x4 __attribute__((address_space(1))) * g4_ptr;
    const g4_ptr zero = (g4_ptr)(const void *)g_zero16;
    const int CPR = a.N / 8;
    const int tid = threadIdx.x;
    const int c = tid % CPR, ty = tid / CPR;
    const bool active = ty < TY;
    const int R = TY * RT;
    const int row0 = blockIdx.x * R;
    const int n = c * 8;
    T *row0v = reinterpret_cast<T *>(rsm);                              // [N] the block's row 0 (shifts)
    float *red = reinterpret_cast<float *>(rsm + ((a.N * 2 + 15) / 16) * 16);  // [TY*CPR][4]
    const BatchOfRow batch_of(a);
    u32x4 outv[RT];
    const u32x4 vb = *(a.bias ? (g4_ptr)(const void *)((const T *)a.bias + n) : zero);
    float b0[8];
    unpack8<T>(vb, b0);
#pragma unroll
    for (int k = 0; k < RT; ++k) {
        const int m = row0 + ty + k * TY;
        outv[k] = u32x4{0u, 0u, 0u, 0u};
        if (!active || m >= a.M) continue;
        const int bi = a.rowbias ? batch_of(m) : 0;
        const u32x4 vb2 = *(a.rowbias ? (g4_ptr)(const void *)((const T *)a.rowbias + (int64_t)bi * a.ld_rowbias + n) : zero);
        const u32x4 vr = *(a.res ? (g4_ptr)(const void *)((const T *)a.res + (int64_t)m * a.ldr + n) : zero);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float *p0 = a.partial + (int64_t)m * a.N + n;
        const int64_t zstride = (int64_t)a.M * a.N;
        for (int z = 0; z < a.splits; ++z) {  // summation order z = 0, 1, 2, ...
            const f32x4 t0 = *reinterpret_cast<const f32x4 *>(p0 + z * zstride), t1 = *reinterpret_cast<const f32x4 *>(p0 + z * zstride + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] += t0[i];
                v[4 + i] += t1[i];
            }
        }
        float b1[8], r[8];
        unpack8<T>(vb2, b1);
        unpack8<T>(vr, r);
        const bool res_now = a.res_before_act || a.act == SFAST_ACT_NONE;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], a.out_scale, b0[i]) + b1[i] + (res_now ? r[i] * a.alpha : -0.0f);
        if (a.act != SFAST_ACT_NONE) {
#pragma unroll 1
            for (int i = 0; i < 8; ++i) v[i] = apply_act(v[i], a.act);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += res_now ? -0.0f : r[i] * a.alpha;
        }
        outv[k] = pack8<T>(v);
        *reinterpret_cast<u32x4 *>((T *)a.out + (int64_t)m * a.ldo + n) = outv[k];
        if (ty == 0 && k == 0) *reinterpret_cast<u32x4 *>(row0v + n) = outv[k];
    }
    __syncthreads();
    const int unit = a.gn_unit;
    const int U0 = n / unit;
    const int nb = min(8, (U0 + 1) * unit - n);
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
    if (active) {
        const float sha = (float)row0v[U0 * unit], shb = nb < 8 ? (float)row0v[(U0 + 1) * unit] : 0.f;
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            if (row0 + ty + k * TY >= a.M) continue;
            float f[8];
            unpack8<T>(outv[k], f);
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
        *reinterpret_cast<f32x4 *>(red + tid * 4) = f32x4{s1a, s2a, s1b, s2b};
    }
    __syncthreads();
    for (int j = tid; j < a.gn_slots; j += blockDim.x) {
        const int lo = j * unit, hi_ = min(a.N, lo + unit);
        float mean = 0.f, m2 = 0.f;
        if (hi_ > lo) {
            const float sh = (float)row0v[lo];
            float s1 = 0.f, s2 = 0.f;
            for (int cc = lo >> 3; cc <= (hi_ - 1) >> 3; ++cc) {
                const int part = ((cc * 8) / unit == j) ? 0 : 2;
                for (int t = 0; t < TY; ++t) {
                    const float *q = red + (t * CPR + cc) * 4 + part;
                    s1 += q[0];
                    s2 += q[1];
                }
            }
            const float cnt = (float)(hi_ - lo) * (float)min(R, a.M - row0);
            mean = sh + s1 / cnt;
            m2 = fmaxf(s2 - s1 * s1 / cnt, 0.f);
        }
        float *o = a.gn_stats + ((int64_t)blockIdx.x * a.gn_slots + j) * 2;
        o[0] = mean;
        o[1] = m2;
    }
}

// rows per workgroup of the statistics-emitting reduce: TY row phases x RT rows per thread
static void reduce_rows_geometry(int M, int N, int rows_per_sample, int &TY, int &RT) {
    const int CPR = N / 8;
    TY = 1;
    while (TY < 8 && CPR * TY * 2 <= 512) TY *= 2;
    RT = 1;
    if (g_batch_ref > 0 && rows_per_sample > 0) M = rows_per_sample * g_batch_ref;  // SFAST_BATCH_INVARIANT: row blocks of a sample do not follow the batch
    while (RT < 4 && M / (TY * RT * 2) >= 128) RT *= 2;  // >= 128 workgroups; fewer, larger row blocks = fewer records to merge
    while (TY * RT > 1 && rows_per_sample % (TY * RT) != 0) {
        if (RT > 1) RT /= 2; else TY /= 2;
    }
}

// ---- host side: variants, heuristics, launch -------------------------------------------------------
// Two main-loop structures share the tile shapes: pipe 0 = register-staged double buffer (this
// file), pipe 1 = LDS-DMA ring (igemm_glds.hip). Variant ids 1..5 select pipe 0, 11..15 pipe 1.
struct Variant {
    int id, BM, BN, WM, WN, pipe, ns;  // ns = LDS ring depth (2 for the register pipe's double buffer)
    float eff;  // relative efficiency of the tile shape (arithmetic intensity / LDS pressure)
};
// BN = weight rows per tile (GEGLU variants produce BN/2 output columns)
static const Variant kVariants[] = {
    {1, 128, 128, 2, 2, 0, 2, 1.00f},  {2, 128, 160, 4, 1, 0, 2, 1.00f},  {3, 64, 64, 2, 2, 0, 2, 0.60f},
    {4, 64, 160, 2, 1, 0, 2, 0.80f},   {5, 256, 128, 4, 2, 0, 2, 1.10f},  {11, 128, 128, 2, 2, 1, 4, 1.00f},
    {12, 128, 160, 4, 1, 1, 4, 1.00f}, {13, 64, 64, 2, 2, 1, 5, 0.60f},   {14, 64, 160, 2, 1, 1, 4, 0.80f},
    {15, 256, 128, 4, 2, 1, 3, 1.10f}, {16, 128, 128, 2, 2, 1, 2, 1.00f}, {17, 128, 160, 4, 1, 1, 2, 1.00f},
    {18, 64, 64, 2, 2, 1, 3, 0.60f},
    // pipe 2: wave-specialised LDS-DMA (igemm_glds_ws.hip): 4 producer waves + WM*WN consumer waves
    {21, 128, 128, 2, 2, 2, 4, 1.00f}, {22, 128, 160, 4, 1, 2, 4, 1.00f}, {23, 64, 64, 2, 2, 2, 4, 0.60f},
    {24, 128, 64, 2, 2, 2, 3, 0.80f},  {25, 64, 128, 2, 2, 2, 3, 0.80f},  // autotuner candidates (ids >= 16 are skipped by the analytic planner)
    {26, 64, 64, 2, 2, 2, 3, 0.60f},   // 48 KB ring: three workgroups per CU
    // pipe 3: LDS-resident input patch for 3x3 / stride 1 / pad 1 convs (conv_patch.hip); autotuner candidates
    {31, 128, 160, 4, 1, 3, 5, 1.00f}, {32, 128, 128, 2, 2, 3, 5, 1.00f}, {34, 128, 64, 2, 2, 3, 5, 0.80f},  // (ring: 3 .. 5, by LDS left)
    // pipe 4: packed weights global -> VGPR, activations through a 4-deep LDS ring (igemm_pk.hip); BM pixels x BN weight rows, 1 x WN waves;
    // autotuner candidates, only when the caller hands over packed copies (caps bit 1)
    {41, 128, 256, 1, 4, 4, 4, 1.10f}, {42, 64, 256, 1, 4, 4, 4, 0.90f},  {43, 64, 320, 1, 5, 4, 4, 0.90f},
    {44, 128, 160, 1, 5, 4, 4, 0.90f}, {45, 64, 160, 1, 5, 4, 4, 0.70f},  {46, 128, 128, 1, 4, 4, 4, 0.90f},
    // pipe 5 (round 6): 256-row ping-pong tiles, 8 waves in two alternating groups (igemm_pp.h); autotuner candidates for large M
    {51, 256, 128, 4, 2, 5, 3, 1.30f}, {52, 256, 160, 4, 2, 5, 3, 1.30f}, {53, 256, 256, 2, 4, 5, 2, 1.40f},
    {55, 256, 128, 4, 2, 5, 3, 1.30f}, {56, 256, 160, 4, 2, 5, 3, 1.30f},  // + four producer waves (12 waves per workgroup)
    {57, 256, 128, 4, 2, 5, 3, 1.30f}, {58, 256, 160, 4, 2, 5, 3, 1.30f},  // producers + LOCKSTEP consumers (one barrier per K-tile)
};
static const Variant kGegluVariants[] = {
    {1, 128, 128, 2, 2, 0, 2, 1.00f},  {3, 64, 128, 2, 2, 0, 2, 0.75f},  {11, 128, 128, 2, 2, 1, 4, 1.00f},
    {13, 64, 128, 2, 2, 1, 5, 0.75f},  {16, 128, 128, 2, 2, 1, 2, 1.00f}, {18, 64, 128, 2, 2, 1, 3, 0.75f},
    {21, 128, 128, 2, 2, 2, 4, 1.00f}, {23, 64, 128, 2, 2, 2, 3, 0.75f},
    {53, 256, 256, 2, 4, 5, 2, 1.40f},  // pipe 5: 256 pixels x (128 h + 128 g) weight rows
    {57, 256, 128, 4, 2, 5, 3, 1.20f},  // pipe 5, producers + lockstep consumers: 256 pixels x (64 h + 64 g) weight rows
};

extern unsigned long long *g_igemm_trace;  // igemm_glds.hip
int igemm_glds_ws_init();  // igemm_glds_ws.hip
int igemm_glds_ws_launch(const IgemmArgs &a, int dtype, int mode, bool geglu, int BM, int BN, int NS, hipStream_t st);
int igemm_pk_init();                                                                                 // igemm_pk.hip
int igemm_pk_launch(const IgemmArgs &a, int dtype, int mode, int BM, int BN, hipStream_t st);       // igemm_pk.hip
int igemm_pp_init();                                                                                 // igemm_pp.hip
int igemm_pp_launch(const IgemmArgs &a, int dtype, int mode, bool geglu, int BN, int pw, hipStream_t st);  // igemm_pp.hip
int igemm_glds_init();                                                                               // igemm_glds.hip
int igemm_glds_launch(const IgemmArgs &a, int dtype, int mode, bool geglu, int BM, int BN, int NS, hipStream_t st);  // igemm_glds.hip

template <typename T, int BM, int BN, int WM, int WN, int MODE, bool GEGLU, bool STAGED = false>
static int launch_one(const IgemmArgs &a, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = igemm_kernel<T, BM, BN, WM, WN, MODE, GEGLU, STAGED>;
    hipLaunchKernelGGL(kern, igemm_grid(a), dim3(WM * WN * 64), smem, st, a);
    return check_launch("igemm");
}

template <typename T, int BM, int BN, int WM, int WN, int MODE, bool GEGLU, bool STAGED = false>
static int set_attr_one() {
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = igemm_kernel<T, BM, BN, WM, WN, MODE, GEGLU, STAGED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(igemm %dx%d): %s", BM, BN, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

#define SFAST_FOR_VARIANTS(T, MODE, OP)              \
    OP(T, 128, 128, 2, 2, MODE, false)               \
    OP(T, 128, 160, 4, 1, MODE, false)               \
    OP(T, 64, 64, 2, 2, MODE, false)                 \
    OP(T, 64, 160, 2, 1, MODE, false)                \
    OP(T, 256, 128, 4, 2, MODE, false)

#define SFAST_FOR_GEGLU_VARIANTS(T, OP) \
    OP(T, 128, 128, 2, 2, 0, true)      \
    OP(T, 64, 128, 2, 2, 0, true)

static int g_pipe_pref = -1;  // -1 auto, 0 force register pipe, 1 force LDS-DMA pipe (SFAST_IGEMM_PIPE)
static int g_xmap_pref = 1;   // SFAST_XCD_MAP=0: legacy block order everywhere; 1 (default): box maps; 2: + contiguous runs (A/B of choose_xcd_map)
static int g_stage_pref = 0;  // 1: stage every eligible output tile through LDS (SFAST_STAGE_OUT=1), not only those that emit statistics

int igemm_grouped_init();
int igemm_init() {
    int rc = 0;
#define INIT_OP(T, BM, BN, WM, WN, MODE, G)                  \
    if (!rc) rc = set_attr_one<T, BM, BN, WM, WN, MODE, G>(); \
    if (!rc && !G) rc = set_attr_one<T, BM, BN, WM, WN, MODE, false, true>();
    SFAST_FOR_VARIANTS(f16, 0, INIT_OP)
    SFAST_FOR_VARIANTS(f16, 1, INIT_OP)
    SFAST_FOR_VARIANTS(bf16, 0, INIT_OP)
    SFAST_FOR_VARIANTS(bf16, 1, INIT_OP)
    SFAST_FOR_GEGLU_VARIANTS(f16, INIT_OP)
    SFAST_FOR_GEGLU_VARIANTS(bf16, INIT_OP)
#undef INIT_OP
    if (!rc) rc = igemm_grouped_init();
    if (!rc) rc = igemm_glds_init();
    if (!rc) rc = igemm_glds_ws_init();
    if (!rc) rc = igemm_pk_init();
    if (!rc) rc = igemm_pp_init();
    const char *so = getenv("SFAST_STAGE_OUT");
    g_stage_pref = (so && so[0] == '1') ? 1 : 0;
    const char *xm = getenv("SFAST_XCD_MAP");
    // 2: + contiguous runs (round 4). Measured and NOT the default: it cuts the modelled XCD-to-XCD duplication of the 32^2-level convs from
    // 62 MB to ~25 MB and changes nothing in the step (184.07 vs 183.99 it/s, SDXL 41.74 vs 41.73: profiles/r04_xcd_run_map_ab_run25.log)
    // -- the fabric duplication behind `roofline.traffic` = 3.9 x algorithmic is not what holds these launches.
#ifdef SFAST_PROBES
    g_xmap_pref = (xm && xm[0] == '0') ? 0 : (xm && xm[0] == '2') ? 2 : 1;
#else
    g_xmap_pref = (xm && xm[0] == '0') ? 0 : 1;  // the run map is a probe-build candidate: "2" means "1" here
#endif
    const char *e = getenv("SFAST_IGEMM_PIPE");
    if (e && e[0] == 'r') g_pipe_pref = 0;
    if (e && e[0] == 'g') g_pipe_pref = 1;
    return rc;
}

template <typename T, int MODE>
static int dispatch_variant(const IgemmArgs &a, const Variant &v, bool geglu, hipStream_t st) {
#define LAUNCH_OP(TT, BM_, BN_, WM_, WN_, MODE_, G_)                                                    \
    if (v.BM == BM_ && v.BN == BN_ && v.WM == WM_ && v.WN == WN_ && geglu == G_) {                     \
        if constexpr (!G_) {                                                                           \
            if (a.stage_out) return launch_one<TT, BM_, BN_, WM_, WN_, MODE_, false, true>(a, st);     \
        }                                                                                              \
        return launch_one<TT, BM_, BN_, WM_, WN_, MODE_, G_>(a, st);                                   \
    }
    if (!geglu) {
        SFAST_FOR_VARIANTS(T, MODE, LAUNCH_OP)
    } else {
        if (MODE == 0) {
            SFAST_FOR_GEGLU_VARIANTS(T, LAUNCH_OP)
        }
    }
#undef LAUNCH_OP
    set_error("igemm: no kernel for variant BM=%d BN=%d", v.BM, v.BN);
    return SFAST_ERR_UNSUPPORTED;
}

struct IgemmPlan {
    Variant v;
    int splits, ktps, tiles_m, tiles_n, ktiles;
};

// Pick main-loop structure, tile shape and split-K factor with a small analytic model (ns):
//   t = max(waves * t_workgroup, t_memory) + per-workgroup prologue/epilogue + split-K slab traffic
// calibrated on MI355X sweeps (profiles/r01_tune_*.json): a CU sustains ~560 MAC/ns with one resident
// workgroup of the register pipe and ~650 MAC/ns with two; the LDS-DMA pipe keeps NS-1 tiles in flight
// and is modelled by its own rate. The 16x16 / 8x8 UNet levels (M <= 512, K up to 23k) are
// weight-streaming bound and want as many K-splits as it takes to put ~2 workgroups on every CU;
// measured optima cluster at 480..640 workgroups.
// Staged-store / statistics kernels exist for the register pipe and the wave-specialised pipe; a forced LDS-DMA ring variant
// is replaced by the same tile shape in one of those (ring depth is a latency knob, the arithmetic is identical).
static int staged_variant(int id) {
    switch (id) {
    case 11: case 16: return 21;
    case 12: case 17: return 22;
    case 13: case 18: return 23;
    case 14: return 4;
    case 15: return 5;
    default: return id;
    }
}

static IgemmPlan igemm_plan(int M, int N, int K, bool geglu, int force_variant, int force_split, int caps) {
    const bool glds_ok = (caps & 1) != 0;
    const int patch_w = (caps >> 8) & 0xfff, patch_h = (caps >> 20) & 0xfff;
    const Variant *vs = geglu ? kGegluVariants : kVariants;
    const int nv = geglu ? (int)(sizeof(kGegluVariants) / sizeof(Variant)) : (int)(sizeof(kVariants) / sizeof(Variant));
    const int ktiles = ceil_div(K, 64);
    static const int kSplitCand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    IgemmPlan best{};
    double best_cost = 1e300;
    for (int i = 0; i < nv; ++i) {
        const Variant &v = vs[i];
        if (force_variant) {
            if (v.id != force_variant) continue;
        } else {
            if (v.BM == 64 && v.BN == 160) continue;  // never the measured optimum on MI355X sweeps
            if (v.id >= 16) continue;                  // shallow rings are autotuner candidates only
            if (v.pipe >= 1 && (!glds_ok || g_pipe_pref == 0)) continue;
            if (v.pipe == 0 && glds_ok && g_pipe_pref == 1) continue;
        }
        if (v.pipe == 5 && !glds_ok && (caps & 4) && v.id >= 55) {
            // fused-upsample conv: the producer waves of pipe 5 address it (igemm_pp.h pp_decode_row); nothing else with LDS-DMA does
        } else if (v.pipe >= 1 && !glds_ok) {
            continue;
        }
        if (v.pipe == 4 && !(caps & 2)) continue;
        if (v.pipe == 5 && (K % 64 != 0 || M < 256)) continue;  // no K tail in the ping-pong pipe; one full tile of rows at least
        if (v.pipe == 3 && !(patch_w > 0 && K % 576 == 0 && conv_patch_fits(patch_h, patch_w, M, v.BM, v.BN))) continue;
        const int bno = geglu ? v.BN / 2 : v.BN;
        const int tm = ceil_div(M, v.BM), tn = ceil_div(N, bno);
        const int tiles = tm * tn;
        const int lds = v.pipe == 4 ? v.ns * v.BM * 128 : v.ns * (v.BM + v.BN) * 128;
        const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
        const double wrows = geglu ? 2.0 * N : (double)N;
        // unique operand bytes stream from HBM (~4 TB/s); panel re-reads by other tiles are served
        // by the XCD L2s / Infinity Cache (~15 TB/s aggregate)
        const double t_mem = (wrows * K + (double)M * K) * 2.0 / 4000.0 +
                             ((double)(tm - 1) * wrows * K + (double)(tn - 1) * M * K) * 2.0 / 15000.0;
        int last_splits = 0;
        for (int ci = 0; ci < (int)(sizeof(kSplitCand) / sizeof(int)); ++ci) {
            int s = force_split ? force_split : kSplitCand[ci];
            if (s > ktiles) s = ktiles;
            if (s < 1) s = 1;
            int ktps = ceil_div(ktiles, s);
            if (v.pipe == 3) ktps = 9 * ceil_div(ktiles / 9, s);  // the patch pipe cuts K between 64-channel slices (9 taps each)
            const int splits = ceil_div(ktiles, ktps);
            if (splits == last_splits) continue;
            last_splits = splits;
            if (!force_split && splits > 1 && ktps < 4) continue;
            const int wgs = tiles * splits;
            const double waves = (double)ceil_div(wgs, 256 * wg_per_cu);
            int share = ceil_div(wgs, 256);
            if (share > wg_per_cu) share = wg_per_cu;
            // MAC/ns available to one workgroup
            double rate;
            if (v.pipe == 0)
                rate = (share == 1 ? 560.0 : 325.0) * v.eff;
            else
                rate = (share == 1 ? 700.0 : 380.0) * v.eff;
            const double t_wg = (double)v.BM * v.BN * 64.0 * ktps / rate + (v.pipe ? 2500.0 : 4000.0);
            double cost = waves * t_wg;
            if (cost < t_mem) cost = t_mem;
            cost += 2000.0;
            if (splits > 1) cost += (double)splits * M * wrows * 8.0 / 3000.0 + 2000.0;
            if (cost < best_cost) {
                best_cost = cost;
                best.v = v;
                best.splits = splits;
                best.ktps = ktps;
                best.tiles_m = tm;
                best.tiles_n = tn;
                best.ktiles = ktiles;
            }
            if (force_split) break;
        }
    }
    if (best_cost == 1e300) {
        // forced variant id unknown / not applicable: smallest register-pipe tile, no split
        best.v = vs[geglu ? 1 : 2];
        const int bno = geglu ? best.v.BN / 2 : best.v.BN;
        best.splits = 1;
        best.ktps = ktiles;
        best.tiles_m = ceil_div(M, best.v.BM);
        best.tiles_n = ceil_div(N, bno);
        best.ktiles = ktiles;
    }
    return best;
}

// whether a split-K plan finishes its tiles inside the GEMM kernel (splitk_join, igemm_device.h) instead of a second launch
static bool splitk_joins(const IgemmPlan &p, bool tickets, bool stats, int rows_per_sample) {
#ifndef SFAST_PROBES
    (void)p; (void)tickets; (void)stats; (void)rows_per_sample;
    return false;  // product build: the join is not compiled in (igemm_device.h) -- SFAST_EXT_WS_TICKETS is accepted and ignored
#else
    return p.splits > 1 && tickets && p.v.pipe != 1 && (int64_t)p.tiles_m * p.tiles_n <= SFAST_WS_TICKET_BYTES / 4 && (!stats || rows_per_sample % p.v.BM == 0);
#endif
}

bool igemm_stats_layout(int M, int N, int K, bool geglu, int variant, int split, int glds_ok, int unit, int rows_per_sample,
                        bool tickets, StatsLayout &out) {
    if (geglu || unit < 8 || rows_per_sample <= 0 || M % rows_per_sample != 0 || N % 8 != 0) return false;
    IgemmPlan p = igemm_plan(M, N, K, geglu, variant, split, glds_ok);
    if (p.v.pipe == 1 && p.splits == 1) p = igemm_plan(M, N, K, geglu, staged_variant(p.v.id), p.splits, glds_ok);
    if (p.splits == 1 || splitk_joins(p, tickets, true, rows_per_sample)) {
        if (rows_per_sample % p.v.BM != 0) return false;
        out.rb_rows = p.v.BM;
        out.bno = p.v.BN;
        out.slots = stats_slots(p.v.BN, unit);
        out.tiles_n = p.tiles_n;
        out.n_rb = M / p.v.BM;
        return true;
    }
    if (N / 8 > 512 || N * 2 + 16 + 512 * 16 > 64 * 1024) return false;
    int ty, rt;
    reduce_rows_geometry(M, N, rows_per_sample, ty, rt);
    out.rb_rows = ty * rt;
    out.bno = N;
    out.slots = ceil_div(N, unit);
    out.tiles_n = 1;
    out.n_rb = M / (ty * rt);
    return true;
}

void igemm_plan_query(int M, int N, int K, bool geglu, int variant, int split, int glds_ok, int out[5]) {
    IgemmPlan p = igemm_plan(M, N, K, geglu, variant, split, glds_ok);
    out[0] = p.v.BM;
    out[1] = p.v.BN;
    out[2] = p.splits;
    out[3] = p.ktps;
    out[4] = p.v.id;
}

size_t igemm_workspace_bytes(int M, int N, int K, bool geglu, int variant, int split, int glds_ok) {
    IgemmPlan p = igemm_plan(M, N, K, geglu, variant, split, glds_ok);
    if (p.splits <= 1) return 0;
    // row-major slabs for the reduce kernel, or whole-tile slabs in fragment order + the ticket block at the end (whichever the
    // caller's flags select at launch time)
    const size_t rows = (size_t)p.splits * M * (geglu ? 2 * (size_t)N : (size_t)N) * sizeof(float);
    const size_t tiles = (size_t)p.splits * p.tiles_m * p.tiles_n * p.v.BM * p.v.BN * sizeof(float);
    return (rows > tiles ? rows : tiles) + SFAST_WS_TICKET_BYTES;
}

// The LDS-DMA pipe takes every linear problem; conv problems need uniform taps per K-tile.
bool igemm_glds_eligible(const IgemmArgs &a, int mode) {
    if (mode == 0) return true;
    // the LDS-DMA kernel addresses activations with 32-bit element offsets
    const int64_t batch = (a.Ho > 0 && a.Wo > 0) ? (int64_t)a.M / ((int64_t)a.Ho * a.Wo) : 0;
    const int64_t elems = batch * a.H * a.W * (int64_t)(a.C1 > a.C2 ? a.C1 : a.C2);
    return !a.ups && a.C1 % 64 == 0 && a.C2 % 64 == 0 && a.KH * a.KW <= 32 && elems < (1ll << 31);
}

template <typename T, int BM, int BN, int WM, int WN> static int set_attr_grouped() {
    constexpr int smem = 2 * (BM + BN) * 128;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(igemm_grouped_kernel<T, BM, BN, WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(igemm_grouped %dx%d): %s", BM, BN, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}
int igemm_grouped_init() {
    int rc = set_attr_grouped<f16, 64, 64, 2, 2>();
    if (!rc) rc = set_attr_grouped<bf16, 64, 64, 2, 2>();
    if (!rc) rc = set_attr_grouped<f16, 128, 128, 2, 2>();
    if (!rc) rc = set_attr_grouped<bf16, 128, 128, 2, 2>();
    return rc;
}

template <typename T, int BM, int BN> static int w8_launch(const IgemmArgs &a, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = igemm_w8_kernel<T, BM, BN, 2, 2>;
    static bool attr_done = false;  // idempotent; re-applied per process (cheap)
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n, 1), dim3(256), smem, st, a);
    return check_launch("igemm_w8");
}

int igemm_run_w8(IgemmArgs &a, int dtype, hipStream_t st) {
    const bool big = (int64_t)ceil_div(a.M, 64) * ceil_div(a.N, 64) > 4 * 256 && a.M >= 128;
    const int BM = big ? 128 : 64, BN = big ? 128 : 64;
    a.tiles_m = ceil_div(a.M, BM);
    a.tiles_n = ceil_div(a.N, BN);
    a.ktiles = ceil_div(a.K, 64);
    a.ktiles_per_split = a.ktiles;
    a.splits = 1;
    a.partial = nullptr;
    a.trace = nullptr;
    a.stage_out = 0;
    a.gn_stats = nullptr;
    a.w_int8 = 1;
    if (a.out_scale == 0.f) a.out_scale = 1.0f;
    set_kernel_name("igemm_w8_%s[%dx%d,reg]", dtype == SFAST_F16 ? "f16" : "bf16", BM, BN);
    if (dtype == SFAST_F16) return big ? w8_launch<f16, 128, 128>(a, st) : w8_launch<f16, 64, 64>(a, st);
    return big ? w8_launch<bf16, 128, 128>(a, st) : w8_launch<bf16, 64, 64>(a, st);
}

// n_groups problems of identical [M, N, K]; per group an activation pointer (usually shared), up to two stacked weight segments, a bias and an output.
int igemm_run_grouped(IgemmArgs &a, int dtype, int n_groups, const void *const *xs, const void *const *w_segs, int n_wseg,
                      const void *const *bias, void *const *out, hipStream_t st) {
    IgemmGroupTab g{};
    for (int i = 0; i < SFAST_MAX_GEMM_GROUPS; ++i) {
        const int j = i < n_groups ? i : 0;
        g.x[i] = xs[j];
        g.w0[i] = w_segs[j * n_wseg];
        g.w1[i] = w_segs[j * n_wseg + (n_wseg > 1 ? 1 : 0)];
        g.bias[i] = bias ? bias[j] : nullptr;
        g.out[i] = out[j];
    }
    // tile: 64x64 unless that makes more than ~6 workgroups per CU (then 128x128 halves the operand re-reads)
    const int64_t wg64 = (int64_t)ceil_div(a.M, 64) * ceil_div(a.N, 64) * n_groups;
    const bool big = wg64 > 6 * 256 && a.M >= 128;
    const int BM = big ? 128 : 64, BN = big ? 128 : 64;
    a.tiles_m = ceil_div(a.M, BM);
    a.tiles_n = ceil_div(a.N, BN);
    a.ktiles = ceil_div(a.K, 64);
    a.ktiles_per_split = a.ktiles;
    a.splits = 1;
    a.partial = nullptr;
    a.trace = nullptr;
    a.out_scale = 1.0f;
    a.stage_out = 0;
    a.gn_stats = nullptr;
    set_kernel_name("igemm_grouped_%s[%dx%d,G=%d,reg]", dtype == SFAST_F16 ? "f16" : "bf16", BM, BN, n_groups);
    const dim3 grid(a.tiles_m * a.tiles_n, 1, n_groups);
#define GROUPED_LAUNCH(T, BM_, BN_) \
    hipLaunchKernelGGL((igemm_grouped_kernel<T, BM_, BN_, 2, 2>), grid, dim3(256), 2 * (BM_ + BN_) * 128, st, a, g)
    if (dtype == SFAST_F16) {
        if (big) GROUPED_LAUNCH(f16, 128, 128); else GROUPED_LAUNCH(f16, 64, 64);
    } else {
        if (big) GROUPED_LAUNCH(bf16, 128, 128); else GROUPED_LAUNCH(bf16, 64, 64);
    }
#undef GROUPED_LAUNCH
    return check_launch("igemm_grouped");
}

// Block -> (tile, K-split) map over the 8 XCD L2s (decode_block, igemm_device.h). Every factorisation 8 = xs * xm * xn into K-split,
// row and column boxes that divides (splits, tiles_m, tiles_n) is priced by the bytes it pulls across the fabric,
// activations * xn + weights * xm, and the cheapest wins; K-split boxes replicate nothing, so split problems hand the whole
// factor to xs when they can (each XCD then streams only its K-slice of both operands: the 1280-channel convs of the 16x16 level
// read their 29-59 MB of weights ONCE instead of once per XCD that owns a tile of the column). Ties go to fewer column boxes
// (an activation panel stays in one L2, as in the legacy order). No dividing factorisation: legacy order (xmap = 0).
// bytes the eight XCD L2s fetch under a block -> (tile, split) map: every distinct (split, tile_n) pair an XCD touches costs one weight
// panel slice, every distinct (split, tile_m) pair one activation panel slice
template <typename F> static double xcd_map_bytes(const IgemmArgs &a, double w_pair, double x_pair, int blocks_per_xcd, F &&decode) {
    double bytes = 0.0;
    std::vector<char> seen_n((size_t)a.splits * a.tiles_n), seen_m((size_t)a.splits * a.tiles_m);
    for (int xcd = 0; xcd < 8; ++xcd) {
        std::fill(seen_n.begin(), seen_n.end(), 0);
        std::fill(seen_m.begin(), seen_m.end(), 0);
        for (int k = 0; k < blocks_per_xcd; ++k) {
            int tm, tn, sp;
            if (!decode(xcd, k, tm, tn, sp)) continue;
            char &n = seen_n[(size_t)sp * a.tiles_n + tn], &m = seen_m[(size_t)sp * a.tiles_m + tm];
            if (!n) bytes += w_pair;
            if (!m) bytes += x_pair;
            n = m = 1;
        }
    }
    return bytes;
}

static void choose_xcd_map(IgemmArgs &a, int mode, bool geglu) {
    a.xmap = 0;
    a.x_per = a.x_order = 0;
    if (!g_xmap_pref) return;
    // unique operand bytes (a conv reads every input pixel once, whatever its im2col row count)
    const double act_bytes = mode ? (double)(a.M / (a.Ho * a.Wo)) * a.H * a.W * (a.ups ? 0.25 : 1.0) * (a.C1 + a.C2) * 2.0 : (double)a.M * a.K * 2.0;
    const double w_bytes = (geglu ? 2.0 : 1.0) * (double)a.N * a.K * 2.0;
    double best = 1e300;
    for (int lxs = 3; lxs >= 0; --lxs) {
        if (a.splits % (1 << lxs)) continue;
        for (int lxn = 0; lxn + lxs <= 3; ++lxn) {
            const int lxm = 3 - lxs - lxn;
            if (a.tiles_m % (1 << lxm) || a.tiles_n % (1 << lxn)) continue;
            const double cost = act_bytes * (1 << lxn) + w_bytes * (1 << lxm);
            if (cost < best) {
                best = cost;
                a.xmap = 1;
                a.x_lxn = lxn;
                a.x_lxm = lxm;
                a.x_tn = a.tiles_n >> lxn;
                a.x_tm = a.tiles_m >> lxm;
                a.x_sp = a.splits >> lxs;
            }
        }
    }
    // round 4: contiguous runs of a linear (split, tile_n, tile_m) order -- no divisibility needed (5 tile columns x 3 splits of the
    // 640 -> 640 @ 32^2 conv: the box map can only cut the 16 tile rows, every XCD fetches ALL weights: 62 MB against ~25 MB)
#ifdef SFAST_PROBES
    if (g_xmap_pref >= 2) {
        const int total = a.tiles_m * a.tiles_n * a.splits;
        if (total >= 16 && total <= 65536) {
            const int per = ceil_div(total, 8);
            const double w_pair = w_bytes / ((double)a.splits * a.tiles_n), x_pair = act_bytes / ((double)a.splits * a.tiles_m);
            double cur = best;
            if (a.xmap == 0) cur = 1e300;
            for (int order = 0; order < 2; ++order) {
                const double c = xcd_map_bytes(a, w_pair, x_pair, per, [&](int xcd, int k, int &tm, int &tn, int &sp) {
                    const int lid = xcd * per + k;
                    if (lid >= total) return false;
                    const int d1 = order ? a.tiles_n : a.tiles_m, d2 = order ? a.tiles_m : a.tiles_n;
                    const int q = lid / d1, s = q / d2, i1 = lid - q * d1, i2 = q - s * d2;
                    tm = order ? i2 : i1;
                    tn = order ? i1 : i2;
                    sp = s;
                    return true;
                });
                if (c < 0.85 * cur) {  // a clear win only: the box map keeps the tile order friendlier to the L2 within an XCD
                    cur = c;
                    a.xmap = 2;
                    a.x_per = per;
                    a.x_order = order;
                }
            }
        }
    }
#endif
}

// entry used by api_gemm_conv.hip. mode: 0 linear, 1 conv. Fills plan fields of `a`.
int igemm_run(IgemmArgs &a, int dtype, int mode, bool geglu, int variant, int split, void *ws, size_t ws_bytes,
              hipStream_t st) {
    const bool glds_elig = igemm_glds_eligible(a, mode);
    const bool patch_elig = mode == 1 && glds_elig && a.KH == 3 && a.KW == 3 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 1 && a.pad_w == 1 &&
                            a.dil_h == 1 && a.dil_w == 1 && !a.ups && a.Ho == a.H && a.Wo == a.W;
    // pipe 4 needs a packed copy of every weight segment the problem touches and row blocks that do not straddle segments
    const int nseg = a.rows_per_seg > 0 ? ceil_div(a.N, a.rows_per_seg) : 1;
    bool packed = glds_elig && !geglu && !a.w_int8 && nseg <= SFAST_MAX_WSEG && (nseg == 1 || a.rows_per_seg % 32 == 0);
    for (int i = 0; i < nseg && packed; ++i) packed = a.wpk[i] != nullptr;
    // pipe 5's producer waves also take a single-source 3x3-style conv with the nearest-2x upsample fused into the gather
    const int64_t ups_elems = (a.Ho > 0 && a.Wo > 0) ? (int64_t)a.M / ((int64_t)a.Ho * a.Wo) * a.H * a.W * a.C1 : 0;
    const bool pp_ups = mode == 1 && a.ups && a.x2 == nullptr && a.C2 == 0 && a.C1 % 64 == 0 && a.dil_h == 1 && a.dil_w == 1 && a.KH * a.KW <= 30 &&
                        ups_elems < (1ll << 31);
    const int glds_ok = igemm_caps(glds_elig, patch_elig ? a.H : 0, patch_elig ? a.W : 0, packed, pp_ups);
    a.pk_ksteps = ceil_div(a.K, 64) * 4;
    IgemmPlan p = igemm_plan(a.M, a.N, a.K, geglu, variant, split, glds_ok);
    const bool want_staged = !geglu && (a.gn_stats != nullptr || g_stage_pref > 0);
    if (want_staged && p.v.pipe == 1 && p.splits == 1) p = igemm_plan(a.M, a.N, a.K, geglu, staged_variant(p.v.id), p.splits, glds_ok);
    a.tiles_m = p.tiles_m;
    a.tiles_n = p.tiles_n;
    a.ktiles = p.ktiles;
    a.ktiles_per_split = p.ktps;
    a.splits = p.splits;
    a.partial = nullptr;
    a.trace = g_igemm_trace;
    choose_xcd_map(a, mode, geglu);
    if (a.out_scale == 0.f) a.out_scale = 1.0f;
    // staged (LDS -> 16-byte row segments) stores need 16-byte aligned output rows; statistics additionally whole tiles per sample
    const int bno_sel = geglu ? p.v.BN / 2 : p.v.BN;
    const bool stage_ok = a.N % 8 == 0 && a.ldo % 8 == 0 && aligned16(a.out);
    int red_ty = 0, red_rt = 0;
    const bool joins = splitk_joins(p, a.tickets != nullptr, a.gn_stats != nullptr, a.gn_rows_per_sample);
    if (!joins) a.tickets = nullptr;
    if (a.gn_stats) {
        SFAST_REQUIRE(!geglu && stage_ok && a.gn_unit >= 8 && a.gn_rows_per_sample > 0 && a.M % a.gn_rows_per_sample == 0, SFAST_ERR_UNSUPPORTED,
                      "igemm: GroupNorm statistics need a non-GEGLU problem with 16-byte aligned output rows and unit >= 8");
        if (p.splits == 1 || joins) {
            SFAST_REQUIRE(a.gn_rows_per_sample % p.v.BM == 0, SFAST_ERR_UNSUPPORTED, "igemm: %d rows per sample do not tile by BM=%d",
                          a.gn_rows_per_sample, p.v.BM);
            a.gn_slots = stats_slots(bno_sel, a.gn_unit);
        } else {
            SFAST_REQUIRE(a.N / 8 <= 512 && a.N * 2 + 16 + 512 * 16 <= 64 * 1024, SFAST_ERR_UNSUPPORTED, "igemm: N=%d too wide for the statistics reduce", a.N);
            reduce_rows_geometry(a.M, a.N, a.gn_rows_per_sample, red_ty, red_rt);
            a.gn_slots = ceil_div(a.N, a.gn_unit);
        }
    }
    // pipe 5 stages every eligible tile: its six-fragment waves would otherwise run the one-fragment-ahead epilogue with 8-byte global
    // stores between the operand fetches, and gfx950 counts loads and stores in ONE vmcnt -- the wait for fragment f + 1's operands also
    // waits for fragment f's stores to be acknowledged (~1.15 us per fragment on a busy chip: 7 us of epilogue behind a 70 us K loop,
    // profiles/r06_pp_loop_probe_run8_timeline.log). Staged, the fragment loop only writes LDS and the tile leaves as 16-byte segments.
    a.stage_out = ((p.splits == 1 || joins) && !geglu && stage_ok && (a.gn_stats != nullptr || g_stage_pref > 0 || p.v.pipe == 5)) ? 1 : 0;
    if (p.splits > 1) {
        const size_t need = joins ? (size_t)p.splits * p.tiles_m * p.tiles_n * p.v.BM * p.v.BN * sizeof(float)
                                  : (size_t)p.splits * a.M * (geglu ? 2 * (size_t)a.N : (size_t)a.N) * sizeof(float);
        SFAST_REQUIRE(ws && ws_bytes >= need, SFAST_ERR_WORKSPACE, "igemm: workspace %zu < %zu", ws_bytes, need);
        a.partial = (float *)ws;
    }
    if (a.gn_out) {
        // sfast_epilogue_ext.gn_out -- the consumer GroupNorm inside the split-K reduce launch. Refused BEFORE anything is launched: an error
        // after the main kernel would leave `out` written, `gn_out` untouched and, inside a capture, a dangling node.
#ifdef SFAST_PROBES
        SFAST_REQUIRE(p.splits > 1 && !joins && !geglu && igemm_reduce_gn_ok(a.M, a.N, a.rows_per_batch, a.gn_groups) && a.ldo % 4 == 0, SFAST_ERR_UNSUPPORTED,
                      "igemm: the fused GroupNorm epilogue needs a split-K plan (got %d splits) and N / G %% 4 == 0, H*W * N / G <= 16384", p.splits);
#else
        SFAST_REQUIRE(false, SFAST_ERR_UNSUPPORTED, "igemm: the fused GroupNorm epilogue (sfast_epilogue_ext.gn_out) measured slower than reduce + "
                      "GroupNorm and lives in the probe build only (-DSFAST_PROBES); %d splits planned, nothing launched", p.splits);
#endif
    }
    char pipe[8];
    snprintf(pipe, sizeof(pipe), p.v.pipe == 5 ? (p.v.id >= 57 ? "ppl%d" : p.v.id >= 55 ? "ppw%d" : "pp%d") : p.v.pipe == 4 ? "pk%d" : p.v.pipe == 3 ? "patch%d" : p.v.pipe == 2 ? "ws%d" : p.v.pipe ? "dma%d" : "reg", p.v.ns);
    char xmap[24] = "";
    if (a.xmap == 2) snprintf(xmap, sizeof(xmap), "@xcdrun%c%d", a.x_order ? 'n' : 'm', a.x_per);  // contiguous runs, tile_m / tile_n fastest
    else if (a.xmap) snprintf(xmap, sizeof(xmap), "@xcd%dx%dx%d", 8 >> (a.x_lxm + a.x_lxn), 1 << a.x_lxm, 1 << a.x_lxn);  // K-split x row x column boxes
    set_kernel_name("igemm_%s_%s%s[%dx%d,split=%d,%s]%s%s%s%s", mode ? "conv" : "lin", dtype == SFAST_F16 ? "f16" : "bf16",
                    geglu ? "_geglu" : "", p.v.BM, p.v.BN, p.splits, pipe, a.gn_stats ? "+gnstats" : (a.stage_out ? "+staged" : ""),
                    joins ? "+join" : "", a.gn_out ? "+gn" : "", xmap);  // +join: split-K finished inside this kernel (no reduce launch)
    int rc;
    if (p.v.pipe == 5)
        rc = igemm_pp_launch(a, dtype, mode, geglu, p.v.BN, p.v.id >= 57 ? 104 : p.v.id >= 55 ? 4 : 0, st);  // ids 55..: four producer waves; 57..: lockstep
    else if (p.v.pipe == 4)
        rc = igemm_pk_launch(a, dtype, mode, p.v.BM, p.v.BN, st);
    else if (p.v.pipe == 3)
        rc = conv_patch_launch(a, dtype, p.v.BM, p.v.BN, st);
    else if (p.v.pipe == 2)
        rc = igemm_glds_ws_launch(a, dtype, mode, geglu, p.v.BM, p.v.BN, p.v.ns, st);
    else if (p.v.pipe == 1)
        rc = igemm_glds_launch(a, dtype, mode, geglu, p.v.BM, p.v.BN, p.v.ns, st);
    else if (dtype == SFAST_F16)
        rc = mode ? dispatch_variant<f16, 1>(a, p.v, geglu, st) : dispatch_variant<f16, 0>(a, p.v, geglu, st);
    else
        rc = mode ? dispatch_variant<bf16, 1>(a, p.v, geglu, st) : dispatch_variant<bf16, 0>(a, p.v, geglu, st);
    if (rc || joins) return rc;
#ifdef SFAST_PROBES
    if (a.gn_out) {  // the GroupNorm that consumes this output rides in the reduce launch (coverage checked BEFORE the main launch, above)
        const int items = ceil_div(a.rows_per_batch * (a.N / a.gn_groups / 4), 512);
        const dim3 grid((unsigned)a.gn_groups, (unsigned)(a.M / a.rows_per_batch)), block(512);
#define RG_LAUNCH(T, I) hipLaunchKernelGGL((splitk_reduce_gn_kernel<T, I>), grid, block, 0, st, a)
        if (dtype == SFAST_F16) {
            if (items <= 2) RG_LAUNCH(f16, 2); else if (items <= 4) RG_LAUNCH(f16, 4); else RG_LAUNCH(f16, 8);
        } else {
            if (items <= 2) RG_LAUNCH(bf16, 2); else if (items <= 4) RG_LAUNCH(bf16, 4); else RG_LAUNCH(bf16, 8);
        }
#undef RG_LAUNCH
        return check_launch("splitk_reduce_gn");
    }
#endif
    if (p.splits > 1 && a.gn_stats) {
        const int R = red_ty * red_rt, CPR = a.N / 8;
        const dim3 grid((unsigned)ceil_div(a.M, R)), block((unsigned)(((CPR * red_ty + 63) / 64) * 64));
        const size_t smem = (size_t)((a.N * 2 + 15) / 16) * 16 + (size_t)block.x * 16;
#define RR_LAUNCH(T, RT_) hipLaunchKernelGGL((splitk_reduce_rows_kernel<T, RT_>), grid, block, smem, st, a, red_ty)
        if (dtype == SFAST_F16) {
            if (red_rt == 4) RR_LAUNCH(f16, 4); else if (red_rt == 2) RR_LAUNCH(f16, 2); else RR_LAUNCH(f16, 1);
        } else {
            if (red_rt == 4) RR_LAUNCH(bf16, 4); else if (red_rt == 2) RR_LAUNCH(bf16, 2); else RR_LAUNCH(bf16, 1);
        }
#undef RR_LAUNCH
        return check_launch("splitk_reduce_rows");
    }
    if (p.splits > 1) {
        const int64_t total = (int64_t)a.M * (a.N / 4);
        const dim3 grid((unsigned)ceil_div64(total, 256));
        if (dtype == SFAST_F16) {
            if (geglu)
                hipLaunchKernelGGL((splitk_reduce_kernel<f16, true>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((splitk_reduce_kernel<f16, false>), grid, dim3(256), 0, st, a);
        } else {
            if (geglu)
                hipLaunchKernelGGL((splitk_reduce_kernel<bf16, true>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((splitk_reduce_kernel<bf16, false>), grid, dim3(256), 0, st, a);
        }
        return check_launch("splitk_reduce");
    }
    return SFAST_OK;
}

// split-K reduce + epilogue over slabs another kernel wrote (gnconv.hip): a.partial [splits][M][N] fp32, sums in order 0 .. splits-1
int igemm_reduce_only(const IgemmArgs &a, int dtype, hipStream_t st) {
    const int64_t total = (int64_t)a.M * (a.N / 4);
    const dim3 grid((unsigned)ceil_div64(total, 256));
    if (dtype == SFAST_F16)
        hipLaunchKernelGGL((splitk_reduce_kernel<f16, false>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16, false>), grid, dim3(256), 0, st, a);
    return check_launch("splitk_reduce");
}

}  // namespace sfast
