// Pipe 4 of the MFMA implicit GEMM (round 4): the weight operand comes from a PRE-PACKED copy of the parameter and goes global ->
// VGPR in MFMA A-fragment order; only the activation operand passes through the LDS ring.
//
// Why (measured: tools/micro/wdirect.hip, profiles/r04_wdirect_probe_run15.log). A wave's A fragment of v_mfma_f32_32x32x16 is 32
// rows x 16 k, 16 bytes per lane; read from a row-major [N][K] weight a buffer_load_dwordx4 touches 32 different 128-byte lines and
// the vector L1 looks up ONE line per clock: 32 clocks per 1 KB = 32 B/clk/CU, and the MFMA stream of the issuing wave waits
// behind it -- 44-47 % of the dense peak with everything L2-resident. The ring kernels (igemm_glds*.hip) avoid that path by staging
// the weights through LDS, and are bound by LDS bandwidth instead (fragment reads + LDS-DMA writes: 476 ns per 128x128x64 tile for
// 213 ns of MFMAs). With the weight stored as contiguous 1 KB fragments
// [row block][k-step][lane][8 elements] the same load touches 8 full lines: 78 % of the dense peak in the probe (4 x 32 pixels x
// 2 x 32 weight rows per wave, activations read from LDS, weights L2-resident), 81 % with 4 x 4 fragments per wave.
//
// The packed copy is made by sfast_hip_pack_weight (below) from the live parameter; the caller owns its freshness (the engine
// re-packs a parameter when its version counter moved, sfast/engine/unet2d.py). Everything else -- implicit im2col of the
// activations, XCD-aware block map, split-K slabs, epilogue incl. staged stores and GroupNorm statistics -- is igemm_device.h's.
//
// Replaces, like the other pipes: sfast::cudnn_convolution_bias[_add] (/root/reference/src/sfast/csrc/operators/cudnn/
// cudnn_convolution_impl.cc:995-998) and sfast::cublas_lowp_linear[_add] (csrc/operators/cublas/cublas_gemm.cpp:798-948); cuDNN /
// cuBLASLt pre-transform filters too (their "filter transform" / weight re-layout kernels run inside the vendor call).
#pragma once
#include <type_traits>

#include "igemm_device.h"

namespace sfast {

typedef const u32x4 __attribute__((address_space(1))) * pk_src_t;
typedef __attribute__((address_space(3))) void *pk_dst_t;

template <int N> __device__ __forceinline__ void pk_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int pk_smem_bytes(int BM, int BN, int NS, bool staged) {
    const int ring = NS * BM * 128, stage = BM * (BN * 2 + 8) + 16 + 320 * 16;  // staged tile + a float4 per flush thread
    return (staged && stage > ring) ? stage : ring;
}

// BM pixels x BN weight rows per workgroup; WN consumer waves, each owns FN = BN / (32 WN) blocks of 32 weight rows over ALL FM = BM / 32
// pixel blocks (no two waves load the same weight bytes), plus ONE producer wave that streams the activation K-tiles into the LDS
// ring (the producer half of igemm_glds_ws.hip). NS = ring depth (K-tiles of 64), PD = weight tiles a consumer holds in registers.
// The first version let the consumer waves issue the LDS-DMA requests themselves: the compiler orders a wave's ds_reads behind its own
// pending LDS-DMA writes with s_waitcnt vmcnt(0), which also waits for the weight tile requested a moment earlier -- a full L2 round
// trip per K-tile, 19 - 33 % MFMA utilisation in the loop (profiles/r04_pk_ab_trace_run17.log). A wave that never issues LDS-DMA keeps
// exact vmcnt counts for its weight loads.
template <typename T, int BM, int BN, int WN, int NS, int PD, int MODE, bool STAGED>
__global__ void __launch_bounds__((WN + 1) * 64, 2) igemm_pk_kernel(const IgemmArgs a) {
    using vec8 = typename Elem<T>::vec8;
    constexpr int NC = WN * 64;
    constexpr int FM = BM / 32, FN = BN / (WN * 32);
    constexpr int XCH = BM / 8;             // LDS-DMA requests of the producer wave per K-tile (8 rows of 128 bytes each)
    constexpr int STAGE = BM * 128;
    constexpr int WNB = FN * 32;
    static_assert(XCH * (NS - 1) <= 63, "vmcnt field");
    static_assert(NS >= 3 && PD >= 2 && PD <= 3, "pipeline shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    touch_args(a);
    if (MODE == 1) touch_conv_args(a);
    trace_mark(a, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    const BlockTile bt = decode_block(a);
    if (bt.tile_m < 0) return;  // surplus block of an xmap = 2 grid (wave-uniform, before any barrier)
    const int m0 = bt.tile_m * BM, n0 = bt.tile_n * BN;
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);

    if (wave == WN) {
        // =============================== producer wave: activations -> LDS ring ========================================
        const int rbase = lane >> 3;
        const pk_src_t zero_src = (pk_src_t)(const void *)g_zero16;
        const T *xrow[XCH];
        int xoffB[XCH], xdAB[XCH];
        unsigned xmask[XCH];
        const PixelDecoder decode(a);
        unsigned rep_all = 0;
        if (MODE == 1)
            for (int r = 0; r < a.KH; ++r) rep_all |= 1u << (r * a.KW);
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int row = rbase + i * 8;
            const int kc = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (LDS-DMA writes lane-linearly)
            const int m = m0 + row;
            if (MODE == 0) {
                xrow[i] = (m < a.M) ? (const T *)a.x + (int64_t)m * a.ldx + kc * 8 : nullptr;
                xoffB[i] = kc * 8;
            } else {
                unsigned mask = 0;
                int pix = 0;
                if (m < a.M) {
                    int b, ho, wo;
                    decode(m, b, ho, wo);
                    const int h0 = ho * a.stride_h - a.pad_h, w0 = wo * a.stride_w - a.pad_w;
                    pix = (b * a.H + h0) * a.W + w0;
                    if (a.dil_h == 1 && a.dil_w == 1) {
                        const int s_lo = max(0, -w0), s_hi = min(a.KW, a.W - w0);
                        const int r_lo = max(0, -h0), r_hi = min(a.KH, a.H - h0);
                        if (s_hi > s_lo && r_hi > r_lo) {
                            const unsigned cols = ((1u << s_hi) - 1u) & ~((1u << s_lo) - 1u);
                            const unsigned lo_bits = r_lo * a.KW, hi_bits = r_hi * a.KW;
                            const unsigned upto = hi_bits >= 32 ? 0xffffffffu : ((1u << hi_bits) - 1u);
                            mask = cols * (rep_all & upto & ~((1u << lo_bits) - 1u));
                        }
                    } else {
                        unsigned cols = 0;
                        for (int s = 0; s < a.KW; ++s) cols |= ((unsigned)(w0 + s * a.dil_w) < (unsigned)a.W ? 1u : 0u) << s;
                        for (int r = 0; r < a.KH; ++r) mask |= ((unsigned)(h0 + r * a.dil_h) < (unsigned)a.H ? cols : 0u) << (r * a.KW);
                    }
                }
                xoffB[i] = pix * a.C2 + kc * 8;
                xdAB[i] = pix * (a.C1 - a.C2);
                xmask[i] = mask;
            }
        }
        const int cin = a.C1 + a.C2;
        int t_tap = 0, t_r = 0, t_s = 0, t_c = 0;
        if (MODE == 1) {
            const int k0 = kt_begin * 64;
            t_tap = k0 / cin;
            t_c = k0 - t_tap * cin;
            t_r = t_tap / a.KW;
            t_s = t_tap - t_r * a.KW;
        }
        int issued = kt_begin, istage = 0;
        auto issue_x = [&]() __attribute__((always_inline)) {  // the XCH requests of the next K-tile; tiles past kt_end are all-zero requests
            char *sx = smem + istage * STAGE;
            const bool tile_ok = issued < kt_end;
            const int k = issued * 64;
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const bool ok = tile_ok & (xrow[i] != nullptr) & (k + xoffB[i] < a.K);
                    const pk_src_t src = ok ? (pk_src_t)(const void *)(xrow[i] + k) : zero_src;
                    __builtin_amdgcn_global_load_lds(src, (pk_dst_t)(sx + i * 1024), 16, 0, 0);
                }
            } else {
                const bool first = t_c < a.C1;
                const T *sbase = first ? (const T *)a.x + ((t_r * a.dil_h * a.W + t_s * a.dil_w) * a.C1 + t_c)
                                       : (const T *)a.x2 + ((t_r * a.dil_h * a.W + t_s * a.dil_w) * a.C2 + (t_c - a.C1));
                const int fmask = first ? -1 : 0;
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int off = xoffB[i] + (xdAB[i] & fmask);
                    const bool ok = tile_ok & (((xmask[i] >> (t_tap & 31)) & 1u) != 0);
                    const pk_src_t src = ok ? (pk_src_t)(const void *)(sbase + off) : zero_src;
                    __builtin_amdgcn_global_load_lds(src, (pk_dst_t)(sx + i * 1024), 16, 0, 0);
                }
                t_c += 64;
                if (t_c >= cin) {
                    t_c -= cin;
                    ++t_tap;
                    if (++t_s == a.KW) {
                        t_s = 0;
                        ++t_r;
                    }
                }
            }
            ++issued;
            istage = (istage + 1 == NS) ? 0 : istage + 1;
        };
        // ring protocol of igemm_glds_ws.hip: NS tiles at the start; the consumers meet barrier k + 1 with every fragment of tile k in
        // registers, so that barrier releases the stage of tile k and it is refilled with tile k + NS
#pragma unroll
        for (int s = 0; s < NS; ++s) issue_x();
        pk_wait_vmcnt<XCH *(NS - 1)>();  // the first tile has landed
        __builtin_amdgcn_s_barrier();
        for (int kt = kt_begin + 1; kt < kt_end; ++kt) {
            pk_wait_vmcnt<XCH *(NS - 2)>();  // tile kt has landed
            __builtin_amdgcn_s_barrier();
            issue_x();
        }
        pk_wait_vmcnt<0>();  // the zero-filled tail requests have landed before the LDS is handed to the epilogue
        return;
    }

    // =================================== consumer waves ====================================================================
    // weight fragments: this wave's FN row blocks, 4 KB per (block, K-tile), contiguous along K
    const char *ap[FN];
    const int rs = a.rows_per_seg;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        int n = n0 + wave * WNB + fn * 32;
        if (n >= a.N) n = 0;  // overhanging block: any valid bytes, the epilogue drops the rows
        const int seg = (n >= rs) + (n - rs >= rs) + (n - rs - rs >= rs);
        const void *base = seg == 0 ? a.wpk[0] : seg == 1 ? a.wpk[1] : seg == 2 ? a.wpk[2] : a.wpk[3];
        const int nbl = (n - seg * rs) >> 5;
        ap[fn] = (const char *)base + ((int64_t)nbl * a.pk_ksteps + (int64_t)kt_begin * 4) * 1024 + lane * 16;
    }
    int a_left = kt_end - kt_begin;  // K-tiles of this split whose weights are not requested yet
    u32x4 wq[PD][FN][4];
    auto load_a = [&](u32x4 (&dst)[FN][4]) __attribute__((always_inline)) {
        // past the split's range the last tile is requested again (valid bytes; never multiplied)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dst[fn][j] = *reinterpret_cast<const u32x4 *>(ap[fn] + j * 1024);
                __builtin_amdgcn_sched_barrier(0);  // issue order = consumption order (exact vmcnt waits)
            }
        if (a_left > 1) {
            --a_left;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) ap[fn] += 4096;
        }
    };
#pragma unroll
    for (int p = 0; p < PD - 1; ++p) load_a(wq[p]);

    constexpr bool EPI_EARLY = FN * FM <= 4;
    EpiOperands<(EPI_EARLY ? FN : 1), (EPI_EARLY ? FM : 1)> epi;
    if constexpr (EPI_EARLY) epilogue_prefetch<T, FN, FM, false>(a, epi, m0, n0 + wave * WNB, l31, hi);
    f32x16 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;
    trace_mark(a, 1);
    trace_mark(a, 2);
    __builtin_amdgcn_s_barrier();  // tile kt_begin has landed
    trace_mark(a, 3);

    int cstage = 0;
    vec8 bf[2][FM];
    auto read_b = [&](const char *xs, int ks, int set) __attribute__((always_inline)) {
        const int chunk = ks * 2 + hi;
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) bf[set][fm] = *reinterpret_cast<const vec8 *>(xs + lds_off(fm * 32 + l31, chunk));
    };
    // one K-tile with its weights in wq[SLOT]; returns with the first activation fragments of the next tile requested
    auto tile = [&](auto slot_tag, int kt) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const char *xs = smem + cstage * STAGE;
        cstage = (cstage + 1 == NS) ? 0 : cstage + 1;
        load_a(wq[(SLOT + PD - 1) % PD]);  // weights of tile kt + PD - 1 into the registers tile kt - 1 has finished with
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
                read_b(xs, ks + 1, (ks + 1) & 1);
            } else if (kt + 1 < kt_end) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every activation fragment of tile kt is in registers
                __builtin_amdgcn_s_barrier();                       // tile kt + 1 has landed, the stage of tile kt is released
                asm volatile("" ::: "memory");
                read_b(smem + cstage * STAGE, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    acc[fn][fm] = mfma32(__builtin_bit_cast(vec8, wq[SLOT][fn][ks]), bf[ks & 1][fm], acc[fn][fm]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    read_b(smem, 0, 0);
    for (int kt = kt_begin;;) {
        tile(std::integral_constant<int, 0>{}, kt);
        if (++kt >= kt_end) break;
        tile(std::integral_constant<int, 1 % PD>{}, kt);
        if (++kt >= kt_end) break;
        if constexpr (PD == 3) {
            tile(std::integral_constant<int, 2>{}, kt);
            if (++kt >= kt_end) break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus weight requests have landed before their registers die
    trace_mark(a, 4);
    run_epilogue<T, BM, BN, FN, FM, false, EPI_EARLY, NC, STAGED>(a, acc, epi, smem, m0, n0, m0, n0 + wave * WNB, l31, hi, tid, bt.split);
    trace_finish(a);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// variant ids 41.. (igemm.hip kVariants, pipe 4): BM pixels x BN weight rows, WN waves
#define SFAST_FOR_PK_VARIANTS(T, MODE, OP) \
    OP(T, 128, 256, 4, MODE)               \
    OP(T, 64, 256, 4, MODE)                \
    OP(T, 64, 320, 5, MODE)                \
    OP(T, 128, 160, 5, MODE)               \
    OP(T, 64, 160, 5, MODE)                \
    OP(T, 128, 128, 4, MODE)

constexpr int PK_NS = 4;
// weight tiles held in registers: 3 where one tile is 4 loads per lane (FN = 1), 2 where it is 8 (FN = 2: the third set does not fit
// under 256 registers beside a 64 x 64 .. 128 x 64 accumulator without spilling)
constexpr int pk_pd(int BM, int BN, int WN) { return BN / (WN * 32) >= 2 ? 2 : 3; }

template <typename T, int BM, int BN, int WN, int MODE, bool STAGED> static int pk_set_attr() {
    auto kern = igemm_pk_kernel<T, BM, BN, WN, PK_NS, pk_pd(BM, BN, WN), MODE, STAGED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pk_smem_bytes(BM, BN, PK_NS, STAGED));
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(igemm_pk %dx%d): %s", BM, BN, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

template <typename T, int MODE> static int pk_init_tm() {
    int rc = 0;
#define INIT_OP(TT, BM, BN, WN, MODE_)                             \
    if (!rc) rc = pk_set_attr<TT, BM, BN, WN, MODE_, false>();     \
    if (!rc) rc = pk_set_attr<TT, BM, BN, WN, MODE_, true>();
    SFAST_FOR_PK_VARIANTS(T, MODE, INIT_OP)
#undef INIT_OP
    return rc;
}

extern unsigned long long *g_igemm_trace;  // igemm_glds.hip

template <typename T, int MODE> static int pk_dispatch(const IgemmArgs &a, int BM_, int BN_, hipStream_t st) {
#define LAUNCH_OP(TT, BM, BN, WN, MODE_)                                                                                            \
    if (BM_ == BM && BN_ == BN) {                                                                                                   \
        if (a.stage_out) {                                                                                                          \
            hipLaunchKernelGGL((igemm_pk_kernel<TT, BM, BN, WN, PK_NS, pk_pd(BM, BN, WN), MODE_, true>), igemm_grid(a), dim3((WN + 1) * 64),          \
                               pk_smem_bytes(BM, BN, PK_NS, true), st, a);                                                          \
            return check_launch("igemm_pk_staged");                                                                                 \
        }                                                                                                                           \
        hipLaunchKernelGGL((igemm_pk_kernel<TT, BM, BN, WN, PK_NS, pk_pd(BM, BN, WN), MODE_, false>), igemm_grid(a), dim3((WN + 1) * 64),             \
                           pk_smem_bytes(BM, BN, PK_NS, false), st, a);                                                             \
        return check_launch("igemm_pk");                                                                                            \
    }
    SFAST_FOR_PK_VARIANTS(T, MODE, LAUNCH_OP)
#undef LAUNCH_OP
    set_error("igemm_pk: no kernel for tile %dx%d", BM_, BN_);
    return SFAST_ERR_UNSUPPORTED;
}

// One translation unit per (dtype, mode): igemm_pk_f16_lin.hip ... (48 instantiations in one file compile for ten minutes)
#define SFAST_PK_UNIT(T, MODE, TAG)                                                                                          \
    namespace sfast {                                                                                                        \
    int igemm_pk_init_##TAG() { return pk_init_tm<T, MODE>(); }                                                              \
    int igemm_pk_launch_##TAG(const IgemmArgs &a, int BM, int BN, hipStream_t st) { return pk_dispatch<T, MODE>(a, BM, BN, st); } \
    }

}  // namespace sfast
