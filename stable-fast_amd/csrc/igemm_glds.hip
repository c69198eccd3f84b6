// Deep-pipelined MFMA implicit-GEMM: direct global->LDS staging (LDS-DMA) over an NS-stage ring.
//
// Why a second main-loop structure: PMC on MI355X showed the register-staged kernel (igemm.hip) at
// ~20 % MFMA busy with 0 LDS bank conflicts -- each K-tile costs one full memory round trip because
// only ONE tile can be in flight (its data sits in VGPRs until the ds_write), and the short-K GEMMs of
// the transformer blocks (K = 320..1280, 5..20 K-tiles) never leave the latency-bound regime.
// Here the tile bytes travel HBM/L2 -> LDS without touching registers
// (`global_load_lds_dwordx4`, 1 KiB per wave-instruction), so NS-1 K-tiles are in flight at any time
// at zero VGPR cost, the ds_write pass disappears, and for K <= (NS-1)*64 the whole operand is
// requested up front: one memory latency per workgroup instead of one per K-tile.
//
//   * LDS image identical to igemm.hip (128-B rows, chunk ^= (row>>1)&7) so the fragment reads stay
//     conflict-free. LDS-DMA writes lane-linearly (wave-uniform base + lane*16), therefore the
//     swizzle is applied on the SOURCE side: the lane that fills physical chunk p of row r fetches
//     logical chunk p ^ ((r>>1)&7).
//   * zero fill (image border taps, rows >= M, k >= K) = the lane's source address is redirected to
//     a 16-byte zero block in device memory; the DMA stays unconditional.
//   * ordering is hand-placed: counted `s_waitcnt vmcnt(L*tiles_in_flight)` (never 0 in steady
//     state), ONE raw `s_barrier` per K-tile, refill of the stage freed by the previous iteration
//     right after the barrier.
//   * conv addressing is hoisted: per tile row a pixel index and a 9-bit (KH*KW <= 32) tap-validity
//     mask are computed once; per K-tile the tap / channel decode is wave-uniform scalar state
//     advanced incrementally (needs Cin % 64 == 0 and C1 % 64 == 0: true for every UNet layer but
//     conv_in). Problems outside that envelope (upsample-fused convs, odd channel counts) keep using
//     the register-staged kernel.
#include "igemm_device.h"
#include <stdlib.h>
#include <type_traits>

namespace sfast {

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef const u32x4 __attribute__((address_space(1))) * glds_src_t;
typedef __attribute__((address_space(3))) void *glds_dst_t;

// EXP != 0: profiling experiments (tools/trace_igemm.py, only instantiated for two tiles): bit0 no MFMAs, bit1 no
// fragment reads, bit2 no in-loop LDS-DMA requests, bit3 no per-tile barrier. Results are garbage.
template <typename T, int BM, int BN, int WM, int WN, int NS, int MODE, bool GEGLU, int EXP = 0, bool STAGED = false>
__global__ void __launch_bounds__(WM *WN * 64, igemm_min_waves(WM *WN * 64, NS *(BM + BN) * 128)) igemm_glds_kernel(const IgemmArgs a) {
    using vec8 = typename Elem<T>::vec8;
    constexpr int NT = WM * WN * 64;
    constexpr int FM = BM / (WM * 32);
    constexpr int FN = BN / (WN * 32);
    constexpr int XCH = BM * 8 / NT;
    constexpr int WCH = BN * 8 / NT;
    constexpr int RPP = NT / 8;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    constexpr int WNB = FN * 32;
    constexpr int L = XCH + WCH;  // LDS-DMA instructions per thread per K-tile
    static_assert(RPP % 16 == 0, "swizzle phase must not depend on the staging pass");
    static_assert(L * (NS > 2 ? NS - 2 : 0) <= 63, "vmcnt field");
    static_assert(NS >= 2 && NS <= 5, "ring depth");
    static_assert(!GEGLU || (FN % 2 == 0), "GEGLU needs paired fragments");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    touch_args(a);
    if (MODE == 1) touch_conv_args(a);
    trace_mark(a, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, hi = lane >> 5;

    const BlockTile bt = decode_block(a);  // XCD-aware (tile, K-split) of this workgroup
    if (bt.tile_m < 0) return;  // surplus block of an xmap = 2 grid (wave-uniform, before any barrier)
    const int tile_n = bt.tile_n, tile_m = bt.tile_m;
    const int m0 = tile_m * BM, n0 = tile_n * BNO;
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);

    // staging role of this thread: tile row (tid>>3) + i*RPP, PHYSICAL chunk tid&7, which must hold
    // LOGICAL chunk kc (source-side swizzle; RPP % 16 == 0 keeps it independent of i)
    const int rbase = tid >> 3;
    const int kc = (tid & 7) ^ ((rbase >> 1) & 7);
    const glds_src_t zero_src = (glds_src_t)(const void *)g_zero16;

    // ---- per-row staging metadata, computed once ------------------------------------------------------------
    // Everything lane-dependent about a load's address is folded into a per-row value here; per K-tile the
    // address is that value plus a wave-uniform (scalar) term. The in-loop cost of one LDS-DMA request is then
    // a validity test, one 64-bit add and the zero-block select -- no integer multiply, no division.
    const T *xrow[XCH];   // MODE 0: row pointer at column kc*8, or nullptr
    int xoffB[XCH];       // MODE 1: element offset of (tap (0,0), channel kc*8) in source 2 (pitch C2, virtual concat)
    int xdAB[XCH];        //         (same offset in source 1, pitch C1) - xoffB: blended in with a uniform mask
    unsigned xmask[XCH];  // MODE 1: bit (r*KW+s) set when that tap is inside the image
    const PixelDecoder decode(a);
    unsigned rep_all = 0;  // bit r*KW set for every tap row r (uniform)
    if (MODE == 1)
        for (int r = 0; r < a.KH; ++r) rep_all |= 1u << (r * a.KW);
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int m = m0 + rbase + i * RPP;
        if (MODE == 0) {
            xrow[i] = (m < a.M) ? (const T *)a.x + (int64_t)m * a.ldx + kc * 8 : nullptr;
        } else {
            unsigned mask = 0;
            int pix = 0;
            if (m < a.M) {
                int b, ho, wo;
                decode(m, b, ho, wo);
                const int h0 = ho * a.stride_h - a.pad_h, w0 = wo * a.stride_w - a.pad_w;
                pix = (b * a.H + h0) * a.W + w0;
                if (a.dil_h == 1 && a.dil_w == 1) {
                    // closed form: taps s in [s_lo, s_hi) x r in [r_lo, r_hi) are inside the image
                    const int s_lo = max(0, -w0), s_hi = min(a.KW, a.W - w0);
                    const int r_lo = max(0, -h0), r_hi = min(a.KH, a.H - h0);
                    if (s_hi > s_lo && r_hi > r_lo) {
                        const unsigned cols = ((1u << s_hi) - 1u) & ~((1u << s_lo) - 1u);
                        const unsigned lo_bits = r_lo * a.KW, hi_bits = r_hi * a.KW;  // <= 32
                        const unsigned upto = hi_bits >= 32 ? 0xffffffffu : ((1u << hi_bits) - 1u);
                        mask = cols * (rep_all & upto & ~((1u << lo_bits) - 1u));  // no carries: cols < 2^KW
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
    const T *wrow[WCH];  // weight row pointer at column kc*8, or nullptr
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int j = rbase + i * RPP;
        if (GEGLU) {
            const int grp = j / WNB, within = j % WNB;
            const int half = within / (WNB / 2), i2 = within % (WNB / 2);
            const int ncol = n0 + grp * (WNB / 2) + i2;
            wrow[i] = (ncol < a.N) ? (const T *)(half ? a.w[1] : a.w[0]) + (int64_t)ncol * a.ldw + kc * 8 : nullptr;
        } else {
            const int n = n0 + j;
            if (n < a.N) {
                const int rs = a.rows_per_seg;
                const int seg = (n >= rs) + (n - rs >= rs) + (n - rs - rs >= rs);
                const void *base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : seg == 2 ? a.w[2] : a.w[3];
                wrow[i] = (const T *)base + (int64_t)(n - seg * rs) * a.ldw + kc * 8;
            } else {
                wrow[i] = nullptr;
            }
        }
    }

    // ---- wave-uniform state of the NEXT tile to be issued ------------------------------------------------------
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

    // One LDS-DMA request of the next tile: slice l < XCH is activation row-pass l, the others weight row-passes.
    // Requests are unconditional; an invalid chunk (image border, row >= M, k >= K) is redirected to the device
    // zero block. `l` is a compile-time constant at every call site.
    auto issue_slice = [&](int l) {
        char *sx = smem + istage * STAGE + wave * 1024;
        char *sw = sx + BM * 128;
        const bool tile_ok = issued < kt_end;  // uniform; tiles past the end are all-zero requests (see the ring below)
        if (l < XCH) {
            glds_src_t src;
            if (MODE == 0) {
                const int k = issued * 64;  // uniform
                const bool ok = tile_ok & (xrow[l] != nullptr) & (k + kc * 8 < a.K);
                src = ok ? (glds_src_t)(const void *)(xrow[l] + k) : zero_src;
            } else {
                const bool first = t_c < a.C1;  // uniform: which concat source this K-tile reads
                const T *sbase = first ? (const T *)a.x + ((t_r * a.dil_h * a.W + t_s * a.dil_w) * a.C1 + t_c)
                                       : (const T *)a.x2 + ((t_r * a.dil_h * a.W + t_s * a.dil_w) * a.C2 + (t_c - a.C1));
                const int off = xoffB[l] + (xdAB[l] & (first ? -1 : 0));  // arithmetic blend: a select of two arrays is lowered through scratch
                const bool ok = tile_ok & (((xmask[l] >> (t_tap & 31)) & 1u) != 0);
                src = ok ? (glds_src_t)(const void *)(sbase + off) : zero_src;
            }
            __builtin_amdgcn_global_load_lds(src, (glds_dst_t)(sx + l * (RPP * 128)), 16, 0, 0);
        } else {
            const int i = l - XCH;
            const int k = issued * 64;  // uniform
            const bool ok = tile_ok & (wrow[i] != nullptr) & (MODE == 1 || k + kc * 8 < a.K);
            const glds_src_t src = ok ? (glds_src_t)(const void *)(wrow[i] + k) : zero_src;
            __builtin_amdgcn_global_load_lds(src, (glds_dst_t)(sw + i * (RPP * 128)), 16, 0, 0);
        }
    };
    // after the last slice of a tile: advance the uniform state by one K-tile (Cin % 64 == 0: at most one wrap)
    auto issue_advance = [&]() {
        if (MODE == 1) {
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

    // One K-tile of MFMAs. Fragment reads run one 16-wide K step AHEAD of the MFMAs that consume them (two register
    // sets). With REFILL the L requests of the tile that re-fills the stage freed by the previous iteration are
    // interleaved one per MFMA: the address arithmetic of a request (~10 VALU/SALU ops) then issues while the matrix
    // pipe is busy with the MFMA before it. Issued as one block ahead of the MFMAs it cost about as long as the MFMAs
    // themselves (0.9 us per 128x160x64 tile against 0.27 us of MFMA time, profiles/r01_igemm_phase_trace.log).
    auto compute = [&](int stage) {
        const char *xs = smem + stage * STAGE;
        const char *ws = xs + BM * 128;
        vec8 af[2][FN], bf[2][FM];
        if constexpr ((EXP & 2) != 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) af[q][fn] = vec8{};
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) bf[q][fm] = vec8{};
            }
        }
        auto read_frags = [&](int ks, int set) {
            if constexpr ((EXP & 2) != 0) return;
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
                for (int fm = 0; fm < FM; ++fm) {
                    if constexpr ((EXP & 1) == 0) acc[fn][fm] = mfma32(af[ks & 1][fn], bf[ks & 1][fm], acc[fn][fm]);
                    const int j = (ks * FN + fn) * FM + fm;  // compile-time after unrolling
                    if (j < L && (EXP & 4) == 0) {
                        issue_slice(j);
                        __builtin_amdgcn_sched_barrier(0);  // pin: MFMA j, request j, MFMA j+1, ...
                    }
                }
        }
        static_assert(L <= 4 * FN * FM, "one request per MFMA");
        issue_advance();
    };

    // ---- NS-stage ring: tiles kt+1 .. kt+NS-2 stay in flight while tile kt is multiplied ------------------
    // Every iteration issues exactly one tile (L requests per thread), tiles past kt_end as all-zero requests into
    // stages nobody reads again: the outstanding-request count is then the same in every iteration and the wait is
    // ONE constant `s_waitcnt vmcnt(L*(NS-2))` -- tile kt has landed, tiles kt+1 .. kt+NS-2 may still be in flight.
    trace_mark(a, 1);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
        for (int l = 0; l < L; ++l) issue_slice(l);
        issue_advance();
    }
    trace_mark(a, 2);
    int cstage = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        wait_vmcnt<((EXP & 4) ? 0 : L *(NS - 2))>();
        if constexpr ((EXP & 8) == 0) __builtin_amdgcn_s_barrier();
        if (kt == kt_begin) trace_mark(a, 3);
        compute(cstage);
        cstage = (cstage + 1 == NS) ? 0 : cstage + 1;
    }
    wait_vmcnt<0>();  // the zero-filled tail requests must have landed before this workgroup's LDS is released

    // ---- epilogue: batched operand loads, fp32 math, 8-byte stores (igemm_device.h) -------------------------
    trace_mark(a, 4);
    run_epilogue<T, BM, BNO, FN, FM, GEGLU, EPI_EARLY, NT, STAGED, false>(a, acc, epi, smem, m0, n0, m0 + wm * (FM * 32), n0 + wn * (GEGLU ? WNB / 2 : WNB), l31, hi,
                                                             tid, bt.split);
    trace_finish(a);
}

// ---- host side ------------------------------------------------------------------------------------------
// glds variants are addressed as variant ids 11..15 (same tile shapes as ids 1..5 of igemm.hip)
#define SFAST_FOR_GLDS_VARIANTS(T, MODE, OP) \
    OP(T, 128, 128, 2, 2, 4, MODE, false)    \
    OP(T, 128, 160, 4, 1, 4, MODE, false)    \
    OP(T, 64, 64, 2, 2, 5, MODE, false)      \
    OP(T, 64, 160, 2, 1, 4, MODE, false)     \
    OP(T, 256, 128, 4, 2, 3, MODE, false)    \
    OP(T, 128, 128, 2, 2, 2, MODE, false)    \
    OP(T, 128, 160, 4, 1, 2, MODE, false)    \
    OP(T, 64, 64, 2, 2, 3, MODE, false)

#define SFAST_FOR_GLDS_GEGLU_VARIANTS(T, OP) \
    OP(T, 128, 128, 2, 2, 4, 0, true)        \
    OP(T, 64, 128, 2, 2, 5, 0, true)         \
    OP(T, 128, 128, 2, 2, 2, 0, true)        \
    OP(T, 64, 128, 2, 2, 3, 0, true)

template <typename T, int BM, int BN, int WM, int WN, int NS, int MODE, bool GEGLU>
static int glds_set_attr() {
    constexpr int smem = NS * (BM + BN) * 128;
    auto kern = igemm_glds_kernel<T, BM, BN, WM, WN, NS, MODE, GEGLU>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(igemm_glds %dx%dx%d): %s", BM, BN, NS, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

int igemm_glds_init() {
    int rc = 0;
#define INIT_OP(T, BM, BN, WM, WN, NS, MODE, G) \
    if (!rc) rc = glds_set_attr<T, BM, BN, WM, WN, NS, MODE, G>();
    SFAST_FOR_GLDS_VARIANTS(f16, 0, INIT_OP)
    SFAST_FOR_GLDS_VARIANTS(f16, 1, INIT_OP)
    SFAST_FOR_GLDS_VARIANTS(bf16, 0, INIT_OP)
    SFAST_FOR_GLDS_VARIANTS(bf16, 1, INIT_OP)
    SFAST_FOR_GLDS_GEGLU_VARIANTS(f16, INIT_OP)
    SFAST_FOR_GLDS_GEGLU_VARIANTS(bf16, INIT_OP)
#undef INIT_OP
    return rc;
}

unsigned long long *g_igemm_trace = nullptr;  // set through sfast_hip_set_trace (profiling only)
int g_igemm_exp = 0;  // SFAST_IGEMM_EXP, latched by sfast_hip_set_trace: selects an experiment instantiation

template <typename T, int MODE>
static int glds_dispatch(const IgemmArgs &a, int BM_, int BN_, int NS_, bool geglu, hipStream_t st) {
#define LAUNCH_OP(TT, BM, BN, WM, WN, NS, MODE_, G_)                                                         \
    if (BM_ == BM && BN_ == BN && NS_ == NS && geglu == G_) {                                                             \
        auto kern = igemm_glds_kernel<TT, BM, BN, WM, WN, NS, MODE_, G_>;                                    \
        hipLaunchKernelGGL(kern, igemm_grid(a), dim3(WM *WN * 64), NS *(BM + BN) * 128, st, a); \
        return check_launch("igemm_glds");                                                                   \
    }
#ifdef SFAST_PROBES  // timing-only instantiations (results are garbage): probe build only (build.py --probes)
    if constexpr (std::is_same<T, f16>::value) {
        if (g_igemm_exp != 0 && !geglu) {  // profiling experiments: two representative tiles only
#define LAUNCH_EXP(BM, BN, WM, WN, NS, MODE_, E)                                                                        \
    if (BM_ == BM && BN_ == BN && NS_ == NS && MODE == MODE_ && g_igemm_exp == E) {                                     \
        auto kern = igemm_glds_kernel<f16, BM, BN, WM, WN, NS, MODE_, false, E>;                                        \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NS *(BM + BN) * 128); \
        hipLaunchKernelGGL(kern, igemm_grid(a), dim3(WM *WN * 64), NS *(BM + BN) * 128, st, a);  \
        return check_launch("igemm_glds_exp");                                                                          \
    }
            LAUNCH_EXP(128, 160, 4, 1, 4, 1, 1) LAUNCH_EXP(128, 160, 4, 1, 4, 1, 2) LAUNCH_EXP(128, 160, 4, 1, 4, 1, 3)
            LAUNCH_EXP(128, 160, 4, 1, 4, 1, 4) LAUNCH_EXP(128, 160, 4, 1, 4, 1, 8) LAUNCH_EXP(128, 160, 4, 1, 4, 1, 7)
            LAUNCH_EXP(64, 64, 2, 2, 3, 0, 1) LAUNCH_EXP(64, 64, 2, 2, 3, 0, 2) LAUNCH_EXP(64, 64, 2, 2, 3, 0, 3)
            LAUNCH_EXP(64, 64, 2, 2, 3, 0, 4) LAUNCH_EXP(64, 64, 2, 2, 3, 0, 8) LAUNCH_EXP(64, 64, 2, 2, 3, 0, 7)
#undef LAUNCH_EXP
        }
    }
#endif
    if (!geglu) {
        SFAST_FOR_GLDS_VARIANTS(T, MODE, LAUNCH_OP)
    } else {
        if (MODE == 0) {
            SFAST_FOR_GLDS_GEGLU_VARIANTS(T, LAUNCH_OP)
        }
    }
#undef LAUNCH_OP
    set_error("igemm_glds: no kernel for tile %dx%d ring %d", BM_, BN_, NS_);
    return SFAST_ERR_UNSUPPORTED;
}

int igemm_glds_launch(const IgemmArgs &a_in, int dtype, int mode, bool geglu, int BM, int BN, int NS, hipStream_t st) {
    IgemmArgs a = a_in;
    a.trace = g_igemm_trace;
    if (dtype == SFAST_F16) return mode ? glds_dispatch<f16, 1>(a, BM, BN, NS, geglu, st) : glds_dispatch<f16, 0>(a, BM, BN, NS, geglu, st);
    return mode ? glds_dispatch<bf16, 1>(a, BM, BN, NS, geglu, st) : glds_dispatch<bf16, 0>(a, BM, BN, NS, geglu, st);
}

}  // namespace sfast
