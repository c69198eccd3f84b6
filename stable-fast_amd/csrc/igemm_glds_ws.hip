// Wave-specialised LDS-DMA implicit GEMM: PW producer waves stream K-tiles into the LDS ring, WM*WN consumer
// waves only read fragments and issue MFMAs.
//
// Why a third main-loop structure. The experiment instantiations of igemm_glds.hip (EXP, tools/trace_igemm.py,
// profiles/r01_igemm_kloop_experiments.log) split one 128x160x64 K-tile at one workgroup per CU into
//     LDS-DMA requests alone 0.50 us   |   MFMA + fragment reads alone 0.52 us   |   both 0.94 us
// i.e. fetch and compute did NOT overlap although three tiles were in flight: a wave that issues
// `global_load_lds` faster than the texture-address path of its CU drains them (~28 cycles per 1 KiB request,
// four waves sharing it) stalls AT ISSUE, in order, and the MFMAs behind the request in its instruction stream
// wait with it. Interleaving one request per MFMA did not help for the same reason. Here the waves that stall
// on the memory pipe are not the waves that feed the matrix pipe:
//   producers   prologue: NS tiles; barrier k (k >= 1): s_waitcnt vmcnt(L*(NS-2)) (tile k landed) -> s_barrier ->
//               L requests of tile k-1+NS into the stage of tile k-1, which the consumers released at that barrier
//   consumers   per K-tile: 3 x (fragment reads one step ahead, FN*FM MFMAs); then, with ALL fragments of the tile in registers,
//               s_barrier (next tile landed / this stage released) -> first fragments of the next tile -> the last FN*FM MFMAs;
//               epilogue (round 3: the barrier sat at the top of a tile before, with an exposed LDS round trip behind it)
// One barrier per K-tile for everybody, same LDS image / swizzle / zero-block redirect / tap masks as
// igemm_glds.hip, same epilogue (igemm_device.h). Register budget: 8 waves per CU -> 256 registers per lane.
#include "igemm_device.h"

namespace sfast {

// experiment (EXP bit 6): the accumulators in AGPRs -- the MFMA reads and writes C / D through the accumulation file's ports instead
// of the arch VGPR file the LDS returns write into (results are correct; MFMA hazards are not tracked inside inline asm: drained by
// hand before the epilogue reads the accumulators)
template <typename T> __device__ __forceinline__ void mfma32_agpr(f32x16 &c, const typename Elem<T>::vec8 &a, const typename Elem<T>::vec8 &b) {
    if constexpr (std::is_same<T, f16>::value)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int N> __device__ __forceinline__ void ws_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef const u32x4 __attribute__((address_space(1))) * ws_src_t;
typedef __attribute__((address_space(3))) void *ws_dst_t;

// Waves per SIMD the register allocator must leave room for. The 64x64 tile with a 3-stage ring (48 KB of LDS) is meant to run
// THREE workgroups per CU (24 waves = 6 per SIMD, <= 80 registers): M = 8192, N = 320 problems have 640 tiles -- 2.5 per CU -- and
// with two resident workgroups a third of the CUs run a second, mostly empty round.
constexpr int ws_min_waves(int threads, int lds_bytes) {
    return (threads == 512 && lds_bytes <= 48 * 1024) ? 6 : igemm_min_waves(threads, lds_bytes);
}

// EXP != 0: timing-only experiment instantiations (tools/ws_loop_probe.py; results are garbage): bit 0 no MFMAs, bit 1 no fragment
// reads, bit 2 no LDS-DMA requests inside the loop (the producers only wait and meet the barrier), bit 3 weight requests only,
// bit 4 / bit 5 (results are CORRECT): s_setprio 3 in the producer / consumer waves; bit 6 (CORRECT): accumulators in AGPRs; bit 7 (CORRECT): the loop header behind the barrier (see the consumer loop).
template <typename T, int BM, int BN, int WM, int WN, int PW, int NS, int MODE, bool GEGLU, bool STAGED = false, int EXP = 0>
__global__ void __launch_bounds__((WM * WN + PW) * 64, ws_min_waves((WM * WN + PW) * 64, NS *(BM + BN) * 128))
    igemm_glds_ws_kernel(const IgemmArgs a) {
    using vec8 = typename Elem<T>::vec8;
    constexpr int NC = WM * WN * 64;  // consumer threads
    constexpr int NP = PW * 64;       // producer threads
    constexpr int FM = BM / (WM * 32);
    constexpr int FN = BN / (WN * 32);
    constexpr int XCH = BM * 8 / NP;
    constexpr int WCH = BN * 8 / NP;
    constexpr int RPP = NP / 8;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    constexpr int WNB = FN * 32;
    constexpr int L = XCH + WCH;  // LDS-DMA requests per producer thread per K-tile
    static_assert((BM * 8) % NP == 0 && (BN * 8) % NP == 0, "staging mismatch");
    static_assert(RPP % 16 == 0, "swizzle phase must not depend on the staging pass");
    static_assert(L * (NS - 1) <= 63, "vmcnt field");
    static_assert(NS >= 3 && NS <= 5, "ring depth");
    static_assert(!GEGLU || (FN % 2 == 0), "GEGLU needs paired fragments");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    touch_args(a);
    if (MODE == 1) touch_conv_args(a);
    trace_mark(a, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const BlockTile bt = decode_block(a);  // XCD-aware (tile, K-split) of this workgroup
    if (bt.tile_m < 0) return;  // surplus block of an xmap = 2 grid (wave-uniform, before any barrier)
    const int tile_n = bt.tile_n, tile_m = bt.tile_m;
    const int m0 = tile_m * BM, n0 = tile_n * BNO;
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);

    if (wave >= WM * WN) {
        if constexpr ((EXP & 16) != 0) __builtin_amdgcn_s_setprio(3);  // experiment: the producers' address arithmetic and requests ahead of the MFMAs
        // =============================== producer wave ===============================================
        const int ptid = tid - NC;
        const int pwave = wave - WM * WN;
        const int rbase = ptid >> 3;
        const int kc = (ptid & 7) ^ ((rbase >> 1) & 7);  // source-side swizzle (LDS-DMA writes lane-linearly)
        const ws_src_t zero_src = (ws_src_t)(const void *)g_zero16;

        const T *xrow[XCH];   // MODE 0: row pointer at column kc*8, or nullptr
        int xoffB[XCH];       // MODE 1: element offset of (tap (0,0), channel kc*8) in source 2 (pitch C2)
        int xdAB[XCH];        //         (the same in source 1, pitch C1) - xoffB
        unsigned xmask[XCH];  // MODE 1: bit (r*KW+s) set when that tap is inside the image
        const PixelDecoder decode(a);
        unsigned rep_all = 0;
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
        const T *wrow[WCH];
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

        // all L requests of the next tile; tiles past kt_end are all-zero requests (constant vmcnt bookkeeping)
        auto issue_tile = [&]() {
            char *sx = smem + istage * STAGE + pwave * 1024;
            char *sw = sx + BM * 128;
            const bool tile_ok = issued < kt_end;
            const int k = issued * 64;
            if constexpr ((EXP & 8) != 0) {
#pragma unroll
                for (int i = 0; i < XCH; ++i) __builtin_amdgcn_global_load_lds(zero_src, (ws_dst_t)(sx + i * (RPP * 128)), 16, 0, 0);
            } else if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const bool ok = tile_ok & (xrow[i] != nullptr) & (k + kc * 8 < a.K);
                    const ws_src_t src = ok ? (ws_src_t)(const void *)(xrow[i] + k) : zero_src;
                    __builtin_amdgcn_global_load_lds(src, (ws_dst_t)(sx + i * (RPP * 128)), 16, 0, 0);
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
                    const ws_src_t src = ok ? (ws_src_t)(const void *)(sbase + off) : zero_src;
                    __builtin_amdgcn_global_load_lds(src, (ws_dst_t)(sx + i * (RPP * 128)), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                const bool ok = tile_ok & (wrow[i] != nullptr) & (MODE == 1 || k + kc * 8 < a.K);
                const ws_src_t src = ok ? (ws_src_t)(const void *)(wrow[i] + k) : zero_src;
                __builtin_amdgcn_global_load_lds(src, (ws_dst_t)(sw + i * (RPP * 128)), 16, 0, 0);
            }
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

        // The consumers meet barrier k + 1 BEFORE the last k-step's MFMAs of tile k, with every fragment of tile k in registers (see
        // their loop): the stage of tile k is free from that barrier on, one K-tile period earlier than its last MFMA. So the ring
        // holds NS tiles at the start, and barrier k + 1 refills the stage of tile k with tile k + NS: NS - 1 tiles in flight.
#pragma unroll
        for (int s = 0; s < NS; ++s) issue_tile();
        ws_wait_vmcnt<((EXP & 4) ? 0 : L *(NS - 1))>();  // the first tile has landed (this wave's share)
        __builtin_amdgcn_s_barrier();
        for (int kt = kt_begin + 1; kt < kt_end; ++kt) {
            ws_wait_vmcnt<((EXP & 4) ? 0 : L *(NS - 2))>();  // tile kt has landed
            __builtin_amdgcn_s_barrier();                    // the consumers hold all of tile kt - 1 in registers: its stage is free
            if constexpr ((EXP & 4) == 0) issue_tile();      // tile kt - 1 + NS -> that stage
        }
        ws_wait_vmcnt<0>();  // the zero-filled tail requests must have landed before the LDS is released
        return;
    }

    // =================================== consumer wave ===============================================
    if constexpr ((EXP & 32) != 0) __builtin_amdgcn_s_setprio(3);  // experiment: the MFMA waves ahead of the producers
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, hi = lane >> 5;

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

    trace_mark(a, 1);
    trace_mark(a, 2);
    // The fragment reads run one k-step ahead of the MFMAs ACROSS the tile boundary: before the last k-step's MFMAs of tile kt the
    // wave has every fragment of the tile in registers (lgkmcnt(0)), so it meets the producers at the barrier there -- tile kt + 1
    // has landed, stage kt is released -- and the first fragments of tile kt + 1 are fetched under those MFMAs. (With the barrier at
    // the top of a tile every tile began with an exposed LDS round trip, ~1/6 of the loop: profiles/r03_conv_patch_ab_run5.log
    // showed the loop indifferent to a third less LDS-DMA traffic, i.e. not bound by it.)
    int cstage = 0;
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
    auto read_frags = [&](const char *xs, int ks, int set) {
        if constexpr ((EXP & 2) != 0) return;
        const char *ws = xs + BM * 128;
        const int chunk = ks * 2 + hi;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
            af[set][fn] = *reinterpret_cast<const vec8 *>(ws + lds_off(wn * WNB + fn * 32 + l31, chunk));
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
            bf[set][fm] = *reinterpret_cast<const vec8 *>(xs + lds_off(wm * (FM * 32) + fm * 32 + l31, chunk));
    };
    auto mfma_set = [&](int set) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = mfma32(af[set][fn], bf[set][fm], acc[fn][fm]);
    };
    if constexpr ((EXP & 128) != 0) {
        // experiment (results are CORRECT): the loop header sits right behind the lgkmcnt(0) + barrier, where no LDS read is
        // outstanding. In the production loop below the header has the next tile's first fragments in flight, the waitcnt pass of
        // the compiler loses their order against the reads issued at the top of the body, and it puts a full `s_waitcnt lgkmcnt(0)`
        // in front of the first MFMA group of EVERY tile -- one exposed LDS round trip per K-tile besides the intended one
        // (hipcc -S: the `s_waitcnt lgkmcnt(0)` behind the first four ds_read_b128 of the loop body).
        __builtin_amdgcn_s_barrier();  // tile kt_begin has landed
        trace_mark(a, 3);
        {
            const char *xs = smem;
            read_frags(xs, 0, 0);
            read_frags(xs, 1, 1);
            mfma_set(0);
            read_frags(xs, 2, 0);
            mfma_set(1);
            read_frags(xs, 3, 1);
            mfma_set(0);
        }
        cstage = (NS > 1) ? 1 : 0;
        for (int kt = kt_begin + 1; kt < kt_end; ++kt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of tile kt - 1 is in registers
            __builtin_amdgcn_s_barrier();                       // tile kt has landed, the stage of tile kt - 1 is released
            asm volatile("" ::: "memory");
            const char *xs = smem + cstage * STAGE;
            cstage = (cstage + 1 == NS) ? 0 : cstage + 1;
            mfma_set(1);               // the last k-step of tile kt - 1
            read_frags(xs, 0, 0);
            read_frags(xs, 1, 1);
            mfma_set(0);
            read_frags(xs, 2, 0);
            mfma_set(1);
            read_frags(xs, 3, 1);
            mfma_set(0);
        }
        mfma_set(1);
    } else {
        __builtin_amdgcn_s_barrier();  // tile kt_begin has landed
        trace_mark(a, 3);
        read_frags(smem, 0, 0);
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const char *xs = smem + cstage * STAGE;
            cstage = (cstage + 1 == NS) ? 0 : cstage + 1;
    #pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
                    read_frags(xs, ks + 1, (ks + 1) & 1);
                } else if (kt + 1 < kt_end) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");  // (the reads below stay below)
                    read_frags(smem + cstage * STAGE, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs (the scheduler sinks them otherwise)
    #pragma unroll
                for (int fn = 0; fn < FN; ++fn)
    #pragma unroll
                    for (int fm = 0; fm < FM; ++fm) {
                        if constexpr ((EXP & 64) != 0) {
                            mfma32_agpr<T>(acc[fn][fm], af[ks & 1][fn], bf[ks & 1][fm]);
                        } else if constexpr ((EXP & 1) == 0) {
                            acc[fn][fm] = mfma32(af[ks & 1][fn], bf[ks & 1][fm], acc[fn][fm]);
                        } else {
                            asm volatile("" ::"v"(af[ks & 1][fn]), "v"(bf[ks & 1][fm]));  // the fragment reads stay
                        }
                    }
            }
        }
    }
    if constexpr ((EXP & 64) != 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs have written their accumulators
    trace_mark(a, 4);
    run_epilogue<T, BM, BNO, FN, FM, GEGLU, EPI_EARLY, NC, STAGED>(a, acc, epi, smem, m0, n0, m0 + wm * (FM * 32), n0 + wn * (GEGLU ? WNB / 2 : WNB), l31, hi,
                                                             tid, bt.split);
    trace_finish(a);
}

// ---- host side ------------------------------------------------------------------------------------------
// variant ids 21.. (igemm.hip kVariants, pipe 2): tile, consumer waves WM x WN, producer waves, ring depth
// (128x64 / 64x128 with a 3-deep ring: 72 KB of LDS -> two workgroups per CU; M = 8192 x N = 320 becomes 320 / 384 workgroups = ONE
//  round on 256 CUs where the 64x64 tile needs 640 = two rounds of ~5 us workgroups)
#define SFAST_FOR_WS_VARIANTS(T, MODE, OP) \
    OP(T, 128, 128, 2, 2, 4, 4, MODE, false) \
    OP(T, 128, 160, 4, 1, 4, 4, MODE, false) \
    OP(T, 64, 64, 2, 2, 4, 4, MODE, false)   \
    OP(T, 64, 64, 2, 2, 4, 3, MODE, false)   \
    OP(T, 128, 64, 2, 2, 4, 3, MODE, false)  \
    OP(T, 64, 128, 2, 2, 4, 3, MODE, false)

#define SFAST_FOR_WS_GEGLU_VARIANTS(T, OP) \
    OP(T, 128, 128, 2, 2, 4, 4, 0, true)   \
    OP(T, 64, 128, 2, 2, 4, 3, 0, true)

template <typename T, int BM, int BN, int WM, int WN, int PW, int NS, int MODE, bool GEGLU, bool STAGED = false>
static int ws_set_attr() {
    constexpr int smem = NS * (BM + BN) * 128;
    auto kern = igemm_glds_ws_kernel<T, BM, BN, WM, WN, PW, NS, MODE, GEGLU, STAGED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(igemm_glds_ws %dx%dx%d): %s", BM, BN, NS, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

int igemm_glds_ws_init() {
    int rc = 0;
#define INIT_OP(T, BM, BN, WM, WN, PW, NS, MODE, G)                  \
    if (!rc) rc = ws_set_attr<T, BM, BN, WM, WN, PW, NS, MODE, G>(); \
    if (!rc && !G) rc = ws_set_attr<T, BM, BN, WM, WN, PW, NS, MODE, false, true>();
    SFAST_FOR_WS_VARIANTS(f16, 0, INIT_OP)
    SFAST_FOR_WS_VARIANTS(f16, 1, INIT_OP)
    SFAST_FOR_WS_VARIANTS(bf16, 0, INIT_OP)
    SFAST_FOR_WS_VARIANTS(bf16, 1, INIT_OP)
    SFAST_FOR_WS_GEGLU_VARIANTS(f16, INIT_OP)
    SFAST_FOR_WS_GEGLU_VARIANTS(bf16, INIT_OP)
#undef INIT_OP
    return rc;
}

extern unsigned long long *g_igemm_trace;  // igemm_glds.hip
extern int g_igemm_exp;                     // igemm_glds.hip (SFAST_IGEMM_EXP, latched by sfast_hip_set_trace)

template <typename T, int MODE>
static int ws_dispatch(const IgemmArgs &a, int BM_, int BN_, int NS_, bool geglu, hipStream_t st) {
#define LAUNCH_OP(TT, BM, BN, WM, WN, PW, NS, MODE_, G_)                                                                       \
    if (BM_ == BM && BN_ == BN && NS_ == NS && geglu == G_) {                                                                  \
        if constexpr (!G_) {                                                                                                   \
            if (a.stage_out) {                                                                                                 \
                auto ks = igemm_glds_ws_kernel<TT, BM, BN, WM, WN, PW, NS, MODE_, false, true>;                                \
                hipLaunchKernelGGL(ks, igemm_grid(a), dim3((WM * WN + PW) * 64), NS *(BM + BN) * 128, st, a); \
                return check_launch("igemm_glds_ws_staged");                                                                   \
            }                                                                                                                  \
        }                                                                                                                      \
        auto kern = igemm_glds_ws_kernel<TT, BM, BN, WM, WN, PW, NS, MODE_, G_>;                                               \
        hipLaunchKernelGGL(kern, igemm_grid(a), dim3((WM * WN + PW) * 64), NS *(BM + BN) * 128, st, a); \
        return check_launch("igemm_glds_ws");                                                                                  \
    }
#ifdef SFAST_PROBES  // timing-only instantiations (results are garbage): probe build only (build.py --probes)
    if constexpr (std::is_same<T, f16>::value && MODE == 1) {
        if (g_igemm_exp != 0 && !geglu && !a.stage_out) {  // timing experiments: two conv tiles only
#define LAUNCH_EXP(BM, BN, WM, WN, NS, E)                                                                                      \
    if (BM_ == BM && BN_ == BN && NS_ == NS && g_igemm_exp == E) {                                                             \
        auto kern = igemm_glds_ws_kernel<f16, BM, BN, WM, WN, 4, NS, 1, false, false, E>;                                      \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NS *(BM + BN) * 128); \
        hipLaunchKernelGGL(kern, igemm_grid(a), dim3((WM * WN + 4) * 64), NS *(BM + BN) * 128, st, a);                         \
        return check_launch("igemm_glds_ws_exp");                                                                              \
    }
            LAUNCH_EXP(128, 128, 2, 2, 4, 1) LAUNCH_EXP(128, 128, 2, 2, 4, 2) LAUNCH_EXP(128, 128, 2, 2, 4, 3)
            LAUNCH_EXP(128, 128, 2, 2, 4, 4) LAUNCH_EXP(128, 128, 2, 2, 4, 7) LAUNCH_EXP(128, 128, 2, 2, 4, 8)
            LAUNCH_EXP(128, 128, 2, 2, 4, 16) LAUNCH_EXP(128, 128, 2, 2, 4, 32) LAUNCH_EXP(128, 160, 4, 1, 4, 16) LAUNCH_EXP(128, 160, 4, 1, 4, 32)
            LAUNCH_EXP(128, 128, 2, 2, 4, 6) LAUNCH_EXP(128, 160, 4, 1, 4, 6) LAUNCH_EXP(128, 128, 2, 2, 4, 5) LAUNCH_EXP(128, 160, 4, 1, 4, 5)
            LAUNCH_EXP(128, 128, 2, 2, 4, 64) LAUNCH_EXP(128, 160, 4, 1, 4, 64) LAUNCH_EXP(128, 128, 2, 2, 4, 68) LAUNCH_EXP(128, 160, 4, 1, 4, 68)
            LAUNCH_EXP(128, 128, 2, 2, 4, 128) LAUNCH_EXP(128, 160, 4, 1, 4, 128) LAUNCH_EXP(128, 128, 2, 2, 4, 132) LAUNCH_EXP(128, 160, 4, 1, 4, 132)
            LAUNCH_EXP(128, 160, 4, 1, 4, 1) LAUNCH_EXP(128, 160, 4, 1, 4, 2) LAUNCH_EXP(128, 160, 4, 1, 4, 3)
            LAUNCH_EXP(128, 160, 4, 1, 4, 4) LAUNCH_EXP(128, 160, 4, 1, 4, 7) LAUNCH_EXP(128, 160, 4, 1, 4, 8)
#undef LAUNCH_EXP
        }
    }
#endif
    if (!geglu) {
        SFAST_FOR_WS_VARIANTS(T, MODE, LAUNCH_OP)
    } else {
        if (MODE == 0) {
            SFAST_FOR_WS_GEGLU_VARIANTS(T, LAUNCH_OP)
        }
    }
#undef LAUNCH_OP
    set_error("igemm_glds_ws: no kernel for tile %dx%d ring %d", BM_, BN_, NS_);
    return SFAST_ERR_UNSUPPORTED;
}

int igemm_glds_ws_launch(const IgemmArgs &a_in, int dtype, int mode, bool geglu, int BM, int BN, int NS, hipStream_t st) {
    IgemmArgs a = a_in;
    a.trace = g_igemm_trace;
    if (dtype == SFAST_F16) return mode ? ws_dispatch<f16, 1>(a, BM, BN, NS, geglu, st) : ws_dispatch<f16, 0>(a, BM, BN, NS, geglu, st);
    return mode ? ws_dispatch<bf16, 1>(a, BM, BN, NS, geglu, st) : ws_dispatch<bf16, 0>(a, BM, BN, NS, geglu, st);
}

}  // namespace sfast
