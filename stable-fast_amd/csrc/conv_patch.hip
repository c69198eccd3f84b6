// 3x3 convolution with an LDS-resident input PATCH (pipe 3 of the implicit-GEMM family).
//
// Replaces the same reference call as the other conv kernels -- sfast::cudnn_convolution_bias[_add][_act]
// (/root/reference/src/sfast/csrc/operators/cudnn/cudnn_convolution_impl.cc:995-998, 1265-1286) -- for the case the UNet spends its
// conv time in: 3x3, stride 1, padding 1, dense NHWC, channel counts that are multiples of 64.
//
// Why it was written. Over the whole SD1.5 step the launch times of the MFMA conv kernels follow  bytes_into_LDS / 13.3 TB/s  (52 GB/s
// per CU) within a few percent -- 64x64 tiles at the 32x32 level: 472 MB -> 35 us predicted, 34.5 measured; the GEGLU GEMM: 419 MB ->
// 31 us, 28 measured (profiles/r03_kernels_per_op_run1.json) -- which reads like a bound on what a CU can pull from L2 into LDS.
// STATUS: measured, and NOT faster than the implicit-im2col pipes on any SD1.5 shape (0.79-1.00x, profiles/r03_conv_ab_run10.log). The
// K-loop probes (tools/ws_loop_probe.py, profiles/r03_ws_loop_probe_run7.log, r03_conv_patch_loop_probe_run10.log) show why the fit
// misled: LDS-DMA time and MFMA time ADD in these loops instead of overlapping, and with two weight-producer waves instead of four
// this kernel's request issue is slower than the one it replaces. The autotuner measures it like any other candidate and does not
// pick it; it stays as that measured candidate.
// The implicit-im2col kernels fetch every input pixel NINE times per tile (once per tap) on top of the weights. Here a workgroup
// owns BM consecutive output pixels = whole image rows, keeps the (rows + 2) x (W + 2) input patch of ONE 64-channel slice in LDS and
// runs all nine taps against it: the activation traffic into LDS drops ~9x (for the 16x16 / 8x8 levels the launch then moves little
// more than the weights once), the weight traffic is unchanged.
//
//   K order of a workgroup: channel slice major, tap minor (unit u = slice * 9 + tap; weights stay [Cout][kh][kw][Cin], so unit u reads
//   weight columns tap * Cin + slice * 64 ...). Split-K cuts between slices.
//   LDS: two patch buffers (slice s is computed while slice s + 1 lands) + a ring of weight tiles (BN rows x 128 B), as deep as the
//   rest of the 160 KiB allows (3 .. 5 stages). Patch rows are 128 B per pixel, XOR-swizzled by the pixel index exactly like the tile
//   rows of the other pipes; image borders (and the two padding rows between images when a tile spans images) are patch pixels loaded
//   from the device zero block, so the consumers never test a border.
//   Waves: two weight-producer and two patch-producer waves issue LDS-DMA only; WM x WN consumer waves read fragments and issue MFMAs;
//   one barrier per unit -- the structure of igemm_glds_ws.hip. Same fragment layout, same epilogue (igemm_device.h: bias / row bias /
//   residual / activation, staged stores, GroupNorm statistics, split-K slabs).
//
// Round 4: this file is EVIDENCE, not product -- the autotuner never selects pipe 3. It is compiled only into the probe build
// (-DSFAST_PROBES: build.py --probes -> libsfast_hip_probes.so, what tools/conv_ab.py and the conv_patch tests load); the product
// library carries the two stubs at the bottom (no patch variant fits, so the planner never proposes one).
#include "igemm_device.h"

namespace sfast {
#ifdef SFAST_PROBES

struct PatchGeom {
    int W2, H2;        // padded image width / height (W + 2, H + 2)
    int pp;            // patch pixels of a tile (patch rows x W2)
    int npi;           // LDS-DMA instructions per patch (8 pixels each, ceil(pp / 8))
    int patch_bytes;   // bytes of one patch buffer (npi * 1024)
    int units_per_split, units;  // K-loop units (slice * 9 + tap) per split / in total
    float r_W, r_H, r_W2, r_H2;  // reciprocals for the index decodes at the top of a workgroup
};

template <int N> __device__ __forceinline__ void cp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef const u32x4 __attribute__((address_space(1))) * cp_src_t;
typedef __attribute__((address_space(3))) void *cp_dst_t;

constexpr int CP_MAXPI = 32;   // patch LDS-DMA instructions per patch-producer wave and slice (two such waves: patches up to 512 pixels)
constexpr int CP_SPREAD = 7;   // the next slice's patch is requested during taps 0 .. 6 of the current one

// PWW weight-producer waves + PWP patch-producer waves. They are separate waves on purpose: a weight producer issues the SAME number of
// requests for every unit, so one counted vmcnt tells it that unit u's tile has landed; the patch producers have requests in taps
// 0 .. 6 only and simply drain (vmcnt(0)) at a slice change, two or more units after their last request. (The first version let every
// producer wave carry both kinds and padded the patch share of every unit with requests into a sink to keep the count constant:
// 72 requests per slice and wave against the 81 of the implicit-im2col kernel, a ring one stage shallower -- and 0.73-0.82x its speed,
// profiles/r03_conv_patch_ab_run4.log. A CU retires one 1-KiB LDS-DMA request per ~28 cycles: requests are the currency.)
// EXP != 0: timing-only experiment instantiations (tools/ws_loop_probe.py --patch; results are garbage): bit 0 no MFMAs, bit 1 no fragment
// reads, bit 2 no weight requests inside the loop, bit 3 no patch requests inside the loop.
template <typename T, int BM, int BN, int WM, int WN, int PWW, int PWP, int NSW, bool STAGED, int EXP = 0>
__global__ void __launch_bounds__((WM * WN + PWW + PWP) * 64, (WM * WN + PWW + PWP) / 4) conv_patch_kernel(const IgemmArgs a, const PatchGeom g) {
    using vec8 = typename Elem<T>::vec8;
    constexpr int NC = WM * WN * 64, NPW = PWW * 64;
    constexpr int FM = BM / (WM * 32), FN = BN / (WN * 32);
    constexpr int WCH = BN * 8 / NPW;  // weight chunks per weight-producer thread and unit
    constexpr int RPP = NPW / 8;
    constexpr int WNB = FN * 32;
    constexpr int PPT = (CP_MAXPI + CP_SPREAD - 1) / CP_SPREAD;  // patch requests per patch-producer wave and unit
    static_assert((BN * 8) % NPW == 0 && RPP % 16 == 0, "weight staging mismatch");
    static_assert(NSW >= 3 && NSW <= 5 && WCH * (NSW - 1) <= 63, "ring depth / vmcnt field");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const pbuf0 = smem;
    char *const wring = smem + 2 * g.patch_bytes;

    touch_args(a);
    touch_conv_args(a);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const BlockTile bt = decode_block(a);
    if (bt.tile_m < 0) return;  // surplus block of an xmap = 2 grid (wave-uniform, before any barrier)
    const int m0 = bt.tile_m * BM, n0 = bt.tile_n * BN;
    const int u_begin = bt.split * g.units_per_split;
    const int u_end = min(g.units, u_begin + g.units_per_split);
    const int cin = a.C1 + a.C2;
    // first patch row in PADDED global row coordinates (image b occupies padded rows b * H2 .. b * H2 + H2 - 1, its own rows at +1)
    const int gy0 = fdiv22(m0, a.W, g.r_W);             // global output row of the tile's first pixel (tiles are whole rows)
    const int b0 = fdiv22(gy0, a.H, g.r_H);
    const int pg_first = b0 * g.H2 + (gy0 - b0 * a.H);  // = padded row of the first output row, minus 1
    const cp_src_t zero_src = (cp_src_t)(const void *)g_zero16;

    if (wave >= WM * WN + PWW) {
        // =============================== patch-producer wave ==========================================
        const int pwave = wave - (WM * WN + PWW);
        const int nbatch = a.M / (a.H * a.W);
        // this lane's share of the patch: instruction ii = pwave + PWP * j covers patch pixels ii * 8 .. + 7, lane -> (pixel, 16-byte slot);
        // packed: source pixel index * 8 + channel chunk, or -1 for a border / padding pixel
        int ppix[CP_MAXPI];
#pragma unroll
        for (int j = 0; j < CP_MAXPI; ++j) {
            const int ii = pwave + PWP * j;
            const int pp = ii * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((pp >> 1) & 7);  // source-side swizzle: LDS-DMA writes lane-linearly
            const int prow = fdiv22(pp, g.W2, g.r_W2);
            const int pcol = pp - prow * g.W2;
            const int pgr = pg_first + prow;
            const int b = fdiv22(pgr, g.H2, g.r_H2);
            const int yy = pgr - b * g.H2 - 1, xx = pcol - 1;
            const bool ok = pp < g.pp && b < nbatch && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            ppix[j] = ok ? (((b * a.H + yy) * a.W + xx) << 3) + chunk : -1;
        }
        // request of patch instruction j for channel slice `cs` into patch buffer cs & 1 (nothing for instructions past the patch)
        auto issue_patch = [&](int j, int cs) __attribute__((always_inline)) {
            const int ii = pwave + PWP * j;
            if (ii >= g.npi) return;  // wave-uniform
            const int c0 = cs * 64;
            const bool first = c0 < a.C1;
            const T *src = first ? (const T *)a.x : (const T *)a.x2;
            const int pitch = first ? a.C1 : a.C2, cc = first ? c0 : c0 - a.C1;
            const int v = ppix[j];
            const cp_src_t s = (v >= 0) ? (cp_src_t)(const void *)(src + (int64_t)(v >> 3) * pitch + cc + (v & 7) * 8) : zero_src;
            __builtin_amdgcn_global_load_lds(s, (cp_dst_t)(pbuf0 + (cs & 1) * g.patch_bytes + ii * 1024), 16, 0, 0);
        };
        int cs = fdiv22(u_begin, 9, 0.11111111938953400f), tap = u_begin - cs * 9;
#pragma unroll
        for (int j = 0; j < CP_MAXPI; ++j) issue_patch(j, cs);
        cp_wait_vmcnt<0>();  // the first patch
        for (int u = u_begin; u < u_end; ++u) {
            if (tap == 0 && u != u_begin) cp_wait_vmcnt<0>();  // slice change: this slice's patch, requested during taps 0 .. 6 of the last one
            __builtin_amdgcn_s_barrier();
            const bool more = (cs + 1) * 9 < u_end;
#pragma unroll
            for (int t = 0; t < CP_SPREAD; ++t) {   // (static indices into ppix: one uniform branch per tap)
                if (tap == t && more && (EXP & 8) == 0) {
#pragma unroll
                    for (int j = 0; j < PPT; ++j)
                        if (t * PPT + j < CP_MAXPI) issue_patch(t * PPT + j, cs + 1);
                }
            }
            if (++tap == 9) {
                tap = 0;
                ++cs;
            }
        }
        cp_wait_vmcnt<0>();
        return;
    }
    if (wave >= WM * WN) {
        // =============================== weight-producer wave =========================================
        const int ptid = tid - NC;
        const int pwave = wave - WM * WN;
        const int rbase = ptid >> 3;
        const int kc = (ptid & 7) ^ ((rbase >> 1) & 7);
        const T *wrow[WCH];
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int n = n0 + rbase + i * RPP;
            wrow[i] = (n < a.N) ? (const T *)a.w[0] + (int64_t)n * a.ldw + kc * 8 : nullptr;
        }
        int wstage = 0;
        auto issue_weights = [&](int u) __attribute__((always_inline)) {
            const bool ok_u = u < u_end;
            const int cs = fdiv22(u, 9, 0.11111111938953400f), tap = u - cs * 9;
            const int k = tap * cin + cs * 64;
            char *sw = wring + wstage * (BN * 128) + pwave * 1024;
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                const bool ok = ok_u & (wrow[i] != nullptr);
                const cp_src_t s = ok ? (cp_src_t)(const void *)(wrow[i] + k) : zero_src;
                __builtin_amdgcn_global_load_lds(s, (cp_dst_t)(sw + i * (RPP * 128)), 16, 0, 0);
            }
            wstage = (wstage + 1 == NSW) ? 0 : wstage + 1;
        };
        // NSW tiles at the start; barrier u + 1 (the consumers then hold all of unit u in registers) refills unit u's stage with
        // unit u + NSW (igemm_glds_ws.hip, producer loop)
        int u_next = u_begin;
#pragma unroll
        for (int s = 0; s < NSW; ++s) issue_weights(u_next++);
        cp_wait_vmcnt<WCH *(NSW - 1)>();
        __builtin_amdgcn_s_barrier();
        for (int u = u_begin + 1; u < u_end; ++u) {
            cp_wait_vmcnt<((EXP & 4) ? 0 : WCH *(NSW - 2))>();  // unit u's tile has landed (this wave's share; requests retire in order)
            __builtin_amdgcn_s_barrier();
            if constexpr ((EXP & 4) == 0) issue_weights(u_next++);
        }
        cp_wait_vmcnt<0>();  // nothing may still be landing when the LDS is released / re-used by the epilogue
        return;
    }

    // =================================== consumer wave ===============================================
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, hi = lane >> 5;

    constexpr bool EPI_EARLY = FN * FM <= 4 && (WM * WN + PWW + PWP) <= 8;  // (12-wave tiles have 168 registers per lane: operands fetched late)
    EpiOperands<(EPI_EARLY ? FN : 1), (EPI_EARLY ? FM : 1)> epi;
    if constexpr (EPI_EARLY) epilogue_prefetch<T, FN, FM, false>(a, epi, m0 + wm * (FM * 32), n0 + wn * WNB, l31, hi);

    // patch pixel of this lane's output pixel for tap (0, 0): base + r * W2 + s is the input pixel of tap (r, s)
    int pbase[FM];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int pix = wm * (FM * 32) + fm * 32 + l31;          // pixel inside the tile
        const int r = fdiv22(pix, a.W, g.r_W), x = pix - r * a.W;
        const int gy = gy0 + r;
        const int b = fdiv22(gy, a.H, g.r_H);
        const int prow_out = b * g.H2 + (gy - b * a.H) + 1 - pg_first;  // patch row of the output pixel's own row
        pbase[fm] = (prow_out - 1) * g.W2 + x;
    }

    f32x16 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;

    int cstage = 0;
    int cs = fdiv22(u_begin, 9, 0.11111111938953400f), tap = u_begin - cs * 9;
    cs = __builtin_amdgcn_readfirstlane(cs);
    tap = __builtin_amdgcn_readfirstlane(tap);
    vec8 af[2][FN], bf[2][FM];
    int prow[FM], pswz[FM];
    const char *ws, *ps;
    // addresses of unit (cs, tap) in ring stage `cstage`
    auto enter_unit = [&]() {
        ws = wring + cstage * (BN * 128);
        ps = pbuf0 + (cs & 1) * g.patch_bytes;
        const int tr = (tap >= 6) ? 2 : (tap >= 3 ? 1 : 0);
        const int toff = tr * g.W2 + (tap - 3 * tr);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int pp = pbase[fm] + toff;
            prow[fm] = pp << 7;
            pswz[fm] = (pp >> 1) & 7;
        }
    };
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
            bf[set][fm] = *reinterpret_cast<const vec8 *>(ps + prow[fm] + ((chunk ^ pswz[fm]) << 4));
    };
    // fragment reads one k-step ahead of the MFMAs across the unit boundary, the barrier before the last k-step's MFMAs
    // (igemm_glds_ws.hip, consumer loop)
    __builtin_amdgcn_s_barrier();
    enter_unit();
    read_frags(0, 0);
    for (int u = u_begin; u < u_end; ++u) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
                read_frags(ks + 1, (ks + 1) & 1);
            } else if (u + 1 < u_end) {
                cstage = (cstage + 1 == NSW) ? 0 : cstage + 1;
                if (++tap == 9) {
                    tap = 0;
                    ++cs;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");  // (the reads below stay below)
                enter_unit();
                read_frags(0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    if constexpr ((EXP & 1) == 0) {
                        acc[fn][fm] = mfma32(af[ks & 1][fn], bf[ks & 1][fm], acc[fn][fm]);
                    } else {
                        asm volatile("" ::"v"(af[ks & 1][fn]), "v"(bf[ks & 1][fm]));  // the fragment reads stay
                    }
                }
        }
    }

    run_epilogue<T, BM, BN, FN, FM, false, EPI_EARLY, NC, STAGED>(a, acc, epi, smem, m0, n0, m0 + wm * (FM * 32), n0 + wn * WNB, l31, hi, tid, bt.split);
}

// ---- host side ------------------------------------------------------------------------------------------
// variant ids 31.. (igemm.hip kVariants, pipe 3): tile, consumer waves WM x WN; two weight-producer + two patch-producer waves; the
// weight ring takes what the patch buffers leave of the 160 KiB (3 .. 5 stages)
#define SFAST_FOR_PATCH_VARIANTS(T, OP) \
    OP(T, 128, 160, 4, 1)               \
    OP(T, 128, 128, 2, 2)               \
    OP(T, 128, 64, 2, 2)

static size_t patch_lds_bytes(int BN, int NSW, int patch_bytes) { return (size_t)2 * patch_bytes + (size_t)NSW * BN * 128; }
static int patch_ring_depth(int BM, int BN, int patch_bytes, bool staged) {
    for (int n = 5; n >= 3; --n) {
        const size_t need = staged ? (size_t)BM * (BN * 2 + 8) + 16 + 256 * 16 : 0;
        if (patch_lds_bytes(BN, n, patch_bytes) <= 160 * 1024 && need <= 160 * 1024) return n;
    }
    return 0;
}

// patch rows of a BM-pixel tile of whole image rows: inside one image rows + 2, over whole images (H + 2) each; 0 = not tileable
int conv_patch_rows(int H, int W, int BM) {
    if (W <= 0 || H <= 0 || BM % W != 0) return 0;
    const int hw = H * W;
    if (hw % BM == 0) return BM / W + 2;
    if (BM % hw == 0) return (BM / hw) * (H + 2);
    return 0;
}

// can variant (BM, BN) run this conv? (3x3 / stride 1 / pad 1 / dense / channel slices of 64 are checked by the caller's caps)
bool conv_patch_fits(int H, int W, int M, int BM, int BN) {
    const int pr = conv_patch_rows(H, W, BM);
    if (!pr || M % BM != 0) return false;
    const int pp = pr * (W + 2), npi = (pp + 7) / 8;
    return npi <= 2 * CP_MAXPI && patch_ring_depth(BM, BN, npi * 1024, false) >= 3;
}

extern int g_igemm_exp;  // igemm_glds.hip (SFAST_IGEMM_EXP, latched by sfast_hip_set_trace)

template <typename T, int BM, int BN, int WM, int WN, int NSW>
static int patch_launch_one(const IgemmArgs &a, const PatchGeom &g, hipStream_t st) {
    const size_t smem = patch_lds_bytes(BN, NSW, g.patch_bytes);
    const dim3 block((WM * WN + 4) * 64);
    if constexpr (std::is_same<T, f16>::value && NSW >= 4 && BN >= 128) {
        if (g_igemm_exp != 0 && !a.stage_out) {  // timing experiments
#define PATCH_EXP(E)                                                                                                              \
    if (g_igemm_exp == E) {                                                                                                       \
        auto kern = conv_patch_kernel<f16, BM, BN, WM, WN, 2, 2, NSW, false, E>;                                                  \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
        hipLaunchKernelGGL(kern, igemm_grid(a), block, smem, st, a, g);                                                           \
        return check_launch("conv_patch_exp");                                                                                    \
    }
            PATCH_EXP(1) PATCH_EXP(2) PATCH_EXP(3) PATCH_EXP(4) PATCH_EXP(8) PATCH_EXP(12) PATCH_EXP(15)
#undef PATCH_EXP
        }
    }
    static bool attr_done = false;  // per instantiation, idempotent: the kernels may use the whole 160 KiB (the size depends on the image width)
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(conv_patch_kernel<T, BM, BN, WM, WN, 2, 2, NSW, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(conv_patch_kernel<T, BM, BN, WM, WN, 2, 2, NSW, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (a.stage_out) {
        const size_t need = (size_t)BM * (BN * 2 + 8) + 16 + (size_t)WM * WN * 64 * 16;
        const size_t sm = smem > need ? smem : need;
        hipLaunchKernelGGL((conv_patch_kernel<T, BM, BN, WM, WN, 2, 2, NSW, true>), igemm_grid(a), block, sm, st, a, g);
        return check_launch("conv_patch_staged");
    }
    hipLaunchKernelGGL((conv_patch_kernel<T, BM, BN, WM, WN, 2, 2, NSW, false>), igemm_grid(a), block, smem, st, a, g);
    return check_launch("conv_patch");
}

// `a` carries the plan of igemm_run (tiles, splits with ktiles_per_split a multiple of 9, block map)
int conv_patch_launch(const IgemmArgs &a, int dtype, int BM, int BN, hipStream_t st) {
    PatchGeom g{};
    g.W2 = a.W + 2;
    g.H2 = a.H + 2;
    g.pp = conv_patch_rows(a.H, a.W, BM) * g.W2;
    g.npi = (g.pp + 7) / 8;
    g.patch_bytes = g.npi * 1024;
    g.units = a.ktiles;
    g.units_per_split = a.ktiles_per_split;
    g.r_W = 1.0f / (float)a.W;
    g.r_H = 1.0f / (float)a.H;
    g.r_W2 = 1.0f / (float)g.W2;
    g.r_H2 = 1.0f / (float)g.H2;
    SFAST_REQUIRE(g.pp > 0 && a.ktiles_per_split % 9 == 0 && (a.C1 + a.C2) * 9 == a.K, SFAST_ERR_UNSUPPORTED, "conv_patch: plan does not fit the patch kernel");
    const int nsw = patch_ring_depth(BM, BN, g.patch_bytes, a.stage_out != 0);
    SFAST_REQUIRE(nsw >= 3, SFAST_ERR_UNSUPPORTED, "conv_patch: the patch leaves no room for a weight ring");
#define PATCH_OP(T, BM_, BN_, WM_, WN_)                                                              \
    if (BM == BM_ && BN == BN_) {                                                                    \
        if (nsw == 5) return patch_launch_one<T, BM_, BN_, WM_, WN_, 5>(a, g, st);                   \
        if (nsw == 4) return patch_launch_one<T, BM_, BN_, WM_, WN_, 4>(a, g, st);                   \
        return patch_launch_one<T, BM_, BN_, WM_, WN_, 3>(a, g, st);                                 \
    }
    if (dtype == SFAST_F16) {
        SFAST_FOR_PATCH_VARIANTS(f16, PATCH_OP)
    } else {
        SFAST_FOR_PATCH_VARIANTS(bf16, PATCH_OP)
    }
#undef PATCH_OP
    set_error("conv_patch: no kernel for tile %dx%d", BM, BN);
    return SFAST_ERR_UNSUPPORTED;
}

#else  // !SFAST_PROBES: the product library has no patch pipe

bool conv_patch_fits(int, int, int, int, int) { return false; }
int conv_patch_launch(const IgemmArgs &, int, int, int, hipStream_t) {
    set_error("conv_patch: this library was built without -DSFAST_PROBES (the patch pipe is a measured, never-selected candidate)");
    return SFAST_ERR_UNSUPPORTED;
}

#endif
}  // namespace sfast
