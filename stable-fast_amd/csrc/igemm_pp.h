// Pipe 5 of the MFMA implicit GEMM (round 6): 256-row tiles, 8 waves in two groups that alternate between a MEMORY phase and an
// MFMA phase ("ping-pong": each SIMD hosts one wave of either group, so its matrix pipe always has a wave inside an MFMA phase while
// the other wave of that SIMD reads fragments and issues the LDS-DMA requests of a later K-tile).
//
// Why (measured, DESIGN.md section 9 rounds 3-5 + round 6): the 64..128-row tiles of pipes 0-4 sit at 20-35 % of the dense MFMA peak
// on every large-M problem. Per K-tile a workgroup pulls (BM + BN) * 128 bytes through its CU's L2 -> LDS path (~56-64 B/clk) for
// BM * BN * 64 MACs: 128 x 128 needs 585 clocks of transfer for 512 clocks of MFMA issue, 256 x 160 needs 930 for 1280, 256 x 256
// 1170 for 2048 -- only the 256-row tiles leave the matrix pipe something to hide the transfer behind. The schedule follows the
// 8-phase template of the platform guide (cdna_hip_programming.md "The 256^2 8-phase template", T3+T4+T5): counted `s_waitcnt vmcnt`
// (never 0 in the steady state), raw `s_barrier`, `s_setprio 1` around the MFMA clusters, LDS XOR swizzle with the inverse
// permutation applied on the LDS-DMA SOURCE address.
//
// Structure. BM = 256 pixels x BN weight rows per workgroup, 512 threads. Waves 0-3 are group 0, waves 4-7 group 1 (a workgroup's
// waves go to the SIMDs cyclically, so waves w and w + 4 share one).
//   BN < 256 : 4 (M) x 2 (N) waves, wave = 64 pixels x {FN0, FN1} * 32 weight rows; the groups split the weight rows, unevenly
//              when BN / 32 is odd (160 = 96 + 64: the SIMD still runs FM * 5 MFMAs per k-step, 3 : 2 between its two waves)
//   BN >= 256: 2 (M) x 4 (N) waves, wave = 128 pixels x BN / 4 weight rows; the groups split the pixels
// A PHASE is KSP 16-wide k-steps of the wave's whole tile (KSP = 4: one phase per K-tile): memory part {KSP * (FM + FN) ds_read_b128
// interleaved with this phase's share of the LDS-DMA requests, the counted vmcnt, lgkmcnt(0)} s_barrier, MFMA part {KSP * FM * FN
// MFMAs} s_barrier. Group 1 runs one barrier behind group 0, so in every barrier interval one group is in its memory part and the
// other in its MFMA part. Phases are long on purpose: the first version used one k-step per phase (8 barriers per K-tile, as the
// guide's template) and measured ~100 clocks of fixed cost per barrier interval against MFMA parts of 128 - 192 clocks
// (profiles/r06_pp_loop_probe_run2.log: the loop with barriers alone 40 us of a 139 us conv; reads, MFMAs and requests ADDED UP
// instead of overlapping).
// Ring: NS stages of one K-tile (64 halves of K; 128-byte rows, so every request reads whole cache lines). During the phases of tile
// u the requests of tile u + NS - 1 go into the stage of tile u - 1. Orderings (interval = span between two consecutive barriers;
// with NP = 4 / KSP phases per tile, group 0 has the memory part of phase p in interval 2p, group 1 in 2p + 1):
//   WAR  every wave waits for ITS fragment reads (lgkmcnt(0)) BEFORE the barrier that ends a memory part, so the stage of tile u - 1
//        is free once group 1 has passed the barrier behind its last memory part of that tile (end of interval 2 NP u - 1); the
//        memory parts of tile u start in interval 2 NP u (group 0).
//   RAW  tile u + 1 is first read in interval 2 NP (u + 1). NS >= 3: every wave waits for its own requests of that tile
//        (`vmcnt(L * (NS - 2))`: only the tile issued during tile u may still be in flight) at the end of its LAST memory part of
//        tile u -- intervals 2 NP (u + 1) - 2 / - 1 -- and a barrier follows either. NS = 2 (256 x 256: two 64 KB stages): the tile
//        issued during tile u IS tile u + 1; group 1 issues its whole share in its first memory part and waits at the end of its
//        last one, group 0 spreads its share and waits at the end of its last MFMA part (interval 2 NP (u + 1) - 1).
// Everything else -- implicit im2col (pixel offsets + tap masks per row, wave-uniform tap / channel state per K-tile), zero-block
// redirect of out-of-range chunks, XCD-aware block map, split-K slabs, the epilogues incl. staged stores and GroupNorm statistics --
// is shared with the other pipes (igemm_device.h).
//
// Replaces, like the other pipes: sfast::cudnn_convolution_bias[_add] (/root/reference/src/sfast/csrc/operators/cudnn/
// cudnn_convolution_impl.cc:947-956, :995-998), sfast::cublas_lowp_linear[_add] (csrc/operators/cublas/cublas_gemm.cpp:798-948) and
// sfast::cutlass_linear_geglu (csrc/operators/cutlass/cutlass_dual_linear_kernel.cu:196-208: a 128 x 64 x 32 threadblock tile there).
#pragma once
#include <type_traits>

#include "igemm_device.h"

namespace sfast {

typedef const u32x4 __attribute__((address_space(1))) * pp_src_t;
typedef __attribute__((address_space(3))) void *pp_dst_t;

template <int N> __device__ __forceinline__ void pp_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN_, int NS_, bool GEGLU_> struct PPShape {
    static constexpr int BM = 256, BN = BN_, NS = NS_;
    static constexpr bool SPLIT_M = BN >= 256;        // groups split the pixels (2 x 4 waves) instead of the weight rows (4 x 2)
    static constexpr int NF = BN / 32;                // 32-row weight fragments of the tile
    static constexpr int FM = SPLIT_M ? 4 : 2;        // pixel fragments per wave
    static constexpr int FN0 = SPLIT_M ? NF / 4 : (NF + 1) / 2;
    static constexpr int FN1 = SPLIT_M ? NF / 4 : NF / 2;
    static constexpr int STAGE = (BM + BN) * 128;
    static constexpr int SMEM = NS * STAGE;
    static constexpr int BNO = GEGLU_ ? BN / 2 : BN;
    static constexpr int WF = BN / 64;                // whole 64-row weight passes (512 threads x 16 bytes)
    static constexpr bool TAIL = (BN % 64) != 0;      // + a 32-row pass issued by group 1 alone
    static_assert(BN % 32 == 0 && (BN % 64 == 0 || BN % 64 == 32), "weight rows per tile");
    static_assert(!SPLIT_M || NF % 4 == 0, "2 x 4 waves need BN % 128 == 0");
    static_assert(!GEGLU_ || (FN0 == FN1 && FN0 % 2 == 0), "GEGLU needs paired fragments in every wave");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

// LDS-DMA requests of group G per K-tile, and how many of them the memory part of phase `ph` issues.
template <typename S, int G, int KSP> struct PPIssue {
    static constexpr int NP = 4 / KSP;  // phases per K-tile
    static constexpr int L = 4 + S::WF + ((S::TAIL && G == 1) ? 1 : 0);
    static constexpr int count(int ph) {
        if (S::NS == 2 && G == 1) return ph == 0 ? L : 0;
        return L / NP + (ph < (L % NP) ? 1 : 0);
    }
    static constexpr int first(int ph) {
        int f = 0;
        for (int k = 0; k < ph; ++k) f += count(k);
        return f;
    }
    static_assert(KSP == 1 || KSP == 2 || KSP == 4, "k-steps per phase");
    static_assert(S::NS >= 3 || NP >= 2, "a two-stage ring needs two phases per tile (a request cannot be waited for where it is issued)");
    static_assert(L * (S::NS - 1) <= 63, "vmcnt field");
};

// EXP != 0: timing-only experiment instantiations (probe build, tools/pp_loop_probe.py; results are garbage): bit 0 no MFMAs, bit 1 no
// fragment reads, bit 2 no LDS-DMA requests inside the loop, bit 3 (results CORRECT) no s_setprio.
template <typename T, int BN, int NS, int KSP, int MODE, bool GEGLU, bool STAGED, int G, int EXP = 0>
__device__ __forceinline__ void pp_group(const IgemmArgs &a, char *smem, const BlockTile &bt, const int tid, const int wave) {
    using S = PPShape<BN, NS, GEGLU>;
    using I = PPIssue<S, G, KSP>;
    using vec8 = typename Elem<T>::vec8;
    constexpr int BM = S::BM, FM = S::FM, FN = G ? S::FN1 : S::FN0, STAGE = S::STAGE, BNO = S::BNO, WF = S::WF, L = I::L;
    constexpr int WNB = FN * 32;  // weight rows of this wave
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wq = wave & 3;      // position inside the group
    const int xr0 = S::SPLIT_M ? G * 128 : wq * 64;                // first tile row (pixel) of this wave
    const int wr0 = S::SPLIT_M ? wq * WNB : (G ? S::FN0 * 32 : 0);  // first weight row of this wave
    const int m0 = bt.tile_m * BM, n0 = bt.tile_n * BNO;
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);

    // ---- staging role of this thread (as igemm_glds.hip): tile row rbase + 64 * pass, PHYSICAL chunk tid & 7, which holds LOGICAL
    // chunk kc (source-side swizzle; the pass stride of 64 rows keeps (row >> 1) & 7 independent of the pass).
    // A request costs its issuing wave a memory part of 128 .. 256 clocks minus the fragment reads, so everything lane-dependent is
    // folded into per-row values here and the per-request work is 1 VALU op (linear operands: 64-bit row pointer + wave-uniform byte
    // offset) or 4 (conv activations: byte offset + tap validity -> the offset of a raw buffer load, whose range check returns the
    // zeros of a border tap). Rows past M / N are CLAMPED to the last valid row instead of zero-filled: they only feed output rows /
    // columns the epilogue drops. K % 64 == 0 is a host-side condition of this pipe (no K tail).
    const int rbase = tid >> 3;
    const int kc = (tid & 7) ^ ((rbase >> 1) & 7);
    const char *xptr[4];  // MODE 0: row pointer at chunk kc
    int xoffB[4], xdAB[4];  // MODE 1: BYTE offset of (tap (0,0), chunk kc) in source 2; (the same in source 1) - xoffB
    unsigned xmask[4];      // MODE 1: bit (r * KW + s) set when that tap is inside the image
    __amdgpu_buffer_rsrc_t rsrc1, rsrc2;
    {
        const PixelDecoder decode(a);
        unsigned rep_all = 0;
        if (MODE == 1) {
            for (int r = 0; r < a.KH; ++r) rep_all |= 1u << (r * a.KW);
            int b_last, ho_, wo_;
            decode(a.M - 1, b_last, ho_, wo_);
            const unsigned pixels = (unsigned)((b_last + 1) * a.H * a.W);
            rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x), 0, pixels * (unsigned)a.C1 * 2u, 0x00020000);
            rsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x2 ? a.x2 : a.x), 0, pixels * (unsigned)(a.x2 ? a.C2 : a.C1) * 2u, 0x00020000);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m0 + rbase + i * 64, a.M - 1);
            if (MODE == 0) {
                xptr[i] = (const char *)a.x + ((int64_t)m * a.ldx + kc * 8) * 2;
            } else {
                unsigned mask = 0;
                int b, ho, wo;
                decode(m, b, ho, wo);
                const int h0 = ho * a.stride_h - a.pad_h, w0 = wo * a.stride_w - a.pad_w;
                const int pix = (b * a.H + h0) * a.W + w0;
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
                xoffB[i] = (pix * a.C2 + kc * 8) * 2;
                xdAB[i] = pix * (a.C1 - a.C2) * 2;
                xmask[i] = mask;
            }
        }
    }
    constexpr int WCH = WF + ((S::TAIL && G == 1) ? 1 : 0);
    const char *wptr[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int j = i < WF ? rbase + i * 64 : WF * 64 + rbase - 32;  // tail pass: threads 256 .. 511 -> rows WF * 64 .. + 31
        if (GEGLU) {
            constexpr int GW = S::SPLIT_M ? WNB : 64;  // weight rows per h / g pairing block (= one wave's rows)
            const int grp = j / GW, within = j % GW;
            const int half = within / (GW / 2), i2 = within % (GW / 2);
            const int ncol = min(n0 + grp * (GW / 2) + i2, a.N - 1);
            wptr[i] = (const char *)(half ? a.w[1] : a.w[0]) + ((int64_t)ncol * a.ldw + kc * 8) * 2;
        } else {
            const int n = min(n0 + j, a.N - 1);
            const int rs = a.rows_per_seg;
            const int seg = (n >= rs) + (n - rs >= rs) + (n - rs - rs >= rs);
            const void *base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : seg == 2 ? a.w[2] : a.w[3];
            wptr[i] = (const char *)base + ((int64_t)(n - seg * rs) * a.ldw + kc * 8) * 2;
        }
    }

    // ---- wave-uniform state of the tile being issued (tiles past the split's range request its last tile again: constant vmcnt
    // bookkeeping, nobody reads those stages) ---------------------------------------------------------------------------------------
    // Conv K-tiles are visited CHANNEL-SLICE major (all taps of input channels [c, c + 64), then the next slice), not in the order of
    // the weight's K axis (tap major): the nine taps of a slice re-read one 256-pixel x 128-byte patch (+ halo), ~40 KB per
    // workgroup, which stays in the XCD's L2 between taps; tap major re-reads the whole 200 KB patch of all channels per tap and 32
    // workgroups per XCD push each other's patches out of the 4 MB L2 (measured: the in-loop requests alone ran at 25 B/clk/CU).
    // The sum over K is the same set of products; only the fp32 summation order differs from the other pipes.
    const int cin = a.C1 + a.C2, ntaps = a.KH * a.KW;
    int t_tap = 0, t_r = 0, t_s = 0, t_c = 0;
    if (MODE == 1) {
        const int cs = kt_begin / ntaps;
        t_tap = kt_begin - cs * ntaps;
        t_c = cs * 64;
        t_r = t_tap / a.KW;
        t_s = t_tap - t_r * a.KW;
    }
    int issued = kt_begin;
    char *istage = smem;   // stage the tile being issued goes to
    int64_t kb = 0;        // byte offset of the tile along K (weights, linear activations)
    int tapoff = 0, fmask = 0, tapsh = 0;
    bool first = true;
    auto tile_state = [&]() __attribute__((always_inline)) {
        kb = MODE == 1 ? (int64_t)(t_tap * cin + t_c) * 2 : (int64_t)issued * 128;
        if (MODE == 1) {
            first = t_c < a.C1;
            const int pixoff = t_r * a.dil_h * a.W + t_s * a.dil_w;
            tapoff = first ? (pixoff * a.C1 + t_c) * 2 : (pixoff * a.C2 + (t_c - a.C1)) * 2;
            fmask = first ? -1 : 0;
            tapsh = 31 - (t_tap & 31);
        }
    };
    tile_state();
    // request l of the tile being issued: l < 4 activation pass l, then the weight passes
    auto issue = [&](int l) __attribute__((always_inline)) {
        if (l < 4) {
            pp_dst_t dst = (pp_dst_t)(istage + l * 8192 + wave * 1024);
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((pp_src_t)(const void *)(xptr[l] + kb), dst, 16, 0, 0);
            } else {
                const int valid = (int)(xmask[l] << tapsh) >> 31;  // -1: the tap is inside the image
                const int voff = (xoffB[l] + (xdAB[l] & fmask) + tapoff) | ~valid;
                if (first)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, dst, 16, voff, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, dst, 16, voff, 0, 0, 0);
            }
        } else {
            const int i = l - 4;
            pp_dst_t dst = (pp_dst_t)(istage + BM * 128 + (i < WF ? i * 8192 + wave * 1024 : WF * 8192 + (wave - 4) * 1024));
            __builtin_amdgcn_global_load_lds((pp_src_t)(const void *)(wptr[i] + kb), dst, 16, 0, 0);
        }
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
        if (issued + 1 < kt_end) {
            ++issued;
            if (MODE == 1) {
                ++t_tap;
                if (++t_s == a.KW) {
                    t_s = 0;
                    ++t_r;
                }
                if (t_tap == ntaps) {
                    t_tap = t_r = t_s = 0;
                    t_c += 64;
                }
            }
            tile_state();
        }
        istage = (istage + STAGE == smem + NS * STAGE) ? smem : istage + STAGE;
    };

    f32x16 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;

    // ---- prologue: tiles 0 .. NS - 2 requested in full, tile 0 landed -------------------------------------------------------------
    trace_mark(a, 1);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
        for (int l = 0; l < L; ++l) issue(l);
        issue_advance();
    }
    trace_mark(a, 2);
    pp_wait_vmcnt<L *(NS - 2)>();
    __builtin_amdgcn_s_barrier();
    if (G == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0
    trace_mark(a, 3);

    // fragment addresses: row = const + l31 with const % 32 == 0, so the swizzle term (row >> 1) & 7 is a per-lane constant
    const int swz = (l31 >> 1) & 7;
    const int xoff = (xr0 + l31) * 128, woff = BM * 128 + (wr0 + l31) * 128;
    const char *cstage = smem;
    vec8 af[KSP][FN], bf[KSP][FM];
    if constexpr ((EXP & 2) != 0) {
#pragma unroll
        for (int q = 0; q < KSP; ++q) {
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) af[q][fn] = vec8{};
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) bf[q][fm] = vec8{};
        }
    }
    // EXP bit 4 (results CORRECT): a timeline of K-tile kt_begin + 8 -- lane 0 of waves 0 and 4 stamps the shader clock at seven points
    // of the phase into the trace buffer (behind the per-workgroup records: tools/pp_loop_probe.py --timeline)
    int cur_kt = 0;
    auto stamp = [&](int point) __attribute__((always_inline)) {
        if constexpr ((EXP & 16) != 0) {
            if (cur_kt == kt_begin + 8 && a.trace != nullptr && (tid & 255) == 0)
                a.trace[16 * 32768 + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + G) * 16 + point] = clock64();
        }
    };
    auto phase = [&](auto ph_tag) __attribute__((always_inline)) {
        constexpr int ph = decltype(ph_tag)::value;
        constexpr int cnt = (EXP & 4) ? 0 : I::count(ph), first = I::first(ph);
        constexpr bool last = ph == I::NP - 1;
        // ---------------- memory part: the requests are spread between the k-steps' fragment reads ----------------
        if (ph == 0) stamp(0);
#pragma unroll
        for (int q = 0; q < KSP; ++q) {
            const int coff = (((ph * KSP + q) * 2 + hi) ^ swz) << 4;
            if constexpr ((EXP & 2) == 0) {
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) af[q][fn] = *reinterpret_cast<const vec8 *>(cstage + woff + fn * 4096 + coff);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) bf[q][fm] = *reinterpret_cast<const vec8 *>(cstage + xoff + fm * 4096 + coff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j * KSP + q < cnt) issue(first + j * KSP + q);  // request r of this phase goes behind the reads of k-step r % KSP
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (cnt > 0 && first + cnt == L) issue_advance();
        if (ph == 0) stamp(1);
        if constexpr (last && (NS >= 3 || G == 1)) pp_wait_vmcnt<((EXP & 4) ? 0 : L *(NS - 2))>();  // this wave's share of the next tile has landed
        if (ph == 0) stamp(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and its fragments of this phase are in registers: the stage is released
        if (ph == 0) stamp(3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (ph == 0) stamp(4);
        // ---------------- MFMA part ----------------
        if constexpr ((EXP & 8) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < KSP; ++q)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    if constexpr ((EXP & 1) == 0)
                        acc[fn][fm] = mfma32(af[q][fn], bf[q][fm], acc[fn][fm]);
                    else
                        asm volatile("" ::"v"(af[q][fn]), "v"(bf[q][fm]));  // the fragment reads stay
                }
        if constexpr ((EXP & 8) == 0) __builtin_amdgcn_s_setprio(0);
        if (ph == 0) stamp(5);
        if constexpr (last && NS == 2 && G == 0) pp_wait_vmcnt<0>();  // two-stage ring: group 0's share of the next tile
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (ph == 0) stamp(6);
    };
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        cur_kt = kt;
        phase(std::integral_constant<int, 0>{});
        if constexpr (I::NP > 1) phase(std::integral_constant<int, 1>{});
        if constexpr (I::NP > 2) {
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
        }
        cstage = (cstage + STAGE == smem + NS * STAGE) ? smem : cstage + STAGE;
    }
    if (G == 0) __builtin_amdgcn_s_barrier();  // the barrier group 1 spent at the start
    pp_wait_vmcnt<0>();                         // the zero-filled tail requests have landed before the LDS is handed to the epilogue
    trace_mark(a, 4);
    // GEGLU runs the EPI_EARLY form of the shared epilogue (the only one that pairs h / g fragments); its two bias vectors are fetched
    // here rather than ahead of the K loop (16 registers for ~0.3 us of a >= 10 us workgroup)
    EpiOperands<(GEGLU ? FN / 2 : 1), (GEGLU ? FM : 1)> epi;
    if constexpr (GEGLU) epilogue_prefetch<T, FN, FM, true>(a, epi, m0 + xr0, n0 + wr0 / 2, l31, hi);
    run_epilogue<T, BM, BNO, FN, FM, GEGLU, GEGLU, 512, STAGED>(a, acc, epi, smem, m0, n0, m0 + xr0, n0 + (GEGLU ? wr0 / 2 : wr0), l31, hi, tid,
                                                                 bt.split);
}

template <typename T, int BN, int NS, int KSP, int MODE, bool GEGLU, bool STAGED, int EXP = 0>
__global__ void __launch_bounds__(512, 2) igemm_pp_kernel(const IgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    touch_args(a);
    if (MODE == 1) touch_conv_args(a);
    trace_mark(a, 0);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const BlockTile bt = decode_block(a);
    if (bt.tile_m < 0) return;  // surplus block of an xmap = 2 grid (wave-uniform, before any barrier)
    if (wave < 4)
        pp_group<T, BN, NS, KSP, MODE, GEGLU, STAGED, 0, EXP>(a, smem, bt, tid, wave);
    else
        pp_group<T, BN, NS, KSP, MODE, GEGLU, STAGED, 1, EXP>(a, smem, bt, tid, wave);
    trace_finish(a);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// variant ids 51.. (igemm.hip kVariants, pipe 5): 256 pixels x BN weight rows, ring depth NS
// (tile columns, ring depth, k-steps per phase)
#define SFAST_FOR_PP_VARIANTS(T, MODE, OP) \
    OP(T, 128, 3, 4, MODE, false)          \
    OP(T, 160, 3, 4, MODE, false)          \
    OP(T, 256, 2, 2, MODE, false)

#define SFAST_FOR_PP_GEGLU_VARIANTS(T, OP) OP(T, 256, 2, 2, 0, true)

constexpr int pp_smem_bytes(int BN, int NS, bool geglu, bool staged) {
    const int ring = NS * (256 + BN) * 128;
    const int bno = geglu ? BN / 2 : BN;
    const int stage = 256 * (bno * 2 + 8) + 16 + 512 * 16;  // staged tile + a float4 per flush thread
    return (staged && stage > ring) ? stage : ring;
}

template <typename T, int BN, int NS, int KSP, int MODE, bool GEGLU, bool STAGED> static int pp_set_attr() {
    auto kern = igemm_pp_kernel<T, BN, NS, KSP, MODE, GEGLU, STAGED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem_bytes(BN, NS, GEGLU, STAGED));
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(igemm_pp 256x%d): %s", BN, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

template <typename T, int MODE> static int pp_init_tm() {
    int rc = 0;
#define INIT_OP(TT, BN, NS, KSP, MODE_, G)                                 \
    if (!rc) rc = pp_set_attr<TT, BN, NS, KSP, MODE_, G, false>();         \
    if (!rc && !G) rc = pp_set_attr<TT, BN, NS, KSP, MODE_, false, true>();
    SFAST_FOR_PP_VARIANTS(T, MODE, INIT_OP)
    if constexpr (MODE == 0) {
        SFAST_FOR_PP_GEGLU_VARIANTS(T, INIT_OP)
    }
#undef INIT_OP
    return rc;
}

extern int g_igemm_exp;  // igemm_glds.hip (SFAST_IGEMM_EXP, latched by sfast_hip_set_trace)

template <typename T, int MODE> static int pp_dispatch(const IgemmArgs &a, int BN_, bool geglu, hipStream_t st) {
#ifdef SFAST_PROBES  // timing-only instantiations (results are garbage): probe build only (build.py --probes)
    if constexpr (std::is_same<T, f16>::value && MODE == 1) {
        if (g_igemm_exp != 0 && !geglu && !a.stage_out) {
#define LAUNCH_EXP(BN, NS, KSP, E)                                                                                                         \
    if (BN_ == BN && g_igemm_exp == E) {                                                                                               \
        auto kern = igemm_pp_kernel<f16, BN, NS, KSP, 1, false, false, E>;                                                                  \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem_bytes(BN, NS, false, false)); \
        hipLaunchKernelGGL(kern, igemm_grid(a), dim3(512), pp_smem_bytes(BN, NS, false, false), st, a);                                \
        return check_launch("igemm_pp_exp");                                                                                           \
    }
            LAUNCH_EXP(160, 3, 4, 1) LAUNCH_EXP(160, 3, 4, 2) LAUNCH_EXP(160, 3, 4, 3) LAUNCH_EXP(160, 3, 4, 4) LAUNCH_EXP(160, 3, 4, 5) LAUNCH_EXP(160, 3, 4, 6)
            LAUNCH_EXP(160, 3, 4, 7) LAUNCH_EXP(160, 3, 4, 8) LAUNCH_EXP(160, 3, 4, 16)
            LAUNCH_EXP(256, 2, 2, 1) LAUNCH_EXP(256, 2, 2, 2) LAUNCH_EXP(256, 2, 2, 3) LAUNCH_EXP(256, 2, 2, 4) LAUNCH_EXP(256, 2, 2, 5) LAUNCH_EXP(256, 2, 2, 6)
            LAUNCH_EXP(256, 2, 2, 7) LAUNCH_EXP(256, 2, 2, 8) LAUNCH_EXP(256, 2, 2, 16)
#undef LAUNCH_EXP
        }
    }
#endif
#define LAUNCH_OP(TT, BN, NS, KSP, MODE_, G)                                                                                           \
    if (BN_ == BN && geglu == G) {                                                                                                  \
        if constexpr (!G) {                                                                                                         \
            if (a.stage_out) {                                                                                                      \
                hipLaunchKernelGGL((igemm_pp_kernel<TT, BN, NS, KSP, MODE_, false, true>), igemm_grid(a), dim3(512), pp_smem_bytes(BN, NS, false, true), st, a); \
                return check_launch("igemm_pp_staged");                                                                             \
            }                                                                                                                       \
        }                                                                                                                           \
        hipLaunchKernelGGL((igemm_pp_kernel<TT, BN, NS, KSP, MODE_, G, false>), igemm_grid(a), dim3(512), pp_smem_bytes(BN, NS, G, false), st, a); \
        return check_launch("igemm_pp");                                                                                            \
    }
    SFAST_FOR_PP_VARIANTS(T, MODE, LAUNCH_OP)
    if constexpr (MODE == 0) {
        SFAST_FOR_PP_GEGLU_VARIANTS(T, LAUNCH_OP)
    }
#undef LAUNCH_OP
    set_error("igemm_pp: no kernel for tile 256x%d%s", BN_, geglu ? " (GEGLU)" : "");
    return SFAST_ERR_UNSUPPORTED;
}

// One translation unit per (dtype, mode): igemm_pp_f16_lin.hip ...
#define SFAST_PP_UNIT(T, MODE, TAG)                                                                                              \
    namespace sfast {                                                                                                            \
    int igemm_pp_init_##TAG() { return pp_init_tm<T, MODE>(); }                                                                  \
    int igemm_pp_launch_##TAG(const IgemmArgs &a, int BN, bool geglu, hipStream_t st) { return pp_dispatch<T, MODE>(a, BN, geglu, st); } \
    }

}  // namespace sfast
