// Pipe 5 of the MFMA implicit GEMM (round 6): 256-row tiles, 8 waves in two groups that alternate between a MEMORY phase and an
// MFMA phase ("ping-pong": each SIMD hosts one wave of either group, so its matrix pipe always has a wave inside an MFMA phase while
// the other wave of that SIMD reads fragments and issues the LDS-DMA requests of a later K-tile).
//
// Why (measured, DESIGN.md section 9 rounds 3-5 + round 6): the 64..128-row tiles of pipes 0-4 sit at 20-35 % of the dense MFMA peak
// on every large-M problem. Per K-tile a workgroup pulls (BM + BN) * 128 bytes through its CU's L2 -> LDS path (~56-64 B/clk) for
// BM * BN * 64 MACs: 128 x 128 needs 585 clocks of transfer for 512 clocks of MFMA issue, 256 x 160 needs 930 for 1280, 256 x 256
// 1170 for 2048 -- only the 256-row tiles leave the matrix pipe something to hide the transfer behind. The schedule follows the
// 8-phase template of the platform guide (cdna_hip_programming.md "The 256^2 8-phase template", T3+T4+T5): counted `s_waitcnt vmcnt`
// (never 0 in the steady state), raw `s_barrier`, LDS XOR swizzle with the inverse
// permutation applied on the LDS-DMA SOURCE address. The template's `s_setprio 1` around the
// MFMA clusters is NOT in the product schedule: raising the priority of either part measured 0 - 1 % slower (EXP bit 3, below;
// profiles/r06_pp_loop_probe_run8_timeline.log).
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
// PW = 4 (round 6, second form): the requests move to FOUR PRODUCER WAVES (waves 8 - 11, one per SIMD) that do nothing else, the memory
// part of a consumer group is fragment reads only. Measured reason (profiles/r06_pp_loop_probe_run8_timeline.log): with the
// consumers issuing, a memory part took 1500 - 1900 clocks (16 - 20 ds_read_b128 + 6 - 7 requests) against MFMA parts of 512 - 768:
// a request blocks its wave until the CU's address path has taken it (~25 clocks per 1 KiB request with four waves queueing,
// tools/micro/ldsdma_rate.hip: 45 - 50 B/clk/CU chip-wide), the reads behind it in program order wait, and the sum of both -- not the
// larger -- is what the other group's MFMA part has to cover. With producers the address path runs beside BOTH consumer parts.
// The producers take part in every barrier (one per interval); tile u + NS - 1 is issued over the 2 NP intervals of tile u and
// `vmcnt` is waited at the end of the last one. 12 waves = three per SIMD: 168 registers per lane, so the 96-register
// accumulator of the 160-wide tile leaves room for KSP = 2 fragment sets only.
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

template <int BN_, int NS_, bool GEGLU_, int PW_> struct PPShape {
    static constexpr int BM = 256, BN = BN_, NS = NS_, PW = PW_;
    static constexpr bool SPLIT_M = BN >= 256;        // groups split the pixels (2 x 4 waves) instead of the weight rows (4 x 2)
    static constexpr int NF = BN / 32;                // 32-row weight fragments of the tile
    static constexpr int FM = SPLIT_M ? 4 : 2;        // pixel fragments per wave
    static constexpr int FN0 = SPLIT_M ? NF / 4 : (NF + 1) / 2;
    static constexpr int FN1 = SPLIT_M ? NF / 4 : NF / 2;
    static constexpr int STAGE = (BM + BN) * 128;
    static constexpr int SMEM = NS * STAGE;
    static constexpr int BNO = GEGLU_ ? BN / 2 : BN;
    static constexpr int THREADS = 512 + PW * 64;
    static constexpr int NIW = PW ? PW : 8;           // issuing waves
    static constexpr int RPP = NIW * 8;               // tile rows one request pass of all issuing waves covers (16 bytes per lane)
    static constexpr int XP = BM / RPP;               // activation passes
    static constexpr int WF = BN / RPP;               // whole weight passes
    static constexpr bool TAIL = (BN % RPP) != 0;     // + half a pass issued by the upper half of the issuing waves alone
    static_assert(BN % 32 == 0 && (BN % RPP == 0 || BN % RPP == RPP / 2), "weight rows per tile");
    static_assert(RPP % 16 == 0, "the swizzle phase (row >> 1) & 7 must not depend on the pass");
    static_assert(!SPLIT_M || NF % 4 == 0, "2 x 4 waves need BN % 128 == 0");
    static_assert(!GEGLU_ || (FN0 == FN1 && FN0 % 2 == 0), "GEGLU needs paired fragments in every wave");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

// ---- LDS-DMA requests of one issuing thread -------------------------------------------------------------------------------------
// Staging role (as igemm_glds.hip): tile row rbase + RPP * pass, PHYSICAL chunk itid & 7, which holds LOGICAL chunk kc (source-side
// swizzle). Everything lane-dependent is folded into per-row values in init(), so a request is 1 VALU op (linear operands: 64-bit row
// pointer + wave-uniform byte offset) or 4 (conv activations: byte offset + tap validity -> the offset of a raw buffer load, whose
// range check returns the zeros of a border tap). Rows past M / N are CLAMPED to the last valid row instead of zero-filled: they only
// feed output rows / columns the epilogue drops. K % 64 == 0 is a host-side condition of this pipe (no K tail). Tiles past the
// split's range request its last tile again (constant vmcnt bookkeeping; nobody reads those stages).
// Conv K-tiles are visited CHANNEL-SLICE major (all taps of input channels [c, c + 64), then the next slice), not in the order of the
// weight's K axis: the taps of a slice re-read one 256-pixel x 128-byte patch (+ halo). The sum over K is the same set of products;
// only the fp32 summation order differs from the other pipes.
// tile row m of a conv: pixel index of tap (0, 0) (may lie outside the image) and the mask of the taps inside it
__device__ __forceinline__ void pp_decode_row(const IgemmArgs &a, const PixelDecoder &decode, unsigned rep_all, int m, int &pix, unsigned &mask) {
    mask = 0;
    int b, ho, wo;
    decode(m, b, ho, wo);
    const int h0 = ho * a.stride_h - a.pad_h, w0 = wo * a.stride_w - a.pad_w;
    // fused nearest-2x upsample (12-wave forms only): taps address the UPSAMPLED image [2H, 2W], tap (r, s) of this row reads source pixel
    // ((h0 + r) >> 1, (w0 + s) >> 1) = (h0 >> 1, w0 >> 1) + (((h0 & 1) + r) >> 1, ((w0 & 1) + s) >> 1): `pix` is the first term, the two
    // parities ride in bits 30 / 31 of the mask (KH * KW <= 30) and the loader adds the second term per K-tile (PPLoader::tile_state)
    const int Hin = a.ups ? 2 * a.H : a.H, Win = a.ups ? 2 * a.W : a.W;
    pix = a.ups ? (b * a.H + (h0 >> 1)) * a.W + (w0 >> 1) : (b * a.H + h0) * a.W + w0;
    if (a.dil_h == 1 && a.dil_w == 1) {
        const int s_lo = max(0, -w0), s_hi = min(a.KW, Win - w0);
        const int r_lo = max(0, -h0), r_hi = min(a.KH, Hin - h0);
        if (s_hi > s_lo && r_hi > r_lo) {
            const unsigned cols = ((1u << s_hi) - 1u) & ~((1u << s_lo) - 1u);
            const unsigned lo_bits = r_lo * a.KW, hi_bits = r_hi * a.KW;
            const unsigned upto = hi_bits >= 32 ? 0xffffffffu : ((1u << hi_bits) - 1u);
            mask = cols * (rep_all & upto & ~((1u << lo_bits) - 1u));
        }
    } else {
        unsigned cols = 0;
        for (int s = 0; s < a.KW; ++s) cols |= ((unsigned)(w0 + s * a.dil_w) < (unsigned)Win ? 1u : 0u) << s;
        for (int r = 0; r < a.KH; ++r) mask |= ((unsigned)(h0 + r * a.dil_h) < (unsigned)Hin ? cols : 0u) << (r * a.KW);
    }
    if (a.ups) mask = (mask & 0x3fffffffu) | ((unsigned)(h0 & 1) << 30) | ((unsigned)(w0 & 1) << 31);
}
__device__ __forceinline__ unsigned pp_rep_all(const IgemmArgs &a) {
    unsigned rep_all = 0;
    for (int r = 0; r < a.KH; ++r) rep_all |= 1u << (r * a.KW);
    return rep_all;
}

template <typename T, typename S, int MODE, bool GEGLU, bool UPPER> struct PPLoader {
    static constexpr int XP = S::XP, WF = S::WF, WCH = S::WF + ((S::TAIL && UPPER) ? 1 : 0), L = XP + WCH;
    const char *xptr[XP];    // MODE 0: row pointer at chunk kc
    int xoffB[XP], xdAB[XP];  // MODE 1: BYTE offset of (tap (0,0), chunk kc) in source 2; (the same in source 1) - xoffB
    unsigned xmask[XP];      // MODE 1: bit (r * KW + s) set when that tap is inside the image
    static constexpr bool UPS = S::PW > 0 && MODE == 1;  // the producer waves take convs with a fused nearest-2x upsample (registers to spare)
    int xph[UPS ? XP : 1], xpw[UPS ? XP : 1];  // ups: byte step of one source row / one source pixel where the row's parity is odd, else 0
    int mh, mw;                                 // ups: -1 where this K-tile's tap row / column index is odd (the parity term applies)
    const char *wptr[WCH];
    __amdgpu_buffer_rsrc_t rsrc1, rsrc2;
    int cin, ntaps, t_tap, t_r, t_s, t_c, issued, kt_end, tapoff, fmask, tapsh, iwave;
    bool first;
    int64_t kb;
    char *istage, *smem;

    __device__ __forceinline__ void tile_state(const IgemmArgs &a) {
        kb = MODE == 1 ? (int64_t)(t_tap * cin + t_c) * 2 : (int64_t)issued * 128;
        if (MODE == 1) {
            first = t_c < a.C1;
            const int pixoff = t_r * a.dil_h * a.W + t_s * a.dil_w;
            tapoff = first ? (pixoff * a.C1 + t_c) * 2 : (pixoff * a.C2 + (t_c - a.C1)) * 2;
            fmask = first ? -1 : 0;
            tapsh = 31 - (t_tap & 31);
            if (UPS && a.ups) {  // (single source, no dilation: host-side conditions)
                tapoff = (((t_r >> 1) * a.W + (t_s >> 1)) * a.C1 + t_c) * 2;
                mh = (t_r & 1) ? -1 : 0;
                mw = (t_s & 1) ? -1 : 0;
            }
        }
    }
    // itid: index among the issuing threads, iwave: its wave among the issuing waves; meta: per-tile-row {pixel index, tap mask} in LDS, or
    // nullptr (every issuing thread decodes its own rows)
    __device__ __forceinline__ void init(const IgemmArgs &a, char *smem_, int itid, int iwave_, int m0, int n0, int kt_begin, int kt_end_,
                                         const u32x2 *meta = nullptr) {
        smem = istage = smem_;
        iwave = iwave_;
        kt_end = kt_end_;
        const int rbase = itid >> 3;
        const int kc = (itid & 7) ^ ((rbase >> 1) & 7);
        {
            const PixelDecoder decode(a);
            unsigned rep_all = 0;
            if (MODE == 1) {
                if (meta == nullptr) rep_all = pp_rep_all(a);
                int b_last, ho_, wo_;
                decode(a.M - 1, b_last, ho_, wo_);
                const unsigned pixels = (unsigned)((b_last + 1) * a.H * a.W);
                rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x), 0, pixels * (unsigned)a.C1 * 2u, 0x00020000);
                rsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x2 ? a.x2 : a.x), 0, pixels * (unsigned)(a.x2 ? a.C2 : a.C1) * 2u, 0x00020000);
            }
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int m = min(m0 + rbase + i * S::RPP, a.M - 1);
                if (MODE == 0) {
                    xptr[i] = (const char *)a.x + ((int64_t)m * a.ldx + kc * 8) * 2;
                } else {
                    int pix;
                    unsigned mask;
                    if (meta != nullptr) {  // decoded by the whole workgroup (pp_decode_rows), one row per thread
                        const u32x2 v = meta[rbase + i * S::RPP];  // (row already clamped to M - 1 by its decoder)
                        pix = (int)v[0];
                        mask = v[1];
                    } else {
                        pp_decode_row(a, decode, rep_all, m, pix, mask);
                    }
                    xoffB[i] = (pix * a.C2 + kc * 8) * 2;
                    xdAB[i] = pix * (a.C1 - a.C2) * 2;
                    xmask[i] = mask;
                    if constexpr (UPS) {
                        xph[i] = (a.ups && (mask & (1u << 30))) ? a.W * a.C1 * 2 : 0;
                        xpw[i] = (a.ups && (mask & (1u << 31))) ? a.C1 * 2 : 0;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int j = i < WF ? rbase + i * S::RPP : WF * S::RPP + rbase - S::RPP / 2;  // tail pass: the upper half of the issuing threads
            if (GEGLU) {
                constexpr int GW = 64;  // weight rows per h / g pairing block (= one wave's rows: 32 h + 32 g)
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
        cin = a.C1 + a.C2;
        ntaps = a.KH * a.KW;
        t_tap = t_r = t_s = t_c = 0;
        if (MODE == 1) {
            const int cs = kt_begin / ntaps;
            t_tap = kt_begin - cs * ntaps;
            t_c = cs * 64;
            t_r = t_tap / a.KW;
            t_s = t_tap - t_r * a.KW;
        }
        issued = kt_begin;
        tapoff = fmask = tapsh = mh = mw = 0;
        first = true;
        tile_state(a);
    }
    // request l of the tile being issued: l < XP activation pass l, then the weight passes (l is a compile-time constant at every call)
    __device__ __forceinline__ void issue(int l) {
        if (l < XP) {
            pp_dst_t dst = (pp_dst_t)(istage + l * (S::RPP * 128) + iwave * 1024);
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((pp_src_t)(const void *)(xptr[l] + kb), dst, 16, 0, 0);
            } else {
                const int valid = (int)(xmask[l] << tapsh) >> 31;  // -1: the tap is inside the image
                int off = xoffB[l] + (xdAB[l] & fmask) + tapoff;
                if constexpr (UPS) off += (xph[l] & mh) + (xpw[l] & mw);
                const int voff = off | ~valid;
                if (first)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, dst, 16, voff, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, dst, 16, voff, 0, 0, 0);
            }
        } else {
            const int i = l - XP;
            pp_dst_t dst = (pp_dst_t)(istage + S::BM * 128 + (i < WF ? i * (S::RPP * 128) + iwave * 1024 : WF * (S::RPP * 128) + (iwave - S::NIW / 2) * 1024));
            __builtin_amdgcn_global_load_lds((pp_src_t)(const void *)(wptr[i] + kb), dst, 16, 0, 0);
        }
    }
    __device__ __forceinline__ void advance(const IgemmArgs &a) {
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
            tile_state(a);
        }
        istage = (istage + S::STAGE == smem + S::NS * S::STAGE) ? smem : istage + S::STAGE;
    }
    // requests first .. first + cnt - 1, spread over `parts` slots of which this is slot `q` (request r goes to slot r % parts)
    template <int first_, int cnt, int parts, int q> __device__ __forceinline__ void issue_slot() {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j * parts + q < cnt) issue(first_ + j * parts + q);
    }
};

// How many requests of a tile the `slot`-th of `NSLOT` issue points of that tile carries. With a two-stage ring (NS == 2) the tile
// being issued is the NEXT tile to be read, so nothing is issued at the point that also waits for it (the last one).
template <int L, int NSLOT, bool TWO_STAGE> struct PPSplit {
    static constexpr int SL = TWO_STAGE ? NSLOT - 1 : NSLOT;
    static constexpr int count(int slot) { return slot >= SL ? 0 : L / SL + (slot < (L % SL) ? 1 : 0); }
    static constexpr int first(int slot) {
        int f = 0;
        for (int k = 0; k < slot; ++k) f += count(k);
        return f;
    }
    static_assert(SL >= 1, "a two-stage ring needs two issue points per tile (a request cannot be waited for where it is issued)");
};

// ---- producer wave (PW > 0) ---------------------------------------------------------------------------------------------------------
template <typename T, typename S, int KSP, int MODE, bool GEGLU, int EXP>
__device__ __forceinline__ void pp_producer(const IgemmArgs &a, char *smem, const BlockTile &bt, const int tid, const int wave, const u32x2 *meta) {
    constexpr int NIV = 2 * (4 / KSP);  // barrier intervals per K-tile
    using LD = PPLoader<T, S, MODE, GEGLU, false>;
    using SP = PPSplit<LD::L, NIV, S::NS == 2>;
    static_assert(!S::TAIL, "producer passes cover the weight rows exactly");
    static_assert(LD::L * (S::NS - 1) <= 63, "vmcnt field");
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);
    LD ld;
    ld.init(a, smem, tid - 512, wave - 8, bt.tile_m * S::BM, bt.tile_n * S::BNO, kt_begin, kt_end, meta);
#pragma unroll
    for (int s = 0; s < S::NS - 1; ++s) {
#pragma unroll
        for (int l = 0; l < LD::L; ++l) ld.issue(l);
        ld.advance(a);
    }
    pp_wait_vmcnt<LD::L *(S::NS - 2)>();
    __builtin_amdgcn_s_barrier();  // tile kt_begin has landed
    auto interval = [&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int cnt = (EXP & 4) ? 0 : SP::count(j), first = SP::first(j);
        ld.template issue_slot<first, cnt, 1, 0>();
        if constexpr (cnt > 0 && first + cnt == LD::L) ld.advance(a);
        if constexpr (j == NIV - 1) pp_wait_vmcnt<((EXP & 4) ? 0 : LD::L *(S::NS - 2))>();  // the next tile has landed (this wave's share)
        __builtin_amdgcn_s_barrier();
    };
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        interval(std::integral_constant<int, 0>{});
        interval(std::integral_constant<int, 1>{});
        if constexpr (NIV > 2) {
            interval(std::integral_constant<int, 2>{});
            interval(std::integral_constant<int, 3>{});
        }
        if constexpr (NIV > 4) {
            interval(std::integral_constant<int, 4>{});
            interval(std::integral_constant<int, 5>{});
            interval(std::integral_constant<int, 6>{});
            interval(std::integral_constant<int, 7>{});
        }
    }
    __builtin_amdgcn_s_barrier();  // the stagger barrier of the consumer groups
    pp_wait_vmcnt<0>();            // the surplus requests have landed before the LDS is handed to the epilogue
}

// ---- consumer wave of group G -------------------------------------------------------------------------------------------------------
// EXP != 0: timing-only experiment instantiations (probe build, tools/pp_loop_probe.py; results are garbage): bit 0 no MFMAs, bit 1 no
// fragment reads, bit 2 no LDS-DMA requests inside the loop, bit 3 (results CORRECT) s_setprio 1 over the MEMORY part (fragment reads + requests), back to 0 for the MFMA part -- measured 0 - 1 %
// slower than no priority change, like the template's opposite arrangement before it --, bit 4 (CORRECT) timeline stamps.
template <typename T, typename S, int KSP, int MODE, bool GEGLU, bool STAGED, int G, int EXP>
__device__ __forceinline__ void pp_consumer(const IgemmArgs &a, char *smem, const BlockTile &bt, const int tid, const int wave) {
    using vec8 = typename Elem<T>::vec8;
    constexpr int BM = S::BM, NS = S::NS, FM = S::FM, FN = G ? S::FN1 : S::FN0, STAGE = S::STAGE, BNO = S::BNO, NP = 4 / KSP;
    constexpr int WNB = FN * 32;  // weight rows of this wave
    constexpr bool ISSUER = S::PW == 0;
    using LD = PPLoader<T, S, MODE, GEGLU, G == 1>;
    using SP = PPSplit<LD::L, NP, NS == 2 && G == 1>;  // two-stage ring: group 1 waits at the end of its last memory part, group 0 a part later
    constexpr int L = LD::L;
    static_assert(!ISSUER || L * (NS - 1) <= 63, "vmcnt field");
    static_assert(KSP == 1 || KSP == 2 || KSP == 4, "k-steps per phase");
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wq = wave & 3;      // position inside the group
    const int xr0 = S::SPLIT_M ? G * 128 : wq * 64;                // first tile row (pixel) of this wave
    const int wr0 = S::SPLIT_M ? wq * WNB : (G ? S::FN0 * 32 : 0);  // first weight row of this wave
    const int m0 = bt.tile_m * BM, n0 = bt.tile_n * BNO;
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);

    LD ld;
    if constexpr (ISSUER) ld.init(a, smem, tid, wave, m0, n0, kt_begin, kt_end);

    f32x16 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;

    // ---- prologue: tiles 0 .. NS - 2 requested in full, tile 0 landed -------------------------------------------------------------
    trace_mark(a, 1);
    if constexpr (ISSUER) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
            for (int l = 0; l < L; ++l) ld.issue(l);
            ld.advance(a);
        }
        trace_mark(a, 2);
        pp_wait_vmcnt<L *(NS - 2)>();
    } else {
        trace_mark(a, 2);
    }
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
    // of the phase into the trace buffer (behind the per-workgroup records: tools/pp_loop_probe.py)
    int cur_kt = 0;
    auto stamp = [&](int point) __attribute__((always_inline)) {
        if constexpr ((EXP & 16) != 0) {
            if (cur_kt == kt_begin + 8 && a.trace != nullptr && (tid & 255) == 0)
                a.trace[16 * 32768 + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + G) * 16 + point] = clock64();
        }
    };
    auto phase = [&](auto ph_tag) __attribute__((always_inline)) {
        constexpr int ph = decltype(ph_tag)::value;
        constexpr int cnt = (!ISSUER || (EXP & 4)) ? 0 : SP::count(ph), first = SP::first(ph);
        constexpr bool last = ph == NP - 1;
        // ---------------- memory part: fragment reads (issuing consumers: the requests spread between the k-steps' reads) --------
        if (ph == 0) stamp(0);
        if constexpr ((EXP & 8) != 0) __builtin_amdgcn_s_setprio(1);
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
            if constexpr (cnt > 0) {
                if (q == 0) ld.template issue_slot<first, cnt, KSP, 0>();
                if (q == 1) ld.template issue_slot<first, cnt, KSP, 1>();
                if (q == 2) ld.template issue_slot<first, cnt, KSP, 2>();
                if (q == 3) ld.template issue_slot<first, cnt, KSP, 3>();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (cnt > 0 && first + cnt == L) ld.advance(a);
        if (ph == 0) stamp(1);
        if constexpr (ISSUER && last && (NS >= 3 || G == 1)) pp_wait_vmcnt<((EXP & 4) ? 0 : L *(NS - 2))>();  // this wave's share of the next tile has landed
        if (ph == 0) stamp(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and its fragments of this phase are in registers: the stage is released
        if (ph == 0) stamp(3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (ph == 0) stamp(4);
        // ---------------- MFMA part ----------------
        if constexpr ((EXP & 8) != 0) __builtin_amdgcn_s_setprio(0);
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
        if (ph == 0) stamp(5);
        if constexpr (ISSUER && last && NS == 2 && G == 0) pp_wait_vmcnt<0>();  // two-stage ring: group 0's share of the next tile
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (ph == 0) stamp(6);
    };
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        cur_kt = kt;
        phase(std::integral_constant<int, 0>{});
        if constexpr (NP > 1) phase(std::integral_constant<int, 1>{});
        if constexpr (NP > 2) {
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
        }
        cstage = (cstage + STAGE == smem + NS * STAGE) ? smem : cstage + STAGE;
    }
    if (G == 0) __builtin_amdgcn_s_barrier();  // the barrier group 1 spent at the start
    if constexpr (ISSUER) pp_wait_vmcnt<0>();  // the surplus requests have landed before the LDS is handed to the epilogue
    trace_mark(a, 4);
    // GEGLU runs the EPI_EARLY form of the shared epilogue (the only one that pairs h / g fragments); its two bias vectors are fetched
    // here rather than ahead of the K loop (16 registers for ~0.3 us of a >= 10 us workgroup). Everything else runs the
    // one-fragment-ahead form: requesting ALL operand vectors of the wave at once (120 registers, possible in the 8-wave form) measured
    // SLOWER (epilogue 7.2 vs 5.8 us per workgroup, profiles/r06_pp_loop_probe_run11.log) -- all workgroups of a launch reach their
    // epilogue together and 2 x 80 KB per workgroup (residual in, tile out) is simply HBM time: 256 CUs x 160 KB in ~6 us = 6.8 TB/s.
    // The 12-wave form (168 registers) fetches and consumes one fragment's operands at a time: two sets in flight spilled 38 - 50 registers.
    EpiOperands<(GEGLU ? FN / 2 : 1), (GEGLU ? FM : 1)> epi;
    if constexpr (GEGLU) epilogue_prefetch<T, FN, FM, true>(a, epi, m0 + xr0, n0 + wr0 / 2, l31, hi);
    run_epilogue<T, BM, BNO, FN, FM, GEGLU, GEGLU, 512, STAGED, kJoinDefault, 2>(a, acc, epi, smem, m0, n0, m0 + xr0, n0 + (GEGLU ? wr0 / 2 : wr0), l31,
                                                                                            hi, tid, bt.split);
}

// ---- KSP = 0: the LOCKSTEP form (round 6, third form) -----------------------------------------------------------------------------------
// Same tile, same wave layout (incl. the uneven 96 + 64 split of the 160-wide tile), same producers, but the eight consumer waves do NOT
// alternate: every wave reads the fragments of k-step q + 1 while its OWN MFMAs of k-step q run (two fragment sets), across the tile
// boundary, and there is ONE barrier per K-tile -- the loop of igemm_glds_ws.hip on a 256-row tile. Why: on gfx950 an LDS read overlaps
// the MFMAs of the SAME wave but barely those of the other wave of its SIMD -- the ablations of the ping-pong loop ADD UP (skeleton 34
// + MFMAs 46 + fragment reads 28 = 108 us against 111 measured with both), and in its timeline the younger group's eight reads take 470
// clocks beside the older group's 384 clocks of MFMAs (profiles/r06_pp_loop_probe_run9.log, _run10.log).
template <typename T, typename S, int MODE, bool GEGLU, int EXP>
__device__ __forceinline__ void pp_producer_ls(const IgemmArgs &a, char *smem, const BlockTile &bt, const int tid, const int wave, const u32x2 *meta) {
    using LD = PPLoader<T, S, MODE, GEGLU, false>;
    constexpr int L = LD::L, NS = S::NS;
    static_assert(!S::TAIL && NS >= 3 && L * (NS - 1) <= 63, "ring");
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);
    LD ld;
    ld.init(a, smem, tid - 512, wave - 8, bt.tile_m * S::BM, bt.tile_n * S::BNO, kt_begin, kt_end, meta);
    auto issue_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < L; ++l) ld.issue(l);
        ld.advance(a);
    };
    // Ring protocol of igemm_glds_ws.hip -- the consumers meet barrier k + 1 with every fragment of tile k in registers, so that barrier
    // releases the stage of tile k and it is refilled with tile k + NS -- with two differences: (a) only NS - 1 tiles are requested ahead
    // of the first barrier (the consumers start as soon as tile 0 has landed; the NS-th tile goes out right behind that barrier instead of
    // queueing 52 KB in front of it), and (b) NO surplus tiles are requested at the end of the K range: the consumers' first epilogue
    // barrier waits until the producer waves have terminated, i.e. until everything they requested has landed -- two surplus tiles were
    // 104 KB of transfers nobody reads, awaited at the top of every workgroup's epilogue. The waits of the last NS - 2 tiles are therefore
    // counted by hand (NS == 3: vmcnt(L) while one more tile is in flight, vmcnt(0) for the last).
    static_assert(NS == 3, "tail waits are written out for a three-stage ring");
    const int ntiles = kt_end - kt_begin;
    issue_tile();  // T0
    issue_tile();  // T1 (a one-tile K range requests its tile twice: the copy lands in a stage nobody reads)
    pp_wait_vmcnt<L>();  // T0 has landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if constexpr ((EXP & 4) == 0) {
        if (2 < ntiles) issue_tile();  // T2
    }
    for (int kt = 1; kt < ntiles; ++kt) {
        if constexpr ((EXP & 4) != 0) {
            pp_wait_vmcnt<0>();
        } else if (kt + 1 < ntiles) {
            pp_wait_vmcnt<L>();  // T_kt has landed; T_kt+1 may be in flight
        } else {
            pp_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();  // the consumers hold all of T_kt-1 in registers: its stage is free
        if constexpr ((EXP & 4) == 0) {
            if (kt + 2 < ntiles) issue_tile();  // T_kt+2 -> that stage
        }
    }
    pp_wait_vmcnt<0>();
}

template <typename T, typename S, int MODE, bool GEGLU, bool STAGED, int G, int EXP>
__device__ __forceinline__ void pp_consumer_ls(const IgemmArgs &a, char *smem, const BlockTile &bt, const int tid, const int wave) {
    using vec8 = typename Elem<T>::vec8;
    constexpr int BM = S::BM, NS = S::NS, FM = S::FM, FN = G ? S::FN1 : S::FN0, STAGE = S::STAGE, BNO = S::BNO;
    constexpr int WNB = FN * 32;
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wq = wave & 3;
    const int xr0 = S::SPLIT_M ? G * 128 : wq * 64;
    const int wr0 = S::SPLIT_M ? wq * WNB : (G ? S::FN0 * 32 : 0);
    const int m0 = bt.tile_m * BM, n0 = bt.tile_n * BNO;
    const int kt_begin = bt.split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_begin + a.ktiles_per_split);
    f32x16 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;
    trace_mark(a, 1);
    trace_mark(a, 2);
    const int swz = (l31 >> 1) & 7;
    const int xoff = (xr0 + l31) * 128, woff = BM * 128 + (wr0 + l31) * 128;
    const char *cstage = smem;
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
    auto read_frags = [&](const char *st, int ks, int set) __attribute__((always_inline)) {
        if constexpr ((EXP & 2) != 0) return;
        const int coff = (((ks * 2) + hi) ^ swz) << 4;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) af[set][fn] = *reinterpret_cast<const vec8 *>(st + woff + fn * 4096 + coff);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) bf[set][fm] = *reinterpret_cast<const vec8 *>(st + xoff + fm * 4096 + coff);
    };
    __builtin_amdgcn_s_barrier();  // tile kt_begin has landed
    trace_mark(a, 3);
    read_frags(cstage, 0, 0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const char *st = cstage;
        cstage = (cstage + STAGE == smem + NS * STAGE) ? smem : cstage + STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
                read_frags(st, ks + 1, (ks + 1) & 1);
            } else if (kt + 1 < kt_end) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of tile kt is in registers
                __builtin_amdgcn_s_barrier();                       // tile kt + 1 has landed, the stage of tile kt is released
                asm volatile("" ::: "memory");
                read_frags(cstage, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // the reads stay above the MFMAs they run beside
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    if constexpr ((EXP & 1) == 0)
                        acc[fn][fm] = mfma32(af[ks & 1][fn], bf[ks & 1][fm], acc[fn][fm]);
                    else
                        asm volatile("" ::"v"(af[ks & 1][fn]), "v"(bf[ks & 1][fm]));
                }
        }
    }
    trace_mark(a, 4);
    EpiOperands<(GEGLU ? FN / 2 : 1), (GEGLU ? FM : 1)> epi;
    if constexpr (GEGLU) epilogue_prefetch<T, FN, FM, true>(a, epi, m0 + xr0, n0 + wr0 / 2, l31, hi);
    run_epilogue<T, BM, BNO, FN, FM, GEGLU, GEGLU, 512, STAGED, kJoinDefault, 2>(a, acc, epi, smem, m0, n0, m0 + xr0, n0 + (GEGLU ? wr0 / 2 : wr0), l31, hi,
                                                                                    tid, bt.split);
}

template <typename T, int BN, int NS, int KSP, int PW, int MODE, bool GEGLU, bool STAGED, int EXP = 0>
__global__ void __launch_bounds__(512 + PW * 64, (PW ? 3 : 2)) igemm_pp_kernel(const IgemmArgs a) {
    using S = PPShape<BN, NS, GEGLU, PW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    touch_args(a);
    if (MODE == 1) touch_conv_args(a);
    trace_mark(a, 0);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const BlockTile bt = decode_block(a);
    if (bt.tile_m < 0) return;  // surplus block of an xmap = 2 grid (wave-uniform, before any barrier)
    if constexpr (PW > 0) {
        // Conv rows are decoded by the WHOLE workgroup, one tile row per thread, and handed to the four producer waves through the
        // stage the prologue requests do not touch: a producer thread owns 8 activation rows, and decoding them serially (two float
        // divisions + the tap mask each) kept the eight consumer waves waiting ~4 us at the first barrier.
        const u32x2 *meta = nullptr;
        if constexpr (MODE == 1) {
            u32x2 *mt = reinterpret_cast<u32x2 *>(smem + (NS - 1) * S::STAGE);
            if (tid < S::BM) {
                const PixelDecoder decode(a);
                int pix;
                unsigned mask;
                pp_decode_row(a, decode, pp_rep_all(a), min(bt.tile_m * S::BM + tid, a.M - 1), pix, mask);
                mt[tid] = u32x2{(unsigned)pix, mask};
            }
            __syncthreads();
            meta = mt;
        }
        if (wave >= 8) {
            if constexpr (KSP == 0)
                pp_producer_ls<T, S, MODE, GEGLU, EXP>(a, smem, bt, tid, wave, meta);
            else
                pp_producer<T, S, KSP, MODE, GEGLU, EXP>(a, smem, bt, tid, wave, meta);
            return;
        }
    }
    if constexpr (KSP == 0) {
        static_assert(PW > 0, "the lockstep form has producer waves");
        if (wave < 4)
            pp_consumer_ls<T, S, MODE, GEGLU, STAGED, 0, EXP>(a, smem, bt, tid, wave);
        else
            pp_consumer_ls<T, S, MODE, GEGLU, STAGED, 1, EXP>(a, smem, bt, tid, wave);
    } else {
        if (wave < 4)
            pp_consumer<T, S, KSP, MODE, GEGLU, STAGED, 0, EXP>(a, smem, bt, tid, wave);
        else
            pp_consumer<T, S, KSP, MODE, GEGLU, STAGED, 1, EXP>(a, smem, bt, tid, wave);
    }
    trace_finish(a);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// variant ids 51.. (igemm.hip kVariants, pipe 5): 256 pixels x BN weight rows; (tile columns, ring depth, k-steps per phase, producers)
#define SFAST_FOR_PP_VARIANTS(T, MODE, OP) \
    OP(T, 128, 3, 4, 0, MODE, false)       \
    OP(T, 160, 3, 4, 0, MODE, false)       \
    OP(T, 256, 2, 2, 0, MODE, false)       \
    OP(T, 128, 3, 2, 4, MODE, false)       \
    OP(T, 160, 3, 2, 4, MODE, false)       \
    OP(T, 128, 3, 0, 4, MODE, false)       \
    OP(T, 160, 3, 0, 4, MODE, false)

#define SFAST_FOR_PP_GEGLU_VARIANTS(T, OP) \
    OP(T, 256, 2, 2, 0, 0, true)           \
    OP(T, 128, 3, 0, 4, 0, true)

constexpr int pp_smem_bytes(int BN, int NS, bool geglu, bool staged) {
    const int ring = NS * (256 + BN) * 128;
    const int bno = geglu ? BN / 2 : BN;
    const int stage = 256 * (bno * 2 + 8) + 16 + 512 * 16;  // staged tile + a float4 per flush thread
    return (staged && stage > ring) ? stage : ring;
}

template <typename T, int BN, int NS, int KSP, int PW, int MODE, bool GEGLU, bool STAGED> static int pp_set_attr() {
    auto kern = igemm_pp_kernel<T, BN, NS, KSP, PW, MODE, GEGLU, STAGED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem_bytes(BN, NS, GEGLU, STAGED));
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(igemm_pp 256x%d): %s", BN, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

template <typename T, int MODE> static int pp_init_tm() {
    int rc = 0;
#define INIT_OP(TT, BN, NS, KSP, PW, MODE_, G)                                  \
    if (!rc) rc = pp_set_attr<TT, BN, NS, KSP, PW, MODE_, G, false>();         \
    if (!rc && !G) rc = pp_set_attr<TT, BN, NS, KSP, PW, MODE_, false, true>();
    SFAST_FOR_PP_VARIANTS(T, MODE, INIT_OP)
    if constexpr (MODE == 0) {
        SFAST_FOR_PP_GEGLU_VARIANTS(T, INIT_OP)
    }
#undef INIT_OP
    return rc;
}

extern int g_igemm_exp;  // igemm_glds.hip (SFAST_IGEMM_EXP, latched by sfast_hip_set_trace)

// pw: producer waves of the variant (0: the consumer groups issue the requests themselves)
template <typename T, int MODE> static int pp_dispatch(const IgemmArgs &a, int BN_, int pw, bool geglu, hipStream_t st) {
    const bool lockstep = pw >= 100;  // pw = 100 + producer waves: the lockstep form (KSP = 0)
    pw %= 100;
#ifdef SFAST_PROBES  // timing-only instantiations (results are garbage): probe build only (build.py --probes)
    if constexpr (std::is_same<T, f16>::value && MODE == 1) {
        if (g_igemm_exp != 0 && !geglu && a.stage_out) {
#define LAUNCH_EXP(BN, NS, KSP, PW, E)                                                                                                 \
    if (BN_ == BN && pw == PW && g_igemm_exp == E && lockstep == (KSP == 0)) {                                                         \
        auto kern = igemm_pp_kernel<f16, BN, NS, KSP, PW, 1, false, true, E>;                                                         \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem_bytes(BN, NS, false, true)); \
        hipLaunchKernelGGL(kern, igemm_grid(a), dim3(512 + PW * 64), pp_smem_bytes(BN, NS, false, true), st, a);                      \
        return check_launch("igemm_pp_exp");                                                                                           \
    }
#define LAUNCH_EXPS(BN, NS, KSP, PW)                                                                                                   \
    LAUNCH_EXP(BN, NS, KSP, PW, 1) LAUNCH_EXP(BN, NS, KSP, PW, 2) LAUNCH_EXP(BN, NS, KSP, PW, 3) LAUNCH_EXP(BN, NS, KSP, PW, 4)      \
    LAUNCH_EXP(BN, NS, KSP, PW, 5) LAUNCH_EXP(BN, NS, KSP, PW, 6) LAUNCH_EXP(BN, NS, KSP, PW, 7) LAUNCH_EXP(BN, NS, KSP, PW, 8)      \
    LAUNCH_EXP(BN, NS, KSP, PW, 16)
            LAUNCH_EXPS(160, 3, 4, 0) LAUNCH_EXPS(160, 3, 2, 4) LAUNCH_EXPS(256, 2, 2, 0) LAUNCH_EXPS(160, 3, 0, 4)
#undef LAUNCH_EXPS
#undef LAUNCH_EXP
        }
    }
#endif
#define LAUNCH_OP(TT, BN, NS, KSP, PW, MODE_, G)                                                                                    \
    if (BN_ == BN && pw == PW && geglu == G && lockstep == (KSP == 0)) {                                                            \
        if constexpr (!G) {                                                                                                         \
            if (a.stage_out) {                                                                                                      \
                hipLaunchKernelGGL((igemm_pp_kernel<TT, BN, NS, KSP, PW, MODE_, false, true>), igemm_grid(a), dim3(512 + PW * 64),  \
                                   pp_smem_bytes(BN, NS, false, true), st, a);                                                      \
                return check_launch("igemm_pp_staged");                                                                             \
            }                                                                                                                       \
        }                                                                                                                           \
        hipLaunchKernelGGL((igemm_pp_kernel<TT, BN, NS, KSP, PW, MODE_, G, false>), igemm_grid(a), dim3(512 + PW * 64),             \
                           pp_smem_bytes(BN, NS, G, false), st, a);                                                                 \
        return check_launch("igemm_pp");                                                                                            \
    }
    SFAST_FOR_PP_VARIANTS(T, MODE, LAUNCH_OP)
    if constexpr (MODE == 0) {
        SFAST_FOR_PP_GEGLU_VARIANTS(T, LAUNCH_OP)
    }
#undef LAUNCH_OP
    set_error("igemm_pp: no kernel for tile 256x%d%s with %d producer waves", BN_, geglu ? " (GEGLU)" : "", pw);
    return SFAST_ERR_UNSUPPORTED;
}

// One translation unit per (dtype, mode): igemm_pp_f16_lin.hip ...
#define SFAST_PP_UNIT(T, MODE, TAG)                                                                                              \
    namespace sfast {                                                                                                            \
    int igemm_pp_init_##TAG() { return pp_init_tm<T, MODE>(); }                                                                  \
    int igemm_pp_launch_##TAG(const IgemmArgs &a, int BN, int pw, bool geglu, hipStream_t st) { return pp_dispatch<T, MODE>(a, BN, pw, geglu, st); } \
    }

}  // namespace sfast
