// GroupNorm(+SiLU) -> 3x3 convolution as ONE weight-streaming launch for the low-resolution levels of the UNet (B*H*W <= 128 pixels:
// SD1.5's 8x8 level at CFG batch 2, where a 1280 -> 1280 conv moves 29.5 MB of weights for 3.8 GFLOP -- 126 flop / byte, HBM-bound).
//
// Replaces, for those layers, the reference's pair  sfast_triton::group_norm_silu  (/root/reference/src/sfast/triton/torch_ops.py:179-189,
// triton/ops/group_norm.py:357-479) -> sfast::cudnn_convolution_bias[_add]  (csrc/operators/cudnn/cudnn_convolution_impl.cc:995-998):
// the reference fuses GroupNorm with SiLU and the conv with its bias / residual; here the normalisation moves INTO the conv because
// at this size the normalised tensor is 0.3 MB and its launch (4.5 - 8 us, profiles/r03_kernels_per_op_run14.json) costs as much as
// streaming a third of the conv's weights.
//
// Structure (measured first: tools/micro/wdirect.hip, profiles/r04_wdirect_probe_run1.log):
//   * work unit = (tile of 32*NB output channels, slice of CS input channels), ALL nine taps, ALL pixels -- every weight element is
//     read from HBM exactly once by exactly one workgroup; units = (Cout / (32 NB)) x (Cin / CS) ~ 320, two workgroups per CU.
//   * the activation slice [pixels][CS] is loaded ONCE into LDS and normalised there: the slice is a whole number of GroupNorm groups
//     (CS % (Cin / G) == 0) and holds every pixel of every sample, so the workgroup computes exact two-pass statistics itself -- no
//     statistics hand-off from the producer, no second kernel. The nine taps then read the same LDS rows shifted by (dy, dx); border
//     taps read one all-zero row.
//   * weights go global -> VGPR in MFMA A-operand layout (lane (r, g): 16 bytes of weight row r at k-group g), D k-steps ahead,
//     through raw buffer loads -- no LDS ring, no barrier in the loop. The four waves of a workgroup take interleaved k-steps of the
//     unit (an intra-workgroup K split: each wave streams its own bytes; the probe's "own rows / wave" line, 5.65 TB/s chip-wide) and
//     add their accumulators through LDS at the end.
//   * fp32 partial tiles go to the split-K slab [slice][pixel][Cout]; the existing splitk_reduce_kernel (igemm.hip) sums the slices
//     in order 0 .. S-1 and runs the epilogue (bias, time-embedding row bias, residual, activation) -- bitwise reproducible.
//
// Round 5: this file is EVIDENCE, not product. In the SD1.5 step the fused launch measured 1 % SLOWER than the two operators it
// replaces (180.1 vs 182.2 it/s, profiles/r04_gnconv_step_ab_run6.log), so it is compiled only
// into the probe build (-DSFAST_PROBES: build.py --probes -> libsfast_hip_probes.so, what the gn_conv2d tests and tools/gnconv_ab.py
// load). The product library carries the stubs at the bottom: sfast_hip_gn_conv2d_supported() == 0, workspace 0,
// sfast_hip_gn_conv2d() == SFAST_ERR_UNSUPPORTED.
#include <type_traits>

#include "igemm_device.h"

namespace sfast {
#ifdef SFAST_PROBES

struct GnConvArgs {
    const void *x, *x2, *gamma, *beta, *w;
    float *partial;
    int B, H, W, C1, C2, Cout;
    int gamma_on;    // 1: normalise (GN instantiation); 0: plain weight-streaming conv
    int P, HW;       // pixels in total (= M), per sample
    int cpg;         // channels per GroupNorm group
    int CS, S, KS;   // channel slice, number of slices, 16-wide k-steps per tap (CS / 16)
    int NIT;         // k-steps of a unit: 9 * KS
    int rowb;        // LDS row pitch in bytes (CS * 2 + 16)
    int64_t ldw;     // elements between weight rows (9 * Cin)
    float eps;
    int silu;
    unsigned long long *trace;  // profiling (sfast_hip_set_trace): 16 slots of 100 MHz wall-clock stamps per workgroup; nullptr in production
};

// slots: 0 entry, 1 weight requests issued, 2 slice in LDS, 3 statistics, 4 normalised, 5 k loop done, 6 accumulators added, 7 slab stored
__device__ __forceinline__ void gc_mark(const GnConvArgs &a, int slot) {
    if (a.trace != nullptr) {
        if (threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + slot] = wall_clock64();
    }
}

// Weight k-steps a wave requests BEFORE its main loop, spread over the prologue (a few ahead of the activation slice's arrival, the
// rest between the statistics and the iterations of the normalisation): the stream of a unit and the ~4 us of VALU work that turn
// the raw slice into the conv's operand then overlap instead of following each other. History (profiles/r04_gnconv_*.log): all
// requests up front blocked the waves for 5 - 7 us at the memory queue (the chip drains them at HBM rate) and left the prologue behind
// them -- 31 us per 1280 -> 1280 conv against 21.5 us for the two operators; 6 / 12 / 24 steps in flight measured the same: the
// launch was never short of requests, it serialised stream and prologue. NB = 1: 24 steps = a wave's whole share of a 160-channel
// slice (9 taps x 10 steps / 4 waves), nothing left to request in the loop; NB = 2 (two fragments per step): 8, refilled in the loop.
template <int NB> struct GcDepth {
    static constexpr int D = NB == 1 ? 24 : 8;
};
constexpr int GC_PRE = 4;  // steps requested right behind the slice, and again during the statistics

// GN = false: the same launch without the normalisation (the input is already the conv's operand): slice -> LDS, weight stream, K loop.
template <typename T, int MB, int NB, bool GN>
__global__ void __launch_bounds__(256, 2) gnconv_kernel(const GnConvArgs a) {
    constexpr int GC_D = GcDepth<NB>::D;
    constexpr int SL_MAX = 10;  // 16-byte chunks of the slice per thread: the planner keeps P * CS / 8 <= 2560 (128 pixels x 160 channels)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using vec8 = typename Elem<T>::vec8;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.x % a.S, nt = blockIdx.x / a.S;
    const int Cin = a.C1 + a.C2;
    const int c0 = s * a.CS;
    const int ROWB = a.rowb;
    const int ZR = MB * 32;  // the all-zero row
    float *stat = reinterpret_cast<float *>(lds + (size_t)(ZR + 1) * ROWB);  // [B * GS][2] {mean, rstd}
    float *gam = stat + 2 * a.B * (a.CS / a.cpg);                             // [CS] gamma (fp32), then [CS] beta
    float *bet = gam + a.CS;
    gc_mark(a, 0);

    // ---- 1. the activation slice and its gamma / beta are REQUESTED first (they come back first: vmcnt is in order), the weight stream
    //         right behind them. The first version asked for the weights first and then sat 6.7 us in front of the slice; gamma / beta
    //         were fetched inside the normalisation loop, ten exposed round trips = 5.4 us (profiles/r04_gnconv_phase_trace_run4.log).
    const bool second = c0 >= a.C1;
    const T *src = second ? (const T *)a.x2 : (const T *)a.x;
    const int Csrc = second ? a.C2 : a.C1, coff = second ? c0 - a.C1 : c0;
    const int CCH = a.CS / 8;
    const int total = a.P * CCH;
    const float rcch = __builtin_amdgcn_rcpf((float)CCH);
    u32x4 sl[SL_MAX];
#pragma unroll
    for (int j = 0; j < SL_MAX; ++j) {
        const int q = tid + j * 256;
        if (q < total) {
            const int p = fdiv22(q, CCH, rcch), cc = q - p * CCH;
            sl[j] = *reinterpret_cast<const u32x4 *>(src + (int64_t)p * Csrc + coff + cc * 8);
        }
    }
    u32x4 gbv = u32x4{0u, 0u, 0u, 0u};
    const bool gb_thread = GN && tid < 2 * CCH;  // threads 0 .. CCH-1: a gamma chunk, CCH .. 2 CCH - 1: a beta chunk
    if (gb_thread) {
        const T *gp = tid < CCH ? (const T *)a.gamma : (const T *)a.beta;
        if (gp) gbv = *reinterpret_cast<const u32x4 *>(gp + c0 + (tid < CCH ? tid : tid - CCH) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. weights: the descriptor covers this unit's rows. A wave owns a CONTIGUOUS range of the unit's 9 * KS k-steps (tap-major), so
    //         four consecutive steps of a tap walk one 128-byte line of every weight row, as the streaming probe does.
    const T *wb = (const T *)a.w + (int64_t)nt * (NB * 32) * a.ldw;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)wb);
    const uint32_t hi32 = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)wb >> 32));
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi32 << 32) | lo), 0,
                                                                        (int)((int64_t)NB * 32 * a.ldw * 2), 0x00020000);
    int voff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) voff[nb] = (int)((((int64_t)nb * 32 + r) * a.ldw + c0 + g * 8) * 2);
    const int step0 = (wave * a.NIT) >> 2, step1 = ((wave + 1) * a.NIT) >> 2;  // this wave's k-steps [step0, step1)
    const float rks = __builtin_amdgcn_rcpf((float)a.KS);
    int l_tap = __builtin_amdgcn_readfirstlane(fdiv22(step0, a.KS, rks));
    int l_ks = step0 - l_tap * a.KS, l_left = step1 - step0;  // next k-step to request, steps not yet requested
    auto load_step = [&](u32x4 (&dst)[NB]) __attribute__((always_inline)) {
        // past the wave's range the last step is requested again (in range of the descriptor; its MFMAs multiply the zero row)
        const int soff = (l_tap * Cin + l_ks * 16) * 2;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dst[nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff[nb], soff, 0));
        if (l_left > 1) {
            --l_left;
            if (++l_ks == a.KS) {
                l_ks = 0;
                ++l_tap;
            }
        }
    };
    u32x4 wq[GC_D][NB];
#pragma unroll
    for (int d = 0; d < (GN ? GC_PRE : GC_D); ++d) {  // without a prologue to hide behind, the whole pre-loop queue goes out at once
        load_step(wq[d]);
        __builtin_amdgcn_sched_barrier(0);  // issue order = consumption order
    }
    gc_mark(a, 1);

    // ---- 3. slice -> LDS (raw), gamma / beta -> LDS (fp32), zero row; statistics; normalise in place ---------------------------------
#pragma unroll
    for (int j = 0; j < SL_MAX; ++j) {
        const int q = tid + j * 256;
        if (q < total) {
            const int p = fdiv22(q, CCH, rcch), cc = q - p * CCH;
            *reinterpret_cast<u32x4 *>(lds + p * ROWB + cc * 16) = sl[j];  // the compiler waits for exactly this load (older than the weights)
        }
    }
    if (gb_thread) {
        float f[8];
        unpack8<T>(gbv, f);
        float *dst = (tid < CCH ? gam + tid * 8 : bet + (tid - CCH) * 8);
        const bool present = tid < CCH ? a.gamma != nullptr : a.beta != nullptr;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = present ? f[e] : (tid < CCH ? 1.f : 0.f);
    }
    for (int q = tid; q < CCH; q += 256) *reinterpret_cast<u32x4 *>(lds + ZR * ROWB + q * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    gc_mark(a, 2);
    const float rhw = __builtin_amdgcn_rcpf((float)a.HW);
    if constexpr (GN) {
    const int GS = a.CS / a.cpg, CPG8 = a.cpg / 8;
    const int npairs = a.B * GS;
    const float inv_n = 1.0f / ((float)a.HW * (float)a.cpg);
#pragma unroll
    for (int d = GC_PRE; d < 2 * GC_PRE; ++d) {  // the next weight steps go out while the statistics run
        load_step(wq[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // Statistics of every (sample, group) of the slice, all 256 threads at once: 32 lanes per pair, a lane sums the group's channels
    // of its pixels (p = lane32, lane32 + 32, ...) as shifted sums about the group's first element (one pass: S1 = sum(x - sh),
    // S2 = sum((x - sh)^2), the shift bounds the cancellation like norm.hip's), then five xor-shuffle steps inside the half wave.
    // Fixed lane / chunk order -> bitwise reproducible. The first version gave a whole wave to a pair and made two passes: 3.2 - 3.8 us.
    const int lane32 = tid & 31;
    for (int pair = tid >> 5; pair < npairs; pair += 8) {
        const int b = pair / GS, gi = pair - b * GS;
        const char *grow = lds + (size_t)(b * a.HW) * ROWB + gi * CPG8 * 16;
        const float sh = Elem<T>::to_f32(*reinterpret_cast<const T *>(grow));
        float s1 = 0.f, s2 = 0.f;
        for (int p = lane32; p < a.HW; p += 32) {
            const char *row = grow + p * ROWB;
            u32x4 v[10];
#pragma unroll
            for (int c = 0; c < 10; ++c)  // all reads of the pixel's group in flight together (C/G <= 80)
                v[c] = *reinterpret_cast<const u32x4 *>(row + (c < CPG8 ? c : 0) * 16);
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                float f[8];
                unpack8<T>(v[c], f);
                const float use = c < CPG8 ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dl = (f[e] - sh) * use;
                    s1 += dl;
                    s2 = fmaf(dl, dl, s2);
                }
            }
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        const float m1 = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - m1 * m1, 0.f);
        if (lane32 == 0) {
            stat[pair * 2] = sh + m1;
            stat[pair * 2 + 1] = rsqrtf(var + a.eps);
        }
    }
    __syncthreads();
    gc_mark(a, 3);
    const float rcpg8 = __builtin_amdgcn_rcpf((float)CPG8);
#pragma unroll
    for (int j = 0; j < SL_MAX; ++j) {
        const int q = tid + j * 256;
        if (q < total) {
            const int p = fdiv22(q, CCH, rcch), cc = q - p * CCH;
            const int b = fdiv22(p, a.HW, rhw), gi = fdiv22(cc, CPG8, rcpg8);
            const float mean = stat[(b * GS + gi) * 2], rstd = stat[(b * GS + gi) * 2 + 1];
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gam + cc * 8), g1 = *reinterpret_cast<const f32x4 *>(gam + cc * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bet + cc * 8), b1 = *reinterpret_cast<const f32x4 *>(bet + cc * 8 + 4);
            float f[8];
            unpack8<T>(sl[j], f);  // the raw chunk is still in registers
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (f[e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
                f[e] = a.silu ? act_silu(v) : v;
            }
            *reinterpret_cast<u32x4 *>(lds + p * ROWB + cc * 16) = pack8<T>(f);
        }
        // the rest of the pre-loop weight steps, spread evenly over the ten iterations (compile-time schedule)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 2 * GC_PRE + (j * (GC_D - 2 * GC_PRE)) / SL_MAX; d < 2 * GC_PRE + ((j + 1) * (GC_D - 2 * GC_PRE)) / SL_MAX; ++d) {
            load_step(wq[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    }  // GN
    gc_mark(a, 4);

    // ---- 4. main loop: no barrier, no LDS write; per k-step MB fragment reads + NB fragments already in registers -> MB * NB MFMAs.
    //         Fragment addresses are recomputed only when the tap changes (three or four times per wave); a step costs MB adds.
    int prow[MB], vmask[MB];
    const float rw = __builtin_amdgcn_rcpf((float)a.W);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int pix = mb * 32 + r;
        const int b = fdiv22(pix, a.HW, rhw), rem = pix - b * a.HW;
        const int y = fdiv22(rem, a.W, rw), xq = rem - y * a.W;
        prow[mb] = pix * ROWB + g * 16;
        int m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = xq + t % 3 - 1;
            m |= ((int)((unsigned)yy < (unsigned)a.H) & (int)((unsigned)xx < (unsigned)a.W) & (int)(pix < a.P)) << t;
        }
        vmask[mb] = m;
    }
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;
    int c_tap = __builtin_amdgcn_readfirstlane(fdiv22(step0, a.KS, rks));
    int c_ks = step0 - c_tap * a.KS, c_left = step1 - step0;  // the k-step whose fragments are read NEXT, steps not yet read
    int baddr[MB], kadd[MB];
    auto set_tap = [&]() __attribute__((always_inline)) {  // fragment addresses of k-step (c_tap, c_ks); all-zero row once the range is used up
        const int tap = c_left > 0 ? c_tap : 9;            // bit 9 of the masks is never set
        const int dy = (tap >= 6 ? 1 : (tap >= 3 ? 0 : -1)), dx = tap - (dy + 1) * 3 - 1;
        const int dlt = (dy * a.W + dx) * ROWB + c_ks * 32;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int hit = (vmask[mb] >> tap) & 1;
            baddr[mb] = hit ? prow[mb] + dlt : ZR * ROWB;
            kadd[mb] = hit ? 32 : 0;
        }
    };
    vec8 bf[2][MB];
    auto read_b = [&](vec8 (&dst)[MB]) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            dst[mb] = *reinterpret_cast<const vec8 *>(lds + baddr[mb]);
            baddr[mb] += kadd[mb];
        }
        --c_left;
        if (++c_ks == a.KS || c_left <= 0) {  // wave-uniform: next tap, or the end of the range
            if (c_ks == a.KS) {
                c_ks = 0;
                ++c_tap;
            }
            set_tap();
        }
    };
    set_tap();
    const int per_wave = (a.NIT + 3) / 4;                    // k-steps of the busiest wave
    const int trips = (per_wave + GC_D - 1) / GC_D;           // every wave runs the same trip count; surplus steps multiply zeros
    read_b(bf[0]);
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int d = 0; d < GC_D; ++d) {
            read_b(bf[(d + 1) & 1]);  // fragments of the NEXT k-step, requested before this step's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = mfma32(__builtin_bit_cast(vec8, wq[d][nb]), bf[d & 1][mb], acc[mb][nb]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < trips) load_step(wq[d]);  // k-step D ahead into the registers just consumed (wave-uniform; none in the last trip)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static_assert(GC_D % 2 == 0, "the fragment double buffer assumes an even prefetch depth");

    // ---- add the four waves' accumulators (fixed tree: (0 + 2) + (1 + 3)) and write the fp32 partial tile ------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus weight requests have landed before their registers die
    gc_mark(a, 5);
    __syncthreads();                                    // every wave is done reading the slice
    constexpr int NQ = MB * NB * 4;                     // float4 quads per lane
    f32x4 *red = reinterpret_cast<f32x4 *>(lds);        // [2][NQ][64]
    auto put = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[(slot * NQ + (mb * NB + nb) * 4 + q) * 64 + lane] =
                        f32x4{acc[mb][nb][q * 4], acc[mb][nb][q * 4 + 1], acc[mb][nb][q * 4 + 2], acc[mb][nb][q * 4 + 3]};
    };
    auto get = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = red[(slot * NQ + (mb * NB + nb) * 4 + q) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mb][nb][q * 4 + e] += v[e];
                }
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) get(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) {
        get(0);
        gc_mark(a, 6);
        // 32x32 MFMA result layout: register 4 q + e of lane (r, g) = output channel 8 q + 4 g + e (of the 32-block), pixel r
        float *slab = a.partial + (int64_t)s * a.P * a.Cout;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 32 + r;
            if (m < a.P) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = nt * (NB * 32) + nb * 32 + q * 8 + g * 4;
                        *reinterpret_cast<f32x4 *>(slab + (int64_t)m * a.Cout + n) =
                            f32x4{acc[mb][nb][q * 4], acc[mb][nb][q * 4 + 1], acc[mb][nb][q * 4 + 2], acc[mb][nb][q * 4 + 3]};
                    }
            }
        }
        if (a.trace != nullptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gc_mark(a, 7);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static int gcd_i(int x, int y) {
    while (y) {
        const int t = x % y;
        x = y;
        y = t;
    }
    return x;
}

// Covers: f16 / bf16, 3x3 / stride 1 / padding 1 / no dilation / no fused upsample, dense NHWC sources and [Cout][3][3][Cin] weights,
// B*H*W <= 128, channels per group a multiple of 8, Cout % 32 == 0 -- and a channel slice that is a whole number of groups, a whole
// number of 16-wide k-steps, at least 80 channels, and does not straddle the two concat sources.
bool gnconv_plan(int B, int H, int W, int C1, int C2, int Cout, int groups, GnConvPlan &pl) {
    const int Cin = C1 + C2, P = B * H * W;
    if (groups < 0 || (groups > 0 && Cin % groups) || P <= 0 || P > 128 || Cout % 32 || H * W < 1 || Cin % 16) return false;
    const int cpg = groups > 0 ? Cin / groups : 16;  // groups == 0: no normalisation, any 16-channel multiple may be a slice
    if (cpg % 8) return false;
    const int unit = cpg / gcd_i(cpg, 16) * 16;  // lcm(cpg, 16)
    pl.MB = P <= 64 ? 2 : 4;
    int best_key = -1;
    for (int CS = unit; CS <= Cin && CS <= 640; CS += unit) {
        if (CS < 80 || Cin % CS || C1 % CS || (C2 && C2 % CS)) continue;
        for (int NB = 1; NB <= 2; ++NB) {
            if (Cout % (32 * NB)) continue;
            // slice rows + the zero row, {mean, rstd} per (sample, group), gamma and beta of the slice as fp32
            const size_t patch = (size_t)(pl.MB * 32 + 1) * (CS * 2 + 16) + (size_t)B * (CS / cpg) * 8 + (size_t)2 * CS * 4;
            const size_t red = (size_t)2 * pl.MB * NB * 4 * 64 * 16;
            const size_t lds = patch > red ? patch : red;
            if (lds > 78 * 1024) continue;  // two workgroups per CU
            if ((size_t)P * (CS / 8) > 2560) continue;  // the slice passes through 10 16-byte registers per thread
            const int S = Cin / CS, units = (Cout / (32 * NB)) * S;
            const int rounds = (units + 511) / 512;
            // fewest rounds of 512 co-resident workgroups; then enough units to fill the chip; then the fewest slices (slab bytes)
            const int key = (rounds == 1 ? 1 : 0) * 1000000 + (units >= 224 ? 1 : 0) * 100000 + (1000 - S) * 10 + (2 - NB);
            if (key > best_key) {
                best_key = key;
                pl.NB = NB;
                pl.CS = CS;
                pl.S = S;
                pl.lds_bytes = lds + 16;
            }
        }
    }
    if (best_key < 0) return false;
    pl.slab_bytes = (size_t)pl.S * P * Cout * sizeof(float);
    return true;
}

template <typename T> static int gnconv_launch(const GnConvArgs &g, const GnConvPlan &pl, hipStream_t st) {
    const dim3 grid((unsigned)((g.Cout / (32 * pl.NB)) * pl.S)), block(256);
#define GC_OP(MB_, NB_)                                                            \
    if (pl.MB == MB_ && pl.NB == NB_) {                                            \
        if (g.cpg > 0 && g.gamma_on)                                                                       \
            hipLaunchKernelGGL((gnconv_kernel<T, MB_, NB_, true>), grid, block, pl.lds_bytes, st, g);     \
        else                                                                                               \
            hipLaunchKernelGGL((gnconv_kernel<T, MB_, NB_, false>), grid, block, pl.lds_bytes, st, g);    \
        return check_launch("gnconv");                                             \
    }
    GC_OP(4, 1) GC_OP(4, 2) GC_OP(2, 1) GC_OP(2, 2)
#undef GC_OP
    set_error("gnconv: no kernel for MB=%d NB=%d", pl.MB, pl.NB);
    return SFAST_ERR_UNSUPPORTED;
}

// `a` carries the conv problem as sfast_hip_conv2d_ex fills it (M, N = Cout, geometry, epilogue operands); the launch writes the
// slab into `ws` and finishes with the split-K reduce + epilogue of igemm.hip.
extern unsigned long long *g_igemm_trace;  // igemm_glds.hip (sfast_hip_set_trace)

int gnconv_run(IgemmArgs &a, int dtype, int B, const void *gamma, const void *beta, int groups, float eps, int silu, void *ws, size_t ws_bytes,
               hipStream_t st) {
    GnConvPlan pl{};
    a.trace = g_igemm_trace;
    SFAST_REQUIRE(gnconv_plan(B, a.H, a.W, a.C1, a.C2, a.N, groups, pl), SFAST_ERR_UNSUPPORTED, "gn_conv2d: shape outside the fused kernel's coverage");
    SFAST_REQUIRE(ws && ws_bytes >= pl.slab_bytes, SFAST_ERR_WORKSPACE, "gn_conv2d: workspace %zu < %zu", ws_bytes, pl.slab_bytes);
    GnConvArgs g{};
    g.x = a.x;
    g.x2 = a.x2;
    g.gamma = gamma;
    g.beta = beta;
    g.w = a.w[0];
    g.partial = (float *)ws;
    g.B = B;
    g.H = a.H;
    g.W = a.W;
    g.C1 = a.C1;
    g.C2 = a.C2;
    g.Cout = a.N;
    g.P = a.M;
    g.HW = a.H * a.W;
    g.cpg = groups > 0 ? (a.C1 + a.C2) / groups : 16;
    g.gamma_on = groups > 0 ? 1 : 0;
    g.CS = pl.CS;
    g.S = pl.S;
    g.KS = pl.CS / 16;
    g.NIT = 9 * g.KS;
    g.rowb = pl.CS * 2 + 16;
    g.ldw = (int64_t)9 * (a.C1 + a.C2);
    g.eps = eps;
    g.silu = silu;
    g.trace = a.trace;
    set_kernel_name("%s_%s[P=%d,%dx%d,slices=%d]", groups > 0 ? "gnconv" : "wsconv", dtype == SFAST_F16 ? "f16" : "bf16", a.M, 32 * pl.NB, pl.CS, pl.S);
    const int rc = dtype == SFAST_F16 ? gnconv_launch<f16>(g, pl, st) : gnconv_launch<bf16>(g, pl, st);
    if (rc) return rc;
    a.splits = pl.S;
    a.partial = (float *)ws;
    if (a.out_scale == 0.f) a.out_scale = 1.0f;
    return igemm_reduce_only(a, dtype, st);
}

// dynamic-LDS limit of every instantiation (80 KiB: two workgroups per CU); per device, from sfast_hip_init
int gnconv_init() {
    hipError_t e = hipSuccess;
#define GC_ATTR(T, MB_, NB_)                                                                                                         \
    if (e == hipSuccess)                                                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(gnconv_kernel<T, MB_, NB_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
    if (e == hipSuccess)                                                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(gnconv_kernel<T, MB_, NB_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    GC_ATTR(f16, 4, 1) GC_ATTR(f16, 4, 2) GC_ATTR(f16, 2, 1) GC_ATTR(f16, 2, 2)
    GC_ATTR(bf16, 4, 1) GC_ATTR(bf16, 4, 2) GC_ATTR(bf16, 2, 1) GC_ATTR(bf16, 2, 2)
#undef GC_ATTR
    if (e != hipSuccess) {
        set_error("gnconv_init: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return SFAST_OK;
}

#else  // !SFAST_PROBES: the product library has no fused GroupNorm -> conv launch

bool gnconv_plan(int, int, int, int, int, int, int, GnConvPlan &) { return false; }
int gnconv_run(IgemmArgs &, int, int, const void *, const void *, int, float, int, void *, size_t, hipStream_t) {
    set_error("gn_conv2d: this library was built without -DSFAST_PROBES (the fused launch is a measured, slower candidate)");
    return SFAST_ERR_UNSUPPORTED;
}
int gnconv_init() { return SFAST_OK; }

#endif
}  // namespace sfast
