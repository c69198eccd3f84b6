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
#include <type_traits>

#include "igemm_device.h"

namespace sfast {

struct GnConvArgs {
    const void *x, *x2, *gamma, *beta, *w;
    float *partial;
    int B, H, W, C1, C2, Cout;
    int P, HW;       // pixels in total (= M), per sample
    int cpg;         // channels per GroupNorm group
    int CS, S, KS;   // channel slice, number of slices, 16-wide k-steps per tap (CS / 16)
    int NIT;         // k-steps of a unit: 9 * KS
    int rowb;        // LDS row pitch in bytes (CS * 2 + 16)
    int64_t ldw;     // elements between weight rows (9 * Cin)
    float eps;
    int silu;
};

constexpr int GC_D = 6;  // weight k-steps in flight per wave

template <typename T, int MB, int NB>
__global__ void __launch_bounds__(256, 2) gnconv_kernel(const GnConvArgs a) {
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

    // ---- weights: the descriptor covers this unit's rows; the first D k-steps are requested before anything else ----------------
    const T *wb = (const T *)a.w + (int64_t)nt * (NB * 32) * a.ldw;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)wb);
    const uint32_t hi32 = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)wb >> 32));
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi32 << 32) | lo), 0,
                                                                        (int)((int64_t)NB * 32 * a.ldw * 2), 0x00020000);
    int voff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) voff[nb] = (int)((((int64_t)nb * 32 + r) * a.ldw + c0 + g * 8) * 2);
    // k-step i of this wave: i = wave + 4 j  ->  (tap, ks); soffset = (tap * Cin + ks * 16) * 2 bytes. Steps past the unit's end re-read
    // the last step (kept in range) and multiply by the zero row.
    int l_tap = 0, l_ks = wave;  // state of the NEXT k-step to request
    while (l_ks >= a.KS) {
        l_ks -= a.KS;
        ++l_tap;
    }
    auto load_step = [&](u32x4 (&dst)[NB]) __attribute__((always_inline)) {
        const int tap = l_tap < 9 ? l_tap : 8, ks = l_tap < 9 ? l_ks : a.KS - 1;
        const int soff = (tap * Cin + ks * 16) * 2;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dst[nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff[nb], soff, 0));
        l_ks += 4;
        if (l_ks >= a.KS) {  // KS >= 5: at most one wrap per step
            l_ks -= a.KS;
            ++l_tap;
        }
    };
    u32x4 wq[GC_D][NB];
#pragma unroll
    for (int d = 0; d < GC_D; ++d) {
        load_step(wq[d]);
        __builtin_amdgcn_sched_barrier(0);  // issue order = consumption order
    }

    // ---- the activation slice -> LDS (raw), then statistics, then normalise in place -------------------------------------------------
    const bool second = c0 >= a.C1;
    const T *src = second ? (const T *)a.x2 : (const T *)a.x;
    const int Csrc = second ? a.C2 : a.C1, coff = second ? c0 - a.C1 : c0;
    const int CCH = a.CS / 8;
    const int total = a.P * CCH;
    const float rcch = __builtin_amdgcn_rcpf((float)CCH);
    for (int q = tid; q < total; q += 256) {
        const int p = fdiv22(q, CCH, rcch), cc = q - p * CCH;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + (int64_t)p * Csrc + coff + cc * 8);
        *reinterpret_cast<u32x4 *>(lds + p * ROWB + cc * 16) = v;
    }
    for (int q = tid; q < CCH; q += 256) *reinterpret_cast<u32x4 *>(lds + ZR * ROWB + q * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int GS = a.CS / a.cpg, CPG8 = a.cpg / 8;
    const int npairs = a.B * GS;
    const float inv_n = 1.0f / ((float)a.HW * (float)a.cpg);
    for (int pair = wave; pair < npairs; pair += 4) {  // one wave per (sample, group): fixed lane / chunk order -> reproducible
        const int b = pair / GS, gi = pair - b * GS;
        float sum = 0.f;
        for (int p = lane; p < a.HW; p += 64) {
            const char *row = lds + (b * a.HW + p) * ROWB + gi * CPG8 * 16;
            for (int c = 0; c < CPG8; ++c) {
                float f[8];
                unpack8<T>(*reinterpret_cast<const u32x4 *>(row + c * 16), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += f[e];
            }
        }
        const float mean = wave_sum(sum) * inv_n;
        float sq = 0.f;
        for (int p = lane; p < a.HW; p += 64) {
            const char *row = lds + (b * a.HW + p) * ROWB + gi * CPG8 * 16;
            for (int c = 0; c < CPG8; ++c) {
                float f[8];
                unpack8<T>(*reinterpret_cast<const u32x4 *>(row + c * 16), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dlt = f[e] - mean;
                    sq += dlt * dlt;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) * inv_n + a.eps);
        if (lane == 0) {
            stat[pair * 2] = mean;
            stat[pair * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    const float rhw = __builtin_amdgcn_rcpf((float)a.HW), rcpg8 = __builtin_amdgcn_rcpf((float)CPG8);
    for (int q = tid; q < total; q += 256) {
        const int p = fdiv22(q, CCH, rcch), cc = q - p * CCH;
        const int b = fdiv22(p, a.HW, rhw), gi = fdiv22(cc, CPG8, rcpg8);
        const float mean = stat[(b * GS + gi) * 2], rstd = stat[(b * GS + gi) * 2 + 1];
        float f[8], ga[8], be[8];
        unpack8<T>(*reinterpret_cast<const u32x4 *>(lds + p * ROWB + cc * 16), f);
        if (a.gamma) {
            unpack8<T>(*reinterpret_cast<const u32x4 *>((const T *)a.gamma + c0 + cc * 8), ga);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ga[e] = 1.f;
        }
        if (a.beta) {
            unpack8<T>(*reinterpret_cast<const u32x4 *>((const T *)a.beta + c0 + cc * 8), be);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) be[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (f[e] - mean) * rstd * ga[e] + be[e];
            f[e] = a.silu ? act_silu(v) : v;
        }
        *reinterpret_cast<u32x4 *>(lds + p * ROWB + cc * 16) = pack8<T>(f);
    }
    __syncthreads();

    // ---- main loop: no barrier, no LDS write; per k-step MB fragment reads + NB fragments already in registers -> MB * NB MFMAs ----------
    // per 32-pixel block: the lane's pixel row and a 9-bit mask of the taps whose source pixel lies inside the image (padding pixels of
    // the last block: no tap) -- a k-step then costs one bit test, one select and one multiply-add per fragment address
    int prow[MB], vmask[MB];
    const float rw = __builtin_amdgcn_rcpf((float)a.W);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int pix = mb * 32 + r;
        const int b = fdiv22(pix, a.HW, rhw), rem = pix - b * a.HW;
        const int y = fdiv22(rem, a.W, rw), xq = rem - y * a.W;
        prow[mb] = pix;
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
    int c_tap = 0, c_ks = wave;  // state of the k-step being CONSUMED
    while (c_ks >= a.KS) {
        c_ks -= a.KS;
        ++c_tap;
    }
    vec8 bf[2][MB];
    auto read_b = [&](vec8 (&dst)[MB]) __attribute__((always_inline)) {
        const int tap = c_tap < 9 ? c_tap : 9;  // 9: past the unit's end -- no mask bit is set, every lane reads the zero row
        const int dy = (tap >= 6 ? 1 : (tap >= 3 ? 0 : -1)), dx = tap - (dy + 1) * 3 - 1;
        const int dlt = (dy * a.W + dx) * ROWB + c_ks * 32 + g * 16;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int hit = (vmask[mb] >> tap) & 1;
            const int addr = hit ? prow[mb] * ROWB + dlt : ZR * ROWB;
            dst[mb] = *reinterpret_cast<const vec8 *>(lds + addr);
        }
        c_ks += 4;
        if (c_ks >= a.KS) {
            c_ks -= a.KS;
            ++c_tap;
        }
    };
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
            load_step(wq[d]);  // k-step D ahead into the registers just consumed
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static_assert(GC_D % 2 == 0, "the fragment double buffer assumes an even prefetch depth");

    // ---- add the four waves' accumulators (fixed tree: (0 + 2) + (1 + 3)) and write the fp32 partial tile ------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus weight requests have landed before their registers die
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
    if (groups <= 0 || Cin % groups || P <= 0 || P > 128 || Cout % 32 || H * W < 1) return false;
    const int cpg = Cin / groups;
    if (cpg % 8) return false;
    const int unit = cpg / gcd_i(cpg, 16) * 16;  // lcm(cpg, 16)
    pl.MB = P <= 64 ? 2 : 4;
    int best_key = -1;
    for (int CS = unit; CS <= Cin && CS <= 640; CS += unit) {
        if (CS < 80 || Cin % CS || C1 % CS || (C2 && C2 % CS)) continue;
        for (int NB = 1; NB <= 2; ++NB) {
            if (Cout % (32 * NB)) continue;
            const size_t patch = (size_t)(pl.MB * 32 + 1) * (CS * 2 + 16) + (size_t)B * (CS / cpg) * 8;
            const size_t red = (size_t)2 * pl.MB * NB * 4 * 64 * 16;
            const size_t lds = patch > red ? patch : red;
            if (lds > 78 * 1024) continue;  // two workgroups per CU
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
        hipLaunchKernelGGL((gnconv_kernel<T, MB_, NB_>), grid, block, pl.lds_bytes, st, g); \
        return check_launch("gnconv");                                             \
    }
    GC_OP(4, 1) GC_OP(4, 2) GC_OP(2, 1) GC_OP(2, 2)
#undef GC_OP
    set_error("gnconv: no kernel for MB=%d NB=%d", pl.MB, pl.NB);
    return SFAST_ERR_UNSUPPORTED;
}

// `a` carries the conv problem as sfast_hip_conv2d_ex fills it (M, N = Cout, geometry, epilogue operands); the launch writes the
// slab into `ws` and finishes with the split-K reduce + epilogue of igemm.hip.
int gnconv_run(IgemmArgs &a, int dtype, int B, const void *gamma, const void *beta, int groups, float eps, int silu, void *ws, size_t ws_bytes,
               hipStream_t st) {
    GnConvPlan pl{};
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
    g.cpg = (a.C1 + a.C2) / groups;
    g.CS = pl.CS;
    g.S = pl.S;
    g.KS = pl.CS / 16;
    g.NIT = 9 * g.KS;
    g.rowb = pl.CS * 2 + 16;
    g.ldw = (int64_t)9 * (a.C1 + a.C2);
    g.eps = eps;
    g.silu = silu;
    set_kernel_name("gnconv_%s[P=%d,%dx%d,slices=%d]", dtype == SFAST_F16 ? "f16" : "bf16", a.M, 32 * pl.NB, pl.CS, pl.S);
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
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(gnconv_kernel<T, MB_, NB_>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    GC_ATTR(f16, 4, 1) GC_ATTR(f16, 4, 2) GC_ATTR(f16, 2, 1) GC_ATTR(f16, 2, 2)
    GC_ATTR(bf16, 4, 1) GC_ATTR(bf16, 4, 2) GC_ATTR(bf16, 2, 1) GC_ATTR(bf16, 2, 2)
#undef GC_ATTR
    if (e != hipSuccess) {
        set_error("gnconv_init: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return SFAST_OK;
}

}  // namespace sfast
