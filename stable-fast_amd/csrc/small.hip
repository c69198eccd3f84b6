// Non-MFMA kernels of the GEMM / conv families:
//   * gemv_small_m   : M <= 16 rows (time-embedding MLP, time_emb_proj): pure weight streaming,
//                      one wave per output column, 16-B loads, fp32 accumulate.
//   * conv_small_n   : Cout <= 8 (conv_out 320->4): one wave per 4 output pixels, K split over lanes.
//   * conv_small_c   : Cin <= 8 (conv_in 4->320): weights staged once per workgroup in LDS as fp32,
//                      thread = (pixel, 8 output channels); reads NCHW or NHWC input through strides.
//   * naive gemm / conv: correctness catch-all for shapes, strides and dtypes (f32) the fast
//                      kernels do not take -- still HIP, there is no ATen/CPU fallback in the library.
#include "small.h"

namespace sfast {

// ---------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void small_epilogue(const SmallGemmArgs &a, int m, int n, float v, float g) {
    // v (and g for geglu) are raw accumulators of output (m, n)
    v *= a.out_scale;
    if (a.geglu) {
        g *= a.out_scale;
        if (a.bias) {
            v += Elem<T>::to_f32(((const T *)a.bias)[n]);
            g += Elem<T>::to_f32(((const T *)a.bias)[a.N + n]);
        }
        v = v * act_gelu_erf(g);
    } else {
        if (a.bias) v += Elem<T>::to_f32(((const T *)a.bias)[n]);
        if (a.rowbias) v += Elem<T>::to_f32(((const T *)a.rowbias)[(int64_t)(m / a.rows_per_batch) * a.ld_rowbias + n]);
        float r = 0.f;
        if (a.res) r = a.alpha * Elem<T>::to_f32(((const T *)a.res)[(int64_t)m * a.ldr + n]);
        if (a.res_before_act) v += r;
        v = apply_act(v, a.act);
        if (!a.res_before_act) v += r;
    }
    ((T *)a.out)[(int64_t)m * a.ldo + n] = Elem<T>::from_f32(v);
}

template <typename T>
__device__ __forceinline__ const T *small_wrow(const SmallGemmArgs &a, int row) {
    // row in [0, N) (or [0, 2N) for geglu, single segment)
    const int seg = row / a.rows_per_seg;
    const void *base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : seg == 2 ? a.w[2] : a.w[3];
    return (const T *)base + (int64_t)(row - seg * a.rows_per_seg) * a.ldw;
}

// one wave per output column n, MB rows at a time. K % 8 == 0, 16-B aligned rows.
template <typename T, int MB>
__global__ void __launch_bounds__(256) gemv_small_m_kernel(const SmallGemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= a.N) return;
    const T *wr = small_wrow<T>(a, n);
    const T *wg = a.geglu ? small_wrow<T>(a, a.N + n) : nullptr;
    const int nch = a.K / 8;
    for (int mb = 0; mb < a.M; mb += MB) {
        float acc[MB], accg[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            acc[i] = 0.f;
            accg[i] = 0.f;
        }
        for (int ch = lane; ch < nch; ch += 64) {
            float wf[8], gf[8];
            unpack8<T>(*reinterpret_cast<const u32x4 *>(wr + ch * 8), wf);
            if (wg) unpack8<T>(*reinterpret_cast<const u32x4 *>(wg + ch * 8), gf);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int m = mb + i;
                if (m < a.M) {
                    float xf[8];
                    unpack8<T>(*reinterpret_cast<const u32x4 *>((const T *)a.x + (int64_t)m * a.ldx + ch * 8), xf);
                    if (a.in_act != SFAST_ACT_NONE) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) xf[j] = apply_act(xf[j], a.in_act);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i] = fmaf(xf[j], wf[j], acc[i]);
                    if (wg) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) accg[i] = fmaf(xf[j], gf[j], accg[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const float s = wave_sum(acc[i]);
            const float sg = wg ? wave_sum(accg[i]) : 0.f;
            const int m = mb + i;
            if (lane == 0 && m < a.M) small_epilogue<T>(a, m, n, s, sg);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Grouped GEMV: out[m][off_g + n] = act(sum_k x[m][k] * W_g[n][k] + bias_g[n]) for G independent weight
// matrices sharing ONE input -- the UNet's 22 `time_emb_proj` layers all consume silu(emb), which depends on
// nothing but the timestep ("batch them into one GEMV over concatenated weights").
// The weights stay where the live parameters are (no concatenated copy): the per-group base pointers travel
// in the kernel-argument block, one wave per output column finds its group with a uniform scan.
struct GroupedGemvArgs {
    const void *x;
    void *out;
    const void *w[SFAST_MAX_GROUPS];
    const void *bias[SFAST_MAX_GROUPS];
    int n_end[SFAST_MAX_GROUPS];  // exclusive prefix end of every group in the concatenated column space
    int n_groups, M, K, Ntot;
    int64_t ldx, ldw, ldo;
    int act, in_act;
};

template <typename T, int MB>
__global__ void __launch_bounds__(256) gemv_grouped_kernel(const GroupedGemvArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * 4 + wave;
    if (n >= a.Ntot) return;
    int g = 0;
    while (g + 1 < a.n_groups && n >= a.n_end[g]) ++g;
    const int nloc = n - (g ? a.n_end[g - 1] : 0);
    const T *wr = (const T *)a.w[g] + (int64_t)nloc * a.ldw;
    const T *bp = (const T *)a.bias[g];
    const int nch = a.K / 8;
    for (int mb = 0; mb < a.M; mb += MB) {
        float acc[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i] = 0.f;
        for (int ch = lane; ch < nch; ch += 64) {
            float wf[8];
            unpack8<T>(*reinterpret_cast<const u32x4 *>(wr + ch * 8), wf);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int m = mb + i;
                if (m < a.M) {
                    float xf[8];
                    unpack8<T>(*reinterpret_cast<const u32x4 *>((const T *)a.x + (int64_t)m * a.ldx + ch * 8), xf);
                    if (a.in_act != SFAST_ACT_NONE) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) xf[j] = apply_act(xf[j], a.in_act);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i] = fmaf(xf[j], wf[j], acc[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            float s = wave_sum(acc[i]);
            const int m = mb + i;
            if (lane == 0 && m < a.M) {
                if (bp) s += Elem<T>::to_f32(bp[nloc]);
                s = apply_act(s, a.act);
                ((T *)a.out)[(int64_t)m * a.ldo + n] = Elem<T>::from_f32(s);
            }
        }
    }
}

template <typename T> static int run_gemv_grouped(const GroupedGemvArgs &a, hipStream_t st) {
    const dim3 grid(ceil_div(a.Ntot, 4));
    if (a.M <= 2)
        hipLaunchKernelGGL((gemv_grouped_kernel<T, 2>), grid, dim3(256), 0, st, a);
    else if (a.M <= 4)
        hipLaunchKernelGGL((gemv_grouped_kernel<T, 4>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((gemv_grouped_kernel<T, 8>), grid, dim3(256), 0, st, a);
    return check_launch("gemv_grouped");
}

// thread per output element, scalar loads; any K / alignment / dtype
template <typename T>
__global__ void __launch_bounds__(256) gemm_naive_kernel(const SmallGemmArgs a) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (n >= a.N || m >= a.M) return;
    const T *xr = (const T *)a.x + (int64_t)m * a.ldx;
    const T *wr = small_wrow<T>(a, n);
    float acc = 0.f, accg = 0.f;
    for (int k = 0; k < a.K; ++k) {
        float xv = Elem<T>::to_f32(xr[k]);
        if (a.in_act != SFAST_ACT_NONE) xv = apply_act(xv, a.in_act);
        acc = fmaf(xv, Elem<T>::to_f32(wr[k]), acc);
    }
    if (a.geglu) {
        const T *wg = small_wrow<T>(a, a.N + n);
        for (int k = 0; k < a.K; ++k) accg = fmaf(Elem<T>::to_f32(xr[k]), Elem<T>::to_f32(wg[k]), accg);
    }
    small_epilogue<T>(a, m, n, acc, accg);
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void conv_store(const SmallConvArgs &a, int b, int ho, int wo, int co, float v) {
    v *= a.out_scale;
    if (a.bias) v += Elem<T>::to_f32(((const T *)a.bias)[co]);
    if (a.rowbias) v += Elem<T>::to_f32(((const T *)a.rowbias)[(int64_t)b * a.ld_rowbias + co]);
    float r = 0.f;
    if (a.z) r = a.alpha * Elem<T>::to_f32(((const T *)a.z)[b * a.zs[0] + ho * a.zs[1] + wo * a.zs[2] + co * a.zs[3]]);
    if (a.res_before_act) v += r;
    v = apply_act(v, a.act);
    if (!a.res_before_act) v += r;
    ((T *)a.out)[b * a.os[0] + ho * a.os[1] + wo * a.os[2] + co * a.os[3]] = Elem<T>::from_f32(v);
}

// 8 consecutive output channels co0..co0+7 of one pixel: operands as 16-byte vectors requested together and
// unconditionally (absent operands read the device zero block), one activation loop, one 16-byte store.
// Caller guarantees the layout (vec_epilogue_ok). The scalar conv_store costs three dependent load round trips and a
// full activation switch PER ELEMENT.
template <typename T>
__device__ __forceinline__ void conv_store8(const SmallConvArgs &a, int b, int ho, int wo, int co0, const float (&acc)[8]) {
    typedef const u32x4 __attribute__((address_space(1))) * g4_ptr;
    const g4_ptr zero = (g4_ptr)(const void *)g_zero16;
    const u32x4 vb = *(a.bias ? (g4_ptr)(const void *)((const T *)a.bias + co0) : zero);
    const u32x4 vrb = *(a.rowbias ? (g4_ptr)(const void *)((const T *)a.rowbias + (int64_t)b * a.ld_rowbias + co0) : zero);
    const u32x4 vz = *(a.z ? (g4_ptr)(const void *)((const T *)a.z + (b * a.zs[0] + ho * a.zs[1] + wo * a.zs[2] + co0)) : zero);
    float fb[8], frb[8], fz[8], v[8];
    unpack8<T>(vb, fb);
    unpack8<T>(vrb, frb);
    unpack8<T>(vz, fz);
    const bool res_now = a.res_before_act || a.act == SFAST_ACT_NONE;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(acc[e], a.out_scale, fb[e]) + frb[e] + (res_now ? a.alpha * fz[e] : -0.0f);
    if (a.act != SFAST_ACT_NONE) {
        f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {  // rolled: one copy of the activation switch
            lo[e] = apply_act(lo[e], a.act);
            hi[e] = apply_act(hi[e], a.act);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = lo[e] + (res_now ? -0.0f : a.alpha * fz[e]);
            v[4 + e] = hi[e] + (res_now ? -0.0f : a.alpha * fz[4 + e]);
        }
    }
    *reinterpret_cast<u32x4 *>((T *)a.out + (b * a.os[0] + ho * a.os[1] + wo * a.os[2] + co0)) = pack8<T>(v);
}
__device__ __forceinline__ bool vec_epilogue_ok(const SmallConvArgs &a) {
    auto al = [](const void *p) { return (((uintptr_t)p) & 15) == 0; };
    bool ok = a.os[3] == 1 && al(a.out) && a.os[0] % 8 == 0 && a.os[1] % 8 == 0 && a.os[2] % 8 == 0;
    ok = ok && (!a.bias || al(a.bias)) && (!a.rowbias || (al(a.rowbias) && a.ld_rowbias % 8 == 0));
    ok = ok && (!a.z || (a.zs[3] == 1 && al(a.z) && a.zs[0] % 8 == 0 && a.zs[1] % 8 == 0 && a.zs[2] % 8 == 0));
    return ok;
}
template <typename T>
__device__ __forceinline__ float conv_load_x(const SmallConvArgs &a, int b, int hi, int wi, int c) {
    // (hi, wi) in the (possibly 2x-upsampled) input frame; returns 0 outside
    const int HH = a.ups ? 2 * a.H : a.H, WW = a.ups ? 2 * a.W : a.W;
    if ((unsigned)hi >= (unsigned)HH || (unsigned)wi >= (unsigned)WW) return 0.f;
    if (a.ups) {
        hi >>= 1;
        wi >>= 1;
    }
    if (c < a.C1) return Elem<T>::to_f32(((const T *)a.x)[b * a.xs[0] + hi * a.xs[1] + wi * a.xs[2] + c * a.xs[3]]);
    return Elem<T>::to_f32(((const T *)a.x2)[b * a.x2s[0] + hi * a.x2s[1] + wi * a.x2s[2] + (c - a.C1) * a.x2s[3]]);
}

template <typename T>
__global__ void __launch_bounds__(256) conv_naive_kernel(const SmallConvArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.B * a.Ho * a.Wo * a.Cout;
    if (idx >= total) return;
    const int co = (int)(idx % a.Cout);
    int64_t t = idx / a.Cout;
    const int wo = (int)(t % a.Wo);
    t /= a.Wo;
    const int ho = (int)(t % a.Ho);
    const int b = (int)(t / a.Ho);
    float acc = 0.f;
    for (int r = 0; r < a.KH; ++r) {
        const int hi = ho * a.stride_h - a.pad_h + r * a.dil_h;
        for (int s = 0; s < a.KW; ++s) {
            const int wi = wo * a.stride_w - a.pad_w + s * a.dil_w;
            for (int c = 0; c < a.Cin; ++c) {
                const float xv = conv_load_x<T>(a, b, hi, wi, c);
                const float wv = Elem<T>::to_f32(((const T *)a.w)[co * a.ws[0] + c * a.ws[1] + r * a.ws[2] + s * a.ws[3]]);
                acc = fmaf(xv, wv, acc);
            }
        }
    }
    conv_store<T>(a, b, ho, wo, co, acc);
}

// Cout <= 8, dense NHWC x (no concat), weights [Cout][KH][KW][Cin] K-contiguous, Cin % 8 == 0.
// One wave per PX consecutive output pixels; lanes split K in 16-B chunks.
template <typename T, int PX>
__global__ void __launch_bounds__(256) conv_small_n_kernel(const SmallConvArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * PX;
    if (p0 >= M) return;
    int pb[PX], ph[PX], pw[PX];
    bool pv[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int64_t p = p0 + i;
        pv[i] = p < M;
        const int64_t pp = pv[i] ? p : 0;
        const int hw = a.Ho * a.Wo;
        pb[i] = (int)(pp / hw);
        const int rem = (int)(pp % hw);
        ph[i] = rem / a.Wo;
        pw[i] = rem % a.Wo;
    }
    float acc[PX][8];
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[i][c] = 0.f;
    const int cpc = a.Cin / 8;           // chunks per tap
    const int nch = a.KH * a.KW * cpc;   // chunks in K
    const int64_t wstride = (int64_t)a.KH * a.KW * a.Cin;
    typedef const u32x4 __attribute__((address_space(1))) * g4_ptr;
    const g4_ptr zero = (g4_ptr)(const void *)g_zero16;
    const float r_cpc = __builtin_amdgcn_rcpf((float)cpc), r_kw = __builtin_amdgcn_rcpf((float)a.KW);
    const int HH = a.ups ? 2 * a.H : a.H, WW = a.ups ? 2 * a.W : a.W;
    for (int j = lane; j < nch; j += 64) {
        // nch <= 2^22 always (KH*KW*Cin/8): reciprocal division, exact after the +-1 fix-up
        int tap = (int)((float)j * r_cpc);
        tap += (j - tap * cpc >= cpc) ? 1 : 0;
        tap -= (j - tap * cpc < 0) ? 1 : 0;
        const int cch = j - tap * cpc;
        int r = (int)((float)tap * r_kw);
        r += (tap - r * a.KW >= a.KW) ? 1 : 0;
        r -= (tap - r * a.KW < 0) ? 1 : 0;
        const int s = tap - r * a.KW;
        // every load of the iteration is issued unconditionally (border taps read a device zero block): a load
        // inside `if (ok)` is an exec-masked branch with its own wait -- PX serial round trips per iteration
        u32x4 wraw[8], xraw[PX];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool cok = c < a.Cout;
            wraw[c] = *(cok ? (g4_ptr)(const void *)((const T *)a.w + c * wstride + (int64_t)j * 8) : zero);
        }
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            int hi = ph[i] * a.stride_h - a.pad_h + r * a.dil_h;
            int wi = pw[i] * a.stride_w - a.pad_w + s * a.dil_w;
            const bool ok = pv[i] & ((unsigned)hi < (unsigned)HH) & ((unsigned)wi < (unsigned)WW);
            if (a.ups) {
                hi >>= 1;
                wi >>= 1;
            }
            xraw[i] = *(ok ? (g4_ptr)(const void *)((const T *)a.x + (((int64_t)pb[i] * a.H + hi) * a.W + wi) * a.Cin + cch * 8) : zero);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < a.Cout) {
                float wf[8];
                unpack8<T>(wraw[c], wf);
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    float xf[8];
                    unpack8<T>(xraw[i], xf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[i][c] = fmaf(xf[e], wf[e], acc[i][c]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < a.Cout) {
                const float s = wave_sum(acc[i][c]);
                if (lane == 0 && pv[i]) conv_store<T>(a, pb[i], ph[i], pw[i], c, s);
            }
        }
    }
}

// Cin*KH*KW small: weights staged in LDS as fp32 [K][Cout]; thread = (pixel, 8 output channels).
// x read through strides (NCHW or NHWC), output dense NHWC-style through strides with os[3] == 1.
// Every global load is issued in a batch of independent, unconditional requests (invalid taps / k >= K read a
// device zero block): the first version had one load per loop iteration -- 72 serial round trips to stage the
// weights and 36 per output vector, 85 us for SD's 4->320 conv_in.
// KMAX > 0: K <= KMAX, the whole receptive field of a pixel is fetched in ONE batch (k decode is scalar work on
// compile-time k). KMAX == 0: any K, 8 channels of one tap per batch.
// G8 = 8-channel chunks per task (1 or 4): with 4 the receptive field of a pixel is fetched once per 32 outputs
// instead of once per 8, and a task's chunks are strided by Cout/32 so that a wave's LDS reads stay lane-contiguous.
template <typename T, int KMAX, int G8>
__global__ void __launch_bounds__(256) conv_small_c_kernel(const SmallConvArgs a, int pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // [K][Cout]
    typedef const T __attribute__((address_space(1))) * gelem_ptr;
    const gelem_ptr zero = (gelem_ptr)(const void *)g_zero16;
    const int K = a.KH * a.KW * a.Cin;
    for (int co = threadIdx.x; co < a.Cout; co += 256) {
        const T *wrow = (const T *)a.w + (int64_t)co * a.ws[0];
        int c = 0, r = 0, s = 0;  // uniform (r, s, c) of the next k, advanced incrementally: scalar ALU, no division
        for (int k0 = 0; k0 < K; k0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u;
                const T *src = wrow + ((int64_t)c * a.ws[1] + (int64_t)r * a.ws[2] + (int64_t)s * a.ws[3]);
                v[u] = Elem<T>::to_f32(*(k < K ? (gelem_ptr)(const void *)src : zero));
                if (++c == a.Cin) {
                    c = 0;
                    if (++s == a.KW) {
                        s = 0;
                        ++r;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < K) wsm[(k0 + u) * a.Cout + ((co >> 2) & 1) * (a.Cout / 2) + (co >> 3) * 4 + (co & 3)] = v[u];
        }
    }
    __syncthreads();
    const int cch = a.Cout / 8 / G8;  // tasks per pixel
    const bool vec_out = vec_epilogue_ok(a);
    const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
    const int64_t p_begin = (int64_t)blockIdx.x * pix_per_block;
    const int ntask = pix_per_block * cch;
    const int HH = a.ups ? 2 * a.H : a.H, WW = a.ups ? 2 * a.W : a.W;
    auto x_ptr = [&](int b, int hs, int wsrc, int c) -> const T * {
        return (c < a.C1) ? (const T *)a.x + (b * a.xs[0] + hs * a.xs[1] + wsrc * a.xs[2] + c * a.xs[3])
                          : (const T *)a.x2 + (b * a.x2s[0] + hs * a.x2s[1] + wsrc * a.x2s[2] + (c - a.C1) * a.x2s[3]);
    };
    for (int t = threadIdx.x; t < ntask; t += 256) {
        const int64_t p = p_begin + t / cch;
        if (p >= M) break;
        const int j0 = t % cch;  // chunks j0 + g*cch, g < G8
        const int hw = a.Ho * a.Wo;
        const int b = (int)(p / hw);
        const int rem = (int)(p % hw);
        const int ho = rem / a.Wo, wo = rem % a.Wo;
        const int h0 = ho * a.stride_h - a.pad_h, w0 = wo * a.stride_w - a.pad_w;
        float acc[G8][8];
#pragma unroll
        for (int g = 0; g < G8; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
        auto fma8 = [&](float xv, int k) {
            // LDS image [k][half][chunk][4]: each of a lane's 16-byte reads is lane-contiguous (stride 16 B)
#pragma unroll
            for (int g = 0; g < G8; ++g) {
                const int ch4 = (j0 + g * cch) * 4;
                const f32x4 w0v = *reinterpret_cast<const f32x4 *>(wsm + k * a.Cout + ch4);
                const f32x4 w1v = *reinterpret_cast<const f32x4 *>(wsm + k * a.Cout + a.Cout / 2 + ch4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[g][q] = fmaf(xv, w0v[q], acc[g][q]);
                    acc[g][4 + q] = fmaf(xv, w1v[q], acc[g][4 + q]);
                }
            }
        };
        if constexpr (KMAX > 0) {
            // fast form (host guarantees: no upsample, no concat, KH*KW <= 32): one 64-bit pixel base per output
            // vector, tap validity as a bit mask, and per k only a wave-uniform (scalar) offset -- the generic
            // 4-stride 64-bit address per element was ~45 instructions each, 80 us for SD's conv_in
            const T *pbase = (const T *)a.x + ((int64_t)b * a.xs[0] + (int64_t)h0 * a.xs[1] + (int64_t)w0 * a.xs[2]);
            unsigned tapmask = 0;
            for (int r = 0; r < a.KH; ++r)
                for (int s = 0; s < a.KW; ++s) {
                    const bool ok = ((unsigned)(h0 + r * a.dil_h) < (unsigned)a.H) & ((unsigned)(w0 + s * a.dil_w) < (unsigned)a.W);
                    tapmask |= (ok ? 1u : 0u) << (r * a.KW + s);
                }
            float xv[KMAX];
            int c = 0, r = 0, s = 0, tap = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int64_t soff = (int64_t)(r * a.dil_h) * a.xs[1] + (int64_t)(s * a.dil_w) * a.xs[2] + (int64_t)c * a.xs[3];  // uniform
                const bool ok = (k < K) & (((tapmask >> (tap & 31)) & 1u) != 0);
                xv[k] = Elem<T>::to_f32(*(ok ? (gelem_ptr)(const void *)(pbase + soff) : zero));
                if (++c == a.Cin) {
                    c = 0;
                    ++tap;
                    if (++s == a.KW) {
                        s = 0;
                        ++r;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) fma8(xv[k], k);
        } else {
            for (int r = 0; r < a.KH; ++r) {
                const int hi = h0 + r * a.dil_h;
                for (int s = 0; s < a.KW; ++s) {
                    const int wi = w0 + s * a.dil_w;
                    const bool ok = ((unsigned)hi < (unsigned)HH) & ((unsigned)wi < (unsigned)WW);
                    const int hs = a.ups ? hi >> 1 : hi, wsrc = a.ups ? wi >> 1 : wi;
                    const int kbase = (r * a.KW + s) * a.Cin;
                    for (int c0 = 0; c0 < a.Cin; c0 += 8) {
                        float xv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int c = c0 + e;
                            const bool v = ok & (c < a.Cin);
                            xv[e] = Elem<T>::to_f32(*(v ? (gelem_ptr)(const void *)x_ptr(b, hs, wsrc, c) : zero));
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (c0 + e < a.Cin) fma8(xv[e], kbase + c0 + e);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G8; ++g) {
            const int co0 = (j0 + g * cch) * 8;
            if (vec_out) {
                conv_store8<T>(a, b, ho, wo, co0, acc[g]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) conv_store<T>(a, b, ho, wo, co0 + e, acc[g][e]);
            }
        }
    }
}

// ---- host launchers -----------------------------------------------------------------------------------
template <typename T> static int run_gemv(const SmallGemmArgs &a, hipStream_t st) {
    const dim3 grid(ceil_div(a.N, 4));
    if (a.M <= 2)
        hipLaunchKernelGGL((gemv_small_m_kernel<T, 2>), grid, dim3(256), 0, st, a);
    else if (a.M <= 4)
        hipLaunchKernelGGL((gemv_small_m_kernel<T, 4>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((gemv_small_m_kernel<T, 8>), grid, dim3(256), 0, st, a);
    return check_launch("gemv_small_m");
}

int small_gemv(const SmallGemmArgs &a, int dtype, hipStream_t st) {
    set_kernel_name("gemv_small_m");
    if (dtype == SFAST_F16) return run_gemv<f16>(a, st);
    if (dtype == SFAST_BF16) return run_gemv<bf16>(a, st);
    set_error("gemv_small_m: dtype %d", dtype);
    return SFAST_ERR_UNSUPPORTED;
}

int small_gemm_naive(const SmallGemmArgs &a, int dtype, hipStream_t st) {
    set_kernel_name("gemm_naive");
    const dim3 grid(ceil_div(a.N, 64), ceil_div(a.M, 4));
    switch (dtype) {
    case SFAST_F16: hipLaunchKernelGGL(gemm_naive_kernel<f16>, grid, dim3(256), 0, st, a); break;
    case SFAST_BF16: hipLaunchKernelGGL(gemm_naive_kernel<bf16>, grid, dim3(256), 0, st, a); break;
    case SFAST_F32: hipLaunchKernelGGL(gemm_naive_kernel<float>, grid, dim3(256), 0, st, a); break;
    default: set_error("gemm_naive: dtype %d", dtype); return SFAST_ERR_UNSUPPORTED;
    }
    return check_launch("gemm_naive");
}

int small_conv_naive(const SmallConvArgs &a, int dtype, hipStream_t st) {
    set_kernel_name("conv_naive");
    const int64_t total = (int64_t)a.B * a.Ho * a.Wo * a.Cout;
    const dim3 grid((unsigned)ceil_div64(total, 256));
    switch (dtype) {
    case SFAST_F16: hipLaunchKernelGGL(conv_naive_kernel<f16>, grid, dim3(256), 0, st, a); break;
    case SFAST_BF16: hipLaunchKernelGGL(conv_naive_kernel<bf16>, grid, dim3(256), 0, st, a); break;
    case SFAST_F32: hipLaunchKernelGGL(conv_naive_kernel<float>, grid, dim3(256), 0, st, a); break;
    default: set_error("conv_naive: dtype %d", dtype); return SFAST_ERR_UNSUPPORTED;
    }
    return check_launch("conv_naive");
}

int small_conv_n(const SmallConvArgs &a, int dtype, hipStream_t st) {
    set_kernel_name("conv_small_n");
    constexpr int PX = 4;
    const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
    const dim3 grid((unsigned)ceil_div64(M, 4 * PX));
    if (dtype == SFAST_F16)
        hipLaunchKernelGGL((conv_small_n_kernel<f16, PX>), grid, dim3(256), 0, st, a);
    else if (dtype == SFAST_BF16)
        hipLaunchKernelGGL((conv_small_n_kernel<bf16, PX>), grid, dim3(256), 0, st, a);
    else {
        set_error("conv_small_n: dtype %d", dtype);
        return SFAST_ERR_UNSUPPORTED;
    }
    return check_launch("conv_small_n");
}

int small_conv_c(const SmallConvArgs &a, int dtype, hipStream_t st) {
    set_kernel_name("conv_small_c");
    const int K = a.KH * a.KW * a.Cin;
    const size_t smem = (size_t)K * a.Cout * sizeof(float);
    const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
    const bool tiny = K <= 40 && !a.ups && a.C1 == a.Cin && a.KH * a.KW <= 32;
    const bool wide = a.Cout % 32 == 0;
    int ppb = 32;
    while (ppb > 4 && ceil_div64(M, ppb) < 256) ppb >>= 1;
    if (tiny && wide) {
        // tasks per pixel = Cout/32: size the block so that its tasks fill the 256 threads about once
        const int tpp = a.Cout / 32;
        int want = 256 / (tpp > 0 ? tpp : 1);
        if (want < 1) want = 1;
        if (want < ppb) ppb = want;
    }
    const dim3 grid((unsigned)ceil_div64(M, ppb));
#define CSC_LAUNCH(T, KM, G)                                                                                      \
    hipLaunchKernelGGL((conv_small_c_kernel<T, KM, G>), grid, dim3(256), smem, st, a, ppb)
#define CSC_DISPATCH(T)        \
    if (tiny && wide)          \
        CSC_LAUNCH(T, 40, 4);  \
    else if (tiny)             \
        CSC_LAUNCH(T, 40, 1);  \
    else                       \
        CSC_LAUNCH(T, 0, 1);
    if (dtype == SFAST_F16) {
        CSC_DISPATCH(f16)
    } else if (dtype == SFAST_BF16) {
        CSC_DISPATCH(bf16)
    } else {
        set_error("conv_small_c: dtype %d", dtype);
        return SFAST_ERR_UNSUPPORTED;
    }
#undef CSC_DISPATCH
#undef CSC_LAUNCH
    return check_launch("conv_small_c");
}

}  // namespace sfast

using namespace sfast;

extern "C" int sfast_hip_gemv_grouped(const void *x, const void *const *w, const void *const *bias, void *out,
                                      const sfast_gemv_grouped_params *p, sfast_stream_t stream) {
    SFAST_REQUIRE(p && x && w && out, SFAST_ERR_INVALID, "gemv_grouped: null argument");
    SFAST_REQUIRE(p->n_groups >= 1 && p->n_groups <= SFAST_MAX_GROUPS, SFAST_ERR_INVALID, "gemv_grouped: n_groups=%d (1..%d)",
                  p->n_groups, SFAST_MAX_GROUPS);
    SFAST_REQUIRE(p->M > 0 && p->M <= 64 && p->K > 0, SFAST_ERR_INVALID, "gemv_grouped: bad shape M=%d K=%d (M <= 64)", p->M, p->K);
    SFAST_REQUIRE(p->dtype == SFAST_F16 || p->dtype == SFAST_BF16, SFAST_ERR_UNSUPPORTED, "gemv_grouped: dtype %d", p->dtype);
    SFAST_REQUIRE(p->K % 8 == 0 && p->ldx % 8 == 0 && p->ldw % 8 == 0 && p->ldx >= p->K && p->ldw >= p->K && aligned16(x),
                  SFAST_ERR_UNSUPPORTED, "gemv_grouped: rows must be 16-byte aligned (K, ldx, ldw multiples of 8)");
    GroupedGemvArgs a{};
    a.x = x;
    a.out = out;
    int tot = 0;
    for (int g = 0; g < p->n_groups; ++g) {
        SFAST_REQUIRE(w[g] && aligned16(w[g]) && p->n_rows[g] > 0, SFAST_ERR_INVALID, "gemv_grouped: bad group %d", g);
        a.w[g] = w[g];
        a.bias[g] = bias ? bias[g] : nullptr;
        tot += p->n_rows[g];
        a.n_end[g] = tot;
    }
    for (int g = p->n_groups; g < SFAST_MAX_GROUPS; ++g) {
        a.w[g] = w[0];
        a.bias[g] = nullptr;
        a.n_end[g] = tot;
    }
    SFAST_REQUIRE(p->ldo >= tot, SFAST_ERR_INVALID, "gemv_grouped: ldo %lld < %d output columns", (long long)p->ldo, tot);
    a.n_groups = p->n_groups;
    a.M = p->M;
    a.K = p->K;
    a.Ntot = tot;
    a.ldx = p->ldx;
    a.ldw = p->ldw;
    a.ldo = p->ldo;
    a.act = p->act;
    a.in_act = p->in_act;
    set_kernel_name("gemv_grouped[G=%d]", p->n_groups);
    if (p->dtype == SFAST_F16) return run_gemv_grouped<f16>(a, (hipStream_t)stream);
    return run_gemv_grouped<bf16>(a, (hipStream_t)stream);
}
