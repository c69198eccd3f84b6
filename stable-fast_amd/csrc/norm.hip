// GroupNorm(+SiLU) and LayerNorm for gfx950 -- HBM-bound kernels.
//
// Replaces the reference's Triton kernels (src/sfast/triton/ops/group_norm.py:111-165 stats,
// :272-349 apply, src/sfast/triton/ops/layer_norm.py:52-133). Differences by design:
//   * NHWC fast path: a workgroup streams whole pixel rows (all channels, 16 B per lane, fully
//     coalesced) instead of one (group, sample) program gathering C/G-wide strips -- the
//     reference launches only (32, N) programs, far fewer than 256 CUs.
//   * statistics are shifted sums in fp32 (shift = first element of the group), so the partials
//     of different workgroups add exactly like Welford merges without carrying means around;
//     the reduction order is fixed -> bitwise reproducible across graph replays.
//   * affine + SiLU are computed in fp32 (the reference applies the affine in fp16).
//   * a virtual channel concat (x | x2) lets the UNet's up-blocks normalise torch.cat([h, skip])
//     without materialising it.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace sfast {

// =================================================================================================
// NHWC fast path: C % 8 == 0, C1 % 8 == 0, C/G >= 8 or == 4 (an 8-channel chunk then touches at most two groups), f16 / bf16.
// thread (tx, ty): tx -> one 8-channel chunk column, ty -> pixel row phase.
// =================================================================================================
struct GnGeom {
    int HW, C, C1, cpg, G;
    int CX;   // C / 8
    int TXB;  // chunk columns handled per pass (threads along channels)
    int TY;   // rows handled concurrently
};

template <typename T>
__device__ __forceinline__ const T *gn_src(const T *x, const T *x2, const GnGeom &g, int b, int64_t row,
                                           int c) {
    // element pointer for (sample b, pixel row, channel c) under the virtual concat
    if (c < g.C1) return x + ((int64_t)b * g.HW + row) * g.C1 + c;
    return x2 + ((int64_t)b * g.HW + row) * (g.C - g.C1) + (c - g.C1);
}

template <typename T>
__global__ void gn_nhwc_stats_kernel(const T *__restrict__ x, const T *__restrict__ x2,
                                     float *__restrict__ partial, GnGeom g, int rows_per_block,
                                     int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NT = g.TXB * g.TY;
    float *red = smem;            // [NT][4]
    float *gsum = smem + NT * 4;  // [G][2]
    const int tid = threadIdx.x;
    const int tx = tid % g.TXB, ty = tid / g.TXB;
    const int b = blockIdx.y, split = blockIdx.x;
    const int r0 = split * rows_per_block;
    const int r1 = min(g.HW, r0 + rows_per_block);

    if (tid < g.G) {
        gsum[tid * 2] = 0.f;
        gsum[tid * 2 + 1] = 0.f;
    }
    for (int cxb = 0; cxb < g.CX; cxb += g.TXB) {
        const int cx = cxb + tx;
        float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
        if (cx < g.CX && tid < NT) {
            const int c = cx * 8;
            const int g0 = c / g.cpg;
            const int nb = min(8, (g0 + 1) * g.cpg - c);
            const float sh0 = (float)*gn_src(x, x2, g, b, 0, g0 * g.cpg);
            const float sh1 = (nb < 8) ? (float)*gn_src(x, x2, g, b, 0, (g0 + 1) * g.cpg) : 0.f;
            auto accumulate = [&](const u32x4 &raw) {
                float f[8];
                unpack8<T>(raw, f);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < nb) {
                        const float d = f[i] - sh0;
                        s1a += d;
                        s2a += d * d;
                    } else {
                        const float d = f[i] - sh1;
                        s1b += d;
                        s2b += d * d;
                    }
                }
            };
            // four independent row loads in flight per thread (a one-load-per-iteration loop left a workgroup with
            // 8 KB outstanding: ~8 GB/s per CU); accumulation order is unchanged, so results are too
            int r = r0 + ty;
            for (; r + 3 * g.TY < r1; r += 4 * g.TY) {
                u32x4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const u32x4 *>(gn_src(x, x2, g, b, r + u * g.TY, c));
#pragma unroll
                for (int u = 0; u < 4; ++u) accumulate(raw[u]);
            }
            for (; r < r1; r += g.TY) accumulate(*reinterpret_cast<const u32x4 *>(gn_src(x, x2, g, b, r, c)));
        }
        if (tid < NT) {
            red[tid * 4 + 0] = s1a;
            red[tid * 4 + 1] = s2a;
            red[tid * 4 + 2] = s1b;
            red[tid * 4 + 3] = s2b;
        }
        __syncthreads();
        // reduce over ty (fixed order)
        if (tid < g.TXB) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int j = 0; j < g.TY; ++j) {
                const float *p = red + (j * g.TXB + tid) * 4;
                a0 += p[0];
                a1 += p[1];
                a2 += p[2];
                a3 += p[3];
            }
            // entry (0, tid) is read by this thread only -> in-place write is race-free
            red[tid * 4 + 0] = a0;
            red[tid * 4 + 1] = a1;
            red[tid * 4 + 2] = a2;
            red[tid * 4 + 3] = a3;
        }
        __syncthreads();
        // gather chunk columns into groups (thread = group, fixed order)
        if (tid < g.G) {
            const int grp = tid;
            int lo = (grp * g.cpg) / 8, hi = ((grp + 1) * g.cpg - 1) / 8;
            lo = max(lo, cxb);
            hi = min(hi, min(g.CX, cxb + g.TXB) - 1);
            float s1 = 0.f, s2 = 0.f;
            for (int cxx = lo; cxx <= hi; ++cxx) {
                const int g0 = (cxx * 8) / g.cpg;
                const float *p = red + (cxx - cxb) * 4;
                if (g0 == grp) {
                    s1 += p[0];
                    s2 += p[1];
                } else if (g0 + 1 == grp) {
                    s1 += p[2];
                    s2 += p[3];
                }
            }
            gsum[grp * 2] += s1;
            gsum[grp * 2 + 1] += s2;
        }
        __syncthreads();
    }
    if (tid < g.G) {
        float *o = partial + (((int64_t)b * nsplit + split) * g.G + tid) * 2;
        o[0] = gsum[tid * 2];
        o[1] = gsum[tid * 2 + 1];
    }
}

template <typename T, bool SILU>
__device__ __forceinline__ void gn_apply_rows(const T *__restrict__ x, const T *__restrict__ x2, const T *__restrict__ gamma,
                                              const T *__restrict__ beta, T *__restrict__ y, const GnGeom &g, int rows_per_block,
                                              const float *mean, const float *rstd, u32x4 ga_raw, u32x4 be_raw);

// Statistics handed over by the PRODUCERS of the tensor (igemm_device.h flush_staged_tile / splitk_reduce_rows_kernel): per
// concat source one array of {mean, M2} records indexed [sample][row block][tile_n][slot].
struct GnPre {
    const float *p[2];
    int rb_rows[2], n_rb[2], bno[2], tiles_n[2], slots[2], unit[2], nch[2], coff[2];
    int kl;  // lanes per record slot in the merge prologue
};

// Chan et al. pairwise update: (n, mean, M2) <- (n, mean, M2) (+) (ni, mi, M2i)
__device__ __forceinline__ void chan_merge(float &n, float &mean, float &m2, float ni, float mi, float m2i) {
    if (ni <= 0.f) return;
    const float nn = n + ni;
    const float d = mi - mean;
    const float f = ni / nn;
    mean = fmaf(d, f, mean);
    m2 = m2 + m2i + d * d * n * f;
    n = nn;
}

template <typename T, bool SILU, bool PRE>
__global__ void gn_nhwc_apply_kernel(const T *__restrict__ x, const T *__restrict__ x2,
                                     const T *__restrict__ gamma, const T *__restrict__ beta,
                                     T *__restrict__ y, const float *__restrict__ partial, GnGeom g,
                                     int rows_per_block, int nsplit, float eps, const GnPre pre) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *mean = smem;               // [G]
    float *rstd = smem + g.G;         // [G]
    float *tmp = smem + 2 * g.G;      // [G][8][2]  (PRE: [G][8][3])
    const int NT = g.TXB * g.TY;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    constexpr int RS = 8;
    if constexpr (PRE) {
        // gamma / beta of the first channel pass go out first, as below
        const int tx0 = tid % g.TXB;
        const bool pre_ok = tx0 < g.CX && tid < g.TXB * g.TY;
        u32x4 ga_raw = {0u, 0u, 0u, 0u}, be_raw = {0u, 0u, 0u, 0u};
        if (pre_ok && gamma) ga_raw = *reinterpret_cast<const u32x4 *>(gamma + tx0 * 8);
        if (pre_ok && beta) be_raw = *reinterpret_cast<const u32x4 *>(beta + tx0 * 8);
        // merge the producers' records in three fixed-order stages (bitwise reproducible):
        //   1. lane k of slot si (KL lanes per slot, all threads busy, loads batched) merges row blocks k, k+KL, ... of that slot;
        //   2. one thread per slot merges its KL lanes;  3. one thread per group merges the slots of the group in channel order.
        // (The first version let 8 threads per group walk every record through dependent loads: 20-50 us of prologue per workgroup.)
        const int S0 = pre.tiles_n[0] * pre.slots[0];
        const int S_tot = S0 + (g.C1 < g.C ? pre.tiles_n[1] * pre.slots[1] : 0);
        const int KL = pre.kl;
        float *lane_rec = tmp;                    // [S_tot][KL][3]
        float *slot_rec = tmp + S_tot * KL * 3;   // [S_tot][3]
        auto slot_geom = [&](int si, int &s_, int &tn, int &jj, float &cnt) {
            s_ = si < S0 ? 0 : 1;
            const int sl = si - (s_ ? S0 : 0);
            tn = sl / pre.slots[s_];
            jj = sl - tn * pre.slots[s_];
            const int U = (tn * pre.bno[s_]) / pre.unit[s_] + jj;
            const int lo = max(tn * pre.bno[s_], U * pre.unit[s_]), hi = min(min((tn + 1) * pre.bno[s_], pre.nch[s_]), (U + 1) * pre.unit[s_]);
            cnt = hi > lo ? (float)(hi - lo) * (float)pre.rb_rows[s_] : 0.f;
        };
        for (int w = tid; w < S_tot * KL; w += blockDim.x) {
            const int si = w / KL, k = w - si * KL;
            int s_, tn, jj;
            float cnt;
            slot_geom(si, s_, tn, jj, cnt);
            float n = 0.f, mu = 0.f, m2 = 0.f;
            if (cnt > 0.f) {
                const float *base = pre.p[s_] + ((((int64_t)b * pre.n_rb[s_]) * pre.tiles_n[s_] + tn) * pre.slots[s_] + jj) * 2;
                const int64_t rstride = (int64_t)pre.tiles_n[s_] * pre.slots[s_] * 2;
                int rb = k;
                for (; rb + 3 * KL < pre.n_rb[s_]; rb += 4 * KL) {
                    float2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2 *>(base + (rb + u * KL) * rstride);
#pragma unroll
                    for (int u = 0; u < 4; ++u) chan_merge(n, mu, m2, cnt, v[u].x, v[u].y);
                }
                for (; rb < pre.n_rb[s_]; rb += KL) {
                    const float2 v = *reinterpret_cast<const float2 *>(base + rb * rstride);
                    chan_merge(n, mu, m2, cnt, v.x, v.y);
                }
            }
            lane_rec[w * 3] = n;
            lane_rec[w * 3 + 1] = mu;
            lane_rec[w * 3 + 2] = m2;
        }
        __syncthreads();
        for (int si = tid; si < S_tot; si += blockDim.x) {
            float n = 0.f, mu = 0.f, m2 = 0.f;
            for (int k = 0; k < KL; ++k) chan_merge(n, mu, m2, lane_rec[(si * KL + k) * 3], lane_rec[(si * KL + k) * 3 + 1], lane_rec[(si * KL + k) * 3 + 2]);
            slot_rec[si * 3] = n;
            slot_rec[si * 3 + 1] = mu;
            slot_rec[si * 3 + 2] = m2;
        }
        __syncthreads();
        if (tid < g.G) {
            float n = 0.f, mu = 0.f, m2 = 0.f;
            int c = tid * g.cpg;
            const int cend = c + g.cpg;
            while (c < cend) {
                const int s_ = c < g.C1 ? 0 : 1;
                const int cl = c - pre.coff[s_];
                const int Ul = cl / pre.unit[s_];
                const int uend = min((Ul + 1) * pre.unit[s_], pre.nch[s_]);
                for (int tn = cl / pre.bno[s_]; tn <= (uend - 1) / pre.bno[s_]; ++tn) {
                    const int jj = Ul - (tn * pre.bno[s_]) / pre.unit[s_];
                    const int si = (s_ ? S0 : 0) + tn * pre.slots[s_] + jj;
                    chan_merge(n, mu, m2, slot_rec[si * 3], slot_rec[si * 3 + 1], slot_rec[si * 3 + 2]);
                }
                c = uend + pre.coff[s_];
            }
            mean[tid] = mu;
            rstd[tid] = rsqrtf(fmaxf(m2 / n, 0.f) + eps);  // biased variance (group_norm.py:48)
        }
        __syncthreads();
        gn_apply_rows<T, SILU>(x, x2, gamma, beta, y, g, rows_per_block, mean, rstd, ga_raw, be_raw);
        return;
    }
    // requests that do not depend on the statistics go out first (shift value, this thread's gamma / beta vectors of
    // the first channel pass): their round trips overlap the partial-sum reduction instead of following it
    const float sh_early = (tid < g.G) ? (float)*gn_src(x, x2, g, b, 0, tid * g.cpg) : 0.f;
    const int tx0 = tid % g.TXB;
    const bool pre_ok = tx0 < g.CX && tid < g.TXB * g.TY;
    u32x4 ga_raw = {0u, 0u, 0u, 0u}, be_raw = {0u, 0u, 0u, 0u};
    if (pre_ok && gamma) ga_raw = *reinterpret_cast<const u32x4 *>(gamma + tx0 * 8);
    if (pre_ok && beta) be_raw = *reinterpret_cast<const u32x4 *>(beta + tx0 * 8);
    // combine the per-split partial sums: (group, lane-of-8) then 8 -> 1, fixed order.
    if (tid < g.G * RS) {
        const int grp = tid / RS, j = tid % RS;
        float s1 = 0.f, s2 = 0.f;
        int s = j;
        for (; s + 3 * RS < nsplit; s += 4 * RS) {  // loads batched, summation order unchanged
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = *reinterpret_cast<const float2 *>(partial + (((int64_t)b * nsplit + s + u * RS) * g.G + grp) * 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s1 += v[u].x;
                s2 += v[u].y;
            }
        }
        for (; s < nsplit; s += RS) {
            const float *p = partial + (((int64_t)b * nsplit + s) * g.G + grp) * 2;
            s1 += p[0];
            s2 += p[1];
        }
        tmp[tid * 2] = s1;
        tmp[tid * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < g.G) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            s1 += tmp[(tid * RS + j) * 2];
            s2 += tmp[(tid * RS + j) * 2 + 1];
        }
        const float n = (float)g.HW * (float)g.cpg;
        const float sh = sh_early;
        const float m1 = s1 / n;
        const float var = fmaxf(s2 / n - m1 * m1, 0.f);  // biased variance (group_norm.py:48)
        mean[tid] = sh + m1;
        rstd[tid] = rsqrtf(var + eps);
    }
    __syncthreads();

    gn_apply_rows<T, SILU>(x, x2, gamma, beta, y, g, rows_per_block, mean, rstd, ga_raw, be_raw);
}

#ifdef SFAST_PROBES
// One-pass apply over producer-emitted statistics, second form of the merge prologue (round 5). The records of one sample -- per concat
// source n_rb row blocks x S slots of {mean, M2}, one contiguous block of memory -- are copied into LDS by all threads with coalesced
// loads (ONE exposed memory round trip), then LPG lanes per group (a power of two, groups never straddle a wave) combine the group's
// records with the weighted two-pass formulas
//      mean = sum(n_i * mean_i) / sum(n_i),     M2 = sum(M2_i) + sum(n_i * (mean_i - mean)^2)
// each lane over its strided share, xor butterflies inside the lane group: fixed order, bitwise reproducible, no serial Chan chain and no
// division per record. The first form (gn_nhwc_apply_kernel<.., PRE = true>) walks three dependent stages -- per-lane Chan merges, a
// 16-long serial merge per slot, a serial merge per group -- and costs ~3 us more than a LayerNorm over the same bytes (7.5 vs 4.65 us
// at [2, 320, 64, 64], profiles/r05_micro_norm_run3.log). Same contract and record layout; kept for sample sizes whose records
// exceed the LDS budget.
// MEASURED (profiles/r05_norm_ab_run4.log, 40 dependent launches in one hipGraph): 7.33 vs 7.59 us at [2, 320, 64, 64], 6.15 vs 6.38 us at
// [2, 640, 32, 32], 5.52 vs 6.74 us at [2, 1280, 8, 8] (where the plan runs gn_small anyway) -- the merge arithmetic is ~0.25 us of the ~3 us
// that separate this launch from a LayerNorm over the same bytes; the rest is the dependent round trip for the records and its barrier.
// Not measured on its own in the step: it lives in the PROBE build (-DSFAST_PROBES, SFAST_GN_MERGE=two-pass), the product keeps the first form.
template <typename T, bool SILU>
__global__ void gn_nhwc_apply2_kernel(const T *__restrict__ x, const T *__restrict__ x2, const T *__restrict__ gamma, const T *__restrict__ beta,
                                      T *__restrict__ y, GnGeom g, int rows_per_block, float eps, const GnPre pre, int lpg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *mean = smem;                                      // [G]
    float *rstd = smem + g.G;                                // [G]
    float2 *rec = reinterpret_cast<float2 *>(smem + 2 * g.G + (g.G & 1) * 2);  // [R0 + R1] (8-byte aligned)
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int tx0 = tid % g.TXB;
    const bool pre_ok = tx0 < g.CX && tid < g.TXB * g.TY;
    u32x4 ga_raw = {0u, 0u, 0u, 0u}, be_raw = {0u, 0u, 0u, 0u};
    if (pre_ok && gamma) ga_raw = *reinterpret_cast<const u32x4 *>(gamma + tx0 * 8);
    if (pre_ok && beta) be_raw = *reinterpret_cast<const u32x4 *>(beta + tx0 * 8);
    const int S0 = pre.tiles_n[0] * pre.slots[0];
    const int S1 = g.C1 < g.C ? pre.tiles_n[1] * pre.slots[1] : 0;
    const int R0 = S0 * pre.n_rb[0], R1 = S1 * pre.n_rb[1];
    for (int w = tid; w < R0 + R1; w += blockDim.x) {
        const int s_ = w < R0 ? 0 : 1;
        const int lw = w - (s_ ? R0 : 0);
        rec[w] = *reinterpret_cast<const float2 *>(pre.p[s_] + ((int64_t)b * (s_ ? R1 : R0) + lw) * 2);
    }
    __syncthreads();
    if (tid < g.G * lpg) {
        const int grp = tid / lpg, j = tid - grp * lpg;
        // every {slot, count} of this group, in channel order; f(record base of the slot, records stride, row blocks, count)
        auto for_each_slot = [&](auto &&f) {
            int c = grp * g.cpg;
            const int cend = c + g.cpg;
            while (c < cend) {
                const int s_ = c < g.C1 ? 0 : 1;
                const int cl = c - pre.coff[s_];
                const int Ul = cl / pre.unit[s_];
                const int uend = min((Ul + 1) * pre.unit[s_], pre.nch[s_]);
                for (int tn = cl / pre.bno[s_]; tn <= (uend - 1) / pre.bno[s_]; ++tn) {
                    const int jj = Ul - (tn * pre.bno[s_]) / pre.unit[s_];
                    const int lo = max(tn * pre.bno[s_], Ul * pre.unit[s_]), hi = min(min((tn + 1) * pre.bno[s_], pre.nch[s_]), (Ul + 1) * pre.unit[s_]);
                    if (hi > lo)
                        f((s_ ? R0 : 0) + tn * pre.slots[s_] + jj, s_ ? S1 : S0, pre.n_rb[s_], (float)(hi - lo) * (float)pre.rb_rows[s_]);
                }
                c = uend + pre.coff[s_];
            }
        };
        float sw = 0.f, sm = 0.f;
        for_each_slot([&](int base, int stride, int nrb, float cnt) {
            for (int rb = j; rb < nrb; rb += lpg) {
                sw += cnt;
                sm = fmaf(cnt, rec[base + rb * stride].x, sm);
            }
        });
        for (int off = lpg >> 1; off >= 1; off >>= 1) {
            sw += __shfl_xor(sw, off, 64);
            sm += __shfl_xor(sm, off, 64);
        }
        const float mu = sm / sw;
        float m2 = 0.f;
        for_each_slot([&](int base, int stride, int nrb, float cnt) {
            for (int rb = j; rb < nrb; rb += lpg) {
                const float2 r = rec[base + rb * stride];
                const float d = r.x - mu;
                m2 += fmaf(cnt * d, d, r.y);
            }
        });
        for (int off = lpg >> 1; off >= 1; off >>= 1) m2 += __shfl_xor(m2, off, 64);
        if (j == 0) {
            mean[grp] = mu;
            rstd[grp] = rsqrtf(fmaxf(m2 / sw, 0.f) + eps);  // biased variance (group_norm.py:48)
        }
    }
    __syncthreads();
    gn_apply_rows<T, SILU>(x, x2, gamma, beta, y, g, rows_per_block, mean, rstd, ga_raw, be_raw);
}

#endif  // SFAST_PROBES

template <typename T, bool SILU>
__device__ __forceinline__ void gn_apply_rows(const T *__restrict__ x, const T *__restrict__ x2, const T *__restrict__ gamma,
                                              const T *__restrict__ beta, T *__restrict__ y, const GnGeom &g, int rows_per_block,
                                              const float *mean, const float *rstd, u32x4 ga_raw, u32x4 be_raw) {
    const int NT = g.TXB * g.TY;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int tx = tid % g.TXB, ty = tid / g.TXB;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(g.HW, r0 + rows_per_block);
    if (tid >= NT) return;
    for (int cxb = 0; cxb < g.CX; cxb += g.TXB) {
        const int cx = cxb + tx;
        if (cx >= g.CX) continue;
        const int c = cx * 8;
        float a[8], bb[8], gaf[8], bef[8];
        if (cxb != 0) {  // later channel passes (C > 8 * 512 only): fetch here
            if (gamma) ga_raw = *reinterpret_cast<const u32x4 *>(gamma + c);
            if (beta) be_raw = *reinterpret_cast<const u32x4 *>(beta + c);
        }
        unpack8<T>(ga_raw, gaf);
        unpack8<T>(be_raw, bef);
        const int g0 = c / g.cpg;               // an 8-channel chunk touches at most two groups (cpg >= 8)
        const int nb = (g0 + 1) * g.cpg - c;    // channels of the chunk that belong to g0
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int grp = g0 + (i >= nb ? 1 : 0);
            const float ga = gamma ? gaf[i] : 1.f;
            const float be = beta ? bef[i] : 0.f;
            a[i] = rstd[grp] * ga;
            bb[i] = be - mean[grp] * a[i];
        }
        auto finish = [&](const u32x4 &raw, int r) {
            float f[8];
            unpack8<T>(raw, f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = fmaf(f[i], a[i], bb[i]);
                if (SILU) v = act_silu(v);
                f[i] = v;
            }
            *reinterpret_cast<u32x4 *>(y + ((int64_t)b * g.HW + r) * g.C + c) = pack8<T>(f);
        };
        int r = r0 + ty;
        for (; r + 3 * g.TY < r1; r += 4 * g.TY) {  // four row loads in flight per thread
            u32x4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const u32x4 *>(gn_src(x, x2, g, b, r + u * g.TY, c));
#pragma unroll
            for (int u = 0; u < 4; ++u) finish(raw[u], r + u * g.TY);
        }
        for (; r < r1; r += g.TY) finish(*reinterpret_cast<const u32x4 *>(gn_src(x, x2, g, b, r, c)), r);
    }
}

// =================================================================================================
// Generic path: any C/G, NHWC (with concat) or NCHW, f16 / bf16 / f32. One workgroup per (n, g).
// =================================================================================================
template <typename T>
__device__ __forceinline__ int64_t gn_generic_index(int layout, int n, int64_t i, int HW, int C, int cpg,
                                                    int grp, int &c_out) {
    if (layout == SFAST_NHWC) {
        const int cc = (int)(i % cpg);
        const int64_t p = i / cpg;
        c_out = grp * cpg + cc;
        return p;  // caller resolves the concat source
    }
    const int64_t p = i % HW;
    const int cc = (int)(i / HW);
    c_out = grp * cpg + cc;
    return p;
}

template <typename T>
__global__ void gn_generic_kernel(const T *__restrict__ x, const T *__restrict__ x2,
                                  const T *__restrict__ gamma, const T *__restrict__ beta,
                                  T *__restrict__ y, int layout, int HW, int C, int C1, int cpg, float eps,
                                  int act) {
    __shared__ float red[2][256];
    __shared__ float stat[2];
    const int tid = threadIdx.x;
    const int grp = blockIdx.x, n = blockIdx.y;
    const int64_t total = (int64_t)HW * cpg;
    auto load = [&](int64_t p, int c) -> float {
        if (layout == SFAST_NHWC) {
            if (c < C1) return Elem<T>::to_f32(x[((int64_t)n * HW + p) * C1 + c]);
            return Elem<T>::to_f32(x2[((int64_t)n * HW + p) * (C - C1) + (c - C1)]);
        }
        return Elem<T>::to_f32(x[((int64_t)n * C + c) * HW + p]);
    };
    const float sh = load(0, grp * cpg);
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = tid; i < total; i += blockDim.x) {
        int c;
        const int64_t p = gn_generic_index<T>(layout, n, i, HW, C, cpg, grp, c);
        const float d = load(p, c) - sh;
        s1 += d;
        s2 += d * d;
    }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] += red[0][tid + s];
            red[1][tid] += red[1][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float nn = (float)total;
        const float m1 = red[0][0] / nn;
        const float var = fmaxf(red[1][0] / nn - m1 * m1, 0.f);
        stat[0] = sh + m1;
        stat[1] = rsqrtf(var + eps);
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    for (int64_t i = tid; i < total; i += blockDim.x) {
        int c;
        const int64_t p = gn_generic_index<T>(layout, n, i, HW, C, cpg, grp, c);
        const float ga = gamma ? Elem<T>::to_f32(gamma[c]) : 1.f;
        const float be = beta ? Elem<T>::to_f32(beta[c]) : 0.f;
        float v = (load(p, c) - mean) * rstd * ga + be;
        if (act == SFAST_ACT_SILU) v = act_silu(v);
        const int64_t oi = (layout == SFAST_NHWC) ? ((int64_t)n * HW + p) * C + c
                                                  : ((int64_t)n * C + c) * HW + p;
        y[oi] = Elem<T>::from_f32(v);
    }
}

// =================================================================================================
// LayerNorm: one wave per row. Fast path N % 8 == 0 (16-B loads, row cached in registers up to
// N = 4096); generic path scalar.
// =================================================================================================
template <typename T, int MAXCH>
__global__ void __launch_bounds__(256) ln_rows_kernel(const T *__restrict__ x, const T *__restrict__ gamma,
                                                      const T *__restrict__ beta, T *__restrict__ y,
                                                      int M, int N, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int nch = N / 8;
    const T *xr = x + (int64_t)row * N;
    u32x4 cache[MAXCH];
    // gamma / beta do not depend on the statistics: requested together with the row (for rows up to 2048 wide), so their round
    // trip overlaps the two reductions instead of following them -- a LayerNorm launch of the UNet is a chain of exposed latencies
    constexpr bool PRE = MAXCH <= 4;
    u32x4 gar[PRE ? MAXCH : 1], ber[PRE ? MAXCH : 1];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = lane + j * 64;
        if (ch < nch) cache[j] = *reinterpret_cast<const u32x4 *>(xr + ch * 8);
    }
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int ch = lane + j * 64;
            gar[j] = u32x4{0u, 0u, 0u, 0u};
            ber[j] = u32x4{0u, 0u, 0u, 0u};
            if (ch < nch && gamma) gar[j] = *reinterpret_cast<const u32x4 *>(gamma + ch * 8);
            if (ch < nch && beta) ber[j] = *reinterpret_cast<const u32x4 *>(beta + ch * 8);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = lane + j * 64;
        if (ch < nch) {
            float f[8];
            unpack8<T>(cache[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += f[i];
        }
    }
    const float mean = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = lane + j * 64;
        if (ch < nch) {
            float f[8];
            unpack8<T>(cache[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = f[i] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)N + eps);
    T *yr = y + (int64_t)row * N;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = lane + j * 64;
        if (ch < nch) {
            float f[8], ga[8], be[8];
            unpack8<T>(cache[j], f);
            if (gamma) {
                if constexpr (PRE) unpack8<T>(gar[j], ga);
                else unpack8<T>(*reinterpret_cast<const u32x4 *>(gamma + ch * 8), ga);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) ga[i] = 1.f;
            }
            if (beta) {
                if constexpr (PRE) unpack8<T>(ber[j], be);
                else unpack8<T>(*reinterpret_cast<const u32x4 *>(beta + ch * 8), be);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) be[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * ga[i] + be[i];
            *reinterpret_cast<u32x4 *>(yr + ch * 8) = pack8<T>(f);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) ln_generic_kernel(const T *__restrict__ x, const T *__restrict__ gamma,
                                                         const T *__restrict__ beta, T *__restrict__ y,
                                                         int M, int N, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const T *xr = x + (int64_t)row * N;
    float s = 0.f;
    for (int i = lane; i < N; i += 64) s += Elem<T>::to_f32(xr[i]);
    const float mean = wave_sum(s) / (float)N;
    float q = 0.f;
    for (int i = lane; i < N; i += 64) {
        const float d = Elem<T>::to_f32(xr[i]) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)N + eps);
    T *yr = y + (int64_t)row * N;
    for (int i = lane; i < N; i += 64) {
        const float ga = gamma ? Elem<T>::to_f32(gamma[i]) : 1.f;
        const float be = beta ? Elem<T>::to_f32(beta[i]) : 0.f;
        yr[i] = Elem<T>::from_f32((Elem<T>::to_f32(xr[i]) - mean) * rstd * ga + be);
    }
}

// =================================================================================================
// Small-group single-kernel path (NHWC): one workgroup per (sample, group) keeps the whole group -- HW x C/G
// halves, at most 32 KB -- in LDS between the statistics pass and the normalisation pass: ONE launch, one read and
// one write of the tensor. The two-kernel path above costs two ~5.7 us launches however small the tensor is (every
// kernel of the step's graph has a ~4 us floor); at the 32x32 / 16x16 / 8x8 levels of the UNet that was the whole cost
// of a GroupNorm. Same arithmetic as the large path (shifted sums, biased variance, fixed reduction order).
// Granularity is 2 channels (one dword): C/G and C1 only need to be even.
// =================================================================================================
template <typename T, bool SILU>
__global__ void __launch_bounds__(512) gn_small_kernel(const T *__restrict__ x, const T *__restrict__ x2, const T *__restrict__ gamma,
                                                       const T *__restrict__ beta, T *__restrict__ y, int HW, int C, int C1, int cpg,
                                                       float eps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t gsm[];  // [HW][cpg/2] dwords
    __shared__ float red[2][8];
    __shared__ uint32_t gb[2][128];  // gamma / beta dwords of this group (cpg <= 256), requested before the statistics pass
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.x, b = blockIdx.y;
    const int dpg = cpg / 2;           // dwords per pixel of this group
    const int total = HW * dpg;
    const int c0 = grp * cpg;          // first channel of the group
    const bool gb_lds = dpg <= 128;
    if (gb_lds && tid < dpg) {
        gb[0][tid] = gamma ? *reinterpret_cast<const uint32_t *>(gamma + c0 + 2 * tid) : 0u;
        gb[1][tid] = beta ? *reinterpret_cast<const uint32_t *>(beta + c0 + 2 * tid) : 0u;
    }
    const int C2 = C - C1;
    typedef typename Elem<T>::vec2 vec2;
    // shift = first element of the group (as in the large path)
    const float sh = (c0 < C1) ? Elem<T>::to_f32(x[(int64_t)b * HW * C1 + c0]) : Elem<T>::to_f32(x2[(int64_t)b * HW * C2 + (c0 - C1)]);
    float s1 = 0.f, s2 = 0.f;
    // element i = tid + 512*k  <->  (pixel p, dword j): advanced incrementally (one division per thread, not per element);
    // eight independent loads in flight per thread (a one-load-per-iteration loop is a chain of exposed round trips)
    const int step_p = 512 / dpg, step_j = 512 - step_p * dpg;
    int p = tid / dpg, j = tid - p * dpg;
    auto src_of = [&](int pp, int jj) -> const T * {
        const int c = c0 + 2 * jj;
        return (c < C1) ? x + ((int64_t)b * HW + pp) * C1 + c : x2 + ((int64_t)b * HW + pp) * C2 + (c - C1);
    };
    auto advance = [&](int &pp, int &jj) {
        pp += step_p;
        jj += step_j;
        if (jj >= dpg) {
            jj -= dpg;
            ++pp;
        }
    };
    int i = tid;
    for (; i + 7 * 512 < total; i += 8 * 512) {
        uint32_t raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            raw[u] = *reinterpret_cast<const uint32_t *>(src_of(p, j));
            advance(p, j);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            gsm[i + u * 512] = raw[u];
            const vec2 v = __builtin_bit_cast(vec2, raw[u]);
            const float d0 = (float)v[0] - sh, d1 = (float)v[1] - sh;
            s1 += d0 + d1;
            s2 += d0 * d0 + d1 * d1;
        }
    }
    for (; i < total; i += 512) {
        const uint32_t raw = *reinterpret_cast<const uint32_t *>(src_of(p, j));
        advance(p, j);
        gsm[i] = raw;
        const vec2 v = __builtin_bit_cast(vec2, raw);
        const float d0 = (float)v[0] - sh, d1 = (float)v[1] - sh;
        s1 += d0 + d1;
        s2 += d0 * d0 + d1 * d1;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        red[0][wave] = s1;
        red[1][wave] = s2;
    }
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        t1 += red[0][w];
        t2 += red[1][w];
    }
    const float n = (float)HW * (float)cpg;
    const float m1 = t1 / n;
    const float var = fmaxf(t2 / n - m1 * m1, 0.f);
    const float mean = sh + m1;
    const float rstd = rsqrtf(var + eps);
    p = tid / dpg;
    j = tid - p * dpg;
    for (int k = tid; k < total; k += 512) {
        const int c = c0 + 2 * j;
        const vec2 v = __builtin_bit_cast(vec2, gsm[k]);
        float ga0 = 1.f, ga1 = 1.f, be0 = 0.f, be1 = 0.f;
        if (gamma) {
            const vec2 g2 = __builtin_bit_cast(vec2, gb_lds ? gb[0][j] : *reinterpret_cast<const uint32_t *>(gamma + c));
            ga0 = (float)g2[0];
            ga1 = (float)g2[1];
        }
        if (beta) {
            const vec2 b2 = __builtin_bit_cast(vec2, gb_lds ? gb[1][j] : *reinterpret_cast<const uint32_t *>(beta + c));
            be0 = (float)b2[0];
            be1 = (float)b2[1];
        }
        float o0 = ((float)v[0] - mean) * rstd * ga0 + be0;
        float o1 = ((float)v[1] - mean) * rstd * ga1 + be1;
        if (SILU) {
            o0 = act_silu(o0);
            o1 = act_silu(o1);
        }
        vec2 o;
        o[0] = Elem<T>::from_f32(o0);
        o[1] = Elem<T>::from_f32(o1);
        *reinterpret_cast<uint32_t *>(y + ((int64_t)b * HW + p) * C + c) = __builtin_bit_cast(uint32_t, o);
        advance(p, j);
    }
}

// applicable: NHWC f16/bf16, even C/G and C1, group at most 32 KB (N x G workgroups of that size finish in a few
// load round trips), and the tensor is small enough that the
// two-kernel path would be launch-bound anyway (<= 4 MB)
static bool gn_small_ok(const sfast_gn_params *p) {
    if (p->layout != SFAST_NHWC || p->dtype == SFAST_F32 || p->G <= 0 || p->C % p->G) return false;
    const int cpg = p->C / p->G;
    if ((cpg & 1) || (p->C1 & 1)) return false;
    const int64_t group_bytes = (int64_t)p->HW * cpg * 2;
    const int64_t tensor_bytes = (int64_t)(g_batch_ref > 0 ? g_batch_ref : p->N) * p->HW * p->C * 2;  // (SFAST_BATCH_INVARIANT: common.h)
    return group_bytes <= 32 * 1024 && tensor_bytes <= (4 << 20);
}

template <typename T>
static int gn_launch_small(const void *x, const void *x2, const void *gamma, const void *beta, void *y, const sfast_gn_params *p,
                           hipStream_t st) {
    const int cpg = p->C / p->G;
    const size_t smem = (size_t)p->HW * cpg * 2;
    const dim3 grid(p->G, p->N);
    if (p->act == SFAST_ACT_SILU)
        hipLaunchKernelGGL((gn_small_kernel<T, true>), grid, dim3(512), smem, st, (const T *)x, (const T *)x2, (const T *)gamma,
                           (const T *)beta, (T *)y, p->HW, p->C, p->C1, cpg, p->eps);
    else
        hipLaunchKernelGGL((gn_small_kernel<T, false>), grid, dim3(512), smem, st, (const T *)x, (const T *)x2, (const T *)gamma,
                           (const T *)beta, (T *)y, p->HW, p->C, p->C1, cpg, p->eps);
    return check_launch("group_norm_small");
}

// (Round 5 measured and REMOVED a lane-group LayerNorm -- LPR lanes per row x 5 chunks each, 64 / LPR rows per wave, every lane busy for the
// 320 / 640 / 1280-wide rows: 4.9 vs 4.5 us at [8192, 320], 4.6 vs 3.4 us at [2048, 640], 180.2 vs 183.4 it/s in the step
// (profiles/r05_norm_ab_run4.log). One row per wave with 8x more waves in flight hides the exposed load -> reduce -> store chain better
// than fewer, fatter waves with 212 registers.)

// ---- host-side planning ----------------------------------------------------------------------
struct GnPlan {
    bool fast;
    GnGeom g;
    int nsplit, rows_stats, rows_apply, napply;
};

// workgroups of the one-pass apply kernel (sfast_hip_group_norm_apply), over all samples: every workgroup merges the sample's
// statistics records in its prologue, so fewer, fatter workgroups trade prologue repeats against streaming width.
// SFAST_GN_APPLY_WGS overrides (A/B knob).
static int gn_apply_wgs_target() {
    static int v = [] {
        const char *e = getenv("SFAST_GN_APPLY_WGS");
        const int n = e ? atoi(e) : 0;
        return n > 0 ? n : 256;
    }();
    return v;
}

static GnPlan gn_plan(const sfast_gn_params *p, int apply_wgs = 256) {
    GnPlan pl{};
    const int cpg = p->G > 0 ? p->C / p->G : 0;
    pl.fast = p->layout == SFAST_NHWC && p->dtype != SFAST_F32 && p->G > 0 && p->C % p->G == 0 &&
              p->C % 8 == 0 && p->C1 % 8 == 0 && (cpg >= 8 || cpg == 4) && p->G * 8 <= 256 && p->HW > 0;
    if (!pl.fast) return pl;
    GnGeom &g = pl.g;
    g.HW = p->HW;
    g.C = p->C;
    g.C1 = p->C1;
    g.cpg = cpg;
    g.G = p->G;
    g.CX = p->C / 8;
    g.TXB = g.CX < 512 ? g.CX : 512;
    g.TY = 512 / g.TXB;
    if (g.TY < 1) g.TY = 1;
    if (g.TY > p->HW) g.TY = p->HW;
    const int max_split = ceil_div(p->HW, g.TY);
    // ~one workgroup per CU in the apply pass, half that in the stats pass: every apply workgroup re-reads
    // all G x nsplit partial sums in its prologue (16 KB at 64 splits, two batched loads per thread); with
    // 128 splits and 684 apply workgroups that prologue traffic exceeded the tensor itself.
    const int nb = g_batch_ref > 0 ? g_batch_ref : p->N;  // SFAST_BATCH_INVARIANT: the statistics partition of a sample must not follow the batch
    int want = ceil_div(256, nb);
    if (want > 64) want = 64;
    pl.nsplit = want < max_split ? want : max_split;
    if (pl.nsplit < 1) pl.nsplit = 1;
    pl.rows_stats = ceil_div(p->HW, pl.nsplit);
    pl.nsplit = ceil_div(p->HW, pl.rows_stats);
    int wanta = ceil_div(apply_wgs, nb);  // (rows per apply workgroup select the 4-row-batched or the single-row loop body: not bit-identical code)
    int na = wanta < max_split ? wanta : max_split;
    if (na < 1) na = 1;
    pl.rows_apply = ceil_div(p->HW, na);
    pl.napply = ceil_div(p->HW, pl.rows_apply);
    return pl;
}

template <typename T>
static int gn_launch_fast(const void *x, const void *x2, const void *gamma, const void *beta, void *y,
                          const sfast_gn_params *p, const GnPlan &pl, float *ws, hipStream_t st) {
    const GnGeom &g = pl.g;
    const int NT = g.TXB * g.TY;
    int threads = ((NT + 63) / 64) * 64;
    if (threads < g.G * 8) threads = ((g.G * 8 + 63) / 64) * 64;  // the partial combine uses G*8 threads
    const size_t smem_stats = (size_t)(NT * 4 + g.G * 2) * sizeof(float);
    hipLaunchKernelGGL(gn_nhwc_stats_kernel<T>, dim3(pl.nsplit, p->N), dim3(threads), smem_stats, st,
                       (const T *)x, (const T *)x2, ws, g, pl.rows_stats, pl.nsplit);
    const size_t smem_apply = (size_t)(2 * g.G + g.G * 24) * sizeof(float);
    const GnPre none{};
    if (p->act == SFAST_ACT_SILU)
        hipLaunchKernelGGL((gn_nhwc_apply_kernel<T, true, false>), dim3(pl.napply, p->N), dim3(threads), smem_apply,
                           st, (const T *)x, (const T *)x2, (const T *)gamma, (const T *)beta, (T *)y, ws, g,
                           pl.rows_apply, pl.nsplit, p->eps, none);
    else
        hipLaunchKernelGGL((gn_nhwc_apply_kernel<T, false, false>), dim3(pl.napply, p->N), dim3(threads), smem_apply,
                           st, (const T *)x, (const T *)x2, (const T *)gamma, (const T *)beta, (T *)y, ws, g,
                           pl.rows_apply, pl.nsplit, p->eps, none);
    return check_launch("group_norm_nhwc");
}

#ifdef SFAST_PROBES
// probe build only: SFAST_GN_MERGE=two-pass selects round 5's lane-group merge prologue (gn_nhwc_apply2_kernel) for the A/B
static bool gn_merge_two_pass() {
    static const bool v = [] {
        const char *e = getenv("SFAST_GN_MERGE");
        return e && e[0] == 't';
    }();
    return v;
}
#endif

// one normalisation pass over statistics the producers left behind
template <typename T>
static int gn_launch_pre(const void *x, const void *x2, const void *gamma, const void *beta, void *y, const sfast_gn_params *p,
                         const GnPlan &pl, const GnPre &pre, hipStream_t st) {
    const GnGeom &g = pl.g;
    const int NT = g.TXB * g.TY;
    int threads = ((NT + 63) / 64) * 64;
    if (threads < g.G * 8) threads = ((g.G * 8 + 63) / 64) * 64;
    GnPre pr = pre;
    const int s_tot = pr.tiles_n[0] * pr.slots[0] + (g.C1 < g.C ? pr.tiles_n[1] * pr.slots[1] : 0);
    pr.kl = threads / (s_tot > 0 ? s_tot : 1);
    if (pr.kl < 1) pr.kl = 1;
    if (pr.kl > 16) pr.kl = 16;
#ifdef SFAST_PROBES
    // second-form prologue (gn_nhwc_apply2_kernel): all records of the sample in LDS, lane groups per GroupNorm group
    const int64_t recs = (int64_t)pr.tiles_n[0] * pr.slots[0] * pr.n_rb[0] + (g.C1 < g.C ? (int64_t)pr.tiles_n[1] * pr.slots[1] * pr.n_rb[1] : 0);
    const size_t smem2 = (size_t)(2 * g.G + 2) * sizeof(float) + (size_t)recs * 8;
    int lpg = 64;
    while (lpg > 1 && lpg * g.G > threads) lpg >>= 1;
    if (gn_merge_two_pass() && smem2 <= 48 * 1024 && lpg * g.G <= threads) {
        if (p->act == SFAST_ACT_SILU)
            hipLaunchKernelGGL((gn_nhwc_apply2_kernel<T, true>), dim3(pl.napply, p->N), dim3(threads), smem2, st, (const T *)x, (const T *)x2,
                               (const T *)gamma, (const T *)beta, (T *)y, g, pl.rows_apply, p->eps, pr, lpg);
        else
            hipLaunchKernelGGL((gn_nhwc_apply2_kernel<T, false>), dim3(pl.napply, p->N), dim3(threads), smem2, st, (const T *)x, (const T *)x2,
                               (const T *)gamma, (const T *)beta, (T *)y, g, pl.rows_apply, p->eps, pr, lpg);
        return check_launch("group_norm_apply2");
    }
#endif
    const size_t smem_apply = (size_t)(2 * g.G + (size_t)s_tot * pr.kl * 3 + (size_t)s_tot * 3) * sizeof(float);
    if (smem_apply > 60 * 1024) {
        set_error("group_norm_apply: %d record slots per sample exceed the merge scratch", s_tot);
        return SFAST_ERR_UNSUPPORTED;
    }
    if (p->act == SFAST_ACT_SILU)
        hipLaunchKernelGGL((gn_nhwc_apply_kernel<T, true, true>), dim3(pl.napply, p->N), dim3(threads), smem_apply, st, (const T *)x,
                           (const T *)x2, (const T *)gamma, (const T *)beta, (T *)y, (const float *)nullptr, g, pl.rows_apply, 0, p->eps, pr);
    else
        hipLaunchKernelGGL((gn_nhwc_apply_kernel<T, false, true>), dim3(pl.napply, p->N), dim3(threads), smem_apply, st, (const T *)x,
                           (const T *)x2, (const T *)gamma, (const T *)beta, (T *)y, (const float *)nullptr, g, pl.rows_apply, 0, p->eps, pr);
    return check_launch("group_norm_apply");
}

template <typename T>
static int gn_launch_generic(const void *x, const void *x2, const void *gamma, const void *beta, void *y,
                             const sfast_gn_params *p, hipStream_t st) {
    hipLaunchKernelGGL(gn_generic_kernel<T>, dim3(p->G, p->N), dim3(256), 0, st, (const T *)x, (const T *)x2,
                       (const T *)gamma, (const T *)beta, (T *)y, p->layout, p->HW, p->C, p->C1, p->C / p->G,
                       p->eps, p->act);
    return check_launch("group_norm_generic");
}


// ---- row softmax ------------------------------------------------------------------------------------
// One workgroup per row, 16-byte chunks, the row is held in registers between the three passes (max, exp-sum,
// normalise): one read and one write of the row. Fixed reduction order -> bitwise reproducible.
template <typename T, int MAXCH>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const T *x, T *y, int M, int N, int64_t ldx,
                                                           int64_t ldy, float scale_log2e) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = N / 8;
    const T *xr = x + (int64_t)row * ldx;
    u32x4 cache[MAXCH];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = tid + j * 256;
        if (ch < nch) cache[j] = *reinterpret_cast<const u32x4 *>(xr + ch * 8);
    }
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = tid + j * 256;
        if (ch < nch) {
            float f[8];
            unpack8<T>(cache[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) mx = fmaxf(mx, f[i]);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mc = mx * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = tid + j * 256;
        if (ch < nch) {
            float f[8];
            unpack8<T>(cache[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += __builtin_amdgcn_exp2f(fmaf(f[i], scale_log2e, -mc));
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (((red[4] + red[5]) + red[6]) + red[7]);
    T *yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = tid + j * 256;
        if (ch < nch) {
            float f[8];
            unpack8<T>(cache[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_exp2f(fmaf(f[i], scale_log2e, -mc)) * inv;
            *reinterpret_cast<u32x4 *>(yr + ch * 8) = pack8<T>(f);
        }
    }
}

template <typename T>
static int softmax_launch(const void *x, void *y, const sfast_softmax_params *p, hipStream_t st) {
    const float c = p->scale * 1.44269504088896340736f;
    const int nch = p->N / 8;
    const dim3 grid(p->M), block(256);
#define SM_LAUNCH(MC) hipLaunchKernelGGL((softmax_rows_kernel<T, MC>), grid, block, 0, st, (const T *)x, (T *)y, p->M, p->N, p->ldx, p->ldy, c)
    if (nch <= 256) SM_LAUNCH(1);
    else if (nch <= 512) SM_LAUNCH(2);
    else if (nch <= 1024) SM_LAUNCH(4);
    else if (nch <= 2048) SM_LAUNCH(8);
    else SM_LAUNCH(16);
#undef SM_LAUNCH
    return check_launch("softmax_rows");
}

}  // namespace sfast

using namespace sfast;

extern "C" size_t sfast_hip_group_norm_workspace_bytes(const sfast_gn_params *p) {
    if (!p) return 0;
    GnPlan pl = gn_plan(p);
    if (!pl.fast) return 0;
    return (size_t)p->N * pl.nsplit * p->G * 2 * sizeof(float);
}

extern "C" int sfast_hip_group_norm(const void *x, const void *x2, const void *gamma, const void *beta,
                                    void *y, const sfast_gn_params *p, void *workspace,
                                    size_t workspace_bytes, sfast_stream_t stream) {
    SFAST_REQUIRE(p && x && y, SFAST_ERR_INVALID, "group_norm: null argument");
    SFAST_REQUIRE(p->N > 0 && p->C > 0 && p->HW > 0 && p->G > 0 && p->C % p->G == 0, SFAST_ERR_INVALID,
                  "group_norm: bad shape N=%d C=%d HW=%d G=%d", p->N, p->C, p->HW, p->G);
    SFAST_REQUIRE(p->act == SFAST_ACT_NONE || p->act == SFAST_ACT_SILU, SFAST_ERR_UNSUPPORTED,
                  "group_norm: act %d unsupported", p->act);
    SFAST_REQUIRE(p->C1 > 0 && p->C1 <= p->C, SFAST_ERR_INVALID, "group_norm: bad C1=%d", p->C1);
    SFAST_REQUIRE(p->C1 == p->C || (x2 && p->layout == SFAST_NHWC), SFAST_ERR_INVALID,
                  "group_norm: concat needs x2 and NHWC");
    hipStream_t st = (hipStream_t)stream;
    GnPlan pl = gn_plan(p, gn_apply_wgs_target());
    const bool ptr_ok = aligned16(x) && aligned16(y) && (p->C1 == p->C || aligned16(x2)) && (!gamma || aligned16(gamma)) &&
                        (!beta || aligned16(beta));
    const bool al4 = ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)(x2 ? x2 : x)) | ((uintptr_t)(gamma ? gamma : x)) |
                       ((uintptr_t)(beta ? beta : x))) & 3) == 0;
    if (gn_small_ok(p) && al4) {
        set_kernel_name("gn_small[cpg=%d,hw=%d]", p->C / p->G, p->HW);
        if (p->dtype == SFAST_F16) return gn_launch_small<f16>(x, x2, gamma, beta, y, p, st);
        return gn_launch_small<bf16>(x, x2, gamma, beta, y, p, st);
    }
    if (pl.fast && ptr_ok) {
        const size_t need = sfast_hip_group_norm_workspace_bytes(p);
        SFAST_REQUIRE(workspace && workspace_bytes >= need, SFAST_ERR_WORKSPACE,
                      "group_norm: workspace %zu < %zu", workspace_bytes, need);
        set_kernel_name("gn_nhwc[TXB=%d,TY=%d,split=%d,apply=%d]", pl.g.TXB, pl.g.TY, pl.nsplit, pl.napply);
        if (p->dtype == SFAST_F16)
            return gn_launch_fast<f16>(x, x2, gamma, beta, y, p, pl, (float *)workspace, st);
        return gn_launch_fast<bf16>(x, x2, gamma, beta, y, p, pl, (float *)workspace, st);
    }
    set_kernel_name("gn_generic");
    switch (p->dtype) {
    case SFAST_F16: return gn_launch_generic<f16>(x, x2, gamma, beta, y, p, st);
    case SFAST_BF16: return gn_launch_generic<bf16>(x, x2, gamma, beta, y, p, st);
    case SFAST_F32: return gn_launch_generic<float>(x, x2, gamma, beta, y, p, st);
    }
    set_error("group_norm: bad dtype %d", p->dtype);
    return SFAST_ERR_UNSUPPORTED;
}

extern "C" int sfast_hip_group_norm_apply(const void *x, const void *x2, const void *gamma, const void *beta, void *y,
                                          const sfast_gn_params *p, const void *stats1, const sfast_gn_stats_layout *l1,
                                          const void *stats2, const sfast_gn_stats_layout *l2, sfast_stream_t stream) {
    SFAST_REQUIRE(p && x && y && stats1 && l1, SFAST_ERR_INVALID, "group_norm_apply: null argument");
    SFAST_REQUIRE(p->N > 0 && p->C > 0 && p->HW > 0 && p->G > 0 && p->C % p->G == 0, SFAST_ERR_INVALID,
                  "group_norm_apply: bad shape N=%d C=%d HW=%d G=%d", p->N, p->C, p->HW, p->G);
    SFAST_REQUIRE(p->act == SFAST_ACT_NONE || p->act == SFAST_ACT_SILU, SFAST_ERR_UNSUPPORTED, "group_norm_apply: act %d unsupported", p->act);
    SFAST_REQUIRE(p->C1 > 0 && p->C1 <= p->C && (p->C1 == p->C || (x2 && stats2 && l2)), SFAST_ERR_INVALID,
                  "group_norm_apply: a concat input needs x2 and its statistics");
    GnPlan pl = gn_plan(p);
    const bool ptr_ok = aligned16(x) && aligned16(y) && (p->C1 == p->C || aligned16(x2)) && (!gamma || aligned16(gamma)) && (!beta || aligned16(beta));
    SFAST_REQUIRE(pl.fast && ptr_ok && p->layout == SFAST_NHWC, SFAST_ERR_UNSUPPORTED,
                  "group_norm_apply: needs the NHWC fast path (f16/bf16, C and C1 multiples of 8, 16-byte aligned)");
    const int cpg = p->C / p->G;
    GnPre pre{};
    const sfast_gn_stats_layout *ls[2] = {l1, p->C1 == p->C ? l1 : l2};
    const void *ps[2] = {stats1, p->C1 == p->C ? stats1 : stats2};
    const int nch[2] = {p->C1, p->C - p->C1}, coff[2] = {0, p->C1};
    for (int s = 0; s < 2; ++s) {
        const sfast_gn_stats_layout *l = ls[s];
        SFAST_REQUIRE(l->unit > 0 && cpg % l->unit == 0 && coff[s] % l->unit == 0 && l->rb_rows > 0 && l->n_rb > 0 && l->bno > 0 &&
                          l->tiles_n > 0 && l->slots > 0 && (int64_t)l->rb_rows * l->n_rb == (int64_t)p->N * p->HW && l->n_rb % p->N == 0,
                      SFAST_ERR_INVALID, "group_norm_apply: statistics layout %d does not tile this tensor (unit %d, C/G %d)", s, l->unit, cpg);
        pre.p[s] = (const float *)ps[s];
        pre.rb_rows[s] = l->rb_rows;
        pre.n_rb[s] = l->n_rb / p->N;
        pre.bno[s] = l->bno;
        pre.tiles_n[s] = l->tiles_n;
        pre.slots[s] = l->slots;
        pre.unit[s] = l->unit;
        pre.nch[s] = nch[s] > 0 ? nch[s] : 1;
        pre.coff[s] = coff[s];
    }
    set_kernel_name("gn_apply[TXB=%d,TY=%d,apply=%d]", pl.g.TXB, pl.g.TY, pl.napply);
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == SFAST_F16) return gn_launch_pre<f16>(x, x2, gamma, beta, y, p, pl, pre, st);
    return gn_launch_pre<bf16>(x, x2, gamma, beta, y, p, pl, pre, st);
}

// Wide rows (4096 < N <= 32768, N % 8 == 0): one 256-thread workgroup per row, the row cached in registers (MAXCH 16-byte chunks
// per thread), two exact passes (mean, then squared deviations) reduced across the four waves through LDS. Not a UNet shape (those
// are <= 1280 wide); it is the reference's own self-check shape, (1151, 8192) (triton/ops/layer_norm.py:522), which used to fall to
// the scalar generic kernel -- measured 166 us against 39 us for the reference's Triton kernel on the same box
// (profiles/r04_ref_triton_tests_run1.log).
template <typename T, int MAXCH>
__global__ void __launch_bounds__(256) ln_wide_kernel(const T *__restrict__ x, const T *__restrict__ gamma, const T *__restrict__ beta,
                                                      T *__restrict__ y, int M, int N, float eps) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int nch = N / 8;
    const T *xr = x + (int64_t)row * N;
    u32x4 cache[MAXCH];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = tid + j * 256;
        if (ch < nch) cache[j] = *reinterpret_cast<const u32x4 *>(xr + ch * 8);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        if (tid + j * 256 < nch) {
            float f[8];
            unpack8<T>(cache[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += f[i];
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        if (tid + j * 256 < nch) {
            float f[8];
            unpack8<T>(cache[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = f[i] - mean;
                q += d * d;
            }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)N + eps);
    T *yr = y + (int64_t)row * N;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int ch = tid + j * 256;
        if (ch < nch) {
            float f[8], ga[8], be[8];
            unpack8<T>(cache[j], f);
            if (gamma) {
                unpack8<T>(*reinterpret_cast<const u32x4 *>(gamma + ch * 8), ga);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) ga[i] = 1.f;
            }
            if (beta) {
                unpack8<T>(*reinterpret_cast<const u32x4 *>(beta + ch * 8), be);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) be[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * ga[i] + be[i];
            *reinterpret_cast<u32x4 *>(yr + ch * 8) = pack8<T>(f);
        }
    }
}

template <typename T>
static int ln_launch(const void *x, const void *gamma, const void *beta, void *y, const sfast_ln_params *p,
                     hipStream_t st, bool fast) {
    const dim3 grid(ceil_div(p->M, 4)), block(256);
    if constexpr (!std::is_same<T, float>::value) {
      if (fast && p->N > 4096) {
        const int nch = p->N / 8;
        if (nch <= 1024)
            hipLaunchKernelGGL((ln_wide_kernel<T, 4>), dim3(p->M), block, 0, st, (const T *)x, (const T *)gamma, (const T *)beta, (T *)y,
                               p->M, p->N, p->eps);
        else if (nch <= 2048)
            hipLaunchKernelGGL((ln_wide_kernel<T, 8>), dim3(p->M), block, 0, st, (const T *)x, (const T *)gamma, (const T *)beta, (T *)y,
                               p->M, p->N, p->eps);
        else
            hipLaunchKernelGGL((ln_wide_kernel<T, 16>), dim3(p->M), block, 0, st, (const T *)x, (const T *)gamma, (const T *)beta, (T *)y,
                               p->M, p->N, p->eps);
        return check_launch("layer_norm_wide");
      }
      if (fast) {
        const int nch = p->N / 8;
        if (nch <= 64)
            hipLaunchKernelGGL((ln_rows_kernel<T, 1>), grid, block, 0, st, (const T *)x, (const T *)gamma,
                               (const T *)beta, (T *)y, p->M, p->N, p->eps);
        else if (nch <= 128)
            hipLaunchKernelGGL((ln_rows_kernel<T, 2>), grid, block, 0, st, (const T *)x, (const T *)gamma,
                               (const T *)beta, (T *)y, p->M, p->N, p->eps);
        else if (nch <= 256)
            hipLaunchKernelGGL((ln_rows_kernel<T, 4>), grid, block, 0, st, (const T *)x, (const T *)gamma,
                               (const T *)beta, (T *)y, p->M, p->N, p->eps);
        else
            hipLaunchKernelGGL((ln_rows_kernel<T, 8>), grid, block, 0, st, (const T *)x, (const T *)gamma,
                               (const T *)beta, (T *)y, p->M, p->N, p->eps);
        return check_launch("layer_norm");
      }
    }
    {
        hipLaunchKernelGGL(ln_generic_kernel<T>, grid, block, 0, st, (const T *)x, (const T *)gamma,
                           (const T *)beta, (T *)y, p->M, p->N, p->eps);
    }
    return check_launch("layer_norm");
}

extern "C" int sfast_hip_layer_norm(const void *x, const void *gamma, const void *beta, void *y,
                                    const sfast_ln_params *p, sfast_stream_t stream) {
    SFAST_REQUIRE(p && x && y, SFAST_ERR_INVALID, "layer_norm: null argument");
    SFAST_REQUIRE(p->M > 0 && p->N > 0, SFAST_ERR_INVALID, "layer_norm: bad shape %d x %d", p->M, p->N);
    hipStream_t st = (hipStream_t)stream;
    const bool fast = p->dtype != SFAST_F32 && p->N % 8 == 0 && p->N <= 32768 && aligned16(x) && aligned16(y) &&
                      (!gamma || aligned16(gamma)) && (!beta || aligned16(beta));
    set_kernel_name(fast ? (p->N > 4096 ? "ln_wide" : "ln_rows") : "ln_generic");
    switch (p->dtype) {
    case SFAST_F16: return ln_launch<f16>(x, gamma, beta, y, p, st, fast);
    case SFAST_BF16: return ln_launch<bf16>(x, gamma, beta, y, p, st, fast);
    case SFAST_F32: return ln_launch<float>(x, gamma, beta, y, p, st, false);
    }
    set_error("layer_norm: bad dtype %d", p->dtype);
    return SFAST_ERR_UNSUPPORTED;
}

extern "C" int sfast_hip_softmax_rows(const void *x, void *y, const sfast_softmax_params *p, sfast_stream_t stream) {
    SFAST_REQUIRE(p && x && y, SFAST_ERR_INVALID, "softmax_rows: null argument");
    SFAST_REQUIRE(p->M > 0 && p->N > 0 && p->N % 8 == 0 && p->N <= 8 * 256 * 16, SFAST_ERR_UNSUPPORTED,
                  "softmax_rows: N=%d must be a multiple of 8, at most 32768", p->N);
    SFAST_REQUIRE(p->ldx % 8 == 0 && p->ldy % 8 == 0 && p->ldx >= p->N && p->ldy >= p->N && aligned16(x) && aligned16(y),
                  SFAST_ERR_UNSUPPORTED, "softmax_rows: rows must be 16-byte aligned");
    SFAST_REQUIRE(p->scale > 0.f, SFAST_ERR_INVALID, "softmax_rows: scale must be positive");
    set_kernel_name("softmax_rows");
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == SFAST_F16) return softmax_launch<f16>(x, y, p, st);
    if (p->dtype == SFAST_BF16) return softmax_launch<bf16>(x, y, p, st);
    set_error("softmax_rows: dtype %d unsupported", p->dtype);
    return SFAST_ERR_UNSUPPORTED;
}
