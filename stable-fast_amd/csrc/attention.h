// Shared between attention.hip (32 query rows per wave; bias / cross-attention / wide heads) and attention_q64.hip (64 query rows per
// wave, one wave per SIMD): kernel arguments, LDS geometry, MFMA wrappers.
#pragma once
#include "common.h"
#include <type_traits>
#include <math.h>

namespace sfast {

struct AttnArgs {
    const void *q, *k, *v;
    void *out;
    int B, H, Sq, Skv, D;
    int64_t qs[3], ks[3], vs[3], os[3];
    float scale, scale_log2e;
    uint32_t kspan, vspan;      // bytes spanned by the K / V rows of one (batch, head): buffer-descriptor ranges
    unsigned long long *trace;  // profiling only (sfast_hip_set_trace): per-workgroup shader-cycle split of the tile loop
    // additive attention bias (xformers attn_bias / diffusers attention_mask): bias[b][h][q][key], key stride 1
    const void *bias;
    int64_t bs[3];        // element strides (b, h, q); 0 = broadcast
    uint32_t bspan;       // bytes spanned by the bias rows of one (batch, head)
    float inv_scale;      // bias enters the RAW scores as bias / scale (the softmax scale is folded into the exp2 argument)
    // XCD-aware block order (flash kernel): xmap = 1 -> 1-D grid of nqb * B * H blocks; the hardware puts block i on XCD i % 8, and
    // all nqb query blocks of one (batch, head) are given to ONE XCD, so that head's K / V (re-read by every query block) cross the
    // fabric once and then hit that XCD's L2. With the (q-block, head, batch) grid the query blocks of a head were spread over all
    // eight L2s: 53 MB of fabric traffic per SD1.5 self-attention launch against 16 MB of operands (profiles/r02_pmc_traffic_run8.log).
    int xmap, nqb, ppx;  // ppx = (batch, head) pairs per XCD
    int o16;             // attention_q64: output rows are 16-byte addressable (staged epilogue)
};

__device__ __forceinline__ f32x16 amfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 amfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int D> struct AttnGeom {
    static constexpr int DP = (D + 15) / 16 * 16;  // QK^T contraction length (zero padded)
    static constexpr int DO = (D + 31) / 32 * 32;  // padded output width
    static constexpr int KSTR = DP + 8;            // K tile row stride (halves): odd number of 16-B slots
    static constexpr int VSTR = 68;                // V^T tile row stride (halves): 136 B
    static constexpr int STAGE = 64 * KSTR * 2 + DO * VSTR * 2;
    static constexpr int LDS = 2 * STAGE;          // double-buffered K / V^T tiles
    static constexpr int LDS_TOTAL = LDS + 4096 + 16 * VSTR;  // + dump area for idle staging lanes
};


// attention_q64.hip: 64 query rows per wave. Returns -1 when the shape is outside its coverage (the caller then takes the 32-row kernel).
int attention_q64_init();
int attention_q64_launch(const AttnArgs &a, int dtype, int xmap_enabled, hipStream_t st);

}  // namespace sfast
