// Argument blocks + host entries of the non-MFMA GEMM / conv kernels (small.hip).
#pragma once
#include "common.h"

namespace sfast {

struct SmallGemmArgs {
    const void *x;
    const void *w[SFAST_MAX_WSEG];
    const void *bias, *rowbias, *res;
    void *out;
    int M, N, K;
    int64_t ldx, ldw, ldo, ldr, ld_rowbias;
    int rows_per_seg, rows_per_batch;
    int geglu, act, res_before_act, in_act;
    float alpha;
    float out_scale;  // accumulator scale (sfast_epilogue_ext); the launchers map 0 to 1
};

struct SmallConvArgs {
    const void *x, *x2, *w, *bias, *rowbias, *z;
    void *out;
    int B, H, W, Cin, C1, Cout, KH, KW, Ho, Wo;
    int stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, ups;
    int64_t xs[4], x2s[4], ws[4], os[4], zs[4];
    int64_t ld_rowbias;
    int act, res_before_act;
    float alpha;
    float out_scale;
};

int small_gemv(const SmallGemmArgs &a, int dtype, hipStream_t st);        // M <= 16, K % 8 == 0, aligned
int small_gemm_naive(const SmallGemmArgs &a, int dtype, hipStream_t st);  // anything
int small_conv_naive(const SmallConvArgs &a, int dtype, hipStream_t st);  // anything
int small_conv_n(const SmallConvArgs &a, int dtype, hipStream_t st);      // Cout <= 8, dense NHWC, Cin % 8 == 0
int small_conv_c(const SmallConvArgs &a, int dtype, hipStream_t st);      // KH*KW*Cin*Cout*4 <= 64 KiB, Cout % 8 == 0

}  // namespace sfast
