// Pipe 4 of the MFMA implicit GEMM: packed weights straight into registers (kernel: igemm_pk.h; one translation unit per dtype and
// mode: igemm_pk_{f16,bf16}_{lin,conv}.hip). This file: the packing kernel, its C entry points, and the launch switch.
#include "igemm.h"

namespace sfast {

// ---- packing ----------------------------------------------------------------------------------------------------------------------
// out fragment (nb, s), lane l = (r = l % 32, g = l / 32): the 8 elements w[nb * 32 + r][s * 16 + g * 8 .. + 8]; zero outside [N) x [K).
// One thread per 16-byte chunk, consecutive threads = consecutive k-steps of one lane slot: reads walk a weight row, 16 bytes at a
// time (coalesced), writes are 16 bytes 1 KB apart -- a one-off per parameter version, not a per-step cost.
template <typename T>
__global__ void __launch_bounds__(256) pack_weight_kernel(const T *__restrict__ w, u32x4 *__restrict__ out, int N, int K, int64_t ldw, int ksteps,
                                                           int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    // idx = ((nb * 32 + r) * 2 + g) * ksteps + s
    const int s = (int)(idx % ksteps);
    const int64_t q = idx / ksteps;
    const int g = (int)(q & 1);
    const int64_t row = q >> 1;
    const int r = (int)(row & 31);
    const int64_t nb = row >> 5;
    const int k = s * 16 + g * 8;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (row < N && k < K) {
        if (k + 8 <= K && (ldw % 8) == 0 && ((uintptr_t)w % 16) == 0) {
            v = *reinterpret_cast<const u32x4 *>(w + row * ldw + k);
        } else {
            alignas(16) T e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = (k + i < K) ? w[row * ldw + k + i] : Elem<T>::from_f32(0.f);
            v = *reinterpret_cast<const u32x4 *>(e);
        }
    }
    out[(nb * ksteps + s) * 64 + g * 32 + r] = v;
}


extern unsigned long long *g_igemm_trace;  // igemm_glds.hip
int igemm_pk_init_f16_lin();
int igemm_pk_init_f16_conv();
int igemm_pk_init_bf16_lin();
int igemm_pk_init_bf16_conv();
int igemm_pk_launch_f16_lin(const IgemmArgs &a, int BM, int BN, hipStream_t st);
int igemm_pk_launch_f16_conv(const IgemmArgs &a, int BM, int BN, hipStream_t st);
int igemm_pk_launch_bf16_lin(const IgemmArgs &a, int BM, int BN, hipStream_t st);
int igemm_pk_launch_bf16_conv(const IgemmArgs &a, int BM, int BN, hipStream_t st);

int igemm_pk_init() {
    int rc = igemm_pk_init_f16_lin();
    if (!rc) rc = igemm_pk_init_f16_conv();
    if (!rc) rc = igemm_pk_init_bf16_lin();
    if (!rc) rc = igemm_pk_init_bf16_conv();
    return rc;
}

int igemm_pk_launch(const IgemmArgs &a_in, int dtype, int mode, int BM, int BN, hipStream_t st) {
    IgemmArgs a = a_in;
    a.trace = g_igemm_trace;
    if (dtype == SFAST_F16) return mode ? igemm_pk_launch_f16_conv(a, BM, BN, st) : igemm_pk_launch_f16_lin(a, BM, BN, st);
    return mode ? igemm_pk_launch_bf16_conv(a, BM, BN, st) : igemm_pk_launch_bf16_lin(a, BM, BN, st);
}

}  // namespace sfast

using namespace sfast;

extern "C" size_t sfast_hip_packed_weight_bytes(int32_t N, int32_t K) {
    if (N <= 0 || K <= 0) return 0;
    return (size_t)((N + 31) / 32) * (size_t)(((K + 63) / 64) * 4) * 1024;
}

extern "C" int sfast_hip_pack_weight(const void *w, void *packed, int32_t N, int32_t K, int64_t ldw, int32_t dtype, sfast_stream_t stream) {
    SFAST_REQUIRE(w && packed && N > 0 && K > 0 && ldw >= K, SFAST_ERR_INVALID, "pack_weight: bad arguments");
    SFAST_REQUIRE(aligned16(packed), SFAST_ERR_UNSUPPORTED, "pack_weight: the packed buffer must be 16-byte aligned");
    const int ksteps = ((K + 63) / 64) * 4;
    const int64_t total = (int64_t)((N + 31) / 32) * 64 * ksteps;
    const dim3 grid((unsigned)ceil_div64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    set_kernel_name("pack_weight[%dx%d]", N, K);
    if (dtype == SFAST_F16)
        hipLaunchKernelGGL(pack_weight_kernel<f16>, grid, dim3(256), 0, st, (const f16 *)w, (u32x4 *)packed, N, K, ldw, ksteps, total);
    else if (dtype == SFAST_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16>, grid, dim3(256), 0, st, (const bf16 *)w, (u32x4 *)packed, N, K, ldw, ksteps, total);
    else {
        set_error("pack_weight: dtype %d", dtype);
        return SFAST_ERR_UNSUPPORTED;
    }
    return check_launch("pack_weight");
}
