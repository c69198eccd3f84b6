// Pipe 5 of the MFMA implicit GEMM: 256-row ping-pong tiles (kernel: igemm_pp.h; one translation unit per dtype and mode:
// igemm_pp_{f16,bf16}_{lin,conv}.hip). This file: the launch switch.
#include "igemm.h"

namespace sfast {

extern unsigned long long *g_igemm_trace;  // igemm_glds.hip
int igemm_pp_init_f16_lin();
int igemm_pp_init_f16_conv();
int igemm_pp_init_bf16_lin();
int igemm_pp_init_bf16_conv();
int igemm_pp_launch_f16_lin(const IgemmArgs &a, int BN, int pw, bool geglu, hipStream_t st);
int igemm_pp_launch_f16_conv(const IgemmArgs &a, int BN, int pw, bool geglu, hipStream_t st);
int igemm_pp_launch_bf16_lin(const IgemmArgs &a, int BN, int pw, bool geglu, hipStream_t st);
int igemm_pp_launch_bf16_conv(const IgemmArgs &a, int BN, int pw, bool geglu, hipStream_t st);

int igemm_pp_init() {
    int rc = igemm_pp_init_f16_lin();
    if (!rc) rc = igemm_pp_init_f16_conv();
    if (!rc) rc = igemm_pp_init_bf16_lin();
    if (!rc) rc = igemm_pp_init_bf16_conv();
    return rc;
}

int igemm_pp_launch(const IgemmArgs &a_in, int dtype, int mode, bool geglu, int BN, int pw, hipStream_t st) {
    IgemmArgs a = a_in;
    a.trace = g_igemm_trace;
    if (dtype == SFAST_F16) return mode ? igemm_pp_launch_f16_conv(a, BN, pw, geglu, st) : igemm_pp_launch_f16_lin(a, BN, pw, geglu, st);
    return mode ? igemm_pp_launch_bf16_conv(a, BN, pw, geglu, st) : igemm_pp_launch_bf16_lin(a, BN, pw, geglu, st);
}

}  // namespace sfast
