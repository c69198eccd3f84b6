// Library-level entry points: version, init, thread-local error / kernel-name strings.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>

namespace sfast {

static thread_local char g_err[512] = "";
static thread_local char g_kernel[256] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void set_kernel_name(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return SFAST_OK;
}

extern unsigned long long *g_igemm_trace;  // igemm_glds.hip
extern int g_igemm_exp;                    // igemm_glds.hip
int igemm_init();      // igemm.hip
int g_batch_ref = 0;    // common.h
int attention_init();  // attention.hip
int gnconv_init();     // gnconv.hip

}  // namespace sfast

extern "C" {

int sfast_hip_abi_version(void) { return SFAST_HIP_ABI_VERSION; }

int sfast_hip_init(void) {
    // kernel attributes (dynamic-LDS limits) are per device: applied once for every device this is called on (the CURRENT
    // device of the calling thread), under a lock so concurrent first calls cannot interleave
    static std::mutex mu;
    static int state[64];  // 0 = not initialised, 1 = ok, < 0 = the error the first attempt returned
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        sfast::set_error("sfast_hip_init: no usable current HIP device");
        return SFAST_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    {
        const char *bi = getenv("SFAST_BATCH_INVARIANT");
        sfast::g_batch_ref = (bi && bi[0] != '0' && bi[0] != '\0') ? 2 : 0;  // reference batch = the CFG pair of one image
    }
    if (state[dev] == 0) {
        int rc = sfast::igemm_init();
        if (rc == 0) rc = sfast::attention_init();
        if (rc == 0) rc = sfast::gnconv_init();
        state[dev] = rc == 0 ? 1 : rc;
    }
    return state[dev] == 1 ? 0 : state[dev];
}

int sfast_hip_set_trace(void *buf) {
    sfast::g_igemm_trace = (unsigned long long *)buf;
#ifdef SFAST_PROBES
    const char *e = getenv("SFAST_IGEMM_EXP");  // timing experiments: probe build only, and only while tracing
    sfast::g_igemm_exp = (buf && e) ? atoi(e) : 0;
#else
    sfast::g_igemm_exp = 0;  // the product library holds no experiment instantiation: nothing an environment variable could select
#endif
    return 0;
}

int sfast_hip_has_probes(void) {
#ifdef SFAST_PROBES
    return 1;
#else
    return 0;
#endif
}

const char *sfast_hip_last_error(void) { return sfast::g_err; }
const char *sfast_hip_last_kernel(void) { return sfast::g_kernel; }

}  // extern "C"
