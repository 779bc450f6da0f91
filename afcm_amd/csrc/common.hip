// Library-level entry points and host-side error reporting.
#include "common.h"

namespace afcm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace afcm

extern "C" int afcm_abi_version(void) { return AFCM_ABI_VERSION; }
extern "C" const char* afcm_last_error(void) { return afcm::g_err; }

// Host-calibration aid (bench.py `host_us_per_launch`): one wave that does nothing, launched like every other entry point.
namespace afcm { __global__ void noop_kernel() {} }
extern "C" int afcm_noop(void* stream) {
    hipLaunchKernelGGL(afcm::noop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
    return afcm::hip_status(hipGetLastError());
}
