// Library-wide plumbing: error state, version, device query.
#include "common.h"

#include <string.h>

namespace nerfail {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return NERFAIL_EHIP;
}

}  // namespace nerfail

extern "C" int nerfail_abi_version(void) { return NERFAIL_ABI_VERSION; }

extern "C" const char* nerfail_last_error(void) { return nerfail::g_err; }

extern "C" int nerfail_device_name(char* buf_host, size_t buf_len) {
    NF_REQUIRE(buf_host != nullptr && buf_len > 0, "buf_host is NULL or empty");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        nerfail::set_error("nerfail_device_name: no HIP device visible");
        buf_host[0] = 0;
        return NERFAIL_ENODEV;
    }
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return nerfail::hip_fail(e, "hipGetDevice");
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return nerfail::hip_fail(e, "hipGetDeviceProperties");
    snprintf(buf_host, buf_len, "%s", p.gcnArchName);
    return NERFAIL_OK;
}
