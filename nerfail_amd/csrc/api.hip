// Library-wide plumbing: error state, version, device query.
#include "common.h"

#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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

static int trace_level_from_env() {
    const char* v = getenv("NERFAIL_TRACE");
    return v ? atoi(v) : 0;
}
int g_trace = trace_level_from_env();

int trace_launch(const char* name) {
    char buf[160];
    int n = snprintf(buf, sizeof(buf), "[nerfail] %s", name);
    if (write(2, buf, n) < 0) return 0;
    if (g_trace >= 2) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            if (write(2, " FAILED\n", 8) < 0) return 0;
            return hip_fail(e, name);
        }
        if (write(2, " ok", 3) < 0) return 0;
    }
    if (write(2, "\n", 1) < 0) return 0;
    return 0;
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
