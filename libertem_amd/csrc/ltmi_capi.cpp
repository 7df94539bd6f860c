// misc C-ABI entry points: version, errors, device enumeration
#include "ltmi_common.h"
#include <string.h>

namespace ltmi {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ltmi

extern "C" int ltmi_version(void) { return LTMI_VERSION; }

extern "C" const char *ltmi_last_error(void) { return ltmi::g_err; }

extern "C" int ltmi_device_count(int *count) {
    if (!count) LTMI_FAIL(LTMI_E_INVALID, "ltmi_device_count: null argument");
    int n = 0;
    (void)hipGetLastError();            // (also forgets the thread's sticky last error: hip.clear_last_runtime_error)
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) { *count = 0; return LTMI_OK; }
    if (e != hipSuccess) LTMI_FAIL((int)e, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    *count = n;
    return LTMI_OK;
}

extern "C" int ltmi_device_info(int device, char *name_out, int *cu_count, int64_t *hbm_bytes,
                                int *gfx_arch) {
    hipDeviceProp_t prop;
    LTMI_HIP(hipGetDeviceProperties(&prop, device));
    if (name_out) { strncpy(name_out, prop.name, 255); name_out[255] = 0; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (gfx_arch) {
        // gcnArchName is like "gfx950:sramecc+:xnack-"
        int v = 0;
        const char *p = strstr(prop.gcnArchName, "gfx");
        if (p) v = (int)strtol(p + 3, nullptr, 16);
        *gfx_arch = v;   // 0x950 for MI355X
    }
    return LTMI_OK;
}

extern "C" int ltmi_host_device_pointer(int device, void *host, void **dev_out) {
    if (!host || !dev_out) LTMI_FAIL(LTMI_E_INVALID, "ltmi_host_device_pointer: null argument");
    LTMI_HIP(hipSetDevice(device));
    void *d = nullptr;
    LTMI_HIP(hipHostGetDevicePointer(&d, host, 0));
    *dev_out = d;
    return LTMI_OK;
}


// LTMI_ABORT_BACKTRACE=1: print the native call stack when the process aborts (a runtime library calling abort(), an
// uncaught C++ exception) -- Python's faulthandler shows the Python frames only
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void ltmi_abort_backtrace(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\nlibltmi: SIGABRT, native frames:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void ltmi_debug_init() {
    const char *e = getenv("LTMI_ABORT_BACKTRACE");
    if (e && e[0] == '1') signal(SIGABRT, ltmi_abort_backtrace);
}
