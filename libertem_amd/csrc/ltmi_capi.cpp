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


// ---- ltmi_host_copy: host -> host memcpy on several threads ------------------------------------------------------
// The staging copy of host-resident frames into the page-locked bounce buffers of the upload path has to keep up
// with the host link (57.6 GB/s measured H2D); one thread's memcpy does 8 - 12 GB/s.  A small persistent pool of
// workers, each copying a contiguous share; the caller copies a share itself and waits for the others.
#include <atomic>
#include <string>
#include <sched.h>
#include <pthread.h>
#include <condition_variable>
#include <mutex>
#include <thread>
namespace {
struct CopyPool {
    std::mutex call_mu;                       // one ltmi_host_copy at a time
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> workers;
    unsigned char *dst = nullptr;
    const unsigned char *src = nullptr;
    size_t bytes = 0, share = 0;
    int n_shares = 0;                         // shares of the current job (share 0 is the caller's)
    uint64_t job = 0;
    int pending = 0;
    bool quit = false;

    // One copy thread per L3 domain (a CCD of an EPYC host: 8 cores behind one link to the memory fabric).  Left to the
    // scheduler, 16 threads may land on one or two CCDs and the copy runs at 50 GB/s -- below the 57 GB/s host link it
    // has to stay ahead of -- where spread threads reach 90 - 200 (profiles/r06_host_upload.txt).  A worker may run on
    // any core of ITS domain; LTMI_COPY_SPREAD=0 leaves the placement to the scheduler.
    static const std::vector<cpu_set_t> &domains() {
        static const std::vector<cpu_set_t> d = [] {
            std::vector<cpu_set_t> out;
            const char *e = getenv("LTMI_COPY_SPREAD");
            if (e && e[0] == '0') return out;
            cpu_set_t allowed;
            CPU_ZERO(&allowed);
            if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
            std::vector<std::string> keys;
            for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
                if (!CPU_ISSET(cpu, &allowed)) continue;
                char path[128], buf[256] = {0};
                snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
                FILE *f = fopen(path, "r");
                if (!f) continue;
                const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
                fclose(f);
                if (!ok) continue;
                size_t k = 0;
                for (; k < keys.size(); ++k)
                    if (keys[k] == buf) break;
                if (k == keys.size()) {
                    keys.push_back(buf);
                    cpu_set_t s;
                    CPU_ZERO(&s);
                    out.push_back(s);
                }
                CPU_SET(cpu, &out[k]);
            }
            if (out.size() < 2) out.clear();
            return out;
        }();
        return d;
    }

    void work(int idx) {
        const std::vector<cpu_set_t> &dom = domains();
        if (!dom.empty()) {
            // worker idx -> domain idx + 1, idx + 1 + n/2, ... : neighbours in the list (one socket) are filled alternately
            const size_t n = dom.size();
            const size_t k = (size_t)(idx + 1) % n;
            const size_t pick = (k % 2) * (n / 2) + k / 2;
            (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &dom[pick % n]);
        }
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return quit || job != seen; });
            if (quit) return;
            seen = job;
            const int my = idx + 1;
            if (my >= n_shares) continue;    // (not needed for this job; `pending` only counts the shares handed out)
            const size_t o = (size_t)my * share;
            const size_t n = o >= bytes ? 0 : (bytes - o < share ? bytes - o : share);
            unsigned char *d = dst + o;
            const unsigned char *s = src + o;
            lk.unlock();
            if (n) memcpy(d, s, n);
            lk.lock();
            if (--pending == 0) cv_done.notify_one();
        }
    }
    void grow(int n) {
        while ((int)workers.size() < n) {
            const int idx = (int)workers.size();
            workers.emplace_back([this, idx] { work(idx); });
        }
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
    }
};
CopyPool *copy_pool() {
    static CopyPool *pool = new CopyPool();  // (leaked on purpose: no destructor order games at interpreter exit)
    return pool;
}
}  // namespace

extern "C" int ltmi_host_copy(void *dst, const void *src, int64_t bytes, int threads) {
    if (bytes < 0) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_host_copy: negative size");
    if (bytes == 0) return LTMI_OK;
    if (!dst || !src) LTMI_FAIL(LTMI_E_INVALID, "ltmi_host_copy: null pointer");
    if (threads <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        threads = hw >= 32 ? 16 : (hw >= 4 ? (int)hw / 2 : 1);
    }
    if (threads > 64) threads = 64;
    // at least 1 MiB per share
    const int64_t max_shares = (bytes + (1 << 20) - 1) >> 20;
    if (threads > max_shares) threads = (int)max_shares;
    if (threads <= 1) {
        memcpy(dst, src, (size_t)bytes);
        return LTMI_OK;
    }
    CopyPool *p = copy_pool();
    std::lock_guard<std::mutex> call(p->call_mu);
    try {
        p->grow(threads - 1);
    } catch (...) {
        memcpy(dst, src, (size_t)bytes);     // no threads to be had: the plain copy
        return LTMI_OK;
    }
    size_t share = ((size_t)bytes + threads - 1) / threads;
    share = (share + 4095) & ~(size_t)4095;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->dst = (unsigned char *)dst;
        p->src = (const unsigned char *)src;
        p->bytes = (size_t)bytes;
        p->share = share;
        p->n_shares = threads;
        p->pending = threads - 1;
        p->job++;
    }
    p->cv_work.notify_all();
    memcpy(dst, src, share < (size_t)bytes ? share : (size_t)bytes);
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->pending == 0; });
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
