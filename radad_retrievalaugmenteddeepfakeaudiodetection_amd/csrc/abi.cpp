// abi.cpp -- error text, version and device probes of libradad_hip.so.
#include "common.h"

#include <mutex>
#include <set>
#include <string>
#include <stdlib.h>

static thread_local std::string g_err;

void radad_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

const char* radad_env_override(const char* name, const char* effect) {
    const char* e = getenv(name);
    if (e) {
        static std::mutex mu;
        static std::set<std::string> said;
        std::lock_guard<std::mutex> lk(mu);
        if (said.insert(name).second) fprintf(stderr, "[libradad_hip] %s=%s is set: %s\n", name, e, effect);
    }
    return e;
}

extern "C" {
int radad_abi_version(void) { return RADAD_ABI_VERSION; }
const char* radad_last_error(void) { return g_err.c_str(); }
int radad_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        radad_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return RADAD_EHIP;
    }
    return n;
}
}
