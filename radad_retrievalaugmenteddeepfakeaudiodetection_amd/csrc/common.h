// common.h -- shared helpers for libradad_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/radad_hip.h"

// thread-local error text returned by radad_last_error()
void radad_set_error(const char* fmt, ...);

#define RADAD_HIP_CHECK(expr)                                                                         \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            radad_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__);    \
            return RADAD_EHIP;                                                                        \
        }                                                                                             \
    } while (0)

#define RADAD_REQUIRE(cond, ...)                                                                      \
    do {                                                                                              \
        if (!(cond)) {                                                                                \
            radad_set_error(__VA_ARGS__);                                                             \
            return RADAD_EINVAL;                                                                      \
        }                                                                                             \
    } while (0)

// Environment overrides of the default kernel choice (RADAD_KNN_HI, RADAD_KNN_SMALLQ_HI, RADAD_WIDE_MIN_Q, RADAD_LOGMEL_F32) select
// CORRECT but slower paths (8x on the scan); a stray variable must not do that silently: the first read of each one that is set
// is announced once per process on stderr.  Returns the variable's value (nullptr when unset).
const char* radad_env_override(const char* name, const char* effect);

// Timing-experiment switches inside kernels (skip a phase and measure the rest): compiled out unless the library is built
// with -DRADAD_DEBUG_HOOKS, so that no environment variable can make the shipped kernels skip work.
#ifdef RADAD_DEBUG_HOOKS
#define RADAD_DBG(flags, bit) ((flags) & (bit))
#else
#define RADAD_DBG(flags, bit) 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same four floats behind a pointer that is only dword-aligned (a segment of a clip that starts at an arbitrary sample):
// global_load_dwordx4 needs no more than that on gfx9+
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// RAII device switch: every entry point runs on the handle's device and restores the caller's.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        target = dev;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != target) (void)hipSetDevice(prev);
    }
    int target = -1;
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave-wide reductions (wave = 64 lanes on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Ring of HIP event pairs recorded around one kernel launch on the caller's stream (bench.py's roofline leg).
struct EventRing {
    static constexpr int N = 64;
    hipEvent_t a[N], b[N];
    bool created = false, enabled = false;
    bool active = true;          // this search is one of the sampled ones (next_search)
    int every = 1, tick = 0;     // events around every `every`-th search only: an event record between dependent kernels costs ~6 us
    int count = 0;
    void next_search() { active = (tick++ % every) == 0; }
    int enable(bool on, int period = 1) {
        if (on && !created) {
            for (int i = 0; i < N; ++i) {
                if (hipEventCreate(&a[i]) != hipSuccess || hipEventCreate(&b[i]) != hipSuccess) return RADAD_EHIP;
            }
            created = true;
        }
        enabled = on;
        every = period > 1 ? period : 1; tick = 0; active = true;
        count = 0;
        return RADAD_OK;
    }
    void begin(hipStream_t st) { if (enabled && active) (void)hipEventRecord(a[count % N], st); }
    void end(hipStream_t st) { if (enabled && active) { (void)hipEventRecord(b[count % N], st); ++count; } }
    int read(float* out, int cap, int* n_out) {
        const int n = count < N ? count : N;
        int w = 0;
        for (int i = 0; i < n && w < cap; ++i) {
            const int slot = (count - n + i) % N;
            if (hipEventSynchronize(b[slot]) != hipSuccess) return RADAD_EHIP;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, a[slot], b[slot]) != hipSuccess) return RADAD_EHIP;
            out[w++] = ms;
        }
        if (n_out) *n_out = w;
        return RADAD_OK;
    }
    void destroy() {
        if (created) for (int i = 0; i < N; ++i) { (void)hipEventDestroy(a[i]); (void)hipEventDestroy(b[i]); }
        created = false;
    }
};
