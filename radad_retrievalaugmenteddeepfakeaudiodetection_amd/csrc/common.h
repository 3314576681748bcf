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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
