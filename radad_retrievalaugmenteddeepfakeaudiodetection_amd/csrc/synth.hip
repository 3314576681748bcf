// synth.hip -- stateless synthetic inputs for bench.py and the tests (device twin of oracle/synth.py).
//
// Every element is a pure function of (seed, row, col): integer hashing, exact int->float conversion and
// single correctly-rounded float32 operations (__fmul_rn/__fadd_rn/__fdiv_rn: no FMA contraction), so this
// kernel and the numpy oracle produce identical bits.  Not part of the reference; it exists so that each
// GPU of a sharded run can materialise its own slice of the store / its own clips without shipping them.
#include "common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t hash3(uint64_t seed, uint64_t row, uint64_t col) {
    const uint64_t r = splitmix64(row ^ splitmix64(seed));   // seed is hashed first: streams do not alias
    return splitmix64(r ^ (col * 0x100000001B3ull));
}
__device__ __forceinline__ int ih4(uint64_t h) {
    return (int)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
}

constexpr float SCALE = (float)(1.0 / 37837.2271);
constexpr float NOISE_SCALE = 0.1f * SCALE;
constexpr float TRI_AMP = 0.3f;

__global__ void k_synth_rows(float* out, int64_t row0, int64_t n_rows, int dim, uint64_t seed) {
    const int64_t total = n_rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dim;
        const int c = (int)(i - r * dim);
        out[i] = __fmul_rn((float)ih4(hash3(seed, (uint64_t)(row0 + r), (uint64_t)c)), SCALE);
    }
}

__global__ void k_synth_audio(float* out, int64_t clip0, int64_t n_clips, int64_t samples, uint64_t seed) {
    const int64_t total = n_clips * samples;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / samples;
        const int64_t n = i - c * samples;
        const uint64_t clip = (uint64_t)(clip0 + c);
        const float noise = __fmul_rn((float)ih4(hash3(seed, clip, (uint64_t)n)), NOISE_SCALE);
        const int64_t period = 20 + (int64_t)(hash3(seed, clip, 1ull << 40) % 181ull);
        const int64_t ph = n % period;
        int64_t a = 2 * ph - period;
        if (a < 0) a = -a;
        const float num = (float)(2 * a - period);
        const float tri = __fdiv_rn(num, (float)period);
        out[i] = __fadd_rn(noise, __fmul_rn(TRI_AMP, tri));
    }
}

}  // namespace

extern "C" {

int radad_synth_rows(float* out_dev, int64_t row0, int64_t n_rows, int dim, uint64_t seed, int device, void* stream) {
    RADAD_REQUIRE(n_rows >= 0 && dim > 0 && row0 >= 0, "radad_synth_rows: bad shape");
    if (n_rows == 0) return RADAD_OK;
    RADAD_REQUIRE(out_dev, "radad_synth_rows: NULL buffer");
    DeviceGuard g(device);
    hipLaunchKernelGGL(k_synth_rows, dim3(4096), dim3(256), 0, (hipStream_t)stream, out_dev, row0, n_rows, dim, seed);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

int radad_synth_audio(float* out_dev, int64_t clip0, int64_t n_clips, int64_t samples_per_clip, uint64_t seed, int device,
                      void* stream) {
    RADAD_REQUIRE(n_clips >= 0 && samples_per_clip > 0 && clip0 >= 0, "radad_synth_audio: bad shape");
    if (n_clips == 0) return RADAD_OK;
    RADAD_REQUIRE(out_dev, "radad_synth_audio: NULL buffer");
    DeviceGuard g(device);
    hipLaunchKernelGGL(k_synth_audio, dim3(4096), dim3(256), 0, (hipStream_t)stream, out_dev, clip0, n_clips, samples_per_clip, seed);
    RADAD_HIP_CHECK(hipGetLastError());
    return RADAD_OK;
}

}  // extern "C"
