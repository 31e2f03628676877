// Shared helpers of libercgraft (host-side error plumbing + device utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ercgraft.h"

#define ERC_WAVE 64

extern "C" void erc_set_error(const char* fmt, ...);

#define ERC_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            erc_set_error(__VA_ARGS__);     \
            return ERC_E_ARG;               \
        }                                   \
    } while (0)

#define ERC_LAUNCH_CHECK(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            erc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return ERC_E_LAUNCH;                                                 \
        }                                                                        \
    } while (0)

static inline int erc_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// full-wave (64 lane) sum, result in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Window graph (track_mm/cogmen_utils.py:109-172): sum_{q<p} deg(q) with deg(q) = min(L-1, q+fwd) - max(0, q-back) + 1 --
// the closed form behind every CSR offset of a dialogue of L utterances.
__device__ __forceinline__ int erc_window_prefix(int p, int L, int back, int fwd) {
    const int a = min(p, max(0, L - fwd));  // #q<p whose upper end is not clipped
    const int c = max(0, p - back);         // #q<p whose lower end is not clipped
    return a * (a - 1) / 2 + a * fwd + (p - a) * (L - 1) - c * (c - 1) / 2 + p;
}

// Counter-based RNG (splitmix64 finaliser over (seed, offset, index)): one
// uniform in [0,1) per element, reproducible between forward and backward.
__device__ __forceinline__ float erc_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint64_t z = seed ^ (offset * 0x9E3779B97F4A7C15ull) ^ (idx + 0xD1B54A32D192ED03ull) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
#endif
