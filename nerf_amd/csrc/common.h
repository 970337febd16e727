// Shared host-side helpers for libnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/nerf_hip.h"

namespace nerf {

// Thread-local message of the last failing call (returned by nerf_last_error()).
char* err_buf();

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define NERF_REQUIRE(cond, ...) \
    do { if (!(cond)) return nerf::fail(NERF_EINVAL, __VA_ARGS__); } while (0)

// Check the launch that was just enqueued.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NERF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NERF_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int WAVE = 64;

// ---- wavefront primitives (64 lanes) ------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

}  // namespace nerf
