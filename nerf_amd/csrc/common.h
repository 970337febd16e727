// Shared host-side helpers for libnerf_hip.so (gfx950 only).
#pragma once
#include <type_traits>
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

// Compile-time loop: f(std::integral_constant<int, I>) for I in [B, E), so that the
// loop index can be used as a template argument.
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- wavefront primitives (64 lanes) ------------------------------------
// Cross-lane moves by DPP (data-parallel primitives: a lane permutation applied to a VALU
// operand, no LDS traffic).  Lanes without a source lane read `fill`.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float fill, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_ROW_SHR = 0x110;   // + n: lane i of each row of 16 reads lane i-n
constexpr int DPP_ROW_SHL = 0x100;   // + n: lane i reads lane i+n
constexpr int DPP_WAVE_SHR1 = 0x138; // lane i reads lane i-1 over the whole wave
constexpr int DPP_WAVE_SHL1 = 0x130; // lane i reads lane i+1 over the whole wave

__device__ __forceinline__ float lane_value(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Sum over the wave, the same value in every lane: an inclusive scan inside each row of 16
// lanes, then the four row totals.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<DPP_ROW_SHR + 1>(0.0f, v);
    v += dpp_move<DPP_ROW_SHR + 2>(0.0f, v);
    v += dpp_move<DPP_ROW_SHR + 4>(0.0f, v);
    v += dpp_move<DPP_ROW_SHR + 8>(0.0f, v);
    return (lane_value(v, 15) + lane_value(v, 31)) + (lane_value(v, 47) + lane_value(v, 63));
}

}  // namespace nerf
