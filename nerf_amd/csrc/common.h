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

// K consecutive floats as one load (dword, x2, x3 or x4 pieces; 4-byte aligned).
template <int K>
struct __attribute__((packed, aligned(4))) FloatRun {
    float v[K];
};

// sin and cos of 2^k x for k = 0..L-1 (x a float, the products 2^k x exact as in the
// reference, src/utils/positional_encoding.py:30-36), correctly rounded to float up to
// ~1e-9: one sin/cos pair in double (quadrant reduction, Taylor polynomials on
// [-pi/4, pi/4], error < 1e-13) and then angle doubling in double, whose error grows by
// two per step.  20 times cheaper than 2L float library calls with their own range
// reductions.  L <= 20.
template <int L>
__device__ __forceinline__ void sincos_octaves(float xf, float (&sn)[L], float (&cs)[L]) {
    static_assert(L <= 20, "angle doubling in double loses a bit per octave");
    const double x = (double)xf;
    const double k = rint(x * 0.63661977236758134308);                       // 2/pi
    double r = fma(k, -1.57079632679489655800e+00, x);                        // pi/2, high part
    r = fma(k, -6.12323399573676603587e-17, r);                               // low part
    const double r2 = r * r;
    double ps = -7.6471637318198164759e-13;                                   // -1/15!
    ps = fma(ps, r2, 1.6059043836821614599e-10);                              // 1/13!
    ps = fma(ps, r2, -2.5052108385441718775e-08);
    ps = fma(ps, r2, 2.7557319223985890653e-06);
    ps = fma(ps, r2, -1.9841269841269841270e-04);
    ps = fma(ps, r2, 8.3333333333333333333e-03);
    ps = fma(ps, r2, -1.6666666666666666667e-01);
    const double sr = fma(ps * r2, r, r);
    double pc = 4.7794773323873852974e-14;                                    // 1/16!
    pc = fma(pc, r2, -1.1470745597729724714e-11);
    pc = fma(pc, r2, 2.0876756987868098979e-09);
    pc = fma(pc, r2, -2.7557319223985890653e-07);
    pc = fma(pc, r2, 2.4801587301587301587e-05);
    pc = fma(pc, r2, -1.3888888888888888889e-03);
    pc = fma(pc, r2, 4.1666666666666666667e-02);
    pc = fma(pc, r2, -0.5);
    const double cr = fma(pc, r2, 1.0);
    const int q = (int)k & 3;
    double s = (q & 1) ? cr : sr, c = (q & 1) ? sr : cr;
    if (q == 1 || q == 2) c = -c;
    if (q >= 2) s = -s;
#pragma unroll
    for (int f = 0; f < L; ++f) {
        sn[f] = (float)s;
        cs[f] = (float)c;
        const double s2 = (s + s) * c, c2 = (c - s) * (c + s);
        s = s2;
        c = c2;
    }
}

// The number of points a kernel works on: P, or the count in device memory when the caller passes one (a step whose
// sizes depend on the data -- samples skipped by occupancy -- launched with the host's upper bound P and no read-back,
// so that it can be captured into a HIP graph).
__device__ __forceinline__ int64_t points_in_use(int64_t P, const int* __restrict__ count) {
    if (!count) return P;
    const int64_t c = *count;
    return c < P ? (c > 0 ? c : 0) : P;
}

// Clears n 32-bit words.  A kernel of the library's own rather than hipMemsetAsync: captured into a HIP graph, the
// runtime's memset node stopped clearing once the device had been synchronised and other kernels launched between two
// replays (observed on ROCm 7.2; tests/test_gpu_graph.py has the sequence).  The queue counters of the table-gradient
// scatter then kept their old values: every queue looked full, the reduce pass added the previous replay's records
// again -- table gradients off by 90 % -- and a replay took 2-20 ms instead of 0.6.  A kernel node carries its own
// arguments.
__global__ void zero_words_kernel(unsigned* __restrict__ p, int n);
inline void launch_zero_words(void* p, int n, hipStream_t stream) {
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<unsigned*>(p), n);
}

}  // namespace nerf
