// What dense f16 MFMA rate does the card sustain once its power management has settled?  A loop of nothing but
// v_mfma_f32_32x32x16_f16 on register operands (no memory, no LDS), one or two waves per SIMD on every CU, for
// ~100 ms per launch, with operands of random normal values (data-dependent switching) or zeros:
//   hipcc -O3 --offload-arch=gfx950 -o mfma_sustained mfma_sustained.hip && ./mfma_sustained [quick]
// Prints TFLOP/s from HIP events and the shader clock seen by the waves (s_memtime ticks per wall_clock64 tick);
// "quick" runs the random-operand, one-wave-per-SIMD case only (4 launches, ~0.4 s).
// Measured on an MI355X: random operands 1,610 TFLOP/s at 1.66 GHz (0.64 of the 2,500 nominal), the same with one
// or two waves per SIMD; all-zero operands 2,473 TFLOP/s at 2.39 GHz.  The nominal peak is a clock the card does
// not hold while its matrix cores switch real data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void spin(const f16x8* __restrict__ ops, int iters, float* out, long long* ticks) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = ops[(2 * i) * 64 + lane];
        b[i] = ops[(2 * i + 1) * 64 + lane];
    }
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(t + rep) & 3], b[t & 3], acc[t], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ticks[0] = c1 - c0;
        ticks[1] = w1 - w0;
    }
}

#define CHECK(call)                                                                         \
    do {                                                                                    \
        const hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                      \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

int main(int argc, char** argv) {
    const bool quick = argc > 1 && strcmp(argv[1], "quick") == 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<_Float16> h(8 * 64 * 8);
    f16x8* ops;
    float* out;
    long long* ticks;
    CHECK(hipMalloc(&ops, h.size() * 2));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&ticks, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int data = 0; data < (quick ? 1 : 3); ++data) {   // 0 random normal, 1 zeros, 2 random with the B operands passed through a ReLU
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            const float u1 = (float)((rand() + 1.0) / (RAND_MAX + 2.0)), u2 = (float)(rand() / (double)RAND_MAX);
            float v = 0.05f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
            const bool is_b = (i / (64 * 8)) & 1;   // operand sets alternate A, B
            if (data == 1 || (data == 2 && is_b && v < 0.0f)) v = 0.0f;
            h[i] = (_Float16)v;
        }
        CHECK(hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (int waves_per_simd = 1; waves_per_simd <= (quick ? 1 : 2); ++waves_per_simd) {
            const int threads = 256 * waves_per_simd, iters = 120000 / waves_per_simd;
            for (int rep = 0; rep < (quick ? 4 : 6); ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(spin, dim3(cus), dim3(threads), 0, 0, ops, iters, out, ticks);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                long long t[2];
                CHECK(hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost));
                const double flop = (double)cus * (threads / 64) * iters * 32.0 * 32768.0;
                printf("%-7s %d wave(s)/SIMD  launch %d: %7.2f ms  %7.1f TFLOP/s  (%.3f of 2500)  shader clock %.0f MHz\n",
                       data == 0 ? "random" : data == 1 ? "zeros" : "relu B", waves_per_simd, rep, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0,
                       100.0 * (double)t[0] / (double)t[1]);
            }
        }
    }
    return 0;
}
