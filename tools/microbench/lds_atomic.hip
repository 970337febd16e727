// LDS atomic throughput on gfx950 by type and address pattern (one 1024-thread workgroup per CU):
//   hipcc -O3 --offload-arch=gfx950 -o lds_atomic lds_atomic.hip && ./lds_atomic
// Result that shaped the hash-grid scatter: ds_add_f32 192 cycles per wave instruction for every pattern,
// ds_add_f64 8-26, ds_add_u32 5-8.  The last three modes update one table ROW per step (two features).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// cycles per wave-instruction of LDS atomics, 16 waves per CU, one workgroup per CU
template <int MODE>
__global__ __launch_bounds__(1024) void k(const unsigned* __restrict__ idx, int iters, float* out, long long* cyc) {
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < 32768; i += 1024) s[i] = 0.f;
    __syncthreads();
    unsigned a[8];
    for (int u = 0; u < 8; ++u) a[u] = idx[(blockIdx.x * 8 + u) * 1024 + threadIdx.x] & 32767u;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned ad = (a[u] + it * 64) & 32767u;
            if (MODE == 0) atomicAdd(&s[ad], 1.0f);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(s) + ad, 1u);
            else if (MODE == 2) atomicAdd(reinterpret_cast<double*>(s) + (ad >> 1), 1.0);
            else if (MODE == 3) { float v = s[ad]; s[ad] = v + 1.0f; }
            else if (MODE == 4) { float r = atomicAdd(&s[ad], 1.0f); if (r == 12345.f) out[0] = r; }
            else if (MODE == 5) {   // one table row of two floats by compare-and-swap on its 8 bytes
                unsigned long long* p = reinterpret_cast<unsigned long long*>(s) + (ad >> 2);   // 8192 rows
                unsigned long long old = *p, assumed;
                do {
                    assumed = old;
                    const float lo = __uint_as_float((unsigned)assumed) + 1.0f, hi = __uint_as_float((unsigned)(assumed >> 32)) + 1.0f;
                    old = atomicCAS(p, assumed, ((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo));
                } while (old != assumed);
            }
            else if (MODE == 6) atomicAdd(reinterpret_cast<unsigned long long*>(s) + (ad >> 1), 1ull);
            else if (MODE == 7) {   // one table row of two doubles: the scatter's pattern
                double* p = reinterpret_cast<double*>(s) + 2 * (ad >> 2);                        // 8192 rows
                atomicAdd(p, 1.0);
                atomicAdd(p + 1, 1.0);
            }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float acc = 0; for (int i = threadIdx.x; i < 32768; i += 1024) acc += s[i];
    if (acc == 1e30f) out[1] = acc;
}
int main() {
    const int blocks = 256, iters = 64;
    std::vector<unsigned> h(blocks * 8 * 1024);
    unsigned *d; float* out; long long* cyc;
    hipMalloc(&d, h.size() * 4); hipMalloc(&out, 64); hipMalloc(&cyc, blocks * 8);
    const char* pat[] = {"random", "lane-linear", "stride-2 (2 lanes/bank)", "same address"};
    for (int p = 0; p < 4; ++p) {
        unsigned x = 12345;
        for (size_t i = 0; i < h.size(); ++i) {
            x = x * 1664525u + 1013904223u;
            unsigned lane = i & 1023;
            h[i] = p == 0 ? (x >> 8) : p == 1 ? lane + ((i >> 10) * 1024) : p == 2 ? lane * 2 : 7;
        }
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 8; ++mode) {
            hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(1024), 131072, 0, d, iters, out, cyc);
                hipDeviceSynchronize();
            }
            long long c[blocks]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
            double waveinstr = 16.0 * iters * 8;   // per CU
            const char* mn[] = {"ds_add_f32", "ds_add_u32", "ds_add_f64", "read+write f32 (racy)", "ds_add_rtn_f32",
                                "float2 row by 64-bit CAS", "ds_add_u64", "double2 row: 2 ds_add_f64"};
            printf("%-24s %-22s %.1f cycles per wave-instruction (clock64 ticks x ?)  raw %lld\n", pat[p], mn[mode], c[0] / waveinstr, c[0]);
        }
    }
    return 0;
}
