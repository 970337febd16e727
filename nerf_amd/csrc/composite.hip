// Alpha compositing (Camera.volume_render, src/Camera.py:143-183) and its
// backward for gfx950.  One wavefront owns one ray; lane l holds IPL consecutive
// samples, the exclusive transmittance product is a lane-local prefix followed
// by a 64-lane shuffle scan, and the backward uses the mirrored suffix scan.
// Bandwidth-bound: 20 B/sample in, 4 B/sample out (forward).
// Compiled with -ffp-contract=off: alpha, 1-alpha+1e-10 and w round like the
// reference's separate ops.
#include "common.h"

namespace nerf {

constexpr int CMP_WAVES = 4;  // rays per workgroup

// Exclusive prefix product across the wave of each lane's total.
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d, 64);
        if (lane >= d) v *= o;
    }
    const float e = __shfl_up(v, 1, 64);
    return lane == 0 ? 1.0f : e;
}

// Exclusive suffix sum across the wave (sum of the totals of all higher lanes).
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(v, d, 64);
        if (lane + d < 64) v += o;
    }
    const float e = __shfl_down(v, 1, 64);
    return lane == 63 ? 0.0f : e;
}

// Loads the lane's IPL samples and computes alpha, t = 1-alpha+1e-10, T and w.
template <int IPL>
struct RaySlice {
    float sig[IPL], dist[IPL], alpha[IPL], tr[IPL], T[IPL], w[IPL], zz[IPL];
    bool ok[IPL];

    __device__ __forceinline__ void load(const float* __restrict__ sigma, const float* __restrict__ z, int64_t ray,
                                         int N, int lane) {
        const int base = lane * IPL;
        float lane_prod = 1.0f;
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const int i = base + k;
            ok[k] = i < N;
            sig[k] = ok[k] ? sigma[ray * N + i] : 0.0f;
            zz[k] = ok[k] ? z[ray * N + i] : 0.0f;
        }
        // z of the next sample: the next register, or the next lane's first one
        const float z_next_lane = __shfl_down(zz[0], 1, 64);
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const int i = base + k;
            const float zn = (k + 1 < IPL) ? zz[(k + 1) % IPL] : z_next_lane;
            dist[k] = (i == N - 1) ? 1e5f : zn - zz[k];         // src/Camera.py:153-154
            alpha[k] = ok[k] ? 1.0f - expf(-sig[k] * dist[k]) : 0.0f;  // :156
            tr[k] = ok[k] ? (1.0f - alpha[k]) + 1e-10f : 1.0f;   // :163
            T[k] = lane_prod;                                     // lane-local exclusive product
            lane_prod *= tr[k];
        }
        const float before = wave_excl_prod(lane_prod, lane);
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            T[k] = before * T[k];
            w[k] = alpha[k] * T[k];                               // :172
        }
    }
};

template <int IPL>
__global__ __launch_bounds__(CMP_WAVES * 64) void composite_fwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ sigma, const float* __restrict__ z, int64_t R, int N,
    int white, float* __restrict__ out_rgb, float* __restrict__ weights, float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * CMP_WAVES + (threadIdx.x >> 6);
    if (ray >= R) return;  // whole wave leaves together
    RaySlice<IPL> s;
    s.load(sigma, z, ray, N, lane);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, ws = 0.f, dz = 0.f;
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const int i = lane * IPL + k;
        if (s.ok[k]) {
            const float* c = rgb + (ray * N + i) * 3;
            c0 += s.w[k] * c[0];
            c1 += s.w[k] * c[1];
            c2 += s.w[k] * c[2];
            ws += s.w[k];
            dz += s.w[k] * s.zz[k];
            weights[ray * N + i] = s.w[k];
        }
    }
    c0 = wave_sum(c0);
    c1 = wave_sum(c1);
    c2 = wave_sum(c2);
    ws = wave_sum(ws);
    if (depth) dz = wave_sum(dz);
    if (lane == 0) {
        const float bg = white ? 1.0f - ws : 0.0f;  // src/Camera.py:175-176
        out_rgb[ray * 3 + 0] = c0 + bg;
        out_rgb[ray * 3 + 1] = c1 + bg;
        out_rgb[ray * 3 + 2] = c2 + bg;
        if (depth) depth[ray] = dz;
    }
}

// dL/dalpha_i = g_i T_i - (sum_{k>i} g_k w_k) / t_i with g_i = dL/dw_i
// (direct + through colour, background and depth); dL/dsigma_i = dalpha_i *
// dist_i * exp(-sigma_i dist_i); dL/drgb_i = w_i * dL/dC.
template <int IPL>
__global__ __launch_bounds__(CMP_WAVES * 64) void composite_bwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ sigma, const float* __restrict__ z,
    const float* __restrict__ d_out, const float* __restrict__ d_weights, const float* __restrict__ d_depth, int64_t R,
    int N, int white, float* __restrict__ d_rgb, float* __restrict__ d_sigma) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * CMP_WAVES + (threadIdx.x >> 6);
    if (ray >= R) return;
    RaySlice<IPL> s;
    s.load(sigma, z, ray, N, lane);
    const float g0 = d_out[ray * 3 + 0], g1 = d_out[ray * 3 + 1], g2 = d_out[ray * 3 + 2];
    const float gbg = white ? -((g0 + g1) + g2) : 0.0f;
    const float gdz = d_depth ? d_depth[ray] : 0.0f;
    float g[IPL], suf[IPL];
    float lane_sum = 0.0f;
#pragma unroll
    for (int k = IPL - 1; k >= 0; --k) {
        const int i = lane * IPL + k;
        g[k] = 0.0f;
        if (s.ok[k]) {
            const float* c = rgb + (ray * N + i) * 3;
            g[k] = ((g0 * c[0] + g1 * c[1]) + g2 * c[2]) + gbg + gdz * s.zz[k];
            if (d_weights) g[k] += d_weights[ray * N + i];
            float* dc = d_rgb + (ray * N + i) * 3;
            dc[0] = s.w[k] * g0;
            dc[1] = s.w[k] * g1;
            dc[2] = s.w[k] * g2;
        }
        suf[k] = lane_sum;  // lane-local exclusive suffix
        lane_sum += g[k] * s.w[k];
    }
    const float after = wave_excl_suffix_sum(lane_sum, lane);
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const int i = lane * IPL + k;
        if (s.ok[k]) {
            const float dalpha = g[k] * s.T[k] - (after + suf[k]) / s.tr[k];
            d_sigma[ray * N + i] = dalpha * s.dist[k] * expf(-s.sig[k] * s.dist[k]);
        }
    }
}

}  // namespace nerf

using namespace nerf;

#define CMP_DISPATCH(KERNEL, ...)                                                                                   \
    do {                                                                                                            \
        const dim3 grid((unsigned)ceil_div(R, CMP_WAVES)), block(CMP_WAVES * 64);                                    \
        const int ipl = (N + 63) / 64;                                                                              \
        if (ipl <= 1) hipLaunchKernelGGL(KERNEL<1>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);              \
        else if (ipl <= 2) hipLaunchKernelGGL(KERNEL<2>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);         \
        else if (ipl <= 3) hipLaunchKernelGGL(KERNEL<3>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);         \
        else if (ipl <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);         \
        else if (ipl <= 8) hipLaunchKernelGGL(KERNEL<8>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);         \
        else hipLaunchKernelGGL(KERNEL<16>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);                      \
    } while (0)

extern "C" int nerf_composite_fwd(const float* rgb, const float* sigma, const float* z, int64_t R, int N,
                                  int white_bkgd, float* out_rgb, float* weights, float* depth, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rgb && sigma && z && out_rgb && weights, "nerf_composite_fwd: null pointer");
    NERF_REQUIRE(R >= 0 && N >= 2 && N <= 1024, "nerf_composite_fwd: R=%lld N=%d (N must be 2..1024)", (long long)R, N);
    CMP_DISPATCH(composite_fwd_kernel, rgb, sigma, z, R, N, white_bkgd, out_rgb, weights, depth);
    return check_launch("nerf_composite_fwd");
}

extern "C" int nerf_composite_bwd(const float* rgb, const float* sigma, const float* z, const float* d_out,
                                  const float* d_weights, const float* d_depth, int64_t R, int N, int white_bkgd,
                                  float* d_rgb, float* d_sigma, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rgb && sigma && z && d_out && d_rgb && d_sigma, "nerf_composite_bwd: null pointer");
    NERF_REQUIRE(R >= 0 && N >= 2 && N <= 1024, "nerf_composite_bwd: R=%lld N=%d (N must be 2..1024)", (long long)R, N);
    CMP_DISPATCH(composite_bwd_kernel, rgb, sigma, z, d_out, d_weights, d_depth, R, N, white_bkgd, d_rgb, d_sigma);
    return check_launch("nerf_composite_bwd");
}
