// Ray generation, stratified sampling, hierarchical (inverse-CDF) sampling and
// the NDC transform for gfx950.  All kernels here are bandwidth-bound; this file
// is compiled with -ffp-contract=off so that every multiply/add rounds exactly
// like the separate PyTorch ops of the reference (bit-exact depths).
#include "common.h"

namespace nerf {

struct Pose {
    float m[12];
};

__device__ __forceinline__ void pixel_ray(const float* __restrict__ c2w, float fx, float fy, float cx, float cy,
                                          int W, int pix, float o[3], float d[3]) {
    // src/Camera.py:62-70: dirs = ((i-cx)/fx, -(j-cy)/fy, -1); rays_d = normalize(R * dirs)
    const int row = pix / W, col = pix - row * W;
    const float dx = ((float)col - cx) / fx;
    const float dy = -((float)row - cy) / fy;
    const float dz = -1.0f;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = (dx * c2w[4 * k + 0] + dy * c2w[4 * k + 1]) + dz * c2w[4 * k + 2];
    const float n = sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = r[k] / n;
        o[k] = c2w[4 * k + 3];
    }
}

__global__ void raygen_kernel(const float* __restrict__ c2w, float fx, float fy, float cx, float cy, int W,
                              const int32_t* __restrict__ pix, int64_t R, float* __restrict__ rays_o,
                              float* __restrict__ rays_d) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float o[3], d[3];
    pixel_ray(c2w, fx, fy, cx, cy, W, pix ? pix[r] : (int)r, o, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_o[3 * r + k] = o[k];
        rays_d[3 * r + k] = d[k];
    }
}

// Depth of sample n: src/Camera.py:133-137 (midpoint strata, lower + (upper-lower)*u).
__device__ __forceinline__ float stratified_depth(const float* __restrict__ z_lin, int N, int n, const float* u,
                                                  int64_t un) {
    const float zc = z_lin[n];
    if (!u) return zc;
    const float lower = (n == 0) ? zc : 0.5f * (zc + z_lin[n - 1]);
    const float upper = (n == N - 1) ? zc : 0.5f * (z_lin[n + 1] + zc);
    return lower + (upper - lower) * u[un];
}

// One thread per (ray, sample).  RAYGEN: derive the ray from the pixel instead of loading it.
template <bool RAYGEN>
__global__ void sample_kernel(const float* __restrict__ c2w, float fx, float fy, float cx, float cy, int W,
                              const int32_t* __restrict__ pix, const float* __restrict__ in_o,
                              const float* __restrict__ in_d, const float* __restrict__ z_lin,
                              const float* __restrict__ u, int64_t R, int N, float* __restrict__ rays_o,
                              float* __restrict__ rays_d, float* __restrict__ z_vals, float* __restrict__ pts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * N) return;
    const int64_t r = t / N;
    const int n = (int)(t - r * N);
    float o[3], d[3];
    if (RAYGEN) {
        pixel_ray(c2w, fx, fy, cx, cy, W, pix ? pix[r] : (int)r, o, d);
        if (n == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (rays_o) rays_o[3 * r + k] = o[k];
                if (rays_d) rays_d[3 * r + k] = d[k];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = in_o[3 * r + k];
            d[k] = in_d[3 * r + k];
        }
    }
    const float z = stratified_depth(z_lin, N, n, u, t);
    if (z_vals) z_vals[t] = z;
    if (pts) {
#pragma unroll
        for (int k = 0; k < 3; ++k) pts[3 * t + k] = o[k] + d[k] * z;  // src/Camera.py:140
    }
}

// ---------------------------------------------------------------------------
// Hierarchical sampling: one wavefront per ray, 4 rays per workgroup.
// LDS per wave: cdf scan ping-pong (2*256), bins (256), merged depths (1024).
// ---------------------------------------------------------------------------
constexpr int PDF_MAX_NC = 256;
constexpr int PDF_MAX_ALL = 1024;
constexpr int PDF_WAVES = 4;

__global__ __launch_bounds__(PDF_WAVES * 64) void pdf_sample_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ z,
    const float* __restrict__ weights, const float* __restrict__ u, int64_t R, int Nc, int Nf,
    float* __restrict__ z_fine, int32_t* __restrict__ idx_out, float* __restrict__ z_all, float* __restrict__ pts) {
    __shared__ float s_scan[PDF_WAVES][2][PDF_MAX_NC];
    __shared__ float s_bins[PDF_WAVES][PDF_MAX_NC];
    __shared__ float s_all[PDF_WAVES][PDF_MAX_ALL];
    __shared__ float s_sorted[PDF_WAVES][PDF_MAX_ALL];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray_raw = (int64_t)blockIdx.x * PDF_WAVES + wv;
    const bool live = ray_raw < R;
    const int64_t ray = live ? ray_raw : R - 1;  // idle waves shadow the last ray (barriers stay uniform)
    const float* zr = z + ray * Nc;
    const float* wr = weights + ray * Nc;
    const int nb = Nc - 1;  // bins and cdf entries
    const int nw = Nc - 2;  // interior weights

    for (int k = lane; k < nb; k += 64) s_bins[wv][k] = 0.5f * (zr[k + 1] + zr[k]);
    for (int k = lane; k < nw; k += 64) s_scan[wv][0][k] = wr[k + 1] + 1e-5f;
    for (int k = lane; k < Nc; k += 64) s_all[wv][k] = zr[k];
    __syncthreads();
    // Log-step inclusive scan; the association order is part of the contract
    // (the CPU oracle performs the same additions).
    int cur = 0;
    for (int dd = 1; dd < nw; dd <<= 1) {
        for (int k = lane; k < nw; k += 64) {
            const float v = s_scan[wv][cur][k];
            s_scan[wv][cur ^ 1][k] = (k >= dd) ? v + s_scan[wv][cur][k - dd] : v;
        }
        __syncthreads();
        cur ^= 1;
    }
    const float total = s_scan[wv][cur][nw - 1];
    __syncthreads();
    // cdf[0] = 0, cdf[k+1] = c[k] / total, stored in the other half of the ping-pong
    float* cdf = s_scan[wv][cur ^ 1];
    for (int k = lane; k < nb; k += 64) cdf[k] = (k == 0) ? 0.0f : s_scan[wv][cur][k - 1] / total;
    __syncthreads();

    for (int f = lane; f < Nf; f += 64) {
        const float uu = u[ray * Nf + f];
        int lo = 0, hi = nb;  // first entry with cdf > uu
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, nb - 1);
        const float c_lo = cdf[below], c_hi = cdf[above];
        const float b_lo = s_bins[wv][below], b_hi = s_bins[wv][above];
        float denom = c_hi - c_lo;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (uu - c_lo) / denom;
        const float zf = b_lo + t * (b_hi - b_lo);
        s_all[wv][Nc + f] = zf;
        if (live) {
            if (z_fine) z_fine[ray * Nf + f] = zf;
            if (idx_out) idx_out[ray * Nf + f] = lo;
        }
    }
    __syncthreads();
    // Rank sort of the merged depths (ties broken by position: a stable sort).
    const int na = Nc + Nf;
    for (int i = lane; i < na; i += 64) {
        const float v = s_all[wv][i];
        int rank = 0;
        for (int j = 0; j < na; ++j) {
            const float w = s_all[wv][j];
            rank += (w < v) || (w == v && j < i);
        }
        s_sorted[wv][rank] = v;
    }
    __syncthreads();
    if (!live) return;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = rays_o[3 * ray + k];
        d[k] = rays_d[3 * ray + k];
    }
    for (int i = lane; i < na; i += 64) {
        const float v = s_sorted[wv][i];
        if (z_all) z_all[ray * na + i] = v;
        if (pts) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pts[(ray * na + i) * 3 + k] = o[k] + d[k] * v;
        }
    }
}

__global__ void ndc_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t R, float ax,
                           float ay, float near_, float* __restrict__ out_o, float* __restrict__ out_d) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = rays_o[3 * r + k];
        d[k] = rays_d[3 * r + k];
    }
    const float t = -(near_ + o[2]) / d[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = o[k] + t * d[k];
    const float oxz = o[0] / o[2], oyz = o[1] / o[2];
    out_o[3 * r + 0] = ax * oxz;
    out_o[3 * r + 1] = ay * oyz;
    out_o[3 * r + 2] = 1.0f + 2.0f * near_ / o[2];
    out_d[3 * r + 0] = ax * (d[0] / d[2] - oxz);
    out_d[3 * r + 1] = ay * (d[1] / d[2] - oyz);
    out_d[3 * r + 2] = -2.0f * near_ / o[2];
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_raygen(const float* c2w, float fx, float fy, float cx, float cy, int H, int W,
                           const int32_t* pix, int64_t R, float* rays_o, float* rays_d, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(c2w && rays_o && rays_d, "nerf_raygen: null pointer");
    NERF_REQUIRE(H > 0 && W > 0 && R >= 0, "nerf_raygen: bad size H=%d W=%d R=%lld", H, W, (long long)R);
    NERF_REQUIRE(pix || R == (int64_t)H * W, "nerf_raygen: R=%lld must equal H*W when pix is NULL", (long long)R);
    NERF_REQUIRE((int64_t)H * W < (1ll << 31), "nerf_raygen: image too large");
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, c2w, fx, fy,
                       cx, cy, W, pix, R, rays_o, rays_d);
    return check_launch("nerf_raygen");
}

extern "C" int nerf_stratified_sample(const float* rays_o, const float* rays_d, const float* z_lin, const float* u,
                                      int64_t R, int N, float* z_vals, float* pts, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rays_o && rays_d && z_lin, "nerf_stratified_sample: null pointer");
    NERF_REQUIRE(R >= 0 && N >= 1, "nerf_stratified_sample: bad size R=%lld N=%d", (long long)R, N);
    hipLaunchKernelGGL(sample_kernel<false>, dim3((unsigned)ceil_div(R * N, 256)), dim3(256), 0, (hipStream_t)stream,
                       nullptr, 0.f, 0.f, 0.f, 0.f, 0, nullptr, rays_o, rays_d, z_lin, u, R, N, nullptr, nullptr,
                       z_vals, pts);
    return check_launch("nerf_stratified_sample");
}

extern "C" int nerf_raygen_sample(const float* c2w, float fx, float fy, float cx, float cy, int H, int W,
                                  const int32_t* pix, int64_t R, const float* z_lin, const float* u, int N,
                                  float* rays_o, float* rays_d, float* z_vals, float* pts, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(c2w && z_lin, "nerf_raygen_sample: null pointer");
    NERF_REQUIRE(H > 0 && W > 0 && R >= 0 && N >= 1, "nerf_raygen_sample: bad size");
    NERF_REQUIRE(pix || R == (int64_t)H * W, "nerf_raygen_sample: R must equal H*W when pix is NULL");
    NERF_REQUIRE((int64_t)H * W < (1ll << 31), "nerf_raygen_sample: image too large");
    hipLaunchKernelGGL(sample_kernel<true>, dim3((unsigned)ceil_div(R * N, 256)), dim3(256), 0, (hipStream_t)stream, c2w,
                       fx, fy, cx, cy, W, pix, nullptr, nullptr, z_lin, u, R, N, rays_o, rays_d, z_vals, pts);
    return check_launch("nerf_raygen_sample");
}

extern "C" int nerf_pdf_sample(const float* rays_o, const float* rays_d, const float* z, const float* weights,
                               const float* u, int64_t R, int Nc, int Nf, float* z_fine, int32_t* idx, float* z_all,
                               float* pts, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rays_o && rays_d && z && weights && u, "nerf_pdf_sample: null pointer");
    NERF_REQUIRE(Nc >= 3 && Nc <= PDF_MAX_NC, "nerf_pdf_sample: Nc=%d outside [3,%d]", Nc, PDF_MAX_NC);
    NERF_REQUIRE(Nf >= 1 && Nc + Nf <= PDF_MAX_ALL, "nerf_pdf_sample: Nc+Nf=%d exceeds %d", Nc + Nf, PDF_MAX_ALL);
    NERF_REQUIRE(R >= 0, "nerf_pdf_sample: R=%lld", (long long)R);
    hipLaunchKernelGGL(pdf_sample_kernel, dim3((unsigned)ceil_div(R, PDF_WAVES)), dim3(PDF_WAVES * 64), 0,
                       (hipStream_t)stream, rays_o, rays_d, z, weights, u, R, Nc, Nf, z_fine, idx, z_all, pts);
    return check_launch("nerf_pdf_sample");
}

extern "C" int nerf_ndc_rays(const float* rays_o, const float* rays_d, int64_t R, int H, int W, float focal,
                             float near_, float* out_o, float* out_d, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rays_o && rays_d && out_o && out_d, "nerf_ndc_rays: null pointer");
    NERF_REQUIRE(R >= 0 && H > 0 && W > 0, "nerf_ndc_rays: bad size");
    const float ax = (float)(-(double)focal / (0.5 * W)), ay = (float)(-(double)focal / (0.5 * H));
    hipLaunchKernelGGL(ndc_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                       R, ax, ay, near_, out_o, out_d);
    return check_launch("nerf_ndc_rays");
}
