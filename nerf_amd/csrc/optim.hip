// Adam step over a list of tensors in one launch for gfx950 (the optimiser of the
// reference's training loops: torch.optim.Adam(lr=5e-4) src/train_nerf_llff.py:82,
// Adam(lr=1e-2, eps=1e-15) src/train_nerf_ngp.py:55).  Bandwidth-bound: 16 B read +
// 12 B written per parameter, one pass; the 16 hash tables and 9 small matrices of the
// hash-grid model (10.4 M parameters) are 290 MB of traffic per step.
// Compiled with -ffp-contract=off so that every product and sum rounds like the separate
// ops of torch's single-tensor implementation.
#include "common.h"

namespace nerf {

constexpr int ADAM_MAX_TENSORS = 64;
constexpr int ADAM_CHUNK = 4096;   // elements per workgroup
struct AdamTensors {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];
    int count;
};

// torch/optim/adam.py _single_tensor_adam: m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
// denom = v.sqrt() / sqrt(bias2) + eps; p.addcdiv_(m, denom, value=-lr/bias1)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1m, float b2, float b2m,
                                         float bias2_sqrt, float eps, float neg_step) {
    m = m + b1m * (g - m);
    v = v * b2 + b2m * g * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p = p + neg_step * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, float b1m, float b2, float b2m, float bias2_sqrt,
                                                   float eps, float neg_step) {
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
    const long long base = (long long)(blockIdx.x - t.first_block[k]) * ADAM_CHUNK;
    const long long n = t.n[k];
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    const bool vec = (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
    constexpr int ROUNDS = ADAM_CHUNK / (256 * 4);
    if (vec && base + ADAM_CHUNK <= n) {   // a whole chunk: all requests go out before the first result is used
        float4 pp[ROUNDS], mm[ROUNDS], vv[ROUNDS], gg[ROUNDS];
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const long long i = base + (long long)(j * 256 + threadIdx.x) * 4;
            pp[j] = *reinterpret_cast<float4*>(p + i);
            mm[j] = *reinterpret_cast<float4*>(m + i);
            vv[j] = *reinterpret_cast<float4*>(v + i);
            gg[j] = *reinterpret_cast<const float4*>(g + i);
        }
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const long long i = base + (long long)(j * 256 + threadIdx.x) * 4;
            adam_one(pp[j].x, gg[j].x, mm[j].x, vv[j].x, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            adam_one(pp[j].y, gg[j].y, mm[j].y, vv[j].y, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            adam_one(pp[j].z, gg[j].z, mm[j].z, vv[j].z, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            adam_one(pp[j].w, gg[j].w, mm[j].w, vv[j].w, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            *reinterpret_cast<float4*>(p + i) = pp[j];
            *reinterpret_cast<float4*>(m + i) = mm[j];
            *reinterpret_cast<float4*>(v + i) = vv[j];
        }
        return;
    }
    for (int j = 0; j < ROUNDS; ++j) {
        const long long i = base + (long long)(j * 256 + threadIdx.x) * 4;
        if (vec && i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i),
                   vv = *reinterpret_cast<float4*>(v + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            adam_one(pp.y, gg.y, mm.y, vv.y, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            adam_one(pp.z, gg.z, mm.z, vv.z, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            adam_one(pp.w, gg.w, mm.w, vv.w, b1m, b2, b2m, bias2_sqrt, eps, neg_step);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (long long e = i; e < i + 4 && e < n; ++e) adam_one(p[e], g[e], m[e], v[e], b1m, b2, b2m, bias2_sqrt, eps, neg_step);
        }
    }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, int64_t step, float lr, float beta1,
                              float beta2, float eps, void* stream) {
    if (count == 0) return NERF_OK;
    NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, "nerf_adam_step: null pointer");
    NERF_REQUIRE(count > 0 && step >= 1, "nerf_adam_step: count=%d step=%lld", count, (long long)step);
    // the scalars as torch computes them (Python floats = double), rounded to float where a tensor op consumes them
    const double bias1 = 1.0 - pow((double)beta1, (double)step), bias2 = 1.0 - pow((double)beta2, (double)step);
    const float b1m = (float)(1.0 - (double)beta1), b2m = (float)(1.0 - (double)beta2);
    const float bias2_sqrt = (float)sqrt(bias2), neg_step = (float)(-((double)lr / bias1));
    for (int first = 0; first < count; first += ADAM_MAX_TENSORS) {
        AdamTensors t;
        t.count = 0;
        int blocks = 0;
        for (int k = first; k < count && t.count < ADAM_MAX_TENSORS; ++k) {
            NERF_REQUIRE(numel[k] >= 0, "nerf_adam_step: tensor %d has %lld elements", k, (long long)numel[k]);
            if (numel[k] == 0) continue;
            NERF_REQUIRE(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k], "nerf_adam_step: tensor %d has a null pointer", k);
            const int c = t.count++;
            t.p[c] = params[k]; t.g[c] = grads[k]; t.m[c] = exp_avg[k]; t.v[c] = exp_avg_sq[k]; t.n[c] = numel[k];
            t.first_block[c] = blocks;
            blocks += (int)ceil_div(numel[k], ADAM_CHUNK);
        }
        if (t.count == 0) continue;
        t.first_block[t.count] = blocks;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, b1m, beta2, b2m, bias2_sqrt, eps,
                           neg_step);
        if (int rc = check_launch("nerf_adam_step")) return rc;
    }
    return NERF_OK;
}
