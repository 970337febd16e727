// The optimiser side of a training step for gfx950.
// Adam over a list of tensors in one launch (the optimiser of the reference's training loops:
// torch.optim.Adam(lr=5e-4, weight_decay=1e-6) src/train_nerf_llff.py:82 and src/train_nerf_skow.py:47,
// Adam(lr=1e-2, eps=1e-15) src/train_nerf_ngp.py:55), and gradient-norm clipping over the same kind of
// list (torch.nn.utils.clip_grad_norm_(..., max_norm=1.0), src/train_nerf.py:165).  Bandwidth-bound: 16 B read +
// 12 B written per parameter, one pass; the 16 hash tables and 9 small matrices of the
// hash-grid model (10.4 M parameters) are 290 MB of traffic per step.
// Compiled with -ffp-contract=off so that every product and sum rounds like the separate
// ops of torch's single-tensor implementation.
#include <atomic>

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
// weight decay: grad = grad.add(param, alpha=wd), one fused multiply-add as torch's add kernel performs it
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1m, float b2, float b2m,
                                         float bias2_sqrt, float eps, float neg_step, float wd) {
    if (wd != 0.0f) g = fmaf(wd, p, g);
    m = m + b1m * (g - m);
    v = v * b2 + b2m * g * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p = p + neg_step * (m / denom);
}

// dev_scalars: null, or the step state in device memory that adam_advance_kernel has just brought up to date
// ([1] = sqrt(bias2), [2] = -lr / bias1), used in place of the two kernel arguments.
__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, float b1m, float b2, float b2m, float bias2_sqrt,
                                                   float eps, float neg_step, float wd, const float* __restrict__ dev_scalars) {
    if (dev_scalars) {
        bias2_sqrt = dev_scalars[1];
        neg_step = dev_scalars[2];
    }
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
            adam_one(pp[j].x, gg[j].x, mm[j].x, vv[j].x, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
            adam_one(pp[j].y, gg[j].y, mm[j].y, vv[j].y, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
            adam_one(pp[j].z, gg[j].z, mm[j].z, vv[j].z, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
            adam_one(pp[j].w, gg[j].w, mm[j].w, vv[j].w, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
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
            adam_one(pp.x, gg.x, mm.x, vv.x, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
            adam_one(pp.y, gg.y, mm.y, vv.y, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
            adam_one(pp.z, gg.z, mm.z, vv.z, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
            adam_one(pp.w, gg.w, mm.w, vv.w, b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (long long e = i; e < i + 4 && e < n; ++e) adam_one(p[e], g[e], m[e], v[e], b1m, b2, b2m, bias2_sqrt, eps, neg_step, wd);
        }
    }
}

// Neither the step count nor the learning rate of an optimiser whose launches are replayed from a HIP graph can be
// a kernel argument (it would be frozen into the graph, and the reference's loops change the rate every epoch:
// src/train_nerf.py:87,204, src/train_nerf_ngp.py:57-62,373-379): state[0] holds the count and state[3] the rate in
// device memory.  One thread advances the count and derives the two step-dependent scalars the way the host does
// for nerf_adam_step (in double, rounded to float).
__global__ void adam_advance_kernel(float* __restrict__ state, float beta1, float beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double step = (double)state[0] + 1.0;
    const double bias1 = 1.0 - pow((double)beta1, step), bias2 = 1.0 - pow((double)beta2, step);
    state[0] = (float)step;
    state[1] = (float)sqrt(bias2);
    state[2] = (float)(-((double)state[3] / bias1));
}

// The rate arrives as a kernel argument and leaves by a vector store: no host buffer that a replayed graph could
// outlive, and ordered on the stream like any launch (issued in front of a replay, never inside a capture).
__global__ void adam_set_lr_kernel(float* __restrict__ state, float lr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[3] = lr;
}

// ---------------------------------------------------------------------------
// Gradient-norm clipping: total = || all gradients ||_2, coef = min(1, max_norm / (total + 1e-6)),
// every gradient *= coef (torch.nn.utils.clip_grad_norm_).  Three launches, no host round trip:
// one partial sum of squares per 4096-element chunk, one workgroup that folds the partials in a
// fixed order (the result is the same from run to run) and publishes total and coef, one scaling
// pass that reads coef from memory.  Scratch: float [0] = total norm, [1] = coef, [2 + b] = partial
// of chunk b.
// ---------------------------------------------------------------------------
constexpr int CLIP_CHUNK = ADAM_CHUNK;
struct ClipTensors {
    float* g[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];
    int count;
};

__device__ __forceinline__ int clip_tensor_of_block(const ClipTensors& t, int block) {
    int k = 0;
    while (k + 1 < t.count && block >= t.first_block[k + 1]) ++k;
    return k;
}

__global__ __launch_bounds__(256) void clip_sumsq_kernel(ClipTensors t, float* __restrict__ partials) {
    __shared__ float s_wave[4];
    const int k = clip_tensor_of_block(t, blockIdx.x);
    const long long base = (long long)(blockIdx.x - t.first_block[k]) * CLIP_CHUNK, n = t.n[k];
    const float* __restrict__ g = t.g[k];
    const bool vec = ((size_t)g & 15) == 0;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < CLIP_CHUNK / 1024; ++j) {
        const long long i = base + (long long)(j * 256 + threadIdx.x) * 4;
        if (vec && i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            for (long long e = i; e < i + 4 && e < n; ++e) s += g[e] * g[e];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
}

__global__ __launch_bounds__(256) void clip_coef_kernel(float* __restrict__ scratch, int n_partials, float max_norm,
                                                        float* __restrict__ total_out) {
    __shared__ double s_part[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += (double)scratch[2 + i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) s_part[threadIdx.x] += s_part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(s_part[0]);
        const float coef = max_norm / (total + 1e-6f);
        scratch[0] = total;
        scratch[1] = (coef < 1.0f || coef != coef) ? coef : 1.0f;   // a NaN norm reaches every gradient, as torch's clamp lets it
        if (total_out) *total_out = total;
    }
}

__global__ __launch_bounds__(256) void clip_scale_kernel(ClipTensors t, const float* __restrict__ scratch) {
    const float coef = scratch[1];
    if (coef == 1.0f) return;   // x * 1 = x: nothing to write
    const int k = clip_tensor_of_block(t, blockIdx.x);
    const long long base = (long long)(blockIdx.x - t.first_block[k]) * CLIP_CHUNK, n = t.n[k];
    float* __restrict__ g = t.g[k];
    const bool vec = ((size_t)g & 15) == 0;
#pragma unroll
    for (int j = 0; j < CLIP_CHUNK / 1024; ++j) {
        const long long i = base + (long long)(j * 256 + threadIdx.x) * 4;
        if (vec && i + 4 <= n) {
            float4 v = *reinterpret_cast<float4*>(g + i);
            v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
            *reinterpret_cast<float4*>(g + i) = v;
        } else {
            for (long long e = i; e < i + 4 && e < n; ++e) g[e] *= coef;
        }
    }
}

// Chunks (= partial sums) the tensors need, or -1 for a bad list.
static long long clip_chunks(int count, float* const* grads, const int64_t* numel) {
    long long chunks = 0;
    for (int k = 0; k < count; ++k) {
        if (numel[k] < 0 || (numel[k] > 0 && !grads[k])) return -1;
        chunks += ceil_div(numel[k], CLIP_CHUNK);
    }
    return chunks;
}

// The three launches over a scratch buffer of 2 + clip_chunks() floats.  SCALE = false: first pass, true: last pass.
template <bool SCALE>
static int clip_pass(int count, float* const* grads, const int64_t* numel, float* scratch, hipStream_t stream) {
    int done = 0;   // chunks of earlier batches
    for (int k = 0; k < count;) {
        ClipTensors t;
        t.count = 0;
        int blocks = 0;
        for (; k < count && t.count < ADAM_MAX_TENSORS; ++k) {
            if (numel[k] == 0) continue;
            const int c = t.count++;
            t.g[c] = grads[k]; t.n[c] = numel[k];
            t.first_block[c] = blocks;
            blocks += (int)ceil_div(numel[k], CLIP_CHUNK);
        }
        if (t.count == 0) continue;
        t.first_block[t.count] = blocks;
        if (SCALE) hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, stream, t, scratch);
        else hipLaunchKernelGGL(clip_sumsq_kernel, dim3(blocks), dim3(256), 0, stream, t, scratch + 2 + done);
        if (int rc = check_launch("nerf_clip_grad_norm")) return rc;
        done += blocks;
    }
    return NERF_OK;
}

static int clip_run(int count, float* const* grads, const int64_t* numel, long long chunks, float max_norm, float* scratch,
                    float* total_norm, hipStream_t stream) {
    if (int rc = clip_pass<false>(count, grads, numel, scratch, stream)) return rc;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, scratch, (int)chunks, max_norm, total_norm);
    if (int rc = check_launch("nerf_clip_grad_norm(coef)")) return rc;
    return clip_pass<true>(count, grads, numel, scratch, stream);
}

// ---------------------------------------------------------------------------
// The same clipping for callers that cannot allocate (code under HIP graph capture, serving or
// training loops with a fixed memory plan): the scratch floats live in this file-scope device
// array instead of a caller workspace.  Each GPU holds its own copy of the array at its own
// address, looked up once per device; `scratch` itself is only a host-side handle and is never
// passed to a kernel or a memset.
// ---------------------------------------------------------------------------
constexpr long long CLIP_STATIC_MAX_CHUNKS = 65536;   // 268,435,456 gradient elements (1 GiB of float32)
static __device__ __align__(256) float scratch[2 + CLIP_STATIC_MAX_CHUNKS + 62];   // 256.25 KiB per GPU
constexpr int CLIP_MAX_DEVICES = 64;
static std::atomic<float*> scratch_ptr[CLIP_MAX_DEVICES];

// Device address of the current device's copy of `scratch` (looked up on the first call on that device).
static int clip_static_scratch(float** out) {
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= CLIP_MAX_DEVICES)
        return fail(NERF_ELAUNCH, "nerf_clip_grad_norm_static: no current device (or device index above %d)", CLIP_MAX_DEVICES - 1);
    float* p = scratch_ptr[device].load(std::memory_order_acquire);
    if (!p) {
        const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(NERF_ELAUNCH, "nerf_clip_grad_norm_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
        scratch_ptr[device].store(p, std::memory_order_release);   // two threads racing here store the same address
    }
    *out = p;
    return NERF_OK;
}

}  // namespace nerf

using namespace nerf;

// step_state: null (the step count is `step`), or 4 floats in device memory whose first holds the count of steps taken.
static int adam_run(const char* who, int count, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, int64_t step, float* step_state, float lr, float beta1,
                    float beta2, float eps, float weight_decay, hipStream_t stream) {
    // the scalars as torch computes them (Python floats = double), rounded to float where a tensor op consumes them
    const float b1m = (float)(1.0 - (double)beta1), b2m = (float)(1.0 - (double)beta2);
    float bias2_sqrt = 1.0f, neg_step = 0.0f;
    if (step_state) {
        hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, stream, step_state, beta1, beta2);
        if (int rc = check_launch(who)) return rc;
    } else {
        const double bias1 = 1.0 - pow((double)beta1, (double)step), bias2 = 1.0 - pow((double)beta2, (double)step);
        bias2_sqrt = (float)sqrt(bias2);
        neg_step = (float)(-((double)lr / bias1));
    }
    for (int k = 0; k < count;) {   // batches of up to ADAM_MAX_TENSORS non-empty tensors; k carries over
        AdamTensors t;
        t.count = 0;
        int blocks = 0;
        for (; k < count && t.count < ADAM_MAX_TENSORS; ++k) {
            NERF_REQUIRE(numel[k] >= 0, "%s: tensor %d has %lld elements", who, k, (long long)numel[k]);
            if (numel[k] == 0) continue;
            NERF_REQUIRE(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k], "%s: tensor %d has a null pointer", who, k);
            const int c = t.count++;
            t.p[c] = params[k]; t.g[c] = grads[k]; t.m[c] = exp_avg[k]; t.v[c] = exp_avg_sq[k]; t.n[c] = numel[k];
            t.first_block[c] = blocks;
            blocks += (int)ceil_div(numel[k], ADAM_CHUNK);
        }
        if (t.count == 0) continue;
        t.first_block[t.count] = blocks;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, t, b1m, beta2, b2m, bias2_sqrt, eps, neg_step,
                           weight_decay, (const float*)step_state);
        if (int rc = check_launch(who)) return rc;
    }
    return NERF_OK;
}

extern "C" int nerf_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, int64_t step, float lr, float beta1,
                              float beta2, float eps, float weight_decay, void* stream) {
    if (count == 0) return NERF_OK;
    NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, "nerf_adam_step: null pointer");
    NERF_REQUIRE(count > 0 && step >= 1, "nerf_adam_step: count=%d step=%lld", count, (long long)step);
    return adam_run("nerf_adam_step", count, params, grads, exp_avg, exp_avg_sq, numel, step, nullptr, lr, beta1, beta2, eps,
                    weight_decay, (hipStream_t)stream);
}

extern "C" int nerf_adam_step_capturable(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                                         float* const* exp_avg_sq, const int64_t* numel, float* step_state,
                                         float beta1, float beta2, float eps, float weight_decay, void* stream) {
    NERF_REQUIRE(step_state, "nerf_adam_step_capturable: null step state");
    NERF_REQUIRE(count >= 0 && (count == 0 || (params && grads && exp_avg && exp_avg_sq && numel)),
                 "nerf_adam_step_capturable: null pointer");
    return adam_run("nerf_adam_step_capturable", count, params, grads, exp_avg, exp_avg_sq, numel, 0, step_state, 0.0f, beta1,
                    beta2, eps, weight_decay, (hipStream_t)stream);
}

extern "C" int nerf_adam_set_lr(float* step_state, float lr, void* stream) {
    NERF_REQUIRE(step_state, "nerf_adam_set_lr: null step state");
    NERF_REQUIRE(lr >= 0.0f, "nerf_adam_set_lr: lr=%g", (double)lr);
    hipLaunchKernelGGL(adam_set_lr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_state, lr);
    return check_launch("nerf_adam_set_lr");
}

extern "C" size_t nerf_clip_grad_norm_workspace_floats(int count, const int64_t* numel) {
    size_t chunks = 0;
    for (int k = 0; k < count; ++k) chunks += numel[k] > 0 ? (size_t)ceil_div(numel[k], CLIP_CHUNK) : 0;
    return 2 + chunks;
}

extern "C" int nerf_clip_grad_norm(int count, float* const* grads, const int64_t* numel, float max_norm, float* workspace,
                                   float* total_norm, void* stream) {
    NERF_REQUIRE(count >= 0 && (count == 0 || (grads && numel)) && workspace, "nerf_clip_grad_norm: null pointer");
    const long long chunks = clip_chunks(count, grads, numel);
    NERF_REQUIRE(chunks >= 0 && chunks < (1ll << 31), "nerf_clip_grad_norm: bad tensor list");
    return clip_run(count, grads, numel, chunks, max_norm, workspace, total_norm, (hipStream_t)stream);
}

extern "C" int nerf_clip_grad_norm_static(int count, float* const* grads, const int64_t* numel, float max_norm,
                                          float* total_norm, void* stream) {
    NERF_REQUIRE(count >= 0 && (count == 0 || (grads && numel)), "nerf_clip_grad_norm_static: null pointer");
    const long long chunks = clip_chunks(count, grads, numel);
    NERF_REQUIRE(chunks >= 0, "nerf_clip_grad_norm_static: bad tensor list");
    NERF_REQUIRE(chunks <= CLIP_STATIC_MAX_CHUNKS,
                 "nerf_clip_grad_norm_static: %lld chunks of %d elements exceed the built-in scratch of %lld chunks "
                 "(%lld elements in all); use nerf_clip_grad_norm with a workspace",
                 chunks, CLIP_CHUNK, CLIP_STATIC_MAX_CHUNKS, CLIP_STATIC_MAX_CHUNKS * CLIP_CHUNK);
    float* ptr = nullptr;
    if (int rc = clip_static_scratch(&ptr)) return rc;
    // nothing in the scratch has to start at zero: every partial that is read is written first by this call
    return clip_run(count, grads, numel, chunks, max_norm, ptr, total_norm, (hipStream_t)stream);
}
