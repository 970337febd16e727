// Alpha compositing (Camera.volume_render, src/Camera.py:143-183; the older free function
// volume_render of src/utils/get_rays.py:198-236 differs only in the distance appended after the
// last sample, 1e10 instead of 1e5, which is the `tail` argument) and its backward for gfx950.
// One wavefront owns a ray at a time; lane l holds IPL consecutive samples, the exclusive
// transmittance product is a lane-local prefix followed by a 64-lane shuffle scan, and the
// backward uses the mirrored suffix scan.
// Bandwidth-bound: 20 B/sample in, 4 B/sample out (forward).
// Compiled with -ffp-contract=off: alpha, 1-alpha+1e-10 and w round like the
// reference's separate ops.
#include "common.h"

namespace nerf {

constexpr int CMP_WAVES = 4;  // rays per workgroup

// Exclusive prefix product across the wave of each lane's total: inclusive scan inside each
// row of 16 lanes, times the product of the rows before, shifted up by one lane.
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    v *= dpp_move<DPP_ROW_SHR + 1>(1.0f, v);
    v *= dpp_move<DPP_ROW_SHR + 2>(1.0f, v);
    v *= dpp_move<DPP_ROW_SHR + 4>(1.0f, v);
    v *= dpp_move<DPP_ROW_SHR + 8>(1.0f, v);
    const float r0 = lane_value(v, 15), r1 = lane_value(v, 31), r2 = lane_value(v, 47);
    const float p1 = r0, p2 = r0 * r1, p3 = p2 * r2;
    const int row = lane >> 4;
    v *= row == 0 ? 1.0f : row == 1 ? p1 : row == 2 ? p2 : p3;
    return dpp_move<DPP_WAVE_SHR1>(1.0f, v);
}

// Exclusive suffix sum across the wave (sum of the totals of all higher lanes).
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
    v += dpp_move<DPP_ROW_SHL + 1>(0.0f, v);
    v += dpp_move<DPP_ROW_SHL + 2>(0.0f, v);
    v += dpp_move<DPP_ROW_SHL + 4>(0.0f, v);
    v += dpp_move<DPP_ROW_SHL + 8>(0.0f, v);
    const float r1 = lane_value(v, 16), r2 = lane_value(v, 32), r3 = lane_value(v, 48);
    const float s2 = r3, s1 = r3 + r2, s0 = s1 + r1;
    const int row = lane >> 4;
    v += row == 3 ? 0.0f : row == 2 ? s2 : row == 1 ? s1 : s0;
    return dpp_move<DPP_WAVE_SHL1>(0.0f, v);
}

// A lane's IPL consecutive samples of one ray: loads, then alpha, t = 1-alpha+1e-10, T and w.
// FULL: N == 64*IPL, every lane is full and the loads are unconditional vector loads.
template <int IPL, bool FULL>
struct RaySlice {
    float sig[IPL], dist[IPL], alpha[IPL], tr[IPL], T[IPL], w[IPL], zz[IPL], c[IPL][3];
    bool ok[IPL];

    __device__ __forceinline__ void load(const float* __restrict__ rgb, const float* __restrict__ sigma,
                                         const float* __restrict__ z, int64_t ray, int N, int lane) {
        const int base = lane * IPL;
        if (FULL) {
            const FloatRun<IPL> s = *reinterpret_cast<const FloatRun<IPL>*>(sigma + ray * N + base);
            const FloatRun<IPL> q = *reinterpret_cast<const FloatRun<IPL>*>(z + ray * N + base);
            const FloatRun<3 * IPL> col = *reinterpret_cast<const FloatRun<3 * IPL>*>(rgb + (ray * N + base) * 3);
#pragma unroll
            for (int k = 0; k < IPL; ++k) {
                ok[k] = true;
                sig[k] = s.v[k];
                zz[k] = q.v[k];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) c[k][ch] = col.v[3 * k + ch];
            }
        } else {
#pragma unroll
            for (int k = 0; k < IPL; ++k) {
                const int i = base + k;
                ok[k] = i < N;
                sig[k] = ok[k] ? sigma[ray * N + i] : 0.0f;
                zz[k] = ok[k] ? z[ray * N + i] : 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) c[k][ch] = ok[k] ? rgb[(ray * N + i) * 3 + ch] : 0.0f;
            }
        }
    }

    __device__ __forceinline__ void compute(int N, int lane, float tail) {
        const int base = lane * IPL;
        float lane_prod = 1.0f;
        // z of the next sample: the next register, or the next lane's first one
        const float z_next_lane = __shfl_down(zz[0], 1, 64);
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const int i = base + k;
            const float zn = (k + 1 < IPL) ? zz[(k + 1) % IPL] : z_next_lane;
            dist[k] = (i == N - 1) ? tail : zn - zz[k];         // src/Camera.py:153-154
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

// A wave takes RPW rays at once, all loads first: a ray of 64 samples is only 1.3 KB,
// too little in flight per wave to cover the memory latency at 8 waves per SIMD.
template <int IPL>
struct RaysPerWave {
    static constexpr int value = IPL == 1 ? 4 : IPL <= 4 ? 2 : 1;
};

template <int IPL, bool FULL>
__global__ __launch_bounds__(CMP_WAVES * 64) void composite_fwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ sigma, const float* __restrict__ z, int64_t R, int N,
    int white, float tail, float* __restrict__ out_rgb, float* __restrict__ weights, float* __restrict__ depth) {
    constexpr int RPW = RaysPerWave<IPL>::value;
    const int lane = threadIdx.x & 63;
    const int64_t ray0 = ((int64_t)blockIdx.x * CMP_WAVES + (threadIdx.x >> 6)) * RPW;
    if (ray0 >= R) return;  // whole wave leaves together
    RaySlice<IPL, FULL> s[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) s[q].load(rgb, sigma, z, min(ray0 + q, R - 1), N, lane);
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int64_t ray = ray0 + q;
        if (ray >= R) break;  // uniform over the wave
        s[q].compute(N, lane, tail);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, ws = 0.f, dz = 0.f;
        float wout[IPL];
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            wout[k] = s[q].w[k];
            if (s[q].ok[k]) {
                c0 += s[q].w[k] * s[q].c[k][0];
                c1 += s[q].w[k] * s[q].c[k][1];
                c2 += s[q].w[k] * s[q].c[k][2];
                ws += s[q].w[k];
                dz += s[q].w[k] * s[q].zz[k];
            }
        }
        if (FULL) {
            FloatRun<IPL> o;
#pragma unroll
            for (int k = 0; k < IPL; ++k) o.v[k] = wout[k];
            *reinterpret_cast<FloatRun<IPL>*>(weights + ray * N + lane * IPL) = o;
        } else {
#pragma unroll
            for (int k = 0; k < IPL; ++k)
                if (s[q].ok[k]) weights[ray * N + lane * IPL + k] = wout[k];
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
}

// dL/dalpha_i = g_i T_i - (sum_{k>i} g_k w_k) / t_i with g_i = dL/dw_i
// (direct + through colour, background and depth); dL/dsigma_i = dalpha_i *
// dist_i * exp(-sigma_i dist_i); dL/drgb_i = w_i * dL/dC.
template <int IPL, bool FULL>
__global__ __launch_bounds__(CMP_WAVES * 64) void composite_bwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ sigma, const float* __restrict__ z,
    const float* __restrict__ d_out, const float* __restrict__ d_weights, const float* __restrict__ d_depth, int64_t R,
    int N, int white, float tail, float* __restrict__ d_rgb, float* __restrict__ d_sigma) {
    constexpr int RPW = RaysPerWave<IPL>::value;
    const int lane = threadIdx.x & 63;
    const int64_t ray0 = ((int64_t)blockIdx.x * CMP_WAVES + (threadIdx.x >> 6)) * RPW;
    if (ray0 >= R) return;
    RaySlice<IPL, FULL> s[RPW];
    float dw[RPW][IPL];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int64_t ray = min(ray0 + q, R - 1);
        s[q].load(rgb, sigma, z, ray, N, lane);
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const int i = lane * IPL + k;
            dw[q][k] = (d_weights && (FULL || i < N)) ? d_weights[ray * N + i] : 0.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int64_t ray = ray0 + q;
        if (ray >= R) break;
        s[q].compute(N, lane, tail);
        const float g0 = d_out[ray * 3 + 0], g1 = d_out[ray * 3 + 1], g2 = d_out[ray * 3 + 2];
        const float gbg = white ? -((g0 + g1) + g2) : 0.0f;
        const float gdz = d_depth ? d_depth[ray] : 0.0f;
        float g[IPL], suf[IPL];
        FloatRun<3 * IPL> dc;
        float lane_sum = 0.0f;
#pragma unroll
        for (int k = IPL - 1; k >= 0; --k) {
            g[k] = 0.0f;
            if (s[q].ok[k]) {
                g[k] = ((g0 * s[q].c[k][0] + g1 * s[q].c[k][1]) + g2 * s[q].c[k][2]) + gbg + gdz * s[q].zz[k];
                if (d_weights) g[k] += dw[q][k];
            }
            dc.v[3 * k + 0] = s[q].w[k] * g0;
            dc.v[3 * k + 1] = s[q].w[k] * g1;
            dc.v[3 * k + 2] = s[q].w[k] * g2;
            suf[k] = lane_sum;  // lane-local exclusive suffix
            lane_sum += g[k] * s[q].w[k];
        }
        if (FULL) {
            *reinterpret_cast<FloatRun<3 * IPL>*>(d_rgb + (ray * N + lane * IPL) * 3) = dc;
        } else {
#pragma unroll
            for (int k = 0; k < IPL; ++k)
                if (s[q].ok[k]) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) d_rgb[(ray * N + lane * IPL + k) * 3 + ch] = dc.v[3 * k + ch];
                }
        }
        const float after = wave_excl_suffix_sum(lane_sum, lane);
        FloatRun<IPL> ds;
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const float dalpha = g[k] * s[q].T[k] - (after + suf[k]) / s[q].tr[k];
            ds.v[k] = dalpha * s[q].dist[k] * expf(-s[q].sig[k] * s[q].dist[k]);
        }
        if (FULL) {
            *reinterpret_cast<FloatRun<IPL>*>(d_sigma + ray * N + lane * IPL) = ds;
        } else {
#pragma unroll
            for (int k = 0; k < IPL; ++k)
                if (s[q].ok[k]) d_sigma[ray * N + lane * IPL + k] = ds.v[k];
        }
    }
}

// Occupancy-grid masking of per-sample values (src/train_nerf_ngp.py:113-117,
// src/utils/analyze_capture_results.py:283-287): cell = trunc(((p + 1) / 2) / cell_size) per axis,
// flat index x + 32 y + 1024 z (the reference's xyz_to_idx keeps its default res = 32 whatever the
// grid resolution is), value kept where grid[index] == 1 and zeroed elsewhere.  An index outside
// the grid, an IndexError in the reference, is clamped to the nearest stored cell.
__device__ __forceinline__ bool cell_occupied(const float* __restrict__ pts, int64_t p, const uint8_t* __restrict__ grid,
                                              int64_t cells, float cell) {
    long long c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = ((pts[3 * p + k] + 1.0f) / 2.0f) / cell;
        c[k] = v >= 9.0e18f ? (long long)9.0e18f : v <= -9.0e18f ? -(long long)9.0e18f : v != v ? 0 : (long long)v;   // .long() truncates
    }
    long long idx = c[0] + c[1] * 32 + c[2] * 1024;
    idx = idx < 0 ? 0 : idx >= cells ? cells - 1 : idx;
    return grid[idx] == 1;
}

__global__ __launch_bounds__(256) void occupancy_mask_kernel(const float* __restrict__ pts, const float* __restrict__ in,
                                                             int64_t P, const uint8_t* __restrict__ grid, int64_t cells,
                                                             float cell, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    out[p] = cell_occupied(pts, p, grid, cells, cell) ? in[p] : 0.0f;
}

// ---------------------------------------------------------------------------
// Skipping the samples the mask would zero.  A masked sample has sigma = 0, hence alpha = 0 and weight 0: it adds
// nothing to the colour and receives no gradient (src/train_nerf_ngp.py:113-117 zeroes it AFTER the model has
// run on it).  Dropping it BEFORE the model gives the same render for a fraction of the work.  Stream compaction in
// the points' own order (so the result does not depend on scheduling): pass 1 counts the kept points of every
// 1024-point block (wave ballots), pass 2 gives block b the offset sum(counts[0..b)) and writes, for every kept point,
// its index, its coordinates and its ray's view direction to consecutive slots; `slot` maps a point back to its
// compact position (-1: dropped).
// ---------------------------------------------------------------------------
constexpr int OCC_BLOCK = 1024;

__global__ __launch_bounds__(OCC_BLOCK) void occupancy_count_kernel(const float* __restrict__ pts, int64_t P,
                                                                    const uint8_t* __restrict__ grid, int64_t cells,
                                                                    float cell, int32_t* __restrict__ block_counts) {
    __shared__ int s_wave[OCC_BLOCK / 64];
    const int64_t p = (int64_t)blockIdx.x * OCC_BLOCK + threadIdx.x;
    const bool keep = p < P && cell_occupied(pts, p, grid, cells, cell);
    const int n = __popcll(__ballot(keep));
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < OCC_BLOCK / 64; ++w) total += s_wave[w];
        block_counts[blockIdx.x] = total;
    }
}

__global__ __launch_bounds__(OCC_BLOCK) void occupancy_compact_kernel(
    const float* __restrict__ pts, int64_t P, const uint8_t* __restrict__ grid, int64_t cells, float cell,
    const float* __restrict__ dirs, int dirs_div, const int32_t* __restrict__ block_counts, int32_t* __restrict__ kept,
    int32_t* __restrict__ slot, float* __restrict__ pts_kept, float* __restrict__ dirs_kept, int32_t* __restrict__ count) {
    __shared__ int s_wave[OCC_BLOCK / 64];
    __shared__ int s_part[OCC_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this block's offset: the kept points of all earlier blocks
    int before = 0;
    for (int b = tid; b < (int)blockIdx.x; b += OCC_BLOCK) before += block_counts[b];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
    if (lane == 0) s_part[wave] = before;
    const int64_t p = (int64_t)blockIdx.x * OCC_BLOCK + tid;
    const bool keep = p < P && cell_occupied(pts, p, grid, cells, cell);
    const unsigned long long votes = __ballot(keep);
    if (lane == 0) s_wave[wave] = __popcll(votes);
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < OCC_BLOCK / 64; ++w) {
        base += s_part[w];
        if (w < wave) base += s_wave[w];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == OCC_BLOCK - 1) {   // the last thread of all: its prefix + its wave = the total
        int total = base + s_wave[wave];
        *count = total;
    }
    if (p >= P) return;
    if (!keep) {
        slot[p] = -1;
        return;
    }
    const int k = base + __popcll(votes & ((1ull << lane) - 1ull));
    slot[p] = k;
    kept[k] = (int32_t)p;
#pragma unroll
    for (int c = 0; c < 3; ++c) pts_kept[3 * (int64_t)k + c] = pts[3 * p + c];
    if (dirs_kept) {
        const int64_t r = p / dirs_div;
#pragma unroll
        for (int c = 0; c < 3; ++c) dirs_kept[3 * (int64_t)k + c] = dirs[3 * r + c];
    }
}

// Model outputs of the kept points back at their samples' places, zeros elsewhere (a dropped sample's density is
// zero as under the mask; its colour, which the mask leaves as the model computed it, is multiplied by a zero
// weight either way).
__global__ __launch_bounds__(256) void expand_kept_kernel(const float* __restrict__ rgb_k, const float* __restrict__ sigma_k,
                                                          const int32_t* __restrict__ slot, int64_t P,
                                                          float* __restrict__ rgb, float* __restrict__ sigma) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int k = slot[p];
    sigma[p] = k >= 0 ? sigma_k[k] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[3 * p + c] = k >= 0 ? rgb_k[3 * (int64_t)k + c] : 0.0f;
}

// The backward of expand_kept: the gradients of the kept samples, in compact order.
__global__ __launch_bounds__(256) void gather_kept_kernel(const float* __restrict__ d_rgb, const float* __restrict__ d_sigma,
                                                          const int32_t* __restrict__ kept, int64_t K,
                                                          float* __restrict__ d_rgb_k, float* __restrict__ d_sigma_k) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int64_t p = kept[k];
    d_sigma_k[k] = d_sigma ? d_sigma[p] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) d_rgb_k[3 * k + c] = d_rgb ? d_rgb[3 * p + c] : 0.0f;
}

// The same from the samples' side, for a caller that does not know K on the host: sample p hands its gradient to
// position slot[p] (kept samples appear in their own order, so the writes of a wave are contiguous but for the gaps).
__global__ __launch_bounds__(256) void gather_by_slot_kernel(const float* __restrict__ d_rgb, const float* __restrict__ d_sigma,
                                                             const int32_t* __restrict__ slot, int64_t P,
                                                             float* __restrict__ d_rgb_k, float* __restrict__ d_sigma_k) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int64_t k = slot[p];
    if (k < 0) return;
    d_sigma_k[k] = d_sigma ? d_sigma[p] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) d_rgb_k[3 * k + c] = d_rgb ? d_rgb[3 * p + c] : 0.0f;
}

// ---------------------------------------------------------------------------
// The training loss of the reference's loops and its gradient, one pass each:
//   loss = mean((rgb - target)^2) [+ mean((rgb_fine - target)^2)] [+ reg * mean_r |1 - sum_n weights[r][n]|]
// (MSE of the rendered colours, src/train_nerf_llff.py:116; plus the weight regulariser of the hash-grid loop,
// src/train_nerf_ngp.py:221-222).  As separate tensor ops this is some twenty launches of a few microseconds each,
// forward and backward, a tenth of a 4096-ray hash-grid step.  The sums are formed in a fixed order (per thread,
// per wave, per workgroup, then one workgroup over the partials): the same bits from run to run.
// ---------------------------------------------------------------------------
constexpr int LOSS_THREADS = 256, LOSS_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(LOSS_THREADS) void render_loss_partial_kernel(const float* __restrict__ rgb,
                                                                           const float* __restrict__ rgb_fine,
                                                                           const float* __restrict__ target,
                                                                           const float* __restrict__ weights, int64_t R, int N,
                                                                           float* __restrict__ partials) {
    __shared__ float s_part[3][LOSS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float se = 0.0f, sf = 0.0f, sr = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * LOSS_THREADS + tid; i < 3 * R; i += (int64_t)gridDim.x * LOSS_THREADS) {
        const float t = target[i], e = rgb[i] - t;
        se += e * e;
        if (rgb_fine) {
            const float f = rgb_fine[i] - t;
            sf += f * f;
        }
    }
    if (weights) {   // one wavefront per ray: its N weights are contiguous
        const int64_t wave_id = (int64_t)blockIdx.x * (LOSS_THREADS / 64) + wave, waves = (int64_t)gridDim.x * (LOSS_THREADS / 64);
        for (int64_t r = wave_id; r < R; r += waves) {
            float s = 0.0f;
            for (int n = lane; n < N; n += 64) s += weights[r * N + n];
            s = wave_sum(s);
            if (lane == 0) sr += fabsf(1.0f - s);
        }
    }
    se = wave_sum(se);
    sf = wave_sum(sf);
    sr = wave_sum(sr);
    if (lane == 0) {
        s_part[0][wave] = se;
        s_part[1][wave] = sf;
        s_part[2][wave] = sr;
    }
    __syncthreads();
    if (tid < 3) {
        float s = 0.0f;
#pragma unroll
        for (int v = 0; v < LOSS_THREADS / 64; ++v) s += s_part[tid][v];
        partials[3 * blockIdx.x + tid] = s;
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void render_loss_final_kernel(const float* __restrict__ partials, int blocks, int64_t R,
                                                                         float reg, float* __restrict__ loss) {
    __shared__ double s_sum[3][LOSS_THREADS];
    const int tid = threadIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = tid; b < blocks; b += LOSS_THREADS)
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += (double)partials[3 * b + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) s_sum[k][tid] = s[k];
    __syncthreads();
    for (int d = LOSS_THREADS / 2; d > 0; d >>= 1) {
        if (tid < d)
#pragma unroll
            for (int k = 0; k < 3; ++k) s_sum[k][tid] += s_sum[k][tid + d];
        __syncthreads();
    }
    if (tid == 0)
        *loss = (float)(s_sum[0][0] / (double)(3 * R)) + (float)(s_sum[1][0] / (double)(3 * R)) +
                reg * (float)(s_sum[2][0] / (double)R);
}

// d_rgb = g * 2 (rgb - target) / (3 R), the same for the fine colours, d_weights[r][n] = -g * reg * sign(1 - sum_n
// weights[r][.]) / R; g = the gradient of the loss itself, read from memory.
__global__ __launch_bounds__(LOSS_THREADS) void render_loss_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ rgb_fine,
                                                                       const float* __restrict__ target,
                                                                       const float* __restrict__ weights, int64_t R, int N,
                                                                       float reg, const float* __restrict__ d_loss,
                                                                       float* __restrict__ d_rgb, float* __restrict__ d_rgb_fine,
                                                                       float* __restrict__ d_weights) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float g = *d_loss;
    const float c = g * (2.0f / (float)(3 * R));
    for (int64_t i = (int64_t)blockIdx.x * LOSS_THREADS + tid; i < 3 * R; i += (int64_t)gridDim.x * LOSS_THREADS) {
        const float t = target[i];
        if (d_rgb) d_rgb[i] = c * (rgb[i] - t);
        if (d_rgb_fine) d_rgb_fine[i] = c * (rgb_fine[i] - t);
    }
    if (d_weights) {
        const float cw = -(g * reg) / (float)R;
        const int64_t wave_id = (int64_t)blockIdx.x * (LOSS_THREADS / 64) + wave, waves = (int64_t)gridDim.x * (LOSS_THREADS / 64);
        for (int64_t r = wave_id; r < R; r += waves) {
            float s = 0.0f;
            for (int n = lane; n < N; n += 64) s += weights[r * N + n];
            s = wave_sum(s);
            const float d = 1.0f - s;
            const float v = d > 0.0f ? cw : d < 0.0f ? -cw : 0.0f;   // sign(0) = 0, as torch's abs backward
            for (int n = lane; n < N; n += 64) d_weights[r * N + n] = v;
        }
    }
}

static int loss_blocks(int64_t R) {
    const int64_t b = ceil_div(R, 64);   // at least 64 rays per workgroup
    return (int)(b < 1 ? 1 : b > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : b);
}

}  // namespace nerf

using namespace nerf;

#define CMP_LAUNCH(KERNEL, IPL, ...)                                                                                \
    do {                                                                                                            \
        const dim3 grid((unsigned)ceil_div(R, (int64_t)CMP_WAVES * RaysPerWave<IPL>::value)), block(CMP_WAVES * 64);  \
        if (N == 64 * IPL) hipLaunchKernelGGL((KERNEL<IPL, true>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<IPL, false>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);              \
    } while (0)
#define CMP_DISPATCH(KERNEL, ...)                                                                                   \
    do {                                                                                                            \
        const int ipl = (N + 63) / 64;                                                                              \
        if (ipl <= 1) CMP_LAUNCH(KERNEL, 1, __VA_ARGS__);                                                           \
        else if (ipl <= 2) CMP_LAUNCH(KERNEL, 2, __VA_ARGS__);                                                      \
        else if (ipl <= 3) CMP_LAUNCH(KERNEL, 3, __VA_ARGS__);                                                      \
        else if (ipl <= 4) CMP_LAUNCH(KERNEL, 4, __VA_ARGS__);                                                      \
        else if (ipl <= 8) CMP_LAUNCH(KERNEL, 8, __VA_ARGS__);                                                      \
        else CMP_LAUNCH(KERNEL, 16, __VA_ARGS__);                                                                   \
    } while (0)

extern "C" int nerf_composite_fwd(const float* rgb, const float* sigma, const float* z, int64_t R, int N,
                                  int white_bkgd, float tail_dist, float* out_rgb, float* weights, float* depth,
                                  void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rgb && sigma && z && out_rgb && weights, "nerf_composite_fwd: null pointer");
    NERF_REQUIRE(R >= 0 && N >= 2 && N <= 1024, "nerf_composite_fwd: R=%lld N=%d (N must be 2..1024)", (long long)R, N);
    CMP_DISPATCH(composite_fwd_kernel, rgb, sigma, z, R, N, white_bkgd, tail_dist, out_rgb, weights, depth);
    return check_launch("nerf_composite_fwd");
}

extern "C" int nerf_composite_bwd(const float* rgb, const float* sigma, const float* z, const float* d_out,
                                  const float* d_weights, const float* d_depth, int64_t R, int N, int white_bkgd,
                                  float tail_dist, float* d_rgb, float* d_sigma, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rgb && sigma && z && d_out && d_rgb && d_sigma, "nerf_composite_bwd: null pointer");
    NERF_REQUIRE(R >= 0 && N >= 2 && N <= 1024, "nerf_composite_bwd: R=%lld N=%d (N must be 2..1024)", (long long)R, N);
    CMP_DISPATCH(composite_bwd_kernel, rgb, sigma, z, d_out, d_weights, d_depth, R, N, white_bkgd, tail_dist, d_rgb, d_sigma);
    return check_launch("nerf_composite_bwd");
}

extern "C" int nerf_occupancy_mask(const float* pts, const float* values, int64_t P, const uint8_t* grid,
                                   int64_t grid_cells, int grid_res, float* out, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(pts && values && grid && out, "nerf_occupancy_mask: null pointer");
    NERF_REQUIRE(P >= 0 && grid_cells >= 1 && grid_res >= 1, "nerf_occupancy_mask: P=%lld cells=%lld res=%d", (long long)P,
                 (long long)grid_cells, grid_res);
    const float cell = (float)(1.0 / (double)grid_res);   // the Python float 1.0/grid_res, rounded where the tensor op consumes it
    hipLaunchKernelGGL(occupancy_mask_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, (hipStream_t)stream, pts,
                       values, P, grid, grid_cells, cell, out);
    return check_launch("nerf_occupancy_mask");
}

extern "C" size_t nerf_occupancy_compact_workspace_ints(int64_t P) { return P > 0 ? (size_t)ceil_div(P, OCC_BLOCK) : 0; }

extern "C" int nerf_occupancy_compact(const float* pts, int64_t P, const uint8_t* grid, int64_t grid_cells, int grid_res,
                                      const float* dirs, int dirs_div, int32_t* workspace, int32_t* kept, int32_t* slot,
                                      float* pts_kept, float* dirs_kept, int32_t* count, void* stream) {
    NERF_REQUIRE(count, "nerf_occupancy_compact: null count");
    if (P == 0) {
        launch_zero_words(count, 1, (hipStream_t)stream);
        return check_launch("nerf_occupancy_compact(clear)");
    }
    NERF_REQUIRE(pts && grid && workspace && kept && slot && pts_kept, "nerf_occupancy_compact: null pointer");
    NERF_REQUIRE(P > 0 && P < (1ll << 31) && grid_cells >= 1 && grid_res >= 1, "nerf_occupancy_compact: P=%lld cells=%lld res=%d",
                 (long long)P, (long long)grid_cells, grid_res);
    NERF_REQUIRE(!dirs_kept || (dirs && dirs_div >= 1), "nerf_occupancy_compact: dirs_kept needs dirs and dirs_div >= 1");
    const float cell = (float)(1.0 / (double)grid_res);
    const unsigned blocks = (unsigned)ceil_div(P, OCC_BLOCK);
    hipLaunchKernelGGL(occupancy_count_kernel, dim3(blocks), dim3(OCC_BLOCK), 0, (hipStream_t)stream, pts, P, grid, grid_cells,
                       cell, workspace);
    if (int rc = check_launch("nerf_occupancy_compact(count)")) return rc;
    hipLaunchKernelGGL(occupancy_compact_kernel, dim3(blocks), dim3(OCC_BLOCK), 0, (hipStream_t)stream, pts, P, grid, grid_cells,
                       cell, dirs, dirs_div, workspace, kept, slot, pts_kept, dirs_kept, count);
    return check_launch("nerf_occupancy_compact");
}

extern "C" int nerf_expand_kept(const float* rgb_kept, const float* sigma_kept, const int32_t* slot, int64_t P, float* rgb,
                                float* sigma, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(slot && rgb && sigma, "nerf_expand_kept: null pointer");
    NERF_REQUIRE(P > 0, "nerf_expand_kept: P=%lld", (long long)P);
    hipLaunchKernelGGL(expand_kept_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, (hipStream_t)stream, rgb_kept,
                       sigma_kept, slot, P, rgb, sigma);
    return check_launch("nerf_expand_kept");
}

extern "C" int nerf_gather_kept(const float* d_rgb, const float* d_sigma, const int32_t* kept, int64_t K, float* d_rgb_kept,
                                float* d_sigma_kept, void* stream) {
    if (K == 0) return NERF_OK;
    NERF_REQUIRE(kept && d_rgb_kept && d_sigma_kept, "nerf_gather_kept: null pointer");
    NERF_REQUIRE(K > 0, "nerf_gather_kept: K=%lld", (long long)K);
    hipLaunchKernelGGL(gather_kept_kernel, dim3((unsigned)ceil_div(K, 256)), dim3(256), 0, (hipStream_t)stream, d_rgb, d_sigma,
                       kept, K, d_rgb_kept, d_sigma_kept);
    return check_launch("nerf_gather_kept");
}

extern "C" int nerf_gather_kept_by_slot(const float* d_rgb, const float* d_sigma, const int32_t* slot, int64_t P,
                                        float* d_rgb_kept, float* d_sigma_kept, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(slot && d_rgb_kept && d_sigma_kept, "nerf_gather_kept_by_slot: null pointer");
    NERF_REQUIRE(P > 0, "nerf_gather_kept_by_slot: P=%lld", (long long)P);
    hipLaunchKernelGGL(gather_by_slot_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, (hipStream_t)stream, d_rgb, d_sigma,
                       slot, P, d_rgb_kept, d_sigma_kept);
    return check_launch("nerf_gather_kept_by_slot");
}

extern "C" size_t nerf_render_loss_workspace_floats(int64_t R) { return R > 0 ? 3 * (size_t)loss_blocks(R) : 0; }

extern "C" int nerf_render_loss_fwd(const float* rgb, const float* rgb_fine, const float* target, const float* weights,
                                    int64_t R, int N, float reg, float* workspace, float* loss, void* stream) {
    NERF_REQUIRE(rgb && target && workspace && loss, "nerf_render_loss_fwd: null pointer");
    NERF_REQUIRE(R > 0 && (!weights || N > 0), "nerf_render_loss_fwd: R=%lld N=%d", (long long)R, N);
    const int blocks = loss_blocks(R);
    hipLaunchKernelGGL(render_loss_partial_kernel, dim3(blocks), dim3(LOSS_THREADS), 0, (hipStream_t)stream, rgb, rgb_fine, target,
                       weights, R, N, workspace);
    if (int rc = check_launch("nerf_render_loss_fwd")) return rc;
    hipLaunchKernelGGL(render_loss_final_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, workspace, blocks, R,
                       weights ? reg : 0.0f, loss);
    return check_launch("nerf_render_loss_fwd(final)");
}

extern "C" int nerf_render_loss_bwd(const float* rgb, const float* rgb_fine, const float* target, const float* weights,
                                    int64_t R, int N, float reg, const float* d_loss, float* d_rgb, float* d_rgb_fine,
                                    float* d_weights, void* stream) {
    NERF_REQUIRE(rgb && target && d_loss, "nerf_render_loss_bwd: null pointer");
    NERF_REQUIRE(R > 0 && (!d_weights || (weights && N > 0)) && (!d_rgb_fine || rgb_fine), "nerf_render_loss_bwd: R=%lld N=%d",
                 (long long)R, N);
    hipLaunchKernelGGL(render_loss_bwd_kernel, dim3(loss_blocks(R)), dim3(LOSS_THREADS), 0, (hipStream_t)stream, rgb, rgb_fine,
                       target, weights, R, N, reg, d_loss, d_rgb, d_rgb_fine, d_weights);
    return check_launch("nerf_render_loss_bwd");
}
