// Kernels shared by the fused MLPs: weight packing, the weight-gradient GEMMs
// (contraction over points on the exact-f32 MFMA) and the small-head gradients.
#include "mlp_common.h"

namespace nerf {

// ------------------------------------------------------------- packing ----
__global__ void pack_kernel(PackJobs jobs, float* __restrict__ packed) {
    const PackJob& jb = jobs.job[blockIdx.y];
    const int total = jb.mode == 2 ? jb.count : jb.KT * 16 * 64 * jb.OT;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (jb.mode == 2) {
            v = jb.src[idx];
        } else {
            const int ot = idx % jb.OT;
            const int lane = (idx / jb.OT) % 64;
            const int r = (idx / (jb.OT * 64)) % 16;
            const int kt = idx / (jb.OT * 64 * 16);
            const int out_i = 32 * ot + (lane & 31);           // row of the packed operand (MFMA row i)
            const int k = 32 * kt + rowmap(r, lane >> 5);      // contraction index
            if (jb.mode == 0) {  // forward: operand[n][k_eff] = W[n][col(k_eff)]
                int col = -1;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (k >= jb.seg_eff[s] && k < jb.seg_eff[s] + jb.seg_len[s]) col = jb.seg_src[s] + (k - jb.seg_eff[s]);
                if (out_i < jb.N && col >= 0) v = jb.src[(size_t)out_i * jb.ld + col];
            } else {             // backward: operand[m][n] = W[n][col_begin + m]
                if (k < jb.N && out_i < jb.col_count) v = jb.src[(size_t)k * jb.ld + jb.col_begin + out_i];
            }
        }
        packed[jb.dst + idx] = v;
    }
}

int launch_pack(const PackJobs& jobs, float* packed, hipStream_t stream) {
    hipLaunchKernelGGL(pack_kernel, dim3(64, jobs.n), dim3(256), 0, stream, jobs, packed);
    return check_launch("pack");
}

// ---------------------------------------------------------------- wgrad ----
// One workgroup (4 waves) owns one point slab of one part; wave w owns n-tiles
// [w*NTW, (w+1)*NTW) and all k-tiles (KT from X plus KT2 from X2), i.e.
// NTW*(KT+KT2) accumulator tiles.  Operands come straight from the row-major
// activation / gradient rows: for one MFMA k-step a wave reads two 128-byte row
// segments per tile.  Loads run one stage (WG_U k-steps) ahead of the MFMAs.
// U = k-steps (2 points each) per software-pipeline stage

template <int NTW, int KT, int KT2, int WG_U>
struct WgradOperands {
    float a[WG_U][NTW];
    float b[WG_U][KT + KT2];
    __device__ __forceinline__ void load(const WgradPart& pt, long long p, int n0, int i, int s) {
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            const long long row = p + 2 * u + s;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) a[u][nt] = pt.dY[row * pt.ldY + n0 + 32 * nt + i];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) b[u][kt] = pt.X[row * pt.ldX + 32 * kt + i];
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt) b[u][KT + kt] = pt.X2[row * pt.ldX2 + 32 * kt + i];
        }
    }
};

template <int NTW, int KT, int KT2, int WG_U>
__device__ __forceinline__ void wgrad_accumulate(const WgradOperands<NTW, KT, KT2, WG_U>& op, f32x16 (&acc)[NTW][KT + KT2],
                                                 float (&bsum)[NTW]) {
#pragma unroll
    for (int u = 0; u < WG_U; ++u)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            bsum[nt] += op.a[u][nt];
#pragma unroll
            for (int kt = 0; kt < KT + KT2; ++kt)
                acc[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(op.a[u][nt], op.b[u][kt], acc[nt][kt], 0, 0, 0);
        }
}

template <int NTW, int KT, int KT2, int WG_U>
__device__ __forceinline__ void wgrad_part(const WgradPart& pt, long long p_begin, long long p_end, int wave, int lane) {
    const int i = lane & 31, s = lane >> 5;
    const int n0 = wave * NTW * 32;
    f32x16 acc[NTW][KT + KT2];
    float bsum[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bsum[nt] = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT + KT2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][kt][r] = 0.0f;
    }
    constexpr int STEP = 2 * WG_U;  // points per stage; slabs are multiples of 2*STEP
    WgradOperands<NTW, KT, KT2, WG_U> opA, opB;
    opA.load(pt, p_begin, n0, i, s);
    for (long long p = p_begin; p < p_end; p += 2 * STEP) {
        opB.load(pt, p + STEP, n0, i, s);  // rows stay inside the slab: p + STEP < p_end
        __builtin_amdgcn_sched_barrier(0);
        wgrad_accumulate(opA, acc, bsum);
        __builtin_amdgcn_sched_barrier(0);
        if (p + 2 * STEP < p_end) opA.load(pt, p + 2 * STEP, n0, i, s);
        __builtin_amdgcn_sched_barrier(0);
        wgrad_accumulate(opB, acc, bsum);
        __builtin_amdgcn_sched_barrier(0);
    }
    // dW[n0 + 32nt + rowmap(r,s)][col + i]: per register two 128-byte row segments
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int kt = 0; kt < KT + KT2; ++kt) {
            const int k = (kt < KT ? 32 * kt : 32 * (kt - KT)) + i;
            const int col = (kt < KT ? pt.col_off : pt.col_off2) + k;
            if (k < (kt < KT ? pt.k_valid : pt.k_valid2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    atomicAdd(pt.dW + (size_t)(n0 + 32 * nt + rowmap(r, s)) * pt.ldW + col, acc[nt][kt][r]);
            }
        }
    if (pt.dbias) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float b = bsum[nt] + __shfl_xor(bsum[nt], 32, 64);
            if (s == 0) atomicAdd(pt.dbias + n0 + 32 * nt + i, b);
        }
    }
}

// shape ids: 0 = <2,8,0>  1 = <2,8,2>  2 = <1,8,1>  3 = <2,2,0>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParts parts) {
    int pi = 0;
#pragma unroll 1
    while (pi + 1 < parts.n && (int)blockIdx.x >= parts.part[pi + 1].first_block) ++pi;
    const WgradPart& pt = parts.part[pi];
    const int slab = blockIdx.x - pt.first_block;
    // slab boundaries in units of 32 points
    const long long tiles = parts.P32 / 32;
    const long long t0 = tiles * slab / pt.slabs, t1 = tiles * (slab + 1) / pt.slabs;
    if (t0 >= t1) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    switch (pt.shape) {
        case 0: wgrad_part<2, 8, 0, 4>(pt, t0 * 32, t1 * 32, wave, lane); break;
        case 1: wgrad_part<2, 8, 2, 2>(pt, t0 * 32, t1 * 32, wave, lane); break;
        case 2: wgrad_part<1, 8, 1, 4>(pt, t0 * 32, t1 * 32, wave, lane); break;
        default: wgrad_part<2, 2, 0, 4>(pt, t0 * 32, t1 * 32, wave, lane); break;
    }
}

// MFMAs per k-step per wave of each shape (the cost model of the slab split).
static const int kShapeCost[4] = {16, 20, 9, 4};

int launch_wgrad(WgradParts& parts, long long P32, hipStream_t stream) {
    // A workgroup takes a whole CU (512 registers per lane), so the grid is sized
    // to exactly two equal-cost rounds of the 256 CUs: slabs proportional to cost.
    long long total_cost = 0;
    for (int i = 0; i < parts.n; ++i) total_cost += kShapeCost[parts.part[i].shape];
    const long long tiles = P32 / 32;
    const int target_blocks = 512;
    int used = 0;
    for (int i = 0; i < parts.n; ++i) {
        long long s = target_blocks * (long long)kShapeCost[parts.part[i].shape] / total_cost;
        if (s < 1) s = 1;
        if (s > tiles) s = tiles;
        parts.part[i].slabs = (int)s;
        used += (int)s;
    }
    // hand the remaining blocks to the parts with the most work per slab
    while (used < target_blocks) {
        int best = -1;
        double best_load = 0.0;
        for (int i = 0; i < parts.n; ++i) {
            const double load = (double)kShapeCost[parts.part[i].shape] / parts.part[i].slabs;
            if (parts.part[i].slabs < tiles && load > best_load) { best = i; best_load = load; }
        }
        if (best < 0) break;
        ++parts.part[best].slabs;
        ++used;
    }
    int first = 0;
    for (int i = 0; i < parts.n; ++i) {
        parts.part[i].first_block = first;
        first += parts.part[i].slabs;
    }
    parts.total_blocks = first;
    parts.P32 = P32;
    hipLaunchKernelGGL(wgrad_kernel, dim3(first), dim3(256), 0, stream, parts);
    return check_launch("wgrad");
}

// -------------------------------------------------------- small heads ----
// Thread k of a block accumulates dW[c][k] over the block's point range.
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(SkinnyPart pt, long long P32) {
    const int k = threadIdx.x;
    const long long per = (P32 + gridDim.x - 1) / gridDim.x;
    const long long p0 = per * blockIdx.x, p1 = min(P32, p0 + per);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long p = p0; p < p1; ++p) {
        const float x = k < pt.K ? pt.X[p * pt.ldX + k] : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < pt.C) {
                const float g = pt.dY[p * pt.ldY + pt.c0 + c];
                acc[c] = fmaf(g, x, acc[c]);
                bs[c] += g;
            }
        }
    }
    if (k < pt.K) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < pt.C) atomicAdd(pt.dW + c * pt.K + k, acc[c]);
    }
    if (k == 0 && pt.dbias) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < pt.C) atomicAdd(pt.dbias + c, bs[c]);
    }
}

int launch_skinny(const SkinnyPart& pt, long long P32, hipStream_t stream) {
    const int blocks = (int)((P32 / 32 < 1024) ? (P32 / 32 > 0 ? P32 / 32 : 1) : 1024);
    hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(blocks), dim3(256), 0, stream, pt, P32);
    return check_launch("skinny_wgrad");
}

}  // namespace nerf
