// Kernels shared by the fused MLPs: weight packing, the weight-gradient GEMMs
// (contraction over points on the exact-f32 MFMA) and the small-head gradients.
#include "mlp_common.h"

namespace nerf {

// ------------------------------------------------------------- packing ----
__global__ void pack_kernel(PackJobs jobs, float* __restrict__ packed) {
    const PackJob& jb = jobs.job[blockIdx.y];
    const int total = jb.mode >= 2 ? jb.count : jb.KT * 16 * 64 * jb.OT;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (jb.mode == 2) {
            v = jb.src[idx];
        } else if (jb.mode == 3) {
            v = 0.0f;
        } else {
            const int ot = idx % jb.OT;
            const int lane = (idx / jb.OT) % 64;
            const int r = (idx / (jb.OT * 64)) % 16;
            const int kt = idx / (jb.OT * 64 * 16);
            int out_i = 32 * ot + (lane & 31);                 // row of the packed operand (MFMA row i)
            int k = 32 * kt + rowmap(r, lane >> 5);            // contraction index
            if (jb.hash_perm) {   // tile position v = rowmap(r,h) <-> feature 16h + r
                int& v32 = jb.mode == 0 ? k : out_i;
                v32 = 16 * ((v32 >> 2) & 1) + (v32 & 3) + 4 * (v32 >> 3);
            }
            if (jb.mode == 0) {  // forward: operand[n][k_eff] = W[n][col(k_eff)]
                int col = -1;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (k >= jb.seg_eff[s] && k < jb.seg_eff[s] + jb.seg_len[s]) col = jb.seg_src[s] + (k - jb.seg_eff[s]);
                if (out_i < jb.N && col >= 0) {
                    v = jb.src[(size_t)out_i * jb.ld + col];
                    if (jb.add_src && k >= jb.seg_eff[0] && k < jb.seg_eff[0] + jb.seg_len[0])
                        v += jb.add_src[(size_t)out_i * jb.add_ld + col];
                }
            } else {             // backward: operand[m][n] = W[n][col_begin + (m - m_first)]
                const int m = out_i - jb.m_first;
                if (k < jb.N && m >= 0 && m < jb.col_count) {
                    v = jb.src[(size_t)k * jb.ld + jb.col_begin + m];
                    if (jb.add_src) v += jb.add_src[(size_t)k * jb.add_ld + jb.col_begin + m];
                }
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
// NTW*(KT+KT2) accumulator tiles (up to 320 registers).
//
// The slab is streamed through LDS in chunks of 32 points: the dY, X and X2 rows
// of a chunk are copied global -> LDS by direct-to-LDS loads (16 B per lane, one
// whole 1 KiB row segment per wave instruction), two chunk buffers deep, so the
// loads of chunk c+1 are in flight during all 16 k-steps of chunk c (about 8 us
// of cover for an HBM miss under load).  MFMA operands are then conflict-free
// ds_read_b32 (32 consecutive floats per lane half), fetched one k-step ahead.
constexpr int WG_CH = 32;                          // points per chunk
constexpr int WG_LDS_FLOATS = 2 * WG_CH * (256 + 256);  // two buffers of the largest shape (128 KiB)

// NW = waves along the output dimension (N = NW*NTW*32 columns of dY); the other
// 4/NW wave groups split the k-steps of every chunk between them.
template <int NTW, int KT, int KT2, int NW = 4>
struct WgradShape {
    static constexpr int N = NW * NTW * 32, XW = KT * 32, X2W = KT2 * 32, KTT = KT + KT2, G = 4 / NW;
    static constexpr int DY_OFF = 0, X_OFF = WG_CH * N, X2_OFF = WG_CH * (N + XW);
    static constexpr int CHUNK = WG_CH * (N + XW + X2W);   // floats per buffer
    static_assert(CHUNK % 256 == 0, "a chunk must be a whole number of 1 KiB pieces");
    static_assert(2 * CHUNK <= WG_LDS_FLOATS, "chunk buffers exceed the LDS allocation");
};

// The source pointers of one part, copied out of the kernel-argument table once
// (a direct-to-LDS load is an opaque memory write to the compiler: anything still
// read through the table would be re-fetched, and waited for, after every piece).
struct WgradSrc {
    const float *dY, *X, *X2;
    long long ldY, ldX, ldX2;
};

// Enqueue the direct-to-LDS copies of the chunk that starts at point p0.  Piece
// `ins` is 1 KiB of the chunk buffer; the sub-block boundaries are multiples of
// 1 KiB, so which source a piece reads is uniform over the wave.
template <int NTW, int KT, int KT2, int NW>
__device__ __forceinline__ void wgrad_stage(const WgradSrc& src, long long p0, float* buf, int wave, int lane) {
    using S = WgradShape<NTW, KT, KT2, NW>;
    static_assert(S::X_OFF % 256 == 0 && S::X2_OFF % 256 == 0, "sub-blocks must start on a 1 KiB piece");
#pragma unroll
    for (int piece = 0; piece < S::CHUNK / 256; piece += 4) {
        const int ins = piece + wave;                 // this wave's piece (wave is uniform)
        if (ins < S::CHUNK / 256) {
            const float* g;
            if (ins * 256 < S::X_OFF) {
                const int off = ins * 256 + lane * 4;
                g = src.dY + (p0 + off / S::N) * src.ldY + off % S::N;
            } else if (KT2 == 0 || ins * 256 < S::X2_OFF) {
                const int o = ins * 256 - S::X_OFF + lane * 4;
                g = src.X + (p0 + o / S::XW) * src.ldX + o % S::XW;
            } else {
                const int o = ins * 256 - S::X2_OFF + lane * 4;
                g = src.X2 + (p0 + o / (S::X2W ? S::X2W : 1)) * src.ldX2 + o % (S::X2W ? S::X2W : 1);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(buf + ins * 256), 16, 0, 0);
        }
    }
}

template <int NTW, int KT, int KT2, int NW>
__device__ __forceinline__ void wgrad_fetch(const float* buf, int u, int n0, int i, int s, float (&a)[NTW],
                                            float (&b)[KT + KT2]) {
    using S = WgradShape<NTW, KT, KT2, NW>;
    const int row = 2 * u + s;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) a[nt] = buf[S::DY_OFF + row * S::N + n0 + 32 * nt + i];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) b[kt] = buf[S::X_OFF + row * S::XW + 32 * kt + i];
#pragma unroll
    for (int kt = 0; kt < KT2; ++kt) b[KT + kt] = buf[S::X2_OFF + row * S::X2W + 32 * kt + i];
}

template <int NTW, int KTT>
__device__ __forceinline__ void wgrad_mfma(const float (&a)[NTW], const float (&b)[KTT], f32x16 (&acc)[NTW][KTT],
                                           float (&bsum)[NTW]) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bsum[nt] += a[nt];
#pragma unroll
        for (int kt = 0; kt < KTT; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nt], b[kt], acc[nt][kt], 0, 0, 0);
    }
}

template <int NTW, int KT, int KT2, int NW = 4>
__device__ __forceinline__ void wgrad_part(const WgradPart& pt, long long p_begin, long long p_end, int wave, int lane,
                                           float* lds) {
    using S = WgradShape<NTW, KT, KT2, NW>;
    constexpr int KTT = KT + KT2;
    constexpr int G = S::G;                       // wave groups sharing the k-steps
    constexpr int STEPS = WG_CH / 2 / G;          // k-steps per wave per chunk
    static_assert(STEPS % 2 == 0, "the operand ping-pong needs an even step count");
    const int i = lane & 31, s = lane >> 5;
    const int n0 = (wave % NW) * NTW * 32;
    const int g = wave / NW;
    f32x16 acc[NTW][KTT];
    float bsum[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bsum[nt] = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KTT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][kt][r] = 0.0f;
    }
    const WgradSrc src{pt.dY, pt.X, pt.X2, pt.ldY, pt.ldX, pt.ldX2};
    wgrad_stage<NTW, KT, KT2, NW>(src, p_begin, lds, wave, lane);
    int cur = 0;
#pragma unroll 1
    for (long long p = p_begin; p < p_end; p += WG_CH) {
        // Every wave's copies of this chunk have landed (the barrier's fence drains the
        // wave's own direct-to-LDS loads first) and every wave is done with the other buffer.
        __syncthreads();
        if (p + WG_CH < p_end) wgrad_stage<NTW, KT, KT2, NW>(src, p + WG_CH, lds + (cur ^ 1) * S::CHUNK, wave, lane);
        const float* buf = lds + cur * S::CHUNK;
        float a0[NTW], b0[KTT], a1[NTW], b1[KTT];
        wgrad_fetch<NTW, KT, KT2, NW>(buf, g, n0, i, s, a0, b0);
#pragma unroll
        for (int j = 0; j < STEPS; j += 2) {   // this wave's k-steps are g, g+G, g+2G, ...
            wgrad_fetch<NTW, KT, KT2, NW>(buf, g + G * (j + 1), n0, i, s, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            wgrad_mfma<NTW, KTT>(a0, b0, acc, bsum);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < STEPS) wgrad_fetch<NTW, KT, KT2, NW>(buf, g + G * (j + 2), n0, i, s, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            wgrad_mfma<NTW, KTT>(a1, b1, acc, bsum);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    }
    // dW[n0 + 32nt + rowmap(r,s)][col + i]: per register two 128-byte row segments
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int kt = 0; kt < KTT; ++kt) {
            const int k = (kt < KT ? 32 * kt : 32 * (kt - KT)) + i;
            const int col = (kt < KT ? pt.col_off : pt.col_off2) + k;
            if (k < (kt < KT ? pt.k_valid : pt.k_valid2) && (kt >= KT || k >= pt.k_first)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + 32 * nt + rowmap(r, s);
                    if (n < pt.n_valid) atomicAdd(pt.dW + (size_t)n * pt.ldW + col, acc[nt][kt][r]);
                }
            }
        }
    if (pt.dbias) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float b = bsum[nt] + __shfl_xor(bsum[nt], 32, 64);
            if (s == 0 && n0 + 32 * nt + i < pt.n_valid) atomicAdd(pt.dbias + n0 + 32 * nt + i, b);
        }
    }
}

// shape ids (NTW,KT,KT2,NW): 0 = <2,8,0,4>  1 = <1,8,1,4>  2 = <2,2,0,4>   (a 320-register <2,8,2> tile set does
// not allocate without spills);  small layers of the hash-grid model: 3 = <1,1,0,2>  4 = <1,2,0,1>  5 = <1,1,1,2>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParts parts) {
    __shared__ __attribute__((aligned(16))) float lds[WG_LDS_FLOATS];
    int pi = 0;
#pragma unroll 1
    while (pi + 1 < parts.n && (int)blockIdx.x >= parts.part[pi + 1].first_block) ++pi;
    const WgradPart& pt = parts.part[pi];
    const int slab = blockIdx.x - pt.first_block;
    // slab boundaries in units of 32 points
    const long long tiles = parts.P32 / 32;
    const long long t0 = tiles * slab / pt.slabs, t1 = tiles * (slab + 1) / pt.slabs;
    if (t0 >= t1) return;   // uniform for the whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    switch (pt.shape) {
        case 0: wgrad_part<2, 8, 0>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;
        case 1: wgrad_part<1, 8, 1>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;
        case 2: wgrad_part<2, 2, 0>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;
        case 3: wgrad_part<1, 1, 0, 2>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;
        case 4: wgrad_part<1, 2, 0, 1>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;
        default: wgrad_part<1, 1, 1, 2>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;
    }
}

// MFMAs per k-step per wave of each shape (the cost model of the slab split).
static const int kShapeCost[6] = {16, 9, 4, 4, 4, 4};

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
