// Fused hash-grid NeRF (FastNGPNeRF.forward, src/models/FastNGPNeRF.py:79-110)
// for gfx950: 16-level hash encoding + density MLP (32->64->16) + colour MLP
// (15+27 -> 64 -> 3 with the bias-free skip projection) in one kernel, the
// 32-float encoding never leaves registers.
//
// The encoding is gathered first, level-major, by a launch that pins every level to one XCD (encode.hip:
// a 4 MiB table then lives in that XCD's 4 MiB L2; gathered inside this kernel, where every wave walks all
// 16 tables, the same loads took 2.7 times as long).  Then one wavefront owns 32 points.  Lane (p, h) reads
// levels 8h .. 8h+7 of point p, which fills
// the 16 registers of the first MFMA's B tile directly: register r of lane half
// h is feature 16h + r, and the packed first-layer weight is permuted to match.
// The backward kernel ends in the same layout: each lane writes the gradient of exactly
// the levels it read, level-major again, for the table-gradient scatter (encode.hip).
// Compiled with -ffp-contract=off: the interpolation rounds like the reference.
#include "hashgrid.h"
#include "mlp_common.h"

namespace nerf {

namespace np {  // float offsets into the packed buffer
constexpr int F_D1 = 0;                     // 32 -> 64   (KT 1, OT 2)
constexpr int F_D2 = F_D1 + 1 * 16 * 64 * 2;  // 64 -> 16(32)  (KT 2, OT 1)
constexpr int F_C1 = F_D2 + 2 * 16 * 64 * 1;  // [raw(32) | pe(dir)(32)] -> 64  (KT 2, OT 2), skip projection folded in
constexpr int B_C1 = F_C1 + 2 * 16 * 64 * 2;  // 64 -> raw(32)   (KT 2, OT 1)
constexpr int B_D2 = B_C1 + 2 * 16 * 64 * 1;  // raw(32) -> 64   (KT 1, OT 2)
constexpr int B_D1 = B_D2 + 1 * 16 * 64 * 2;  // 64 -> 32 features (KT 2, OT 1), rows permuted like the encoding
constexpr int S_BD1 = B_D1 + 2 * 16 * 64 * 1;
constexpr int S_BD2 = S_BD1 + 64;           // 16 values + 16 zeros
constexpr int S_BC1 = S_BD2 + 32;
constexpr int S_WC2 = S_BC1 + 64;           // 3 x 64
constexpr int S_BC2 = S_WC2 + 192;
constexpr int TOTAL = S_BC2 + 4;
}  // namespace np

namespace nw {  // workspace layout, floats per point
constexpr int E = 0;        // [P32][32] encoding, natural feature order (level-major)
constexpr int HD = 32;      // [P32][64]
constexpr int RAW = 96;     // [P32][32]  (column 0 sigma_raw, 1..15 features, rest zero)
constexpr int DE = 128;     // [P32][32]
constexpr int HC = 160;     // [P32][64]
constexpr int MASK = 224;   // [P32/32][2][64] u32
constexpr int PER_POINT = 228;
// backward workspace
constexpr int DOUT4 = 0;    // [P32][4] = (d sigma_pre, d rgb_pre[3]), for the colour head's gradient
constexpr int DENC = 4;     // [16][P32][2] gradient of the encoding, level-major like the encoding itself
constexpr int PART = 36;    // the backward kernel's partial weight-gradient sums: one slot of 8,512 floats per workgroup
constexpr int PER_POINT2 = PART + 68;   // (a workgroup per 128 points at most: 66.5 floats per point)
}  // namespace nw


struct NgpIdx {
    long long tile, gp, lp, P32;
    int lane, h;
    bool valid;
};

// 32 bits: 1 where the post-ReLU activation of a 64-wide layer is > 0 (this lane's 32 neurons).
__device__ __forceinline__ unsigned relu_bits2(const f32x16 (&a)[2]) {
    unsigned m = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m |= min(__float_as_uint(a[t][r]), 1u) << (16 * t + r);
    return m;
}
__device__ __forceinline__ void apply_bits2(f32x16 (&a)[2], unsigned m) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int keep = ((int)(m << (31 - (16 * t + r)))) >> 31;
            a[t][r] = __uint_as_float(__float_as_uint(a[t][r]) & (unsigned)keep);
        }
}

// store 16 consecutive floats (this lane half's part of a 32-wide row)
__device__ __forceinline__ void store16(float* __restrict__ dst, const f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// A workgroup walks tiles of 32 points per wave (grid: up to three workgroups per CU) with the forward weights and
// biases in LDS for all of them: read from L2 by every wave for every tile (32 KB per 32 points) they were most of
// the kernel's memory traffic and their latency sat between the MFMAs.
template <bool SAVE>
__global__ __launch_bounds__(256, 3) void ngp_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ dirs,
                                                      int dirs_div, long long P,
                                                      const float* __restrict__ pk_arg, float* __restrict__ rgb,
                                                      float* __restrict__ sigma, float* __restrict__ ws,
                                                      const int* __restrict__ count) {
    // a wave's 32 x 64 (or 32 x 32) block of a saved matrix is contiguous in memory: its register tiles go through a
    // per-wave LDS transposition buffer and leave as whole rows, 16 bytes per lane (stored straight from the register
    // layout an instruction writes 64 pieces of 16 bytes, 256 bytes apart)
    __shared__ __attribute__((aligned(16))) float s_tile[4][TILE_LDS_FLOATS];
    constexpr int W_FWD = np::B_C1 - np::F_D1, W_CONST = np::TOTAL - np::S_BD1;   // the three layers; biases and colour head
    __shared__ __attribute__((aligned(16))) float s_w[W_FWD + W_CONST];
    static_assert(W_FWD % 4 == 0 && W_CONST % 4 == 0 && np::S_BD1 % 4 == 0, "16-byte copies");
    for (int k = threadIdx.x * 4; k < W_FWD; k += 256 * 4)
        *reinterpret_cast<float4*>(s_w + k) = *reinterpret_cast<const float4*>(pk_arg + np::F_D1 + k);
    for (int k = threadIdx.x * 4; k < W_CONST; k += 256 * 4)
        *reinterpret_cast<float4*>(s_w + W_FWD + k) = *reinterpret_cast<const float4*>(pk_arg + np::S_BD1 + k);
    __syncthreads();
    const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // P sizes the arrays (their strides are P32); with a count in device memory only the first `used` points are walked
    const long long P32 = (P + 31) / 32 * 32, used = points_in_use(P, count), tiles = (used + 31) / 32;
#pragma unroll 1
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < tiles; tile += (long long)gridDim.x * 4) {
        // the lane index is opaque per tile: otherwise every lane's weight offsets (and the weights) are formed in front of
        // the loop, hundreds of registers of them
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5;
        NgpIdx ix;
        ix.lane = lane; ix.tile = tile; ix.h = h; ix.gp = tile * 32 + (lane & 31); ix.valid = ix.gp < used;
        ix.lp = ix.valid ? ix.gp : used - 1; ix.P32 = P32;
        auto pk_at = [&](int packed_off) { return packed_off >= np::S_BD1 ? s_w + W_FWD + (packed_off - np::S_BD1) : s_w + (packed_off - np::F_D1); };
        const TileBuf tb = tile_buf(s_tile[wave], lane);
        // ---- hash encoding: levels 8h .. 8h+7 of this lane's point, gathered level by level by the launch before
        // this one (enc: [16][P32][2]); 32 consecutive lanes read 256 contiguous bytes per level
        f32x16 e[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 v = *reinterpret_cast<const float2*>(enc + 2 * ((8 * h + j) * ix.P32 + ix.lp));
            e[0][2 * j] = v.x;
            e[0][2 * j + 1] = v.y;
        }
        if (SAVE) {
            // the saved encoding leaves as whole rows too: lane (p, h) holds features 16h .. 16h+15 of point p, which go into
            // row p of the wave's transposition buffer, and the buffer goes out 128 contiguous bytes per 8 lanes
            const unsigned wr_e = lds_addr(s_tile[wave] + (lane & 31) * TILE_LD + 16 * h);
            static_for<0, 4>([&](auto cq) {
                constexpr int q = decltype(cq)::value;
                lds_write4<16 * q>(wr_e, f32x4{e[0][4 * q], e[0][4 * q + 1], e[0][4 * q + 2], e[0][4 * q + 3]});
            });
            f32x4 v[4];
            static_for<0, 4>([&](auto ck) { v[ck] = lds_read4<8 * decltype(ck)::value * TILE_LD * 4>(tb.rd); });
            lgkm_wait();
            float* rows0 = ws + ix.P32 * nw::E + ix.tile * (32 * 32);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("" : "+v"(v[k]));
                __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(rows0 + (size_t)(8 * k + tb.row) * 32 + tb.col));
            }
        }
        // ---- density MLP: raw = W2 relu(W1 e + b1) + b2        (FastNGPNeRF.py:90)
        f32x16 hd[2];
        fill_rows<2>(hd, pk_at(np::S_BD1), h);
        dense<1, 2>(pk_at(np::F_D1), e, hd, lane);
        relu_tiles<2>(hd);
        // every saved matrix leaves right after it is formed, under the layers that follow
        if (SAVE) save_tiles<2>(tb, ws + ix.P32 * nw::HD + ix.tile * (32 * 64), 64, hd);
        f32x16 cin[2];   // [raw | pe(dir)]
        {
            f32x16 raw[1];
            fill_rows<1>(raw, pk_at(np::S_BD2), h);
            dense<2, 1>(pk_at(np::F_D2), hd, raw, lane);
            cin[0] = raw[0];
            if (SAVE) save_tiles<1>(tb, ws + ix.P32 * nw::RAW + ix.tile * (32 * 32), 32, raw);
        }
        if (ix.valid && h == 0) sigma[ix.gp] = softplus_f(cin[0][0] + 1.5f);   // row 0 lives in lane half 0, register 0  (:91-92)
        // ---- view direction: normalise, encode with 4 bands (:98-99); zeros when no directions are given (:100-103)
        {
            f32x16 de[1];
            if (dirs) {
                const long long dr = ix.lp / dirs_div;
                float d0 = dirs[3 * dr], d1 = dirs[3 * dr + 1], d2 = dirs[3 * dr + 2];
                const float n = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
                d0 /= n; d1 /= n; d2 /= n;
                pe_tiles<1, 4>(de, d0, d1, d2, h);
            } else {
                zero_tiles<1>(de);
            }
            cin[1] = de[0];
            if (SAVE) save_tiles<1>(tb, ws + ix.P32 * nw::DE + ix.tile * (32 * 32), 32, de);
        }
        // ---- colour MLP: sigmoid(W4 relu(W3 [feat | pe] + Wskip feat + b3) + b4)   (:105-108)
        f32x16 hc[2];
        fill_rows<2>(hc, pk_at(np::S_BC1), h);
        dense<2, 2>(pk_at(np::F_C1), cin, hc, lane);
        relu_tiles<2>(hc);
        if (SAVE) {
            save_tiles<2>(tb, ws + ix.P32 * nw::HC + ix.tile * (32 * 64), 64, hc);
            unsigned* mask = reinterpret_cast<unsigned*>(ws + ix.P32 * nw::MASK) + ix.tile * 128 + lane;
            mask[0] = relu_bits2(hd);
            mask[64] = relu_bits2(hc);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = dot_rows<2>(hc, pk_at(np::S_WC2 + 64 * c), h) + pk_at(np::S_BC2)[c];
            if (ix.valid && h == 0) rgb[3 * ix.gp + c] = sigmoid_f(v);
        }
        // the wave's accesses to its transposition buffer are in program order; nothing else is shared between tiles
    }
}

// ---------------------------------------------------------------------------
// Backward: the backward-data chain (d rgb, d sigma -> d encoding) with the weight gradients formed in the same
// kernel.  As separate launches the chain wrote d(pre-activation) of its three layers to memory (0.8 KB per point)
// for a work list of GEMMs to read back together with the saved inputs: 2.8 KB per point of traffic for 7,059
// parameters (1.12 + 1.07 ms per 4.2 M points; this kernel: 1.16).  Here a wave keeps all of its weight-gradient tiles in registers (8 tiles = 128 accumulators) across the tiles
// of points it walks, adds them to the gradients once at the end, and nothing but the gradient of the encoding
// (for the table scatter) and d(sigma_pre, rgb_pre) (for the colour head's 3 x 64 gradient) is written.
//   dW[n][k] = sum_p dY[p][n] X[p][k] on v_mfma_f32_32x32x2_f32: A = dY^T (row = neuron, contraction = point),
//   B = X (contraction = point, column = input feature).  MFMA step j takes point 16 s + j from lane half s (any
//   order of the contraction will do as long as A and B agree).
//   B comes straight from the row-major matrices the forward kernel saved: lane (i, s) loads X[16 s + j][i],
//   j = 0..15 -- 32 consecutive floats per lane half and instruction -- requested a whole chain stage before
//   they are used.  A is the chain's own register tile (lane = point) turned through a per-wave LDS buffer
//   (16 ds_write_b32 as [neuron][point], 4 ds_read_b128 per tile).
// Biases: column sums of dY, summed from the A operands.
// ---------------------------------------------------------------------------
struct NgpGrads {
    float *w_d1, *b_d1, *w_d2, *b_d2, *w_c1, *b_c1, *w_skip;   // fc_density.0/.2, fc_color.0, skip_proj (see nerf_mlp_ngp_bwd)
};
constexpr int NGP_PART_FLOATS = 8 * 1024 + 5 * 64;   // a workgroup's weight-gradient sums (see ngp_bwd_fused_kernel)
constexpr int AT_PITCH = 36;                       // floats per [neuron] row of the A staging buffer (32 points + pad)
constexpr int AT_FLOATS = 2 * 32 * AT_PITCH;       // two tiles per wave

// The wave's earlier LDS accesses are complete and visible to all of its lanes (a wave is its own only partner here).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);   // and the scheduler does not blend the stages (their live ranges add up to spills)
}
// register tiles (lane = point p, register r = neuron rowmap(r, h)) -> s_a[t][neuron][point]
template <int NT>
__device__ __forceinline__ void stage_a(float* s_a, const f32x16 (&a)[NT], int p, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_a[(32 * t + rowmap(r, h)) * AT_PITCH + p] = a[t][r];
}
// lane (i, s): X[point 16 s + j][col0 + i], j = 0..15, of the 32-point tile whose first row is rows0 (row length ld)
__device__ __forceinline__ void fetch_b(const float* __restrict__ rows0, int ld, int col0, float (&o)[16], int i, int s) {
    const float* base = rows0 + (size_t)(16 * s) * ld + i;   // one address per lane; (j, col0) are immediate offsets
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = __builtin_nontemporal_load(base + j * ld + col0);
}
// acc[nt][kt] += dY_tile(nt)^T X_tile(kt) over the 32 points; A operands one tile at a time from the staging buffer:
// lane (i, s) reads dY[point 16 s + j][neuron 32 nt + i], j = 0..15
template <int NT, int KT>
__device__ __forceinline__ void wgrad_tiles(const float* s_a, const float (&b)[KT][16], f32x16 (&acc)[NT][KT], float (&bsum)[NT],
                                            int i, int s) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float a[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(s_a + (32 * nt + i) * AT_PITCH + 16 * s + 4 * q);
            a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            bsum[nt] += a[j];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[kt][j], acc[nt][kt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// dW[32 nt + rowmap(r, s)][col_off + k] += acc[nt][r] for k = k0 + i in [k_first, k_valid), rows below n_valid
__device__ __forceinline__ void add_tile(float* __restrict__ dW, int ldW, int col_off, int nt, int k0, int k_first, int k_valid,
                                         int n_valid, const f32x16& acc, int i, int s) {
    const int k = k0 + i;
    if (k < k_first || k >= k_valid) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 32 * nt + rowmap(r, s);
        if (n < n_valid) atomicAdd(dW + (size_t)n * ldW + col_off + k, acc[r]);
    }
}

__global__ __launch_bounds__(256, 2) void ngp_bwd_fused_kernel(const float* __restrict__ d_rgb, const float* __restrict__ d_sigma,
                                                               long long P, const float* __restrict__ pk_arg,
                                                               const float* __restrict__ rgb, const float* __restrict__ sigma,
                                                               const float* __restrict__ ws, float* __restrict__ ws2, NgpGrads g,
                                                               float* __restrict__ partials, const int* __restrict__ count,
                                                               unsigned* __restrict__ clear, int clear_words) {
    // the queue counters of the table-gradient scatter that follows on the stream start from zero: cleared here, a
    // launch of 5 us saved per step
    for (int k = blockIdx.x * 256 + threadIdx.x; k < clear_words; k += gridDim.x * 256) clear[k] = 0u;
    __shared__ __attribute__((aligned(16))) float s_at[4][AT_FLOATS];
    // The backward weights (three transposed layers and the colour head's 3 x 64) stay in LDS for all the tiles the
    // workgroup walks: read from L2 per tile their latency showed between the chain's MFMAs at two waves per SIMD.
    constexpr int W_CHAIN = np::S_BD1 - np::B_C1, W_ALL = W_CHAIN + 192;
    __shared__ __attribute__((aligned(16))) float s_w[W_ALL];
    for (int k = threadIdx.x * 4; k < W_CHAIN; k += 256 * 4)
        *reinterpret_cast<float4*>(s_w + k) = *reinterpret_cast<const float4*>(pk_arg + np::B_C1 + k);
    if (threadIdx.x < 192) s_w[W_CHAIN + threadIdx.x] = pk_arg[np::S_WC2 + threadIdx.x];
    __syncthreads();
    const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long P32 = (P + 31) / 32 * 32, used = points_in_use(P, count), tiles = (used + 31) / 32;
    f32x16 acc_c1[2][2], acc_d2[1][2], acc_d1[2][1];
    float b_c1[2] = {0.f, 0.f}, b_d2[1] = {0.f}, b_d1[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc_c1[a][0][r] = acc_c1[a][1][r] = 0.0f;
            acc_d2[0][a][r] = 0.0f;
            acc_d1[a][0][r] = 0.0f;
        }
    const float* RAW = ws + P32 * nw::RAW;
    const float* DE = ws + P32 * nw::DE;
    const float* HD = ws + P32 * nw::HD;
    const float* E = ws + P32 * nw::E;
#pragma unroll 1
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < tiles; tile += (long long)gridDim.x * 4) {
        // The packed weights and every lane's offsets into them and into the saved matrices are the same for every tile:
        // left to itself the compiler forms all of them (some 200 registers per lane of weights, as many of 64-bit
        // addresses) in front of the loop and spills.  The lane index is made opaque per tile, so that addresses are
        // formed where they are used and the weights are read again per tile (from LDS, see above).
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5, p = lane & 31;
        const int i = p, s = h;   // the lane's role in the weight-gradient MFMAs: column i, point half s
        float* s_a = s_at[wave];
        const float* w_c1t = s_w, *w_d2t = s_w + (np::B_D2 - np::B_C1), *w_d1t = s_w + (np::B_D1 - np::B_C1), *w_c2 = s_w + W_CHAIN;
        const long long gp = tile * 32 + p;
        const bool valid = gp < used;
        const long long lp = valid ? gp : used - 1;
        const unsigned* mask = reinterpret_cast<const unsigned*>(ws + P32 * nw::MASK) + tile * 128 + lane;
        const unsigned m_hd = mask[0], m_hc = mask[64];
        float dpre[3], dps = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float o = rgb[3 * lp + c];
            dpre[c] = valid ? d_rgb[3 * lp + c] * o * (1.0f - o) : 0.0f;
        }
        if (valid) dps = d_sigma[lp] * (1.0f - expf(-sigma[lp]));
        if (h == 0) *reinterpret_cast<float4*>(ws2 + P32 * nw::DOUT4 + gp * 4) = make_float4(dps, dpre[0], dpre[1], dpre[2]);
        // Every stage: the chain's tile goes to the staging buffer, the inputs of that layer's weight gradient are
        // requested, the chain's next layer runs (its MFMAs cover the requests), then the weight-gradient MFMAs.
        // ---- d hc = W4^T d rgb_pre, masked; d raw through (W3 + Wskip)^T, density through softplus'
        f32x16 draw[1];
        float bx[2][16];
        {
            f32x16 dyc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float4 w = *reinterpret_cast<const float4*>(w_c2 + 64 * c + 32 * t + 8 * q + 4 * h);
                        sum.x = fmaf(w.x, dpre[c], sum.x);
                        sum.y = fmaf(w.y, dpre[c], sum.y);
                        sum.z = fmaf(w.z, dpre[c], sum.z);
                        sum.w = fmaf(w.w, dpre[c], sum.w);
                    }
                    dyc[t][4 * q + 0] = sum.x;
                    dyc[t][4 * q + 1] = sum.y;
                    dyc[t][4 * q + 2] = sum.z;
                    dyc[t][4 * q + 3] = sum.w;
                }
            apply_bits2(dyc, m_hc);
            stage_a<2>(s_a, dyc, p, h);
            fetch_b(RAW + tile * (32 * 32), 32, 0, bx[0], i, s);
            fetch_b(DE + tile * (32 * 32), 32, 0, bx[1], i, s);
            zero_tiles<1>(draw);
            dense<2, 1>(w_c1t, dyc, draw, lane);
            if (h == 0) draw[0][0] += dps;
        }
        wave_lds_fence();
        wgrad_tiles<2, 2>(s_a, bx, acc_c1, b_c1, i, s);        // fc_color.0 (and skip_proj): dYC x [raw | pe(dir)]
        wave_lds_fence();
        // ---- d hd = W2^T d raw, masked
        f32x16 dyd[2];
        {
            float bh[2][16];   // the hidden layer of the density MLP: input of fc_density.2's gradient
            stage_a<1>(s_a, draw, p, h);
            fetch_b(HD + tile * (32 * 64), 64, 0, bh[0], i, s);
            fetch_b(HD + tile * (32 * 64), 64, 32, bh[1], i, s);
            zero_tiles<2>(dyd);
            dense<1, 2>(w_d2t, draw, dyd, lane);
            apply_bits2(dyd, m_hd);
            wave_lds_fence();
            wgrad_tiles<1, 2>(s_a, bh, acc_d2, b_d2, i, s);    // fc_density.2: dRAW x hd
            wave_lds_fence();
        }
        // ---- d encoding = W1^T d hd
        {
            float be[1][16];   // the encoding: input of fc_density.0's gradient
            stage_a<2>(s_a, dyd, p, h);
            fetch_b(E + tile * (32 * 32), 32, 0, be[0], i, s);
            f32x16 de[1];
            zero_tiles<1>(de);
            dense<2, 1>(w_d1t, dyd, de, lane);   // register r = d feature (16h + r): this lane's own levels
            // the gradient of the encoding, level-major, for the table scatter: levels 8h .. 8h+7 of this lane's point
            float* denc = ws2 + P32 * nw::DENC;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<float2*>(denc + 2 * ((8 * h + j) * P32 + gp)) = make_float2(de[0][2 * j], de[0][2 * j + 1]);
            wave_lds_fence();
            wgrad_tiles<2, 1>(s_a, be, acc_d1, b_d1, i, s);    // fc_density.0: dYD x encoding
            wave_lds_fence();
        }
    }
    const int i = lane0 & 31, s = lane0 >> 5;
    if (partials) {
        // ---- the workgroup's sums go to its slot of `partials` ([tile 8][register 16][lane 64], then 5 x 64 bias sums);
        // ngp_wgrad_reduce_kernel adds the slots to the gradients.  (One atomic per accumulator and wave instead --
        // 2048 waves x 8192 atomics on 221 cache lines -- took longer than the kernel's arithmetic at 262,144 points.)
        // The four waves' tiles meet in LDS two at a time: 4 x 2 x 1024 floats fill the staging buffers.
        float* s_red = &s_at[0][0];
        float* slot = partials + (size_t)blockIdx.x * NGP_PART_FLOATS;
        auto round = [&](const f32x16& t0, const f32x16& t1, int first_tile) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s_red[(wave * 2 + 0) * 1024 + r * 64 + lane0] = t0[r];
                s_red[(wave * 2 + 1) * 1024 + r * 64 + lane0] = t1[r];
            }
            __syncthreads();
#pragma unroll
            for (int e = threadIdx.x; e < 2048; e += 256) {
                const int t = e >> 10, q = e & 1023;
                const float v = (s_red[(0 * 2 + t) * 1024 + q] + s_red[(1 * 2 + t) * 1024 + q]) +
                                (s_red[(2 * 2 + t) * 1024 + q] + s_red[(3 * 2 + t) * 1024 + q]);
                slot[(first_tile + t) * 1024 + q] = v;
            }
        };
        round(acc_c1[0][0], acc_c1[0][1], 0);
        round(acc_c1[1][0], acc_c1[1][1], 2);
        round(acc_d2[0][0], acc_d2[0][1], 4);
        round(acc_d1[0][0], acc_d1[1][0], 6);
        __syncthreads();
        const float bs[5] = {b_c1[0], b_c1[1], b_d2[0], b_d1[0], b_d1[1]};
#pragma unroll
        for (int k = 0; k < 5; ++k) s_red[(wave * 5 + k) * 64 + lane0] = bs[k];
        __syncthreads();
        for (int e = threadIdx.x; e < 320; e += 256)
            slot[8192 + e] = (s_red[e] + s_red[320 + e]) + (s_red[640 + e] + s_red[960 + e]);
        return;
    }
    // ---- no room for partial sums (a handful of tiles): one atomic per accumulator
    // fc_color.0 [64][42]: raw column k (1..15) is feature k - 1 -> column k - 1; pe(dir) column k (0..26) -> column 15 + k;
    // skip_proj [64][15] receives the feature columns as well (hc = W_c1 [feat | pe] + W_skip feat)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        add_tile(g.w_c1, 42, -1, nt, 0, 1, 16, 64, acc_c1[nt][0], i, s);
        add_tile(g.w_skip, 15, -1, nt, 0, 1, 16, 64, acc_c1[nt][0], i, s);
        add_tile(g.w_c1, 42, 15, nt, 0, 0, 27, 64, acc_c1[nt][1], i, s);
        add_tile(g.w_d1, 32, 0, nt, 0, 0, 32, 64, acc_d1[nt][0], i, s);
        const float bc = b_c1[nt] + __shfl_xor(b_c1[nt], 32, 64), bd = b_d1[nt] + __shfl_xor(b_d1[nt], 32, 64);
        if (s == 0) {
            atomicAdd(g.b_c1 + 32 * nt + i, bc);
            atomicAdd(g.b_d1 + 32 * nt + i, bd);
        }
    }
    add_tile(g.w_d2, 64, 0, 0, 0, 0, 32, 16, acc_d2[0][0], i, s);    // fc_density.2 [16][64]
    add_tile(g.w_d2, 64, 0, 0, 32, 32, 64, 16, acc_d2[0][1], i, s);
    {
        const float b2 = b_d2[0] + __shfl_xor(b_d2[0], 32, 64);
        if (s == 0 && i < 16) atomicAdd(g.b_d2 + i, b2);
    }
}

// grads += sum over the workgroups' slots.  Element e of a slot: tile e / 1024 (c1[0][raw], c1[0][pe], c1[1][raw], c1[1][pe],
// d2[hd 0..31], d2[hd 32..63], d1[0], d1[1]), register (e / 64) % 16, lane e % 64 = (column i, half s), i.e. row
// rowmap(register, s) of the tile; then the bias sums per lane.  One thread per element and 16th of the slots, plain
// additions in a fixed order: the same bits from run to run.
constexpr int RED_E = 64, RED_G = 16;   // (a row of 64 bias sums must be one block's elements)
__global__ __launch_bounds__(RED_E * RED_G) void ngp_wgrad_reduce_kernel(const float* __restrict__ partials, int slots, NgpGrads g) {
    __shared__ float s_sum[RED_G][RED_E];
    const int el = threadIdx.x % RED_E, grp = threadIdx.x / RED_E;
    const int e = blockIdx.x * RED_E + el;
    float sum = 0.0f;
    for (int k = grp; k < slots; k += RED_G) sum += partials[(size_t)k * NGP_PART_FLOATS + e];
    s_sum[grp][el] = sum;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int k = 1; k < RED_G; ++k) sum += s_sum[k][el];
        s_sum[0][el] = sum;
    }
    __syncthreads();
    if (grp != 0) return;
    if (e < 8192) {
        const int t = e >> 10, r = (e >> 6) & 15, l = e & 63, i = l & 31, row = rowmap(r, l >> 5);
        if (t < 4) {
            const int n = 32 * (t >> 1) + row;
            if ((t & 1) == 0) {
                if (i >= 1 && i < 16) {
                    g.w_c1[n * 42 + i - 1] += sum;
                    g.w_skip[n * 15 + i - 1] += sum;
                }
            } else if (i < 27) {
                g.w_c1[n * 42 + 15 + i] += sum;
            }
        } else if (t < 6) {
            if (row < 16) g.w_d2[row * 64 + 32 * (t - 4) + i] += sum;
        } else {
            g.w_d1[(32 * (t - 6) + row) * 32 + i] += sum;
        }
    } else if (el < 32) {   // a row of 64 bias sums = this block's 64 elements: neuron el, both point halves
        const int k = (e - 8192) >> 6;
        const float both = s_sum[0][el] + s_sum[0][el + 32];
        if (k == 0) g.b_c1[el] += both;
        else if (k == 1) g.b_c1[32 + el] += both;
        else if (k == 2) { if (el < 16) g.b_d2[el] += both; }
        else if (k == 3) g.b_d1[el] += both;
        else g.b_d1[32 + el] += both;
    }
}

}  // namespace nerf

using namespace nerf;

extern "C" size_t nerf_mlp_ngp_pack_floats(void) { return np::TOTAL; }
extern "C" size_t nerf_mlp_ngp_ws_floats_per_point(void) { return nw::PER_POINT; }
extern "C" size_t nerf_mlp_ngp_ws2_floats_per_point(void) { return nw::PER_POINT2; }

// Contraction index k of the first layer <-> natural feature: k = rowmap(r, h) holds feature 16h + r.
static PackJob with_hash_perm(PackJob j) {
    j.hash_perm = 1;
    return j;
}

extern "C" int nerf_mlp_ngp_pack(const float* const* params, float* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_ngp_pack: null pointer");
    for (int i = 0; i < NERF_NGP_NPARAM; ++i) NERF_REQUIRE(params[i], "nerf_mlp_ngp_pack: parameter %d is null", i);
    const float *Wd1 = params[0], *bd1 = params[1], *Wd2 = params[2], *bd2 = params[3], *Wc1 = params[4],
                *bc1 = params[5], *Wc2 = params[6], *bc2 = params[7], *Wsk = params[8];
    PackJobs J{};
    int n = 0;
    J.job[n++] = with_hash_perm(pack_fwd_job(Wd1, 32, 64, np::F_D1, 1, 2, 0, 32, 0));
    J.job[n++] = pack_fwd_job(Wd2, 64, 16, np::F_D2, 2, 1, 0, 64, 0);
    {   // colour layer: raw columns 1..15 are the features (skip projection added), column 32+j is pe(dir)[j]
        PackJob j = pack_fwd_job(Wc1, 42, 64, np::F_C1, 2, 2, 1, 15, 0, 32, 27, 15);
        j.add_src = Wsk;
        j.add_ld = 15;
        J.job[n++] = j;
    }
    {
        PackJob j = pack_bwd_job(Wc1, 42, 64, np::B_C1, 2, 1, 0, 15);
        j.m_first = 1;
        j.add_src = Wsk;
        j.add_ld = 15;
        J.job[n++] = j;
    }
    J.job[n++] = pack_bwd_job(Wd2, 64, 16, np::B_D2, 1, 2, 0, 64);
    J.job[n++] = with_hash_perm(pack_bwd_job(Wd1, 32, 64, np::B_D1, 2, 1, 0, 32));
    J.job[n++] = pack_copy_job(bd1, np::S_BD1, 64);
    J.job[n++] = pack_copy_job(bd2, np::S_BD2, 16);
    J.job[n++] = pack_zero_job(np::S_BD2 + 16, 16);
    J.job[n++] = pack_copy_job(bc1, np::S_BC1, 64);
    J.job[n++] = pack_copy_job(Wc2, np::S_WC2, 192);
    J.job[n++] = pack_copy_job(bc2, np::S_BC2, 3);
    J.n = n;
    return launch_pack(J, packed, (hipStream_t)stream);
}

static int check_levels(const int* res, const int* sizes, const char* who) {
    for (int l = 0; l < 16; ++l)
        if (res[l] < 1 || sizes[l] < 1) return fail(NERF_EINVAL, "%s: level %d has res=%d size=%d", who, l, res[l], sizes[l]);
    return NERF_OK;
}

static int ngp_fwd(const float* x, const float* dirs, int dirs_div, int64_t P, const float* const* tables, const int* res,
                   const int* sizes, const float* packed, float* rgb, float* sigma, float* ws, float* enc,
                   const int32_t* count, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && tables && res && sizes && packed && rgb && sigma && enc, "nerf_mlp_ngp_fwd: null pointer");
    NERF_REQUIRE(P >= 0 && dirs_div >= 1, "nerf_mlp_ngp_fwd: P=%lld dirs_div=%d", (long long)P, dirs_div);
    if (int rc = check_levels(res, sizes, "nerf_mlp_ngp_fwd")) return rc;
    const long long P32 = (P + 31) / 32 * 32;
    if (int rc = launch_hashgrid_gather(x, P, tables, res, sizes, 16, enc, true, P32, (hipStream_t)stream, count)) return rc;
    const long long wgs = ceil_div(P, 128);
    const dim3 grid((unsigned)(wgs < 768 ? wgs : 768)), block(256);   // three workgroups per CU, each walking its share of the tiles
    if (ws)
        hipLaunchKernelGGL(ngp_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, enc, dirs, dirs_div, (long long)P,
                           packed, rgb, sigma, ws, count);
    else
        hipLaunchKernelGGL(ngp_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, enc, dirs, dirs_div, (long long)P,
                           packed, rgb, sigma, ws, count);
    return check_launch("nerf_mlp_ngp_fwd");
}

extern "C" int nerf_mlp_ngp_fwd(const float* x, const float* dirs, int dirs_div, int64_t P, const float* const* tables,
                                const int* res, const int* sizes, const float* packed, float* rgb, float* sigma,
                                float* ws, float* enc, void* stream) {
    return ngp_fwd(x, dirs, dirs_div, P, tables, res, sizes, packed, rgb, sigma, ws, enc, nullptr, stream);
}

extern "C" int nerf_mlp_ngp_fwd_counted(const float* x, const float* dirs, int dirs_div, int64_t P_max, const int32_t* count,
                                        const float* const* tables, const int* res, const int* sizes, const float* packed,
                                        float* rgb, float* sigma, float* ws, float* enc, void* stream) {
    NERF_REQUIRE(count, "nerf_mlp_ngp_fwd_counted: null count");
    return ngp_fwd(x, dirs, dirs_div, P_max, tables, res, sizes, packed, rgb, sigma, ws, enc, count, stream);
}

static int ngp_bwd(const float* x, const float* d_rgb, const float* d_sigma, int64_t P, const int* res, const int* sizes,
                   const float* packed, const float* rgb, const float* sigma, float* ws, float* ws2, float* const* d_tables,
                   float* const* grads, void* scatter_ws, size_t scatter_ws_bytes, int phases, const int32_t* count,
                   void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && d_rgb && d_sigma && res && sizes && packed && rgb && sigma && ws && ws2 && d_tables && grads,
                 "nerf_mlp_ngp_bwd: null pointer");
    NERF_REQUIRE(phases >= 1 && phases <= 3, "nerf_mlp_ngp_bwd: phases=%d", phases);
    for (int i = 0; i < NERF_NGP_NPARAM; ++i) NERF_REQUIRE(grads[i], "nerf_mlp_ngp_bwd: gradient %d is null", i);
    if (int rc = check_levels(res, sizes, "nerf_mlp_ngp_bwd")) return rc;
    for (int l = 0; l < 16; ++l) NERF_REQUIRE(d_tables[l], "nerf_mlp_ngp_bwd: gradient table %d is null", l);
    hipStream_t st = (hipStream_t)stream;
    const long long P32 = (P + 31) / 32 * 32;
    // The backward kernel leaves one slot of partial weight-gradient sums per workgroup in ws2: as many workgroups as
    // there is room for slots (2 per CU at most); with no room at all (a handful of tiles) it adds to `grads` directly.
    const NgpGrads g{grads[0], grads[1], grads[2], grads[3], grads[4], grads[5], grads[8]};
    long long wgs = ceil_div(P32 / 32, 4);
    wgs = wgs < 512 ? wgs : 512;
    const long long room = P32 * (nw::PER_POINT2 - nw::PART) / NGP_PART_FLOATS;
    float* partials = room >= 1 ? ws2 + P32 * nw::PART : nullptr;
    if (partials && wgs > room) wgs = room;
    if (phases & 1) {
        // the backward-data chain with the weight-gradient sums of the three layers it walks, then the table gradients:
        // d_tables is complete when these launches are
        const size_t need = hashgrid_scatter_workspace_bytes(P, sizes, 16);
        const bool binned = scatter_ws && need && scatter_ws_bytes >= need && (size_t)scatter_ws % 16 == 0;
        hipLaunchKernelGGL(ngp_bwd_fused_kernel, dim3((unsigned)wgs), dim3(256), 0, st, d_rgb, d_sigma,
                           (long long)P, packed, rgb, sigma, ws, ws2, g, partials, count,
                           reinterpret_cast<unsigned*>(binned ? scatter_ws : nullptr), binned ? hashgrid_scatter_counter_words() : 0);
        if (int rc = check_launch("nerf_mlp_ngp_bwd(chain)")) return rc;
        if (int rc = launch_hashgrid_scatter(x, P, ws2 + P32 * nw::DENC, P32, res, sizes, 16, d_tables, scatter_ws, scatter_ws_bytes, st, count,
                                             binned)) return rc;
    }
    if (!(phases & 2)) return NERF_OK;
    if (partials) {   // grads += the workgroups' sums, in a fixed order
        static_assert(NGP_PART_FLOATS % RED_E == 0, "whole blocks of elements");
        hipLaunchKernelGGL(ngp_wgrad_reduce_kernel, dim3(NGP_PART_FLOATS / RED_E), dim3(RED_E * RED_G), 0, st, partials, (int)wgs, g);
        if (int rc = check_launch("nerf_mlp_ngp_bwd(reduce)")) return rc;
    }
    SkinnyPart c2{ws2 + P32 * nw::DOUT4, ws + P32 * nw::HC, grads[6], grads[7], 4, 64, 1, 3, 64};   // fc_color.2: d(sigma, rgb)_pre x hc
    c2.count = count;
    return launch_skinny(c2, P32, st);
}

extern "C" int nerf_mlp_ngp_bwd(const float* x, const float* d_rgb, const float* d_sigma, int64_t P, const int* res,
                                const int* sizes, const float* packed, const float* rgb, const float* sigma, float* ws,
                                float* ws2, float* const* d_tables, float* const* grads, void* scatter_ws,
                                size_t scatter_ws_bytes, int phases, void* stream) {
    return ngp_bwd(x, d_rgb, d_sigma, P, res, sizes, packed, rgb, sigma, ws, ws2, d_tables, grads, scatter_ws, scatter_ws_bytes,
                   phases, nullptr, stream);
}

extern "C" int nerf_mlp_ngp_bwd_counted(const float* x, const float* d_rgb, const float* d_sigma, int64_t P_max,
                                        const int32_t* count, const int* res, const int* sizes, const float* packed,
                                        const float* rgb, const float* sigma, float* ws, float* ws2, float* const* d_tables,
                                        float* const* grads, void* scatter_ws, size_t scatter_ws_bytes, int phases,
                                        void* stream) {
    NERF_REQUIRE(count, "nerf_mlp_ngp_bwd_counted: null count");
    return ngp_bwd(x, d_rgb, d_sigma, P_max, res, sizes, packed, rgb, sigma, ws, ws2, d_tables, grads, scatter_ws,
                   scatter_ws_bytes, phases, count, stream);
}
