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
// The backward kernel ends in the same layout, so each lane scatters the
// gradient of exactly the levels it gathered (float atomics into the tables).
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
constexpr int DYC = 0;      // [P32][64]
constexpr int DRAW = 64;    // [P32][32]
constexpr int DYD = 96;     // [P32][64]
constexpr int DOUT4 = 160;  // [P32][4]
constexpr int DENC = 164;   // [16][P32][2] gradient of the encoding, level-major like the encoding itself
constexpr int PER_POINT2 = 196;
}  // namespace nw


struct NgpIdx {
    long long tile, gp, lp, P32;
    int lane, h;
    bool valid;
};
__device__ __forceinline__ bool ngp_index(long long P, NgpIdx& ix) {
    ix.lane = threadIdx.x & 63;
    ix.tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    ix.h = ix.lane >> 5;
    ix.gp = ix.tile * 32 + (ix.lane & 31);
    ix.valid = ix.gp < P;
    ix.lp = ix.valid ? ix.gp : P - 1;
    ix.P32 = (P + 31) / 32 * 32;
    return ix.tile * 32 < P;
}

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

template <bool SAVE>
__global__ __launch_bounds__(256, 3) void ngp_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ dirs,
                                                      int dirs_div, long long P,
                                                      const float* __restrict__ pk, float* __restrict__ rgb,
                                                      float* __restrict__ sigma, float* __restrict__ ws) {
    // a wave's 32 x 64 (or 32 x 32) block of a saved matrix is contiguous in memory: its register tiles go through a
    // per-wave LDS transposition buffer and leave as whole rows, 16 bytes per lane (stored straight from the register
    // layout an instruction writes 64 pieces of 16 bytes, 256 bytes apart)
    __shared__ __attribute__((aligned(16))) float s_tile[4][TILE_LDS_FLOATS];
    NgpIdx ix;
    if (!ngp_index(P, ix)) return;
    const int lane = ix.lane, h = ix.h;
    const TileBuf tb = tile_buf(s_tile[threadIdx.x >> 6], lane);
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
        const unsigned wr_e = lds_addr(s_tile[threadIdx.x >> 6] + (lane & 31) * TILE_LD + 16 * h);
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
    fill_rows<2>(hd, pk + np::S_BD1, h);
    dense<1, 2>(pk + np::F_D1, e, hd, lane);
    relu_tiles<2>(hd);
    // every saved matrix leaves right after it is formed, under the layers that follow
    if (SAVE) save_tiles<2>(tb, ws + ix.P32 * nw::HD + ix.tile * (32 * 64), 64, hd);
    f32x16 cin[2];   // [raw | pe(dir)]
    {
        f32x16 raw[1];
        fill_rows<1>(raw, pk + np::S_BD2, h);
        dense<2, 1>(pk + np::F_D2, hd, raw, lane);
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
    fill_rows<2>(hc, pk + np::S_BC1, h);
    dense<2, 2>(pk + np::F_C1, cin, hc, lane);
    relu_tiles<2>(hc);
    if (SAVE) {
        save_tiles<2>(tb, ws + ix.P32 * nw::HC + ix.tile * (32 * 64), 64, hc);
        unsigned* mask = reinterpret_cast<unsigned*>(ws + ix.P32 * nw::MASK) + ix.tile * 128 + lane;
        mask[0] = relu_bits2(hd);
        mask[64] = relu_bits2(hc);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = dot_rows<2>(hc, pk + np::S_WC2 + 64 * c, h) + pk[np::S_BC2 + c];
        if (ix.valid && h == 0) rgb[3 * ix.gp + c] = sigmoid_f(v);
    }
}

// Backward-data chain ending in the hash-table scatter.
__global__ __launch_bounds__(256, 4) void ngp_bwd_data_kernel(const float* __restrict__ x, const float* __restrict__ d_rgb,
                                                           const float* __restrict__ d_sigma, long long P,
                                                           const float* __restrict__ pk,
                                                           const float* __restrict__ rgb,
                                                           const float* __restrict__ sigma,
                                                           const float* __restrict__ ws, float* __restrict__ ws2) {
    __shared__ __attribute__((aligned(16))) float s_tile[4][TILE_LDS_FLOATS];   // as in the forward kernel
    NgpIdx ix;
    if (!ngp_index(P, ix)) return;
    const int lane = ix.lane, h = ix.h;
    const TileBuf tb = tile_buf(s_tile[threadIdx.x >> 6], lane);
    const unsigned* mask = reinterpret_cast<const unsigned*>(ws + ix.P32 * nw::MASK) + ix.tile * 128 + lane;
    float dpre[3], dps = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float o = rgb[3 * ix.lp + c];
        dpre[c] = ix.valid ? d_rgb[3 * ix.lp + c] * o * (1.0f - o) : 0.0f;
    }
    if (ix.valid) dps = d_sigma[ix.lp] * (1.0f - expf(-sigma[ix.lp]));
    if (h == 0) *reinterpret_cast<float4*>(ws2 + ix.P32 * nw::DOUT4 + ix.gp * 4) = make_float4(dps, dpre[0], dpre[1], dpre[2]);
    // d hc = W4^T d rgb_pre, masked
    f32x16 dyc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 w = *reinterpret_cast<const float4*>(pk + np::S_WC2 + 64 * c + 32 * t + 8 * q + 4 * h);
                s.x = fmaf(w.x, dpre[c], s.x);
                s.y = fmaf(w.y, dpre[c], s.y);
                s.z = fmaf(w.z, dpre[c], s.z);
                s.w = fmaf(w.w, dpre[c], s.w);
            }
            dyc[t][4 * q + 0] = s.x;
            dyc[t][4 * q + 1] = s.y;
            dyc[t][4 * q + 2] = s.z;
            dyc[t][4 * q + 3] = s.w;
        }
    apply_bits2(dyc, mask[64]);
    save_tiles<2>(tb, ws2 + ix.P32 * nw::DYC + ix.tile * (32 * 64), 64, dyc);
    // d raw: features through (W3 + Wskip)^T, density through softplus'
    f32x16 draw[1];
    zero_tiles<1>(draw);
    dense<2, 1>(pk + np::B_C1, dyc, draw, lane);
    if (h == 0) draw[0][0] += dps;
    save_tiles<1>(tb, ws2 + ix.P32 * nw::DRAW + ix.tile * (32 * 32), 32, draw);
    f32x16 dyd[2];
    zero_tiles<2>(dyd);
    dense<1, 2>(pk + np::B_D2, draw, dyd, lane);
    apply_bits2(dyd, mask[0]);
    save_tiles<2>(tb, ws2 + ix.P32 * nw::DYD + ix.tile * (32 * 64), 64, dyd);
    f32x16 de[1];
    zero_tiles<1>(de);
    dense<2, 1>(pk + np::B_D1, dyd, de, lane);   // register r = d feature (16h + r): this lane's own levels
    // The table gradients are scattered by further launches (launch_hashgrid_scatter), which walk the levels one
    // after the other: level-major, so that each reads a contiguous run.  This lane holds d feature 16h + r in
    // register r, i.e. levels 8h .. 8h+7; 32 consecutive lanes write 256 contiguous bytes per level.
    float* denc = ws2 + ix.P32 * nw::DENC;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        *reinterpret_cast<float2*>(denc + 2 * ((8 * h + j) * ix.P32 + ix.gp)) = make_float2(de[0][2 * j], de[0][2 * j + 1]);
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

extern "C" int nerf_mlp_ngp_fwd(const float* x, const float* dirs, int dirs_div, int64_t P, const float* const* tables,
                                const int* res, const int* sizes, const float* packed, float* rgb, float* sigma,
                                float* ws, float* enc, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && tables && res && sizes && packed && rgb && sigma && enc, "nerf_mlp_ngp_fwd: null pointer");
    NERF_REQUIRE(P >= 0 && dirs_div >= 1, "nerf_mlp_ngp_fwd: P=%lld dirs_div=%d", (long long)P, dirs_div);
    if (int rc = check_levels(res, sizes, "nerf_mlp_ngp_fwd")) return rc;
    const long long P32 = (P + 31) / 32 * 32;
    if (int rc = launch_hashgrid_gather(x, P, tables, res, sizes, 16, enc, true, P32, (hipStream_t)stream)) return rc;
    const dim3 grid((unsigned)ceil_div(P, 128)), block(256);
    if (ws)
        hipLaunchKernelGGL(ngp_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, enc, dirs, dirs_div, (long long)P,
                           packed, rgb, sigma, ws);
    else
        hipLaunchKernelGGL(ngp_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, enc, dirs, dirs_div, (long long)P,
                           packed, rgb, sigma, ws);
    return check_launch("nerf_mlp_ngp_fwd");
}

extern "C" int nerf_mlp_ngp_bwd(const float* x, const float* d_rgb, const float* d_sigma, int64_t P, const int* res,
                                const int* sizes, const float* packed, const float* rgb, const float* sigma, float* ws,
                                float* ws2, float* const* d_tables, float* const* grads, void* scatter_ws,
                                size_t scatter_ws_bytes, int phases, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && d_rgb && d_sigma && res && sizes && packed && rgb && sigma && ws && ws2 && d_tables && grads,
                 "nerf_mlp_ngp_bwd: null pointer");
    NERF_REQUIRE(phases >= 1 && phases <= 3, "nerf_mlp_ngp_bwd: phases=%d", phases);
    for (int i = 0; i < NERF_NGP_NPARAM; ++i) NERF_REQUIRE(grads[i], "nerf_mlp_ngp_bwd: gradient %d is null", i);
    if (int rc = check_levels(res, sizes, "nerf_mlp_ngp_bwd")) return rc;
    for (int l = 0; l < 16; ++l) NERF_REQUIRE(d_tables[l], "nerf_mlp_ngp_bwd: gradient table %d is null", l);
    hipStream_t st = (hipStream_t)stream;
    const long long P32 = (P + 31) / 32 * 32;
    if (phases & 1) {   // the backward-data chain and the table gradients, which are complete when these launches are
        hipLaunchKernelGGL(ngp_bwd_data_kernel, dim3((unsigned)ceil_div(P, 128)), dim3(256), 0, st, x, d_rgb, d_sigma,
                           (long long)P, packed, rgb, sigma, ws, ws2);
        if (int rc = check_launch("nerf_mlp_ngp_bwd(data)")) return rc;
        if (int rc = launch_hashgrid_scatter(x, P, ws2 + P32 * nw::DENC, P32, res, sizes, 16, d_tables, scatter_ws, scatter_ws_bytes, st)) return rc;
    }
    if (!(phases & 2)) return NERF_OK;

    const float *E = ws + P32 * nw::E, *HD = ws + P32 * nw::HD, *RAW = ws + P32 * nw::RAW, *DE = ws + P32 * nw::DE,
                *HC = ws + P32 * nw::HC;
    const float *DYC = ws2 + P32 * nw::DYC, *DRAW = ws2 + P32 * nw::DRAW, *DYD = ws2 + P32 * nw::DYD,
                *DOUT4 = ws2 + P32 * nw::DOUT4;
    WgradParts parts{};
    int n = 0;
    parts.part[n++] = wgrad_part_desc(DYD, 64, E, 32, grads[0], 32, 0, 32, grads[1], 3);                    // fc_density.0
    {
        WgradPart p = wgrad_part_desc(DRAW, 32, HD, 64, grads[2], 64, 0, 64, grads[3], 4);                  // fc_density.2
        p.n_valid = 16;
        parts.part[n++] = p;
    }
    {
        // fc_color.0, and skip_proj with it: hc = W_c1 [feat | pe(dir)] + W_skip feat, so the gradient of W_skip is the
        // feature columns of the gradient of W_c1 -- the same sums, added to both
        WgradPart p = wgrad_part_desc(DYC, 64, RAW, 32, grads[4], 42, -1, 16, grads[5], 5, DE, 32, 15, 27);
        p.k_first = 1;
        p.dW2 = grads[8];
        p.ldW2 = 15;
        p.col_off2nd = -1;
        parts.part[n++] = p;
    }
    parts.n = n;
    if (int rc = launch_wgrad(parts, P32, st)) return rc;
    SkinnyPart c2{DOUT4, HC, grads[6], grads[7], 4, 64, 1, 3, 64};                                           // fc_color.2
    return launch_skinny(c2, P32, st);
}
